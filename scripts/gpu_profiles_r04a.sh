#!/bin/bash
# Round-4 profile set, part 1 of 2 (one gpurun call each; writes gpurun_out/profiles_r04/, copy to profiles/r04/):
#   C3 default command: kernel stats (default + one stream), PMC passes incl. the VALU-class counters, bench line;  C3 nbatch=1 (the scalar call):
#   PMC + kernel stats;  the FUSED event kernel (--fused 2) at 128 draws and at one draw per call: kernel stats + traffic / instruction counters
export CHIMERA_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r04; mkdir -p $O
python3 scripts/collect_profiles.py r04 > $O/collect_C3.log 2>&1; tail -14 $O/collect_C3.log
python3 scripts/collect_profiles.py r04 --tag nbatch1 -- --nbatch 1 --no-graph --steps 200 --warmup 20 > $O/collect_nb1.log 2>&1; tail -8 $O/collect_nb1.log
python3 scripts/collect_profiles.py r04 --tag fused --passes 0,1,3,5 -- --fused 2 --steps 20 > $O/collect_fused.log 2>&1; tail -6 $O/collect_fused.log
python3 scripts/collect_profiles.py r04 --tag fused_nbatch1 --passes 0,1,3 -- --fused 2 --nbatch 1 --no-graph --steps 200 --warmup 20 > $O/collect_fused_nb1.log 2>&1; tail -6 $O/collect_fused_nb1.log
