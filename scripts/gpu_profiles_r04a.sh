#!/bin/bash
# Round-4 profile set in one gpurun call (writes gpurun_out/profiles_r04/, copy to profiles/r04/):
#   C3 default command: kernel stats (default + one stream), PMC passes incl. the VALU-class counters, bench line;  C3 nbatch=1 (the scalar call):
#   PMC + kernel stats;  the FUSED event kernel (--fused 2) at 128 draws and at one draw per call: kernel stats + traffic / instruction counters;
#   full mode: kernel stats + bench line;  C4;  bench lines of C1, C2, C5, approximate;  scalar-call timeline
export CHIMERA_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r04; mkdir -p $O
python3 scripts/collect_profiles.py r04 > $O/collect_C3.log 2>&1; tail -14 $O/collect_C3.log
python3 scripts/collect_profiles.py r04 --tag nbatch1 -- --nbatch 1 --no-graph --steps 200 --warmup 20 > $O/collect_nb1.log 2>&1; tail -8 $O/collect_nb1.log
python3 scripts/collect_profiles.py r04 --tag fused --passes 0,1,3,5 -- --fused 2 --steps 20 > $O/collect_fused.log 2>&1; tail -6 $O/collect_fused.log
python3 scripts/collect_profiles.py r04 --tag fused_nbatch1 --passes 0,1,3 -- --fused 2 --nbatch 1 --no-graph --steps 200 --warmup 20 > $O/collect_fused_nb1.log 2>&1; tail -6 $O/collect_fused_nb1.log
python3 scripts/collect_profiles.py r04 --tag full --passes 0,1,3,5 -- --mode full --nbatch 4 --steps 10 --warmup 3 > $O/collect_full.log 2>&1; tail -6 $O/collect_full.log
