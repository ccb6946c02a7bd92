#!/bin/bash
# Round-5 profile set, part 1 of 2 (one gpurun call each; writes gpurun_out/profiles_r05/, copy to profiles/r05/):
#   C3 default command: kernel stats (default + one stream), PMC passes incl. the VALU-class counters, bench line;  C3 nbatch=1 (the scalar call): PMC + kernel stats;
#   scalar-call timelines: the whole workload and the per-rank share of an 8-GPU run (125 events, 12 500 injections)
export CHIMERA_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r05; mkdir -p $O
python3 scripts/collect_profiles.py r05 > $O/collect_C3.log 2>&1; tail -14 $O/collect_C3.log | cut -c1-400
python3 scripts/collect_profiles.py r05 --tag nbatch1 -- --nbatch 1 --no-graph --steps 200 --warmup 20 > $O/collect_nb1.log 2>&1; tail -8 $O/collect_nb1.log | cut -c1-400
python3 scripts/timeline_scalar.py $O/timeline_scalar_call.txt > /dev/null 2>&1
python3 scripts/timeline_scalar.py $O/timeline_scalar_call_shard125.txt --events 125 --inj 12500 > /dev/null 2>&1
cat $O/timeline_scalar_call.txt $O/timeline_scalar_call_shard125.txt
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --events 125 --inj 12500 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); print('125-event shard: ms/step %.4f, scalar call %.4f ms' % (j['ms_per_step'], j['single_call_ms']))" | tee -a $O/timeline_scalar_call_shard125.txt
