#!/bin/bash
# round 6, gpurun call AB: event groups small enough for a group's (z, w) to stay in the memory-side cache (256 MB = 31 events x 128 draws): 4 (default) / 8 / 16 / 32 / 40 / 64 groups at C3
OUT=gpurun_out/r06ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('%-10s %-12s ms_per_step=%.4f step_median=%.4f eval_timed=%.4f last=%r' % ('$1', '$2', j['ms_per_step'], j['step_ms']['median'] if j.get('step_ms') else -1, s['eval_timed'], j['last_log_hyper']))"; }
for rep in 1 2; do
  for g in 4 8 16 32 40 64; do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --no-extra --steps 40 --warmup 5 --groups $g 2>/dev/null | line "groups $g" "C3 rep$rep" || exit 1
  done
done 2>&1 | tee $OUT/ab_groups_mall.txt
