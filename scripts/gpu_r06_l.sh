#!/bin/bash
# round 6, gpurun call L: the draws read by k_tables from pinned memory for calls of EVERY size (base) against the H2D copy in front of k_tables above 8 draws (zc8), 125-event share
OUT=gpurun_out/r06l; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']; m = j.get('multi_gpu') or {}
i2 = (m.get('inflight2') or {})
print('%-10s %-34s ms_per_step=%.4f step_median=%.4f eval=%.4f inflight2=%s last=%r' % ('$1', '$2', j['ms_per_step'], j['step_ms']['median'] if j.get('step_ms') else -1, s['eval_timed'], i2.get('ms_per_step', i2.get('error')), j['last_log_hyper']))"; }
for rep in 1 2 3 4; do
  for l in zc8 base; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 200 python3 bench.py --force-comm --no-cpu-baseline --no-single-call --no-extra --steps 400 --warmup 5 --events 125 --inj 12500 2>/dev/null | line $l "shard125 1-rank RCCL rep$rep" || exit 1
  done
done 2>&1 | tee $OUT/ab_shard_zc.txt
for rep in 1 2; do
  for l in zc8 base; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --no-extra --steps 30 --warmup 3 2>/dev/null | line $l "C3 rep$rep" || exit 1
  done
done 2>&1 | tee -a $OUT/ab_shard_zc.txt
