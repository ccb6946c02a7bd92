#!/bin/bash
# round 6, gpurun call AI: GW kernel, an event's blocks on 8 (= the release's mapping through the experiment's code), 2 or 1 of the eight XCDs; five event groups of 200 events (the experiment's mapping needs E_cnt % 8 == 0)
OUT=gpurun_out/r06ai; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('%-10s %-22s ms_per_step=%.4f step_median=%.4f kde_integrate(one lane)=%.4f samples=%.4f last=%r' % ('$1', '$2', j['ms_per_step'], j['step_ms']['median'] if j.get('step_ms') else -1, s['kde_integrate'], s['samples'], j['last_log_hyper']))"; }
for rep in 1 2; do
  for l in base gk8 gk2 gk1; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --no-extra --steps 40 --warmup 5 --groups 5 2>/dev/null | line $l "C3 5 groups rep$rep" || exit 1
  done
done 2>&1 | tee $OUT/ab_gw_xcdk.txt
