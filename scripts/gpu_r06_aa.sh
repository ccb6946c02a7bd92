#!/bin/bash
# round 6, gpurun call AA: the GPU suite on the release with the streaming (z, w) stores; same-box A/B against the plain-store build (code object 78a72e19...) at C3 and at the 125-event shard; the clocks
OUT=gpurun_out/r06aa; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
export CHIMERA_NO_REBUILD=1
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']; m = j.get('multi_gpu') or {}
i2 = (m.get('inflight2') or {})
print('%-6s %-28s ms_per_step=%.4f step_median=%.4f kde_integrate(one lane)=%.4f samples=%.4f inflight2=%s last=%r' % ('$1', '$2', j['ms_per_step'], j['step_ms']['median'] if j.get('step_ms') else -1, s['kde_integrate'], s['samples'], i2.get('ms_per_step', i2.get('error')), j['last_log_hyper']))"; }
for rep in 1 2 3; do
  for l in plain base; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --no-extra --steps 40 --warmup 5 2>/dev/null | line $l "C3 rep$rep" || exit 1
    timeout -k 10 200 python3 bench.py --force-comm --no-cpu-baseline --no-single-call --no-extra --steps 200 --warmup 5 --events 125 --inj 12500 2>/dev/null | line $l "shard125 1-rank RCCL rep$rep" || exit 1
  done
done 2>&1 | tee $OUT/ab_zw_stream.txt
export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_clock.so
timeout -k 10 300 python3 scripts/clock_under_load.py --events 1000 --seconds 4 > $OUT/clock_production.json 2> $OUT/clock.err || { tail -20 $OUT/clock.err; exit 1; }
timeout -k 10 300 python3 scripts/clock_under_load.py --events 4 --draws 16 --inj 4000 --seconds 3 > $OUT/clock_probes.json 2>> $OUT/clock.err || { tail -20 $OUT/clock.err; exit 1; }
python3 -c "
import json
for f in ('clock_production', 'clock_probes'):
    d = json.load(open('$OUT/' + f + '.json'))
    for k, v in d.items():
        if isinstance(v, dict) and 'GHz_median' in v: print(f, k, 'GHz_median', round(v['GHz_median'], 3), v.get('inside_the_kernels'))"
