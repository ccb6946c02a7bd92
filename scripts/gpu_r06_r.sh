#!/bin/bash
# round 6, gpurun call R: (z, w) past the caches in the many-draw call (non-temporal stores in the sample stage, non-temporal loads in the GW kernel), same box A/B at C3;
# the loop-free probes over 10 s (does the clock they run at hold?)
OUT=gpurun_out/r06r; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('%-6s %-22s ms_per_step=%.4f step_median=%.4f kde_integrate(one lane)=%.4f samples=%.4f last=%r' % ('$1', '$2', j['ms_per_step'], j['step_ms']['median'] if j.get('step_ms') else -1, s['kde_integrate'], s['samples'], j['last_log_hyper']))"; }
for rep in 1 2 3; do
  for l in base ntst ntzw ntboth; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --no-extra --steps 40 --warmup 5 2>/dev/null | line $l "C3 rep$rep" || exit 1
  done
done 2>&1 | tee $OUT/ab_nt_zw.txt
CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_probe.so timeout -k 10 240 python3 scripts/run_probes.py --events 4 --draws 4 --seconds 9.6 --out $OUT/probe_long.json > $OUT/probe_long.txt 2> $OUT/probe.err || { tail -20 $OUT/probe.err; exit 1; }
grep sustained $OUT/probe_long.txt
python3 -c "
import json; d = json.load(open('$OUT/probe_long.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'launch_ms' in v: print(k, [round(x, 1) for x in v['launch_ms']])"
