#!/bin/bash
# round 6, gpurun call N: draws per call 128 against 256 (whole C3 workload and the 125-event share, same box); a long fuzz campaign on the final library
OUT=gpurun_out/r06n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); m = j.get('multi_gpu') or {}
i2 = (m.get('inflight2') or {})
print('%-34s value=%.1f evals/s ms_per_step=%.4f per-draw us=%.2f inflight2=%s' % ('$1', j['value'], j['ms_per_step'], 1e3 * j['ms_per_step'] / j['config']['nbatch'], i2.get('value', i2.get('error'))))"; }
for rep in 1 2; do
  for nb in 128 256; do
    timeout -k 10 300 python3 bench.py --nbatch $nb --no-cpu-baseline --no-single-call --no-extra --steps 30 --warmup 3 2>/dev/null | line "C3 nbatch $nb rep$rep" || exit 1
    timeout -k 10 300 python3 bench.py --nbatch $nb --force-comm --no-cpu-baseline --no-single-call --no-extra --steps 200 --warmup 5 --events 125 --inj 12500 2>/dev/null | line "shard125 nbatch $nb rep$rep" || exit 1
  done
done 2>&1 | tee $OUT/nbatch_128_256.txt
FUZZ_PGW=1 FUZZ_HOSTILE=0.15 FUZZ_EXTREME=0.2 FUZZ_MANY_EVERY=25 timeout -k 10 700 python3 scripts/fuzz_parity.py 20000 8500000 640 > $OUT/fuzz_campaign_5.txt 2>&1; echo "fuzz 5 rc $?"; tail -3 $OUT/fuzz_campaign_5.txt | cut -c1-900
