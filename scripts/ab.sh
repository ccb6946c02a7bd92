#!/bin/bash
# A/B of an environment toggle in ONE gpurun call (box-to-box variation is ~5 %): scripts/ab.sh VAR [bench args]
VAR=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in off on; do
    if [ $v = on ]; then export $VAR=1; else unset $VAR; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('$VAR=$v rep$rep ms_per_step=%.4f eval=%.4f samples=%.4f kde=%.4f events_wall=%.4f reduce=%.4f' % (j['ms_per_step'], s['eval'], s['samples'], s['kde_integrate'], s['events_wall'], s['reduce']))" || exit 1
  done
done
