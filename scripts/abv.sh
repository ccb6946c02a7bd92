#!/bin/bash
# sweep of an environment variable's values in ONE gpurun call: scripts/abv.sh VAR "v1 v2 ..." [bench args]
VAR=$1; VALS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in $VALS; do
    export $VAR=$v
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('$VAR=$v rep$rep ms_per_step=%.4f eval=%.4f samples=%.4f kde=%.4f events_wall=%.4f reduce=%.4f' % (j['ms_per_step'], s['eval'], s['samples'], s['kde_integrate'], s['events_wall'], s['reduce']))" || exit 1
  done
done
