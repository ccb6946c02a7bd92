#!/bin/bash
# scalar-call latency A/B in ONE gpurun call over environment switches of the library: scripts/ab_scalar.sh "LABEL:VAR=val,VAR=val" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%:*}; vars=${spec#*:}
    env $(echo $vars | tr ',' ' ') python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['single_call']
print('%-28s rep$rep single_call median=%.4f q25=%.4f q75=%.4f ms  last=%r' % ('$label', s['median_ms'], s['q25_ms'], s['q75_ms'], j['last_log_hyper']))" || exit 1
  done
done
