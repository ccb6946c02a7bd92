#!/bin/bash
# scalar-call latency A/B of library builds in ONE gpurun call: scripts/abl_scalar.sh "base nologs ..."; also the one-draw kernel times (--nbatch 1 --no-graph)
LIBS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
for rep in 1 2; do
  for l in $LIBS; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['single_call']
print('%-10s rep$rep single_call median=%.4f q25=%.4f q75=%.4f ms  last=%r' % ('$l', s['median_ms'], s['q25_ms'], s['q75_ms'], j['last_log_hyper']))" || exit 1
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --nbatch 1 --no-graph --steps 200 --warmup 20 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('%-10s rep$rep nbatch1 ms_per_step=%.4f samples=%.4f kde=%.4f tables=%.4f' % ('$l', j['ms_per_step'], s['samples'], s['kde_integrate'], s['tables']))" || exit 1
  done
done
