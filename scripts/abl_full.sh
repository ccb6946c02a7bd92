#!/bin/bash
# A/B of library builds on the full-mode bench in ONE gpurun call: scripts/abl_full.sh "base lk16 lk48" [bench args]
LIBS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for l in $LIBS; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 300 python3 bench.py --mode full --nbatch 4 --no-cpu-baseline --no-single-call --steps 5 --warmup 2 "$@" 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']; k = j['roofline']['kernels'][0]
print('%-8s rep$rep value=%.1f ms_per_step=%.3f kde=%.3f samples=%.3f Gpairs/s=%.1f pair_frac=%.3f last=%r' % ('$l', j['value'], j['ms_per_step'], s['kde_integrate'], s['samples'], k['Gpairs_s'], k['pair_frac'], j['last_log_hyper']))" || exit 1
  done
done
