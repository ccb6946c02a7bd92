#!/bin/bash
# A/B of library builds in ONE gpurun call (box-to-box variation is a few per cent): scripts/abl.sh "libA.so libB.so ..." [bench args]
# Variant builds: scripts/build_variant.sh NAME -DFLAG...  ->  chimera_amd/lib/variants/libchimera_hip_NAME.so
LIBS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for l in $LIBS; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('%-12s rep$rep ms_per_step=%.4f eval=%.4f samples=%.4f kde=%.4f sel=%.4f last=%r' % ('$l', j['ms_per_step'], s['eval'], s['samples'], s['kde_integrate'], s['selection'], j['last_log_hyper']))" || exit 1
  done
done
