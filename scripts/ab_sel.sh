#!/bin/bash
# A/B of the selection kernels on C4 (1e6 injections) in ONE gpurun call: scripts/ab_sel.sh "base plds ..." (variant builds of scripts/build_variant.sh);
# "generic" = the general kernel of the base build
LIBS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for l in $LIBS; do
    unset CHIMERA_LIB CHM_SELECTION_GENERIC
    if [ $l = generic ]; then export CHM_SELECTION_GENERIC=1; elif [ $l != base ]; then export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    CHM_SERIAL=1 timeout -k 10 200 python3 bench.py --config C4 --no-cpu-baseline --no-single-call --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('%-10s rep$rep value=%.1f ms_per_step=%.4f sel=%.4f last=%r' % ('$l', j['value'], j['ms_per_step'], s['selection'], j['last_log_hyper']))" || exit 1
  done
done
