#!/bin/bash
# A/B of the block count of k_zfactors (CHM_ZF_TARGET builds): step time of the C3 bench, two repetitions, same box
export CHIMERA_NO_REBUILD=1
for rep in 1 2; do
  for v in "" zf1536 zf1024; do
    if [ -n "$v" ]; then export CHIMERA_LIB=$PWD/chimera_amd/lib/variants/libchimera_hip_$v.so; else unset CHIMERA_LIB; fi
    python3 bench.py --no-cpu-baseline --no-single-call --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); s=j['roofline']['stage_ms']; print('${v:-zf2048 (default)}', 'rep $rep', 'ms_per_step', round(j['ms_per_step'],3), 'value', round(j['value'],1))"
  done
done
