#!/bin/bash
# step time vs draws per call, at the full workload and at the per-rank share of an 8-GPU run
OUT=gpurun_out/nb_probe.txt
: > $OUT
for cfg in "1000 100000" "125 12500"; do
  set -- $cfg
  for nb in 16 64 128; do
    timeout -k 10 200 python3 bench.py --events $1 --inj $2 --nbatch $nb --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('E=$1 nb=$nb ms_per_step=%.4f us_per_eval=%.2f value=%.1f' % (j['ms_per_step'], 1e3*j['ms_per_step']/$nb, j['value']))
" >> $OUT || exit 1
  done
done
cat $OUT
