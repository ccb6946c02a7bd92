#!/bin/bash
# Emulates the per-rank work of an N-GPU strong-scaling run on ONE GPU: bench.py on E/N events and I/N injections.
# (No all-reduce; shows the fixed per-step overhead that bounds the scaling.)  Output: gpurun_out/scale_probe.txt
OUT=gpurun_out/scale_probe.txt
: > $OUT
for n in 1 2 4 8; do
  E=$((1000 / n)); I=$((100000 / n))
  timeout -k 10 200 python3 bench.py --events $E --inj $I --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('N=$n E=$E I=$I ms_per_step=%.4f value=%.1f stage=%s' % (j['ms_per_step'], j['value'], json.dumps(j['roofline']['stage_ms'])))
" >> $OUT || exit 1
done
cat $OUT
