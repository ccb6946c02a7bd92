#!/bin/bash
# The other BASELINE.json configurations and call modes (parity-test cases, not the bench line): gpurun_out/variants/*.json
OUT=gpurun_out/variants; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 500 python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -3 $OUT/$name.err; return 1; }
  python3 -c "
import json; j=json.loads(open('$OUT/$name.json').read().strip().split('\n')[-1]); print('%-22s value %10.1f evals/s  ms_per_step %8.3f  nbatch %3d  single_call_ms %s' % ('$name', j['value'], j['ms_per_step'], j['config']['nbatch'], j['single_call_ms']))"; }
run nbatch1 --nbatch 1 --steps 50 --warmup 5 &&
run nbatch16 --nbatch 16 --single-call &&
run approximate --mode approximate &&
run full_mode --mode full --nbatch 4 --steps 5 --warmup 1 &&
run C1 --config C1 &&
run C2 --config C2 &&
run C4 --config C4 --nbatch 16 &&
run C5_events2000 --config C5 --events 2000 --nbatch 16 --steps 5 --warmup 2
