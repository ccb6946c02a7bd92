#!/bin/bash
# round 6, gpurun call J: the whole GPU suite on the final tree; the full-mode bench line (per-kernel times of its eager calls); the default line once more
OUT=gpurun_out/r06j; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
export CHIMERA_NO_REBUILD=1
timeout -k 10 300 python3 bench.py --mode full --nbatch 4 --steps 10 --warmup 3 --no-extra > $OUT/bench_full.json 2> $OUT/bench_full.err || { tail $OUT/bench_full.err; exit 1; }
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python3 -c "
import json
for f in ('$OUT/bench_full.json', '$OUT/bench.json'):
  j = json.loads(open(f).read().strip().split('\n')[-1]); r = j['roofline']
  print(f, 'value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'single', j['single_call_ms'], 'frac', r['frac'], 'frac_of_sustained', r.get('frac_of_sustained'), 'pmc fresh', r['pmc_matches_loaded_code_object'], 'kernel_ms', r['kernel_ms'])"
