#!/bin/bash
# round 6, last gpurun call: what the driver runs at round end, on the final tree -- pytest -m gpu, smoke(), bench.py
OUT=gpurun_out/r06o; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
python3 -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python3 -c "
import json
j = json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]); r = j['roofline']
print('value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'single', j['single_call_ms'], 'frac', r['frac'], 'frac_of_sustained', r.get('frac_of_sustained'), 'hbm_call_frac', r.get('hbm_call_frac'), 'pmc fresh', r['pmc_matches_loaded_code_object'], 'parity', j['parity_full_size']['abs_diff'], 'cpu', j['cpu_baseline']['value'], 'extra', sorted(j['extra']['configs']))"
