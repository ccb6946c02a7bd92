#!/bin/bash
# round 6, gpurun call M: the whole GPU suite on the final library (draws from pinned memory for every call size) + a fuzz campaign + the default line
OUT=gpurun_out/r06m; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
export CHIMERA_NO_REBUILD=1
FUZZ_PGW=1 FUZZ_HOSTILE=0.3 FUZZ_EXTREME=0.3 FUZZ_MANY_EVERY=40 timeout -k 10 460 python3 scripts/fuzz_parity.py 12000 8400000 400 > $OUT/fuzz_campaign_4.txt 2>&1; echo "fuzz 4 rc $?"; tail -3 $OUT/fuzz_campaign_4.txt | cut -c1-900
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python3 -c "
import json
j = json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]); r = j['roofline']
print('value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'single', j['single_call_ms'], 'frac', r['frac'], 'frac_of_sustained', r.get('frac_of_sustained'), 'pmc fresh', r['pmc_matches_loaded_code_object'], 'parity', j['parity_full_size']['abs_diff'])"
