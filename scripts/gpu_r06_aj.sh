#!/bin/bash
# round 6, gpurun call AJ: the final tree once more -- default bench line; six ranks on one GPU through the host sockets (the N > 1 code path: shards, inflight-2 leg, line)
OUT=gpurun_out/r06aj; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python3 -c "
import json
j = json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]); r = j['roofline']
print('value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'single', j['single_call_ms'], 'frac', r['frac'], 'frac_of_sustained', r.get('frac_of_sustained'), 'pmc fresh', r['pmc_matches_loaded_code_object'], 'parity', j['parity_full_size']['abs_diff'])"
export CHIMERA_NO_REBUILD=1
T0=$(date +%s.%N)
timeout -k 10 400 python3 bench.py --gpus 6 --host-comm --steps 20 --warmup 3 > $OUT/rehearse6.json 2> $OUT/rehearse6.err || { tail -20 $OUT/rehearse6.err; exit 1; }
T1=$(date +%s.%N); echo "bench.py --gpus 6 --host-comm wall time: $(python3 -c "print('%.1f s' % ($T1 - $T0))")" | tee $OUT/rehearse6.time
python3 -c "
import json; j = json.loads(open('$OUT/rehearse6.json').read().strip().split('\n')[-1])
print('n_gpus', j['n_gpus'], 'value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'scaling', j['scaling'], 'inflight2', (j.get('multi_gpu') or {}).get('inflight2'))"
