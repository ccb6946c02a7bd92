#!/bin/bash
# round 6, gpurun call K: the roofline block of an N > 1 line (two ranks on one GPU through the host sockets, whole C3 workload; one rank through RCCL); stress + launcher tests
OUT=gpurun_out/r06k; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "results_behind_completion_flags or starts_its_ranks or randomized or findings" > $OUT/pytest_sub.txt 2>&1 || { tail -40 $OUT/pytest_sub.txt; exit 1; }
tail -3 $OUT/pytest_sub.txt
export CHIMERA_NO_REBUILD=1
show() { python3 -c "
import json; j = json.loads(open('$1').read().strip().split('\n')[-1]); r = j['roofline']
print('$1', 'n_gpus', j['n_gpus'], 'value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'frac', r['frac'], 'frac_of_sustained', r.get('frac_of_sustained'), 'kernel_ms', r['kernel_ms'], 'traffic_source', r.get('traffic_source'))
for k in r['kernels']: print('   ', k['kernel'], k['kernel_ms'], k.get('useful_frac'), (k.get('sustained') or {}).get('frac_of_sustained'), k.get('pmc_counters_scaled_by'))
print('   inflight2', (j.get('multi_gpu') or {}).get('inflight2'))"; }
timeout -k 10 400 python3 bench.py --gpus 2 --host-comm --steps 10 --warmup 3 > $OUT/two_ranks_host.json 2> $OUT/two_ranks_host.err || { tail -20 $OUT/two_ranks_host.err; exit 1; }
show $OUT/two_ranks_host.json
timeout -k 10 400 python3 bench.py --force-comm --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/one_rank_rccl.json 2> $OUT/one_rank_rccl.err || { tail -20 $OUT/one_rank_rccl.err; exit 1; }
show $OUT/one_rank_rccl.json
python3 -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
