#!/bin/bash
# Round-6 profile set, part 2 of 2: full mode (kernel stats + PMC + bench line);  C4;  C5 (whole workload on ONE GPU) bench line
export CHIMERA_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r06; mkdir -p $O
python3 scripts/collect_profiles.py r06 --tag full --passes 0,1,3,5 -- --mode full --nbatch 4 --steps 10 --warmup 3 --no-extra > $O/collect_full.log 2>&1; tail -6 $O/collect_full.log | cut -c1-400
python3 scripts/collect_profiles.py r06 --tag C4 --passes 0,1,3,5 -- --config C4 --no-extra > $O/collect_C4.log 2>&1; tail -4 $O/collect_C4.log | cut -c1-400
timeout -k 10 600 python3 bench.py --config C5 --nbatch 16 --steps 10 --warmup 2 --cpu-evals 3 > $O/bench_C5.json 2> $O/bench_C5.err
for f in $O/bench_C4.json $O/bench_C5.json $O/bench_full.json; do python3 -c "
import json,sys; j=json.loads(open('$f').read().strip().split('\n')[-1]); print('$f', round(j['value'],1), 'evals/s', round(j['ms_per_step'],3), 'ms/step single', j['single_call_ms'], 'cpu', j.get('cpu_baseline',{}).get('value'), 'parity', (j.get('parity_full_size') or {}).get('abs_diff'))"; done
