#!/bin/bash
# Round-5 profile set, part 2 of 2: full mode (kernel stats + PMC + bench line + board power beside the run);  C4;  bench lines of C1, C2, C5 (whole workload on one GPU), approximate
export CHIMERA_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r05; mkdir -p $O
python3 scripts/collect_profiles.py r05 --tag full --passes 0,1,3,5 -- --mode full --nbatch 4 --steps 10 --warmup 3 > $O/collect_full.log 2>&1; tail -6 $O/collect_full.log | cut -c1-400
( for i in $(seq 1 50); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' '; echo; sleep 0.2; done ) > $O/power_samples_full_mode.txt &
SMI=$!
timeout -k 10 200 python3 bench.py --mode full --nbatch 4 --steps 400 --warmup 3 --no-cpu-baseline --no-single-call > /dev/null 2>&1
wait $SMI
sort $O/power_samples_full_mode.txt | uniq -c | sort -rn | grep -v "24[0-9].0\|25[0-9].0\|23[0-9].0" | head -5
python3 scripts/collect_profiles.py r05 --tag C4 --passes 0,1,3,5 -- --config C4 > $O/collect_C4.log 2>&1; tail -4 $O/collect_C4.log | cut -c1-400
for c in C1 C2; do timeout -k 10 300 python3 bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout -k 10 600 python3 bench.py --config C5 --nbatch 16 --steps 10 --warmup 2 --cpu-evals 3 > $O/bench_C5.json 2> $O/bench_C5.err
timeout -k 10 300 python3 bench.py --mode approximate > $O/bench_approximate.json 2> $O/bench_approximate.err
for f in $O/bench_C1.json $O/bench_C2.json $O/bench_C4.json $O/bench_C5.json $O/bench_approximate.json $O/bench_full.json; do python3 -c "
import json,sys; j=json.loads(open('$f').read().strip().split('\n')[-1]); print('$f', round(j['value'],1), 'evals/s', round(j['ms_per_step'],3), 'ms/step single', j['single_call_ms'], 'cpu', j.get('cpu_baseline',{}).get('value'), 'parity', (j.get('parity_full_size') or {}).get('abs_diff'))"; done
