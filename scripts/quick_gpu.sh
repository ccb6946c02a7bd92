#!/bin/bash
# tests + bench + kernel stats in one gpurun call; writes under gpurun_out/quick/
OUT=gpurun_out/quick; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; rc=$?
tail -5 $OUT/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 -c "
import json; j=json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]); print('value', j['value'], 'ms/step', j['ms_per_step'], j['roofline']['stage_ms'], 'frac', j['roofline']['frac'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/trace
cut -c1-60,200-400 $OUT/kernel_stats.csv | head -12
python3 - <<PY
import csv
for r in csv.DictReader(open('$OUT/kernel_stats.csv')):
    print('%-40s calls %4s avg_us %10.1f pct %6s' % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
