#!/bin/bash
# standalone kernel durations: everything on one stream (CHM_SERIAL=1) under rocprofv3 --kernel-trace --stats
OUT=gpurun_out/serial; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHM_SERIAL=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.err || exit 1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/trace
python3 - <<PY
import csv, json
j=json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1]); print('value', j['value'], 'ms/step', j['ms_per_step'])
for r in csv.DictReader(open('$OUT/kernel_stats.csv')):
    print('%-40s calls %4s avg_us %10.1f pct %6s' % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
