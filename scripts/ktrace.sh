#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of a short bench run; extra args go to bench.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/kt.log 2>&1
python3 - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob('gpurun_out/kt/*/*kernel_stats.csv')[0])):
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f}  {r['Percentage']:>6s}%")
PY
rm -rf gpurun_out/kt
