#!/bin/bash
# one gpurun call: rocprofv3 kernel stats of the bench command with every kernel on one stream -> gpurun_out/kstats/kernel_stats_serial.csv
export CHIMERA_NO_REBUILD=1 CHM_SERIAL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/kstats; mkdir -p $O; rm -rf $O/trace
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-single-call --steps 10 --warmup 2 ${BENCH_ARGS} > $O/run.log 2>&1
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats_serial.csv && rm -rf $O/trace
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/kstats/kernel_stats_serial.csv')))
for r in rows[:12]:
  print(r['Name'].split('(')[0][:60].ljust(60), r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1), '%', r['Percentage'])
PY
tail -2 $O/run.log | cut -c1-300
