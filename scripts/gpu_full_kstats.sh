#!/bin/bash
# one gpurun call: rocprofv3 kernel stats of the full-mode bench for each library build in $LIBS -> gpurun_out/full_kstats/
export CHIMERA_NO_REBUILD=1 CHM_SERIAL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/full_kstats; mkdir -p $O
for l in ${LIBS:-base}; do
  if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
  rm -rf $O/trace
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --mode full --nbatch 4 --no-cpu-baseline --no-single-call --steps 5 --warmup 2 ${BENCH_ARGS} > $O/run_$l.log 2>&1 || exit 1
  cp $O/trace/*/*kernel_stats.csv $O/kernel_stats_$l.csv && rm -rf $O/trace
  echo "== $l"; python3 - $O/kernel_stats_$l.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
  print(r['Name'].split('(')[0][:60].ljust(60), r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1), '%', r['Percentage'])
PY
done
