#!/bin/bash
# instruction-cache counters of the hot kernels (one rocprofv3 --pmc pass of the default bench command) -> gpurun_out/icache/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1 CHM_GROUPS=1
O=gpurun_out/icache; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/t -- python3 bench.py --no-cpu-baseline --no-single-call --steps 3 --warmup 1 > $O/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/icache/t/*/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
  k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
  agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
  if v.get('SQ_WAVE_CYCLES', 0) > 1e8:
    print(k.ljust(46), {a: '%.3g' % b for a, b in v.items()})
PY
