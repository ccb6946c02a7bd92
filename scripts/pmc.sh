#!/bin/bash
# usage: scripts/pmc.sh <outdir> <counters...>   (separate pass; no tracing flags besides kernel-trace)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out.log 2>&1
python3 - <<PY
import csv, glob, collections
f=glob.glob('$out/*/*counter_collection.csv')
if not f: print('no counter file', glob.glob('$out/*/*')); raise SystemExit
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
