#!/bin/bash
# Timeline of ONE chm_eval call (kernel start/end from rocprofv3 --kernel-trace): where a call's time goes between kernels.
# usage: bash scripts/timeline.sh [bench args]   (default: the per-rank share of an 8-GPU C3 run)   -> gpurun_out/timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="${@:---events 125 --inj 12500}"
rm -rf gpurun_out/tl; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline $ARGS > gpurun_out/tl.log 2>&1 || exit 1
python3 - <<PY > gpurun_out/timeline.txt
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/tl/*/*kernel_trace.csv')[0])))
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-28:], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows]
rows.sort()
# calls start with k_tables
starts = [i for i, r in enumerate(rows) if 'k_tables' in r[2]]
i0, i1 = starts[-2], starts[-1]
t0 = rows[i0][0]
prev_end = t0
print('last full call: %d kernels, %.1f us from first start to last end; next call starts %.1f us after this one ends' % (i1 - i0, (max(r[1] for r in rows[i0:i1]) - t0) / 1e3, (rows[i1][0] - max(r[1] for r in rows[i0:i1])) / 1e3))
for s, e, n, q in rows[i0:i1]:
    print('%-28s q %-3s start %9.1f  end %9.1f  dur %8.1f us' % (n, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
cat gpurun_out/timeline.txt; rm -rf gpurun_out/tl
