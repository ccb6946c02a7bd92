#!/bin/bash
# Kernel timeline of one 128-draw call at the per-rank share of an 8-GPU C3 run (125 events, 12 500 injections) -> gpurun_out/r05q/timeline_shard125_batched.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
mkdir -p gpurun_out/r05q
rm -rf gpurun_out/tl; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-single-call --events 125 --inj 12500 > gpurun_out/tl.log 2>&1 || exit 1
python3 - <<PY > gpurun_out/r05q/timeline_shard125_batched.txt
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/tl/*/*kernel_trace.csv')[0])))
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows]
rows.sort()
starts = [i for i, r in enumerate(rows) if 'k_tables' in r[2]] + [len(rows)]
calls = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
# the timed calls: those with the most kernels; take the third of them
nmax = max(b - a for a, b in calls)
big = [c for c in calls if c[1] - c[0] == nmax]
i0, i1 = big[min(3, len(big) - 1)]
t0 = rows[i0][0]
print('one 128-draw call of the 125-event shard: %d kernels, %.1f us from first start to last end' % (i1 - i0, (max(r[1] for r in rows[i0:i1]) - t0) / 1e3))
for s, e, n, q in rows[i0:i1]:
    print('%-40s q %-3s start %9.1f  end %9.1f  dur %8.1f us' % (n, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
cat gpurun_out/r05q/timeline_shard125_batched.txt; rm -rf gpurun_out/tl
