#!/bin/bash
# round 6, gpurun call A: GPU suite; same-box A/B of the host side of the batched call (round-5 host code vs round 6) on the per-rank share of an 8-GPU C3 run
# (125 events, 12 500 injections) and on the whole of C3; the N > 1 code path on one rank (--force-comm: RCCL + the inflight-2 leg); kernel timeline of the share
#   scripts/gpu_r06_a.sh [tests|notests]
OUT=gpurun_out/r06a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "${1:-tests}" = tests ]; then
  timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
  tail -3 $OUT/pytest_gpu.txt
fi
export CHIMERA_NO_REBUILD=1
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']; m = j.get('multi_gpu') or {}
print('%-10s %-28s ms_per_step=%.4f step_median=%.4f eval=%.4f samples=%.4f kde=%.4f last=%r %s' % ('$1', '$2', j['ms_per_step'], j['step_ms']['median'] if j.get('step_ms') else -1, s['eval'], s['samples'], s['kde_integrate'], j['last_log_hyper'], json.dumps(m.get('inflight2')) if m else ''))"; }
for rep in 1 2 3; do
  for l in r05host base; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --no-extra --steps 200 --warmup 5 --events 125 --inj 12500 2>/dev/null | line $l "shard125 rep$rep" || exit 1
  done
done 2>&1 | tee $OUT/ab_shard.txt
for rep in 1 2; do
  for l in r05host base; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --no-extra --steps 30 --warmup 3 2>/dev/null | line $l "C3 rep$rep" || exit 1
  done
done 2>&1 | tee $OUT/ab_c3.txt
unset CHIMERA_LIB
# the N > 1 code path on one rank: rendezvous + RCCL communicator + the guarded inflight-2 leg
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --force-comm --no-cpu-baseline --no-extra --steps 200 --warmup 5 --events 125 --inj 12500 2>$OUT/force_comm.err | tee $OUT/force_comm_$rep.json | line base "shard125 1-rank RCCL rep$rep" || { tail -20 $OUT/force_comm.err; exit 1; }
done 2>&1 | tee $OUT/force_comm.txt
timeout -k 10 300 python3 bench.py --force-comm --inflight 2 --no-cpu-baseline --no-extra --no-single-call --steps 200 --warmup 5 --events 125 --inj 12500 2>/dev/null | line base "shard125 1-rank RCCL inflight2" | tee -a $OUT/force_comm.txt
# kernel timeline of one 128-draw call of the share
rm -rf $OUT/tl; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-single-call --no-extra --events 125 --inj 12500 > $OUT/tl.log 2>&1 || exit 1
python3 - <<PY > $OUT/timeline_shard125_batched.txt
import csv, glob
rows = list(csv.DictReader(open(glob.glob('$OUT/tl/*/*kernel_trace.csv')[0])))
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows]
rows.sort()
starts = [i for i, r in enumerate(rows) if 'k_tables' in r[2]] + [len(rows)]
calls = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
nmax = max(b - a for a, b in calls)
big = [c for c in calls if c[1] - c[0] >= nmax - 1]
i0, i1 = big[min(3, len(big) - 1)]
t0 = rows[i0][0]
print('one 128-draw call of the 125-event shard: %d kernels, %.1f us from first start to last end' % (i1 - i0, (max(r[1] for r in rows[i0:i1]) - t0) / 1e3))
for s, e, n, q in rows[i0:i1]:
    print('%-40s q %-3s start %9.1f  end %9.1f  dur %8.1f us' % (n, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
cat $OUT/timeline_shard125_batched.txt; rm -rf $OUT/tl
