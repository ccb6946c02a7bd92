#!/bin/bash
# round 6, gpurun call D: GPU suite on the SPLIT layout; shard step with the early / late selection join; probes on the new layout; 6-rank host-socket rehearsal; the default bench line
OUT=gpurun_out/r06d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
export CHIMERA_NO_REBUILD=1
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']; m = j.get('multi_gpu') or {}
i2 = (m.get('inflight2') or {})
print('%-10s %-34s ms_per_step=%.4f step_median=%.4f eval=%.4f inflight2=%s last=%r' % ('$1', '$2', j['ms_per_step'], j['step_ms']['median'] if j.get('step_ms') else -1, s['eval_timed'], i2.get('ms_per_step', i2.get('error')), j['last_log_hyper']))"; }
for rep in 1 2 3; do
  for l in latesel base; do
    if [ $l = base ]; then unset CHIMERA_LIB; else export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_$l.so; fi
    timeout -k 10 200 python3 bench.py --force-comm --no-cpu-baseline --no-single-call --no-extra --steps 200 --warmup 5 --events 125 --inj 12500 2>/dev/null | line $l "shard125 1-rank RCCL rep$rep" || exit 1
  done
done 2>&1 | tee $OUT/ab_shard_sel.txt
unset CHIMERA_LIB
for rep in 1 2; do timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --no-extra --steps 30 --warmup 3 2>/dev/null | line base "C3 rep$rep" || exit 1; done 2>&1 | tee $OUT/c3.txt
rm -rf $OUT/tl; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python3 bench.py --force-comm --no-inflight2 --steps 4 --warmup 2 --no-cpu-baseline --no-single-call --no-extra --events 125 --inj 12500 > $OUT/tl.log 2>&1 || exit 1
python3 - <<PY > $OUT/timeline_shard125_rccl.txt
import csv, glob
rows = list(csv.DictReader(open(glob.glob('$OUT/tl/*/*kernel_trace.csv')[0])))
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-44:], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows if 'copyBuffer' not in r['Kernel_Name']]
rows.sort()
starts = [i for i, r in enumerate(rows) if 'k_tables' in r[2]] + [len(rows)]
calls = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
nmax = max(b - a for a, b in calls)
big = [c for c in calls if c[1] - c[0] >= nmax - 1]
i0, i1 = big[min(3, len(big) - 1)]
t0 = rows[i0][0]
print('one 128-draw call of the 125-event shard through a one-rank RCCL communicator: %d kernels, %.1f us from first start to last end' % (i1 - i0, (max(r[1] for r in rows[i0:i1]) - t0) / 1e3))
for s, e, n, q in rows[i0:i1]:
    print('%-44s q %-3s start %9.1f  end %9.1f  dur %8.1f us' % (n, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
cat $OUT/timeline_shard125_rccl.txt; rm -rf $OUT/tl
# the default line (what the driver runs), wall time
/usr/bin/time -v -o $OUT/bench_default.time timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
grep -E "Elapsed|Maximum resident" $OUT/bench_default.time
python3 -c "
import json; j = json.loads(open('$OUT/bench_default.json').read().strip().split('\n')[-1])
print('value', j['value'], 'ms/step', j['ms_per_step'], 'single', j['single_call_ms'], 'frac', j['roofline']['frac'], 'cpu', j['cpu_baseline']['value'], 'parity', j['parity_full_size']['abs_diff'])
for k, v in j['extra']['configs'].items(): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a not in ('workload', 'note')})
print('extra wall', j['extra']['wall_s'])"
# six ranks on this one GPU through the host sockets (a rehearsal of the launcher and of the rank-side code of an N-GPU run; the pool allows six processes on the card)
/usr/bin/time -v -o $OUT/rehearse6.time timeout -k 10 400 python3 bench.py --gpus 6 --host-comm --steps 20 --warmup 3 > $OUT/rehearse6.json 2> $OUT/rehearse6.err || { tail -20 $OUT/rehearse6.err; exit 1; }
grep -E "Elapsed" $OUT/rehearse6.time
python3 -c "
import json; j = json.loads(open('$OUT/rehearse6.json').read().strip().split('\n')[-1])
print('n_gpus', j['n_gpus'], 'value', j['value'], 'ms/step', j['ms_per_step'], json.dumps(j['multi_gpu'])[:900])"
# probes on the new layout
export CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_probe.so
timeout -k 10 240 python3 scripts/run_probes.py --events 4 --draws 4 --seconds 1.5 --out $OUT/probe_E4_nb4.json > $OUT/probe_E4_nb4.txt 2> $OUT/probe.err || { tail -20 $OUT/probe.err; exit 1; }
grep sustained $OUT/probe_E4_nb4.txt
rm -rf $OUT/pp; timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pp -- python3 scripts/run_probes.py --events 4 --draws 4 --seconds 1.5 > $OUT/probe_pmc.log 2>&1 || { tail -20 $OUT/probe_pmc.log; exit 1; }
python3 - <<PY | tee $OUT/probe_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob('$OUT/pp/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_probe' in r['Kernel_Name']: agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$OUT/pp/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_probe' in r['Kernel_Name']: dur[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
for k, v in agg.items():
    n = len(dur[k]); h = n // 2
    ms = sorted(dur[k][h:])[len(dur[k][h:]) // 2]
    c = {c_: sorted(x[h:])[len(x[h:]) // 2] for c_, x in v.items()}
    print(k, 'launches', n, 'median ms (second half) %.2f' % ms, ' '.join('%s=%.6g' % kv for kv in sorted(c.items())), 'VALU winst/s = %.4g' % (c.get('SQ_INSTS_VALU', 0) / (ms * 1e-3)),
          'clock GHz = %.3f' % (c.get('GRBM_GUI_ACTIVE', 0) / 8 / (ms * 1e-3) / 1e9))
PY
rm -rf $OUT/pp
