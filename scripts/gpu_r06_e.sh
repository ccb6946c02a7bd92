#!/bin/bash
# round 6, gpurun call E: the default bench line (what the driver runs) with its wall time; six ranks through the host sockets on one GPU; probes on the final layout
OUT=gpurun_out/r06e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
T0=$(date +%s.%N)
timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
T1=$(date +%s.%N); echo "default bench.py wall time: $(python3 -c "print('%.1f s' % ($T1 - $T0))")" | tee $OUT/bench_default.time
python3 -c "
import json; j = json.loads(open('$OUT/bench_default.json').read().strip().split('\n')[-1])
print('value', j['value'], 'ms/step', j['ms_per_step'], 'single', j['single_call_ms'], 'frac', j['roofline']['frac'], 'cpu', j['cpu_baseline']['value'], 'parity', j['parity_full_size']['abs_diff'])
for k, v in j['extra']['configs'].items(): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a not in ('workload', 'note')})
print('extra wall', j['extra']['wall_s'])"
T0=$(date +%s.%N)
timeout -k 10 400 python3 bench.py --gpus 6 --host-comm --steps 20 --warmup 3 > $OUT/rehearse6.json 2> $OUT/rehearse6.err || { tail -20 $OUT/rehearse6.err; exit 1; }
T1=$(date +%s.%N); echo "bench.py --gpus 6 --host-comm wall time: $(python3 -c "print('%.1f s' % ($T1 - $T0))")" | tee $OUT/rehearse6.time
python3 -c "
import json; j = json.loads(open('$OUT/rehearse6.json').read().strip().split('\n')[-1])
print('n_gpus', j['n_gpus'], 'value', j['value'], 'ms/step', j['ms_per_step'], json.dumps(j['multi_gpu'])[:1200])"
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
