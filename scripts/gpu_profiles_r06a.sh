#!/bin/bash
# Round-6 profile set, part 1 of 2 (one gpurun call each; writes gpurun_out/profiles_r06/, copy to profiles/r06/):
#   probes of the two hot kernel bodies (plain + one --pmc pass) -> probe_ceilings.json tied to the release binary;
#   C3 default command: kernel stats (default + one stream), PMC passes, the bench line (with frac_of_sustained and the extra legs);
#   C3 nbatch = 1 (the scalar call): PMC + kernel stats; scalar-call timelines: the whole workload and the per-rank share of an 8-GPU run
export CHIMERA_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r06; mkdir -p $O profiles/r06
CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_probe.so timeout -k 10 240 python3 scripts/run_probes.py --events 4 --draws 4 --seconds 1.5 --out $O/probe_E4_nb4.json > $O/probe_E4_nb4.txt 2> $O/probe.err || { tail -20 $O/probe.err; exit 1; }
grep sustained $O/probe_E4_nb4.txt
CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_probe.so timeout -k 10 240 python3 scripts/run_probes.py --events 16 --draws 8 --seconds 1.5 --out $O/probe_E16_nb8.json > $O/probe_E16_nb8.txt 2>> $O/probe.err || { tail -20 $O/probe.err; exit 1; }
grep sustained $O/probe_E16_nb8.txt
rm -rf $O/pp; CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_probe.so timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pp -- python3 scripts/run_probes.py --events 4 --draws 4 --seconds 1.5 > $O/probe_pmc.log 2>&1 || { tail -20 $O/probe_pmc.log; exit 1; }
python3 - <<PY | tee $O/probe_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob('$O/pp/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_probe' in r['Kernel_Name']: agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$O/pp/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_probe' in r['Kernel_Name']: dur[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
for k, v in agg.items():
    n = len(dur[k]); h = n // 2
    ms = sorted(dur[k][h:])[len(dur[k][h:]) // 2]
    c = {c_: sorted(x[h:])[len(x[h:]) // 2] for c_, x in v.items()}
    print(k, 'launches', n, 'median ms (second half) %.2f' % ms, ' '.join('%s=%.6g' % kv for kv in sorted(c.items())), 'VALU winst/s = %.4g' % (c.get('SQ_INSTS_VALU', 0) / (ms * 1e-3)),
          'clock GHz = %.3f' % (c.get('GRBM_GUI_ACTIVE', 0) / 8 / (ms * 1e-3) / 1e9))
PY
rm -rf $O/pp
python3 scripts/make_probe_ceilings.py $O r06 > $O/make_probe_ceilings.log 2>&1 && cp profiles/r06/probe_ceilings.json $O/probe_ceilings.json
python3 scripts/collect_profiles.py r06 > $O/collect_C3.log 2>&1; tail -14 $O/collect_C3.log | cut -c1-400
python3 scripts/collect_profiles.py r06 --tag nbatch1 -- --nbatch 1 --no-graph --steps 200 --warmup 20 > $O/collect_nb1.log 2>&1; tail -8 $O/collect_nb1.log | cut -c1-400
python3 scripts/timeline_scalar.py $O/timeline_scalar_call.txt > /dev/null 2>&1
python3 scripts/timeline_scalar.py $O/timeline_scalar_call_shard125.txt --events 125 --inj 12500 > /dev/null 2>&1
cat $O/timeline_scalar_call.txt $O/timeline_scalar_call_shard125.txt
