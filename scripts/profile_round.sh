#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun); writes under gpurun_out/profiles_rNN/.
#   scripts/profile_round.sh r01
set -u
R=${1:-r01}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. bench line (no profiler)
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
# 2. kernel trace + stats of the same command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
# 2b. the same with every kernel on ONE stream (CHM_SERIAL=1): standalone durations (in the default run k_selection and
#     k_zfactors overlap the sample stage on their own streams, so their trace durations are stretched)
CHM_SERIAL=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- python3 bench.py --no-cpu-baseline > $OUT/bench_serial_under_rocprof.json 2> $OUT/trace_serial.err
cp $OUT/trace_serial/*/*kernel_stats.csv $OUT/kernel_stats_serial.csv 2>/dev/null
# 3. PMC passes (separate runs, counters only)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for f in glob.glob('$OUT/pmc_*/*/*counter_collection.csv'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
json.dump(out, open('$OUT/pmc_per_launch.json', 'w'), indent=1, sort_keys=True)
b = json.loads(open('$OUT/bench.json').read().strip().split('\n')[-1])
nb = b['config']['nbatch']
KN = 'k_kde_marg_sub2<32, 4>'
k = out.get(KN, {})
if 'FETCH_SIZE' in k and 'WRITE_SIZE' in k:
    # gfx950: FETCH_SIZE counts 64 B per 128-B request of wide (16 B/lane) streaming reads -> doubled (MI355X_MICROARCH.md, HBM);
    # the kernel also issues 8 B/lane reads, for which the counter is uncalibrated: the doubled figure is an upper estimate.
    per_launch = (2 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024
    json.dump({'kernel': KN, 'E': b['config']['E'], 'P': b['config']['P'], 'Z': b['config']['Z'], 'nbatch': nb, 'FETCH_SIZE_KB_per_launch': k['FETCH_SIZE'],
               'WRITE_SIZE_KB_per_launch': k['WRITE_SIZE'], 'bytes_per_launch': per_launch, 'bytes_per_draw': per_launch / nb,
               'note': 'bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads'},
              open('$OUT/pmc_traffic.json', 'w'), indent=1)
print(open('$OUT/kernel_stats.csv').read()[:3000])
print(json.dumps(out.get(KN, {})), json.dumps(out.get('k_samples<true, false>', {})))
PY
rm -rf $OUT/trace $OUT/trace_serial $OUT/pmc_*/ 2>/dev/null
ls -la $OUT
