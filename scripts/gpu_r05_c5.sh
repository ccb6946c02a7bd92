#!/bin/bash
# round 5, one gpurun call: [GPU test suite] + BASELINE config 5 (mg_flrw): the whole workload on one GPU, its per-rank shard of an 8-GPU run (1250 events,
# 12 500 injections) with stage times, and rocprofv3 kernel stats + PMC passes of that shard (scripts/collect_profiles.py --tag C5shard)
#   scripts/gpu_r05_c5.sh [tests|notests]
mkdir -p gpurun_out/r05e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "${1:-tests}" = tests ]; then
  timeout -k 10 900 python3 -m pytest tests -q -m gpu > gpurun_out/r05e/pytest_gpu.txt 2>&1
  tail -5 gpurun_out/r05e/pytest_gpu.txt | cut -c1-300
fi
export CHIMERA_NO_REBUILD=1
echo "C5 per-rank shard of an 8-GPU run on ONE GPU (1250 events, 12 500 injections, mg_flrw; 128 draws per call): stage times in ms" > gpurun_out/r05e/shard_C5_one_gpu.txt
for a in "--config C5 --events 1250 --inj 12500 --nbatch 128" "--config C5 --events 1250 --inj 12500 --nbatch 16"; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 $a 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = j['roofline']['stage_ms']
print('$a', 'ms/step %.4f' % j['ms_per_step'], 'evals/s %.0f' % j['value'], 'scalar call ms', j.get('single_call_ms'), {k: round(v,4) for k,v in s.items() if isinstance(v,float)})" >> gpurun_out/r05e/shard_C5_one_gpu.txt
done
cat gpurun_out/r05e/shard_C5_one_gpu.txt
timeout -k 10 500 python3 scripts/collect_profiles.py r05 --tag C5shard -- --config C5 --events 1250 --inj 12500 --nbatch 128 --steps 10 --warmup 2 > gpurun_out/r05e/collect_C5shard.log 2>&1
tail -12 gpurun_out/r05e/collect_C5shard.log | cut -c1-400
cp profiles/r05/pmc_per_launch_C5shard.json gpurun_out/r05e/ 2>/dev/null
