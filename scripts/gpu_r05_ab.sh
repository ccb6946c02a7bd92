#!/bin/bash
# one gpurun call of round 5: [tests] + same-box A/B of library builds on the C3 bench (+ board power sampled by rocm-smi beside one run)
#   scripts/gpu_r05_ab.sh OUTDIR "libA libB ..." [tests|notests] [bench args]
OUT=$1; LIBS=$2; T=${3:-tests}; shift 3
mkdir -p gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$T" = tests ]; then
  timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/$OUT/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/$OUT/pytest_gpu.txt; exit 1; }
  tail -3 gpurun_out/$OUT/pytest_gpu.txt
fi
bash scripts/abl.sh "$LIBS" "$@" 2>&1 | tee gpurun_out/$OUT/ab.txt
# board power / clocks while the default build runs the bench (rocm-smi every ~0.25 s; hwmon's power1_input does not follow the load on this box)
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' '; echo; sleep 0.2; done ) > gpurun_out/$OUT/power_samples.txt &
SMI=$!
unset CHIMERA_LIB
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 600 --warmup 5 "$@" > gpurun_out/$OUT/bench_long.json 2> gpurun_out/$OUT/bench_long.err
wait $SMI
sort gpurun_out/$OUT/power_samples.txt | uniq -c | sort -rn | head -12
