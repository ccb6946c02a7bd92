#!/bin/bash
# is the p_gw pattern mismatch of seed 8302601 (campaign 3: hostile input, every distance of an event far beyond the table) older than round 6?  the round-5 library on the same seed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1 FUZZ_PGW=1 FUZZ_HOSTILE=0.5 FUZZ_EXTREME=0.5 FUZZ_INF_RATE=0.2
mkdir -p gpurun_out/r06h
echo "round-6 library:" | tee gpurun_out/r06h/seed_8302601.txt; timeout 120 python3 scripts/fuzz_parity.py 1 8302601 2>&1 | tail -4 | cut -c1-900 | tee -a gpurun_out/r06h/seed_8302601.txt
echo "round-5 library (commit 49d667d):" | tee -a gpurun_out/r06h/seed_8302601.txt; CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_r05.so timeout 120 python3 scripts/fuzz_parity.py 1 8302601 2>&1 | tail -4 | cut -c1-900 | tee -a gpurun_out/r06h/seed_8302601.txt
bash scripts/gpu_profiles_r06a.sh
