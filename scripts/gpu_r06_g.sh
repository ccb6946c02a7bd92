#!/bin/bash
# round 6, gpurun call G: the flag / result-arrival stress test; fuzz campaigns 2 and 3 on the fixed binary
OUT=gpurun_out/r06g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "results_behind_completion_flags or starts_its_ranks or last_timing or option" > $OUT/pytest_sub.txt 2>&1 || { tail -40 $OUT/pytest_sub.txt; exit 1; }
tail -3 $OUT/pytest_sub.txt
export CHIMERA_NO_REBUILD=1
FUZZ_PGW=1 FUZZ_HOSTILE=0.3 FUZZ_EXTREME=0.3 FUZZ_MANY_EVERY=40 timeout -k 10 500 python3 scripts/fuzz_parity.py 12000 8200000 440 > $OUT/fuzz_campaign_2.txt 2>&1; echo "fuzz 2 rc $?"; tail -3 $OUT/fuzz_campaign_2.txt | cut -c1-700
FUZZ_PGW=1 FUZZ_HOSTILE=0.5 FUZZ_EXTREME=0.5 FUZZ_INF_RATE=0.2 timeout -k 10 500 python3 scripts/fuzz_parity.py 12000 8300000 440 > $OUT/fuzz_campaign_3.txt 2>&1; echo "fuzz 3 rc $?"; tail -3 $OUT/fuzz_campaign_3.txt | cut -c1-700
