#!/bin/bash
# round 6, gpurun call Z: memory-path counters of the sample stage with plain stores (the round's earlier release), non-temporal stores (release) and no stores (diagnostic)
OUT=gpurun_out/r06z; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 scripts/pmc_memory_path.py $OUT/pmc_memory_path.txt plain base nost 2>&1 | tee $OUT/log.txt || { tail -30 $OUT/log.txt; exit 1; }
cat $OUT/pmc_memory_path.txt
