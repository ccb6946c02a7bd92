#!/bin/bash
# round 6, gpurun call B2: same-box A/B WITH counters of the GW kernel's LDS layouts and the sample stage's LDS tables on C3 (scripts/ab_counters.py)
OUT=gpurun_out/r06c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
timeout -k 10 1150 python3 scripts/ab_counters.py $OUT/ab_counters.txt "$@" > $OUT/ab_counters.log 2>&1 || { tail -30 $OUT/ab_counters.log; exit 1; }
cat $OUT/ab_counters.txt
