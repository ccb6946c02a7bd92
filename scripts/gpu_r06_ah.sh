#!/bin/bash
# round 6, gpurun call AH: a fuzz campaign on the final release (code object b5509c45...), new seed, same checker settings as campaigns 4 and 5
OUT=gpurun_out/r06ah; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FUZZ_PGW=1 FUZZ_HOSTILE=0.3 FUZZ_EXTREME=0.3 FUZZ_MANY_EVERY=40 timeout -k 10 500 python3 scripts/fuzz_parity.py 12000 8600000 400 > $OUT/fuzz_campaign_6.txt 2>&1; echo "fuzz 6 rc $?"; tail -3 $OUT/fuzz_campaign_6.txt | cut -c1-900
