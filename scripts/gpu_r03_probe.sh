#!/bin/bash
# Round-3 probe (one gpurun call): the issue-cost table of the card; optionally PMC passes of the default bench command (PASSES=5,6).
export CHIMERA_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_probe; mkdir -p $O
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/issue_cost scripts/issue_cost.hip && timeout -k 10 400 /tmp/issue_cost > $O/issue_cost.txt 2> $O/issue_cost.err
tail -3 $O/issue_cost.txt
if [ -n "$PASSES" ]; then timeout -k 10 900 python3 scripts/collect_profiles.py r03 --skip-trace --passes $PASSES > $O/collect.log 2>&1; tail -15 $O/collect.log; fi
