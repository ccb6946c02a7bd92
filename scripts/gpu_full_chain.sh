#!/bin/bash
# one gpurun call: the full-mode parity tests on the default build, then the A/B of the chain-kernel variants on the full-mode bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/full_chain; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "full or groups" > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
export CHIMERA_NO_REBUILD=1
bash scripts/abl_full.sh "${LIBS:-base mw2 lk16}" 2>&1 | tee $O/ab.txt
CHM_FULL_CHAIN=0 bash scripts/abl_full.sh "base" 2>&1 | tee -a $O/ab.txt
