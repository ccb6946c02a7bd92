#!/bin/bash
# round 6, gpurun call F: chunks per block of the sample stage at the 125-event share; six ranks through the host sockets with the inflight-2 leg; the launcher test;
# a fuzz campaign with the tightened checker (p_gw atol 1e-12 of the largest density, sanity bound on the events not compared, total over the well-conditioned events)
OUT=gpurun_out/r06f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CHIMERA_NO_REBUILD=1
CHIMERA_LIB=$GRAFT_REPO_ROOT/chimera_amd/lib/variants/libchimera_hip_diag.so timeout -k 10 300 python3 scripts/try_cpb.py 125 12500 2>&1 | tee $OUT/try_cpb.txt || exit 1
T0=$(date +%s.%N)
timeout -k 10 400 python3 bench.py --gpus 6 --host-comm --steps 20 --warmup 3 > $OUT/rehearse6.json 2> $OUT/rehearse6.err || { tail -20 $OUT/rehearse6.err; exit 1; }
T1=$(date +%s.%N); echo "bench.py --gpus 6 --host-comm wall time: $(python3 -c "print('%.1f s' % ($T1 - $T0))")" | tee $OUT/rehearse6.time
python3 -c "
import json; j = json.loads(open('$OUT/rehearse6.json').read().strip().split('\n')[-1])
print('n_gpus', j['n_gpus'], 'value', j['value'], 'ms/step', j['ms_per_step'], json.dumps(j['multi_gpu'])[:1500])" | tee -a $OUT/rehearse6.time
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "starts_its_ranks or rccl or shared_gpu or two_processes or ticket" > $OUT/pytest_sub.txt 2>&1 || { tail -40 $OUT/pytest_sub.txt; exit 1; }
tail -3 $OUT/pytest_sub.txt
FUZZ_PGW=1 FUZZ_HOSTILE=0.3 FUZZ_EXTREME=0.3 FUZZ_MANY_EVERY=40 timeout -k 10 700 python3 scripts/fuzz_parity.py 12000 8100000 560 > $OUT/fuzz_campaign_1.txt 2>&1; echo "fuzz rc $?"; tail -4 $OUT/fuzz_campaign_1.txt | cut -c1-600
