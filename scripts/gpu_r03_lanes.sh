#!/bin/bash
# Round-3 rehearsal of two evaluations in flight per rank (one gpurun call) -> gpurun_out/lanes/rehearse_two_in_flight.txt
#   the lane tests, then bench.py with --inflight 1 / 2 on the whole C3 workload and on the per-rank share of an 8-GPU run
#   (125 events, 12 500 injections), and the same through a one-rank RCCL communicator (two lanes = two communicators)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lanes; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lanes or rccl_single or sharded or graph" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && exit $rc
export CHIMERA_NO_REBUILD=1
line() { python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-44s value=%9.1f evals/s  ms_per_step=%.4f  last=%r  %s' % ('$1', j['value'], j['ms_per_step'], j['last_log_hyper'], j['config']['parallelism']))"; }
{
for rep in 1 2; do
  for fl in 1 2; do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --steps 30 --warmup 4 --inflight $fl 2>/dev/null | line "C3 whole workload, inflight $fl (rep $rep)" || exit 1
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-single-call --steps 60 --warmup 4 --events 125 --inj 12500 --inflight $fl 2>/dev/null | line "125-event shard, inflight $fl (rep $rep)" || exit 1
  done
done
for fl in 1 2; do
  timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-comm --no-cpu-baseline --no-single-call --steps 60 --warmup 4 --events 125 --inj 12500 --inflight $fl 2>/dev/null | line "125-event shard, 1-rank RCCL, inflight $fl" || exit 1
done
} | tee $O/rehearse_two_in_flight.txt
