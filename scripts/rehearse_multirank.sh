#!/bin/bash
# Rehearse the N > 1 path of bench.py on a ONE-GPU box (no PyTorch in the ranks; torch.distributed.run is only the launcher the
# driver uses): two ranks sharing GPU 0 with the partial sums reduced over the host sockets, and one rank through a real RCCL
# communicator (ncclAllReduce + k_combine inside chm_eval: the very code path of the 8-GPU job).
#   scripts/rehearse_multirank.sh [bench args]
A="--steps 5 --warmup 2 --no-cpu-baseline --no-single-call $*"
show() { python3 -c "
import sys, json
l=[x for x in sys.stdin.read().strip().split('\n') if x.startswith('{')]
j=json.loads(l[-1]); print('$1', 'value %.1f ms/step %.3f n_gpus %d' % (j['value'], j['ms_per_step'], j['n_gpus']), '|', j['config']['parallelism'], '| log_hyper', j['last_log_hyper'])"; }
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py $A 2>/dev/null | show "1 rank, no comm   " &&
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --host-comm $A 2>/dev/null | show "2 ranks, host comm" &&
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --force-comm $A 2>/dev/null | show "1 rank, RCCL      "
