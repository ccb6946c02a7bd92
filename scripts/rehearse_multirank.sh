#!/bin/bash
# Rehearsal of bench.py's N > 1 paths on a ONE-GPU box (gpurun): two ranks sharing the GPU through the host communicator,
# one rank through RCCL (the in-stream all-reduce + k_combine path), and the plain run.  All three must print the same
# last_log_hyper; stdout of each run must be exactly one JSON line.
show() { python3 -c "
import sys, json
lines = sys.stdin.read().strip().split(chr(10))
assert len(lines) == 1, 'stdout must carry ONE line, got %d' % len(lines)
j = json.loads(lines[0]); print('$1:', round(j['value'], 1), 'evals/s  n_gpus', j['n_gpus'], ' last_log_hyper', repr(j['last_log_hyper']), '|', j['config']['parallelism'])"; }
A="--steps 5 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --host-comm $A 2>/dev/null | show "2 ranks, host comm" &&
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --force-comm $A 2>/dev/null | show "1 rank, RCCL      " &&
timeout -k 10 200 python3 bench.py $A 2>/dev/null | show "no communicator   "
