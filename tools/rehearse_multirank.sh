#!/bin/bash
# Rehearses bench.py's N > 1 code path on a ONE-GPU box: two ranks, both on device 0,
# gloo instead of RCCL for the barrier / max-time reduce (BENCH_REHEARSE=1).  The numbers
# mean nothing (two ranks share one GPU and one CPU quota); the point is that the
# distributed control flow runs and prints one JSON line.
set -e
N=${1:-2}
BENCH_REHEARSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 2 --warmup 1
