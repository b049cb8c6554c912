#!/bin/bash
# Rehearses bench.py --strong (configs[2]: 4K frames sharded by superblock rows) on a ONE-GPU
# box: N ranks on device 0, gloo between them - the strips travel as the same packed buffers
# od_hip_gather_strips sends over RCCL, through host memory (BENCH_REHEARSE=1).  The rate
# means nothing (the ranks share one GPU and one CPU quota); the point is that the sharded
# control flow runs end to end and that rank 0's packets equal the pure reference build's.
set -e
N=${1:-2}
python bench.py --gpus 1 --strong --steps 1 --warmup 0 --strong-frames 2
BENCH_REHEARSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
  --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus $N --strong --steps 1 --warmup 0 --strong-frames 2
