#!/bin/bash
set -x
cd /root/repo
export G3_BENCH_BACKEND=gloo G3_BENCH_SHARE_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --blocks 2 2>&1 | tail -5
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 8 --steps 1 --warmup 1 --blocks 1 2>&1 | tail -5
