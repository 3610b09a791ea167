#!/bin/bash
# round 6: the driver's --gpus 8 command line with 8 ranks SHARING the one GPU over gloo (plumbing of the whole multi-GPU path at full size: autotune agreement, RunGuard, the
# local_first + fused-QKV default, gather of the output) - RCCL refuses two ranks per GPU, so the collectives themselves go through the host here
cd "$(dirname "$0")/.." && export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
( time G3_BENCH_BACKEND=gloo G3_BENCH_SHARE_GPU=1 timeout 1700 python bench.py --gpus 8 --steps 2 --warmup 1 > gpurun_out/r6_bench8_share_gpu_line.json 2> gpurun_out/r6_bench8_share_gpu.err ) 2>&1 | tail -4
echo "rc=$?"; tail -c 1500 gpurun_out/r6_bench8_share_gpu_line.json; tail -5 gpurun_out/r6_bench8_share_gpu.err
