#!/bin/bash
set -x
cd /root/repo
timeout 600 python tools/microbench.py gemm 2>&1 | tee gpurun_out/micro_pp2.txt
timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tee gpurun_out/bench_v8.txt
