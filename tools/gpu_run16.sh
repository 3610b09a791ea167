#!/bin/bash
set -x
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -5
timeout 600 python tools/microbench.py gemm 2>&1 | tee gpurun_out/micro_pp.txt
