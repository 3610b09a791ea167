#!/bin/bash
# round 5, GPU call 2: first run of the deferred-epilogue GEMM kernel (every command under its own timeout: a hang must not take the box)
mkdir -p gpurun_out/r5c2
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "deferred" > gpurun_out/r5c2/t_w4e.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c2/t_w4e.log
tail -25 gpurun_out/r5c2/t_w4e.log
timeout 300 python tools/microbench.py gemm > gpurun_out/r5c2/microbench_gemm.txt 2>&1; echo "rc $?" >> gpurun_out/r5c2/microbench_gemm.txt
cat gpurun_out/r5c2/microbench_gemm.txt
