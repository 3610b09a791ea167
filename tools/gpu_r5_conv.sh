#!/bin/bash
# round 5: tokenizer convolutions after the K-loop bookkeeping rewrite - tokenizer / conv tests, timing
mkdir -p gpurun_out/r5v
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tokenizer_gpu.py tests/test_kernels_gpu.py -q -x -k "conv or tokenizer or Conv" > gpurun_out/r5v/t_conv.log 2>&1; tail -3 gpurun_out/r5v/t_conv.log
for i in 1 2 3; do timeout 200 python tools/bench_tokenizer_single.py 2>&1 | grep "^tokenizer"; done | tee gpurun_out/r5v/tokenizer_timing.txt
timeout 300 python tools/conv_probe.py > gpurun_out/r5v/conv_probe.txt 2>&1; grep -v "^gemm\|amdgpu.ids" gpurun_out/r5v/conv_probe.txt | tail -14
