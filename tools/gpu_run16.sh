#!/bin/bash
set -x
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_tokenizer_gpu.py -x -q -m gpu -k "gemm or conv" 2>&1 | tail -5
timeout 600 python tools/gemm_epilogue_probe.py 2>&1 | tee gpurun_out/gemm_epi_probe.txt
