#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { name=$1; shift; echo "=== $name: $*"; timeout 900 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
TAILN=4 run t_gemm python -m pytest tests/test_kernels_gpu.py tests/test_tokenizer_gpu.py -m gpu -q --tb=short -k "gemm or tokenizer"
TAILN=6 run micro python tools/microbench.py gemm
