#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { name=$1; shift; echo "=== $name: $*"; timeout 900 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
TAILN=8 run t_gemm python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "gemm"
TAILN=40 run t_attn python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -s -k "attn"
TAILN=6 run t_dit python -m pytest tests/test_dit_gpu.py -m gpu -q --tb=short -s
TAILN=12 run micro python tools/microbench.py attn gemm
TAILN=3 run bench python bench.py --steps 2 --warmup 1 --no-cpu-baseline
