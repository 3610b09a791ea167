#!/bin/bash
# First GPU contact: kernel parity tests in separate processes (a fault in one must not hide the others),
# then smoke, then a short bench. Everything is logged under gpurun_out/.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { name=$1; shift; echo "=== $name: $*"; timeout 600 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n 25 gpurun_out/$name.log; }
rocm-smi --showproductname 2>/dev/null | head -8
run t_gemm python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -s -k "gemm"
run t_attn python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -s -k "attn"
run t_misc python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -s -k "not gemm and not attn"
run t_dit python -m pytest tests/test_dit_gpu.py -m gpu -q --tb=short -s
run smoke python __graft_entry__.py --smoke
run bench2 python bench.py --blocks 2 --steps 1 --warmup 1 --no-cpu-baseline
run bench python bench.py --steps 2 --warmup 1
