#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { name=$1; shift; echo "=== $name: $*"; timeout 900 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
TAILN=4 run t_attn python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -k "attn"
TAILN=4 run micro python tools/microbench.py attn
TAILN=3 run bench python bench.py --steps 2 --warmup 1 --no-cpu-baseline
