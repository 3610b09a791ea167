#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { name=$1; shift; echo "=== $name: $*"; timeout 900 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
TAILN=3 run bench python bench.py --steps 2 --warmup 1 --no-cpu-baseline
TAILN=4 run t_dit python -m pytest tests/test_dit_gpu.py tests/test_cp_gpu.py -m gpu -q --tb=short
