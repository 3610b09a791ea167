#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { name=$1; shift; echo "=== $name: $*"; timeout 900 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
TAILN=30 run t_cli python -m pytest tests/test_cli_gpu.py -m gpu -q --tb=short -s
