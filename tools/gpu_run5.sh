#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { name=$1; shift; echo "=== $name: $*"; timeout 900 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
TAILN=12 run t_render python -m pytest tests/test_render_gpu.py -m gpu -q --tb=short -s
TAILN=12 run cp2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/cp_check.py
TAILN=12 run cp4 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 tools/cp_check.py
