#!/bin/bash
# whole -m gpu suite + smoke on the round-6 tree
cd "$(dirname "$0")/.." && export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -m gpu -q --tb=short -x ) > gpurun_out/r6_suite.log 2>&1; echo "suite rc=$?"; tail -8 gpurun_out/r6_suite.log
python __graft_entry__.py --smoke 2>&1 | tail -2
