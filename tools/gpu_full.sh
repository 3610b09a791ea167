#!/bin/bash
# Full round-end style check: whole GPU test-suite, smoke, default bench, rocprofv3 kernel stats of the bench.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; timeout 1500 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
TAILN=6 run t_all python -m pytest tests -m gpu -q --tb=short
TAILN=3 run smoke python __graft_entry__.py --smoke
TAILN=2 run bench python bench.py
ROOTD=$(pwd); cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/prof_final -o bench -- python $ROOTD/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $ROOTD/gpurun_out/prof_final.log 2>&1
echo "rc=$? rocprof"; cd $ROOTD; ls -la gpurun_out/prof_final | head -5
