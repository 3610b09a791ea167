#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { name=$1; shift; echo "=== $name: $*"; timeout 900 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n 12 gpurun_out/$name.log; }
run smoke python __graft_entry__.py --smoke
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA" > gpurun_out/lscpu.txt; cat gpurun_out/lscpu.txt
for t in 16 32 64 128; do
  run cpu_$t python -c "
import bench
print(bench.cpu_baseline($t))"
done
export TMPDIR=/tmp
ROOTD=$(pwd)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/prof_r1 -o bench -- python $ROOTD/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $ROOTD/gpurun_out/prof_bench.log 2>&1
echo "rc=$? (rocprof)"; tail -n 3 $ROOTD/gpurun_out/prof_bench.log
cd $ROOTD
find gpurun_out/prof_r1 -type f | head -20
f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); echo $f; head -40 "$f"
# keep only the small summaries (the trace itself can be big)
find gpurun_out/prof_r1 -name "*kernel_trace.csv" -size +20M -delete
