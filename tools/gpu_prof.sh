#!/bin/bash
# rocprofv3 kernel-trace summary of an arbitrary command: tools/gpu_prof.sh <name> <cmd...>  -> gpurun_out/<name>_kernel_stats.csv
# (rocprofv3 runs from /tmp as the microarch guide prescribes; arguments naming files of the repo are made absolute first)
name=$1; shift
mkdir -p gpurun_out; ROOTD=$(pwd); export TMPDIR=/tmp PYTHONUNBUFFERED=1
args=()
for a in "$@"; do if [ -e "$ROOTD/$a" ] && [[ "$a" != /* ]]; then args+=("$ROOTD/$a"); else args+=("$a"); fi; done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/prof_$name -o $name -- "${args[@]}" > $ROOTD/gpurun_out/prof_$name.log 2>&1
echo "rc=$? rocprof"; cd $ROOTD
db=$(find gpurun_out/prof_$name -name "*.db" | head -1)
python tools/rocpd_summary.py $db gpurun_out/${name}_kernel_stats.csv
