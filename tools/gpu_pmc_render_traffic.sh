#!/bin/bash
# Memory-side traffic of the renderer kernels (product configuration, tools/bench_render_single.py): FETCH_SIZE and WRITE_SIZE in their own
# rocprofv3 --pmc passes (--kernel-trace only), per kernel and launch -> gpurun_out/pmc_render_traffic.txt
mkdir -p gpurun_out/pmc_rt
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOTD/gpurun_out/pmc_rt/$c -o p -- python $ROOTD/tools/bench_render_single.py > $ROOTD/gpurun_out/pmc_rt/$c.log 2>&1
  echo "rc=$? ($c)"
done
cd $ROOTD
python tools/pmc_summary.py gpurun_out/pmc_rt gpurun_out/pmc_render_traffic.csv warp_ mesh_ > gpurun_out/pmc_render_traffic.txt 2>&1
grep -v "VGPR\|LDS_bytes" gpurun_out/pmc_render_traffic.txt
