#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration passes (tools/pmc_calibrate.py) -> gpurun_out/pmc_calibration.txt
mkdir -p gpurun_out/pmc_cal
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOTD/gpurun_out/pmc_cal/$c -o p -- python $ROOTD/tools/pmc_calibrate.py > $ROOTD/gpurun_out/pmc_cal/$c.log 2>&1
  echo "rc=$? ($c)"
done
cd $ROOTD
{ python tools/pmc_calibrate.py --expected; python tools/pmc_summary.py gpurun_out/pmc_cal gpurun_out/pmc_calibration.csv add_inplace reliable_mask unproject_kernel | grep -v "VGPR\|LDS_bytes"; } > gpurun_out/pmc_calibration.txt 2>&1
cat gpurun_out/pmc_calibration.txt
