#!/bin/bash
mkdir -p gpurun_out/pmc
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
cd /tmp
rocprofv3 -L > $ROOTD/gpurun_out/pmc/counters_list.txt 2>&1
grep -c . $ROOTD/gpurun_out/pmc/counters_list.txt
pass() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOTD/gpurun_out/pmc/$name -o p -- python $ROOTD/tools/pmc_probe.py > $ROOTD/gpurun_out/pmc/$name.log 2>&1; echo "rc=$? ($name)"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass grbm GRBM_GUI_ACTIVE
cd $ROOTD
find gpurun_out/pmc -name "*.csv" | head -20
for f in $(find gpurun_out/pmc -name "*counter_collection.csv"); do echo "== $f"; head -3 $f | cut -c1-400; done
