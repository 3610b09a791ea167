#!/bin/bash
# rocprofv3 PMC passes (separate runs, --kernel-trace only, as the microarch guide prescribes) over the two MFMA kernels at the
# benchmark shapes (tools/pmc_probe.py) -> gpurun_out/pmc/<pass>/ + gpurun_out/pmc_summary.csv
mkdir -p gpurun_out/pmc
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
cd /tmp
pass() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOTD/gpurun_out/pmc/$name -o p -- python $ROOTD/tools/pmc_probe.py > $ROOTD/gpurun_out/pmc/$name.log 2>&1; echo "rc=$? ($name)"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass grbm GRBM_GUI_ACTIVE
cd $ROOTD
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc_summary.csv flash_attn gemm_bf16 spatial_attn > gpurun_out/pmc_summary.txt 2>&1
cat gpurun_out/pmc_summary.txt | head -80
