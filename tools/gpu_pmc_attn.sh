#!/bin/bash
# SQ counter passes over the self-attention kernel variants $@ (default "4 9") -> gpurun_out/pmc_attn_v<variant>.csv
mkdir -p gpurun_out/pmc_attn
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
cd /tmp
for v in ${@:-4 9}; do
  G3_ATTN_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $ROOTD/gpurun_out/pmc_attn/v$v -o p -- python $ROOTD/tools/pmc_attn.py > $ROOTD/gpurun_out/pmc_attn/v$v.log 2>&1
  G3_ATTN_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $ROOTD/gpurun_out/pmc_attn/w$v -o p -- python $ROOTD/tools/pmc_attn.py > $ROOTD/gpurun_out/pmc_attn/w$v.log 2>&1
done
cd $ROOTD
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_attn/*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "flash_attn" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), vals in sorted(acc.items()):
        print(d.split("/")[-2], k, c, len(vals), sum(vals) / len(vals))
PY
