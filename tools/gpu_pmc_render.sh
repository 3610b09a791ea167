#!/bin/bash
# SQ counter pass over the renderer kernels (tools/bench_render.py) -> gpurun_out/pmc_render/
mkdir -p gpurun_out/pmc_render
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
BENCH=${1:-tools/bench_render.py}  # tools/bench_render_single.py: the product configuration only
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $ROOTD/gpurun_out/pmc_render/a -o p -- python $ROOTD/$BENCH > $ROOTD/gpurun_out/pmc_render/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $ROOTD/gpurun_out/pmc_render/b -o p -- python $ROOTD/$BENCH > $ROOTD/gpurun_out/pmc_render/b.log 2>&1
cd $ROOTD
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_render/*/")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "warp_" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), vals in sorted(acc.items()):
        print(d.split("/")[-2], k, c, len(vals), "%.4g" % (sum(vals) / len(vals)))
PY
