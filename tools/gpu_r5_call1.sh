#!/bin/bash
# round 5, GPU call 1: whole GPU suite on the hygiene tree, then the attention K / V^T re-read experiment (timing + FETCH_SIZE / SQ / clock passes)
mkdir -p gpurun_out/r5c1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r5c1/t_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c1/t_all.log
tail -3 gpurun_out/r5c1/t_all.log
timeout 300 python tools/attn_kv_reread_ab.py > gpurun_out/r5c1/reread_timing.txt 2>&1
cat gpurun_out/r5c1/reread_timing.txt
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  G3_REREAD_PMC=1 timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $ROOTD/gpurun_out/r5c1/pmc_$tag -o p -- python $ROOTD/tools/attn_kv_reread_ab.py > $ROOTD/gpurun_out/r5c1/pmc_$tag.log 2>&1
done
cd $ROOTD
python - <<'PY' > gpurun_out/r5c1/reread_pmc.txt 2>&1
import csv, glob, collections
rows = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r5c1/pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "flash_attn" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
arms = ["A", "A8", "B", "B2", "C"]
for c, v in rows.items():
    v.sort()
    print(c, "dispatches", len(v))
    for i, (d, val, ns) in enumerate(v):
        print(f"   arm {arms[(i // 2) % 5]} launch {i % 2}: {val:.6g}   ({ns / 1e6:.3f} ms under the profiler)")
PY
cat gpurun_out/r5c1/reread_pmc.txt | head -80
