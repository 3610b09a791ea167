#!/bin/bash
# round 5: renderer with the projection fused into the splat - tests, alternating timing A/B, traffic counters of both forms
mkdir -p gpurun_out/r5r
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_pipeline_gpu.py -q -x > gpurun_out/r5r/t_render.log 2>&1; tail -3 gpurun_out/r5r/t_render.log
for rep in 1 2 3; do for arm in 1 0; do G3_RENDER_FUSED_ARM=$arm G3_RENDER_ONLY_FG=1 timeout 120 python tools/bench_render_single.py 2>/dev/null | grep "^render"; done; done > gpurun_out/r5r/render_ab.txt
cat gpurun_out/r5r/render_ab.txt
cd /tmp
for arm in 1 0; do for c in FETCH_SIZE WRITE_SIZE; do
  G3_RENDER_FUSED_ARM=$arm G3_RENDER_ONLY_FG=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOTD/gpurun_out/r5r/pmc_fused$arm/$c -o p -- python $ROOTD/tools/bench_render_single.py > $ROOTD/gpurun_out/r5r/pmc_fused${arm}_$c.log 2>&1
done; done
cd $ROOTD
for arm in 1 0; do echo "== fused_projection=$arm"; python tools/pmc_summary.py gpurun_out/r5r/pmc_fused$arm gpurun_out/r5r/pmc_fused$arm.csv warp_ mesh_ 2>&1 | grep -v "VGPR\|LDS_bytes"; done > gpurun_out/r5r/render_traffic.txt
cat gpurun_out/r5r/render_traffic.txt
