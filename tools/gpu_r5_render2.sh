#!/bin/bash
# round 5: renderer, single-writer form (extent pre-pass + texels resolved inside the splat) - tests, alternating timing A/B, traffic counters of both forms
mkdir -p gpurun_out/r5x
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_pipeline_gpu.py -q -x -s > gpurun_out/r5x/t_render.log 2>&1; tail -3 gpurun_out/r5x/t_render.log; grep "hand-over" gpurun_out/r5x/t_render.log
# arms: x = single-writer form; f = default (full extents, accumulator read per texel); o = round-4 form (clamped extents, accumulator read per dirty item)
for rep in 1 2 3; do for arm in x f o; do
  case $arm in x) E="G3_RENDER_EXCLUSIVE=1";; f) E="G3_RENDER_EXCLUSIVE=0";; o) E="G3_RENDER_EXCLUSIVE=0 G3_RENDER_FULL_EXTENT=0";; esac
  echo -n "arm=$arm "; env $E G3_RENDER_ONLY_FG=1 timeout 120 python tools/bench_render_single.py 2>/dev/null | grep "^render"; done; done > gpurun_out/r5x/render_ab.txt
cat gpurun_out/r5x/render_ab.txt
cd /tmp
for arm in 1 0 o; do for c in FETCH_SIZE WRITE_SIZE; do
  E="G3_RENDER_EXCLUSIVE=$arm"; [ $arm = o ] && E="G3_RENDER_EXCLUSIVE=0 G3_RENDER_FULL_EXTENT=0"
  env $E G3_RENDER_ONLY_FG=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOTD/gpurun_out/r5x/pmc_excl$arm/$c -o p -- python $ROOTD/tools/bench_render_single.py > $ROOTD/gpurun_out/r5x/pmc_excl${arm}_$c.log 2>&1
done; done
cd $ROOTD
for arm in 1 0 o; do echo "== render_exclusive=$arm (o: exclusive 0 + full_extent 0 = the round-4 form)"; python tools/pmc_summary.py gpurun_out/r5x/pmc_excl$arm gpurun_out/r5x/pmc_excl$arm.csv warp_ mesh_ 2>&1 | grep -v "VGPR\|LDS_bytes"; done > gpurun_out/r5x/render_traffic.txt
cat gpurun_out/r5x/render_traffic.txt
