#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -s -k "zero_tail" > gpurun_out/r6_c4_ztail.log 2>&1; echo "ztail rc=$?"; grep "^\[zero" gpurun_out/r6_c4_ztail.log | head -12; tail -3 gpurun_out/r6_c4_ztail.log
python -m pytest tests/test_dit_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x -s > gpurun_out/r6_c4_dit.log 2>&1; echo "dit rc=$?"; grep "zero-padded" gpurun_out/r6_c4_dit.log; tail -3 gpurun_out/r6_c4_dit.log
python -m pytest tests/test_reference_fixtures_gpu.py -m gpu -q -s > gpurun_out/r6_c4_fixtures.log 2>&1; echo "fixtures rc=$?"; grep "^\[" gpurun_out/r6_c4_fixtures.log; tail -3 gpurun_out/r6_c4_fixtures.log
python __graft_entry__.py --smoke 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 > gpurun_out/r6_c4_bench.json 2> gpurun_out/r6_c4_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_c4_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','cross_attention_zero_tail')})
print(d.get('video_wallclock',{}).get('value'), d.get('roofline',{}).get('frac'), d.get('roofline_gemm',{}).get('frac'))
PY
