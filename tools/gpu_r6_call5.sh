#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -s -k "zero_tail or q_norm_inside" > gpurun_out/r6_c5_cross.log 2>&1; echo "cross rc=$?"; grep "^\[cross" gpurun_out/r6_c5_cross.log | head -12; tail -3 gpurun_out/r6_c5_cross.log
python -m pytest tests/test_dit_gpu.py tests/test_reference_fixtures_gpu.py -m gpu -q -x -s > gpurun_out/r6_c5_dit.log 2>&1; echo "dit+fixtures rc=$?"; grep "^\[render\|^\[tokenizer on\|zero-padded" gpurun_out/r6_c5_dit.log; tail -3 gpurun_out/r6_c5_dit.log
for i in 1 2; do
G3_CROSS_Q_NORM_IN_ATTENTION=0 python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r6_c5_bench_sep_$i.json 2>/dev/null; echo "bench sep rc=$?"
python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r6_c5_bench_fused_$i.json 2>/dev/null; echo "bench fused rc=$?"
done
python - <<'PY'
import json
for n in ("sep_1","fused_1","sep_2","fused_2"):
    d=json.loads(open(f'gpurun_out/r6_c5_bench_{n}.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_gemm']['total_ms_per_step'])
PY
