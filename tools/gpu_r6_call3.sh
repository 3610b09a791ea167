#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/render_fixture_diag.py > gpurun_out/r6_c3_render_diag.log 2>&1; echo "diag rc=$?"; tail -12 gpurun_out/r6_c3_render_diag.log
python tools/gemm_split_probe.py > gpurun_out/r6_c3_gemm_split.log 2>&1; echo "split rc=$?"; cat gpurun_out/r6_c3_gemm_split.log | tail -16
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "operand_swap" > gpurun_out/r6_c3_swap.log 2>&1; echo "swap tests rc=$?"; tail -3 gpurun_out/r6_c3_swap.log
for i in 1 2; do
G3_V_OPERAND_SWAP=0 python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r6_c3_bench_noswap_$i.json 2>/dev/null; echo "bench noswap rc=$?"
python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r6_c3_bench_swap_$i.json 2>/dev/null; echo "bench swap rc=$?"
done
python - <<'PY'
import json
for n in ("noswap_1","swap_1","noswap_2","swap_2"):
    d=json.loads(open(f'gpurun_out/r6_c3_bench_{n}.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_gemm']['total_ms_per_step'], [(c['N'],c['K'],c['M'],c['avg_ms'],c['achieved']) for c in d['roofline_gemm']['classes']])
PY
