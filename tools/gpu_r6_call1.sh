#!/bin/bash
# round 6, call 1: new parity tests (reference fixture inputs, default QKV chain at full size), one-GPU-as-one-rank emulation of cp = 1/2/4/8, bench line with video_wallclock
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_reference_fixtures_gpu.py -m gpu -q -s -x > gpurun_out/r6_c1_fixtures.log 2>&1; echo "fixtures rc=$?"
tail -15 gpurun_out/r6_c1_fixtures.log
python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -x -k "qkv or attention_launch" > gpurun_out/r6_c1_fullsize.log 2>&1; echo "fullsize rc=$?"
tail -5 gpurun_out/r6_c1_fullsize.log
python tools/cp_rank_emulate.py --out gpurun_out/r6_cp_rank_shapes.json > gpurun_out/r6_c1_cp_emulate.log 2>&1; echo "emulate rc=$?"
tail -30 gpurun_out/r6_c1_cp_emulate.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r6_c1_bench.json 2> gpurun_out/r6_c1_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_c1_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','video_wallclock')})
print(d.get('roofline_render'), d.get('roofline_tokenizer',{}).get('encode'))
PY
