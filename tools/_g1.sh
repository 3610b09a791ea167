export G3_WRITE_FULLSIZE_FIXTURE=gpurun_out/tokenizer_fullsize_samples.npz
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fullsize_gpu.py::test_tokenizer_full_clip_vs_fp32_oracle -x -q -s > gpurun_out/r4_tok_full.log 2>&1; echo "tokfull rc=$?"
tail -8 gpurun_out/r4_tok_full.log
timeout 700 python -m pytest tests/test_cp_gpu.py -x -q -k "survives or null_line or autotunes" -s > gpurun_out/r4_cp_guard.log 2>&1; echo "cpguard rc=$?"
tail -5 gpurun_out/r4_cp_guard.log
timeout 700 python -m pytest tests/test_dit_gpu.py tests/test_pipeline_gpu.py tests/test_render_gpu.py tests/test_tokenizer_gpu.py -x -q > gpurun_out/r4_sub.log 2>&1; echo "sub rc=$?"
tail -3 gpurun_out/r4_sub.log
for i in 1 2; do
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r4_bench_timers_$i.json 2>gpurun_out/err.log
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timers > gpurun_out/r4_bench_notimers_$i.json 2>>gpurun_out/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_bench_*timers_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('achieved'), (d.get('roofline_gemm') or {}).get('achieved'))
    except Exception as e: print(f, 'ERR', e)
PY
