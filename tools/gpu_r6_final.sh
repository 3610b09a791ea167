#!/bin/bash
# round 6, final-tree measurement: race screen, driver-style bench line (with the in-run PMC passes, video_wallclock, cross_attention_zero_tail), rocprofv3 kernel stats of the
# bench, PMC passes of the two MFMA kernels, one-GPU-as-one-rank CP emulation at cp = 1 / 2 / 4 / 8
mkdir -p gpurun_out/r6f
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python tools/race_screen.py > gpurun_out/r6f/race_screen.txt 2>&1; echo "race rc $?" >> gpurun_out/r6f/race_screen.txt
grep -c "^ok" gpurun_out/r6f/race_screen.txt; grep "DIFF\|RACE\|rc " gpurun_out/r6f/race_screen.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6f/bench_line.json 2> gpurun_out/r6f/bench_err.log; echo "bench rc $?"
python -c "
import json
d=json.loads(open('gpurun_out/r6f/bench_line.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_gemm']['achieved'], d['roofline_gemm']['frac'])
for c in d['roofline_gemm']['classes'][:8]: print('  ', c['epilogue'], c['M'], c['N'], c['K'], c['avg_ms'], c['achieved'])
print(d.get('video_wallclock')); print(d.get('cross_attention_zero_tail')); print(d.get('roofline_tokenizer',{}).get('encode'), d.get('roofline_tokenizer',{}).get('decode')); print(d.get('roofline_render'))"
bash tools/gpu_prof.sh r6_bench python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r6f/prof.log 2>&1; tail -2 gpurun_out/r6f/prof.log
cp gpurun_out/r6_bench_kernel_stats.csv gpurun_out/r6f/ 2>/dev/null; head -14 gpurun_out/r6f/r6_bench_kernel_stats.csv
bash tools/gpu_pmc.sh > gpurun_out/r6f/pmc.log 2>&1; cp gpurun_out/pmc_summary.txt gpurun_out/r6f/ 2>/dev/null; head -30 gpurun_out/r6f/pmc_summary.txt
timeout 900 python tools/cp_rank_emulate.py --cps 1,2,4,8 --configs "4,auto,local_first;2,auto,local_first;4,auto,gather_first;2,auto,gather_first;1,auto,gather_first" --out gpurun_out/r6f/r6_cp_rank_shapes.json > gpurun_out/r6f/cp_emulate.log 2>&1; echo "emulate rc=$?"; tail -22 gpurun_out/r6f/cp_emulate.log
