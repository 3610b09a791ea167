#!/bin/bash
# round 5, final-tree validation: whole GPU suite, smoke, race screen, driver-style bench line, same-box A/B vs the non-deferred GEMM, rocprofv3 kernel stats of the bench, PMC passes
mkdir -p gpurun_out/r5f
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
timeout 1200 python -m pytest tests -m gpu -x -q -s > gpurun_out/r5f/t_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5f/t_all.log
tail -3 gpurun_out/r5f/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5f/smoke.log 2>&1; tail -2 gpurun_out/r5f/smoke.log
timeout 600 python tools/race_screen.py > gpurun_out/r5f/race_screen.txt 2>&1; echo "rc $?" >> gpurun_out/r5f/race_screen.txt
grep -c "^ok" gpurun_out/r5f/race_screen.txt; grep "DIFF\|RACE\|rc " gpurun_out/r5f/race_screen.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5f/bench_line.json 2> gpurun_out/r5f/bench_err.log; echo "bench rc $?"
G3_GEMM_DEFERRED=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r5f/bench_line_deferred0.json 2>> gpurun_out/r5f/bench_err.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r5f/bench_line_deferred1.json 2>> gpurun_out/r5f/bench_err.log
python -c "
import json
d=json.load(open('gpurun_out/r5f/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_gemm']['achieved'], d['roofline_gemm']['frac'])
for c in d['roofline_gemm']['classes'][:6]: print('  ', c['epilogue'], c['N'], c['K'], c['avg_ms'], c['achieved'])
for f in ('deferred0','deferred1'):
    d=json.load(open('gpurun_out/r5f/bench_line_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline_gemm']['achieved'], d['roofline_gemm']['total_ms_per_step'])"
bash tools/gpu_prof.sh r5_bench python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r5f/prof.log 2>&1; tail -2 gpurun_out/r5f/prof.log
cp gpurun_out/r5_bench_kernel_stats.csv gpurun_out/r5f/ 2>/dev/null; head -8 gpurun_out/r5f/r5_bench_kernel_stats.csv
bash tools/gpu_pmc.sh > gpurun_out/r5f/pmc.log 2>&1; cp gpurun_out/pmc_summary.txt gpurun_out/r5f/ 2>/dev/null; head -40 gpurun_out/r5f/pmc_summary.txt
