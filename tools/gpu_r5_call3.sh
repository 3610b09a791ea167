#!/bin/bash
# round 5, GPU call 3: whole GPU suite with the deferred-epilogue GEMM as the default, race screen, driver-style bench line + kernel trace
mkdir -p gpurun_out/r5c3
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
ROOTD=$(pwd)
timeout 1200 python -m pytest tests -m gpu -x -q -s > gpurun_out/r5c3/t_all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c3/t_all.log
tail -4 gpurun_out/r5c3/t_all.log
timeout 600 python tools/race_screen.py --no-tokenizer > gpurun_out/r5c3/race_screen.txt 2>&1; echo "rc $?" >> gpurun_out/r5c3/race_screen.txt
grep -c "^ok" gpurun_out/r5c3/race_screen.txt; grep "DIFF\|RACE\|rc " gpurun_out/r5c3/race_screen.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5c3/bench_line.json 2> gpurun_out/r5c3/bench_err.log; echo "bench rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r5c3/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], json.dumps(d['roofline_gemm'])[:1500])"
G3_GEMM_DEFERRED=0 timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r5c3/bench_line_deferred0.json 2>> gpurun_out/r5c3/bench_err.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r5c3/bench_line_deferred1.json 2>> gpurun_out/r5c3/bench_err.log
python -c "
import json
for f in ('deferred0','deferred1'):
    d=json.load(open('gpurun_out/r5c3/bench_line_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline_gemm']['achieved'], d['roofline_gemm']['total_ms_per_step'])"
