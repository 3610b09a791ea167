#!/bin/bash
# Round-3 measurement pass: GPU test-suite (with the printed parity numbers), smoke, bench line, kernel stats of the bench, single-configuration
# kernel stats of tokenizer / renderer, per-op tokenizer breakdown. Everything lands in gpurun_out/.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
T0=$(date +%s)
run() { name=$1; shift; echo "=== $name: $* [t+$(( $(date +%s) - T0 ))s]"; timeout ${TMO:-1500} "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
TAILN=25 run t_all python -m pytest tests -m gpu -q --tb=short -rP -x
grep -h "^\[" gpurun_out/t_all.log | sort | uniq > gpurun_out/parity_prints.txt; wc -l gpurun_out/parity_prints.txt
TAILN=3 run smoke python __graft_entry__.py --smoke
TAILN=2 run bench python bench.py --steps ${STEPS:-3} --warmup 1
TAILN=40 run tok_breakdown python tools/tokenizer_breakdown.py
if [ -z "$SKIP_PROF" ]; then
  bash tools/gpu_prof.sh r3_bench python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras | tail -14
  bash tools/gpu_prof.sh r3_tokenizer python tools/bench_tokenizer_single.py | tail -16
  bash tools/gpu_prof.sh r3_render python tools/bench_render_single.py | tail -10
fi
echo "=== done [t+$(( $(date +%s) - T0 ))s]"
