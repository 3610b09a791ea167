#!/bin/bash
# Round-4 full check on the GPU box: whole GPU test-suite, smoke, race screen (incl. the full-size tokenizer), default-style bench, rocprofv3 kernel
# stats of the bench and of the tokenizer / renderer single-configuration runs, PMC passes (MFMA kernels + renderer traffic).
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; timeout 1500 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
TAILN=6 run r4_t_all python -m pytest tests -m gpu -q --tb=short -s
TAILN=3 run r4_smoke python __graft_entry__.py --smoke
TAILN=12 run r4_race python tools/race_screen.py
TAILN=2 run r4_bench python bench.py --steps 10 --warmup 3
ROOTD=$(pwd); cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/r4_prof_bench -o bench -- python $ROOTD/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $ROOTD/gpurun_out/r4_prof_bench.log 2>&1; echo "rc=$? rocprof bench"
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/r4_prof_tok -o tok -- python $ROOTD/tools/bench_tokenizer_single.py > $ROOTD/gpurun_out/r4_prof_tok.log 2>&1; echo "rc=$? rocprof tokenizer"
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/r4_prof_render -o render -- python $ROOTD/tools/bench_render_single.py > $ROOTD/gpurun_out/r4_prof_render.log 2>&1; echo "rc=$? rocprof render"
cd $ROOTD
bash tools/gpu_pmc.sh > gpurun_out/r4_pmc.log 2>&1; echo "rc=$? pmc"; tail -5 gpurun_out/r4_pmc.log
bash tools/gpu_pmc_render_traffic.sh > gpurun_out/r4_pmc_render.log 2>&1; echo "rc=$? pmc render"
find gpurun_out -name "*.db" -size +8M -delete 2>/dev/null
ls gpurun_out/r4_prof_bench gpurun_out/r4_prof_tok 2>/dev/null | head
