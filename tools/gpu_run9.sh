#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name: $*"; timeout 900 "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
TAILN=6 run tok_small python tools/bench_tokenizer.py 17 704 1280
TAILN=6 run tok_full python tools/bench_tokenizer.py
ROOTD=$(pwd); cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/prof_tok -o tok -- python $ROOTD/tools/bench_tokenizer.py > $ROOTD/gpurun_out/prof_tok.log 2>&1
echo "rc=$? rocprof"; cd $ROOTD; ls gpurun_out/prof_tok | head
