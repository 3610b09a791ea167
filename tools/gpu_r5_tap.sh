#!/bin/bash
# round 5: tap change in the MFMA gaps vs between statements, same box
mkdir -p gpurun_out/r5t
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 python tools/conv_tap_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5t/conv_tap_ab.txt
for i in 1 2; do for arm in 1 2; do echo -n "conv_w4=$arm "; G3_CONV_W4=$arm timeout 200 python tools/bench_tokenizer_single.py 2>&1 | grep "^tokenizer" | tr '\n' ' '; echo; done; done | tee gpurun_out/r5t/tokenizer_ab.txt
