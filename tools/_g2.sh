mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_tokenizer_gpu.py -x -q -s -k "flash or spatial or golden or wide" > gpurun_out/r4_flash_tests.log 2>&1; echo "flash tests rc=$?"
grep -E "flash spatial|spatial attn|three-kernel|passed|failed|Error|error" gpurun_out/r4_flash_tests.log | tail -15
for f in 1 0 1 0; do G3_TOK_FLASH_ATTN=$f timeout 300 python tools/bench_tokenizer.py 2>&1 | grep "pingpong=2" | tail -2 | sed "s/^/flash=$f /"; done | tee gpurun_out/r4_tok_flash_ab.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py::test_tokenizer_full_clip_vs_fp32_oracle tests/test_reference_precision_gpu.py -x -q -s > gpurun_out/r4_refprec.log 2>&1; echo "refprec rc=$?"
grep -E "^\[|^  |rel-L2|passed|failed|Error" gpurun_out/r4_refprec.log | cut -c1-400 | tail -30
timeout 900 python -m pytest tests/test_cp_gpu.py -x -q -k "survives or null_line" > gpurun_out/r4_cp_guard2.log 2>&1; echo "cpguard rc=$?"; tail -3 gpurun_out/r4_cp_guard2.log
timeout 900 python -m pytest tests/test_cli_gpu.py -x -q -s -k "serving or checkpoint_layout" > gpurun_out/r4_cli.log 2>&1; echo "cli rc=$?"
grep -E "psnr|PSNR|passed|failed|Error" gpurun_out/r4_cli.log | cut -c1-300 | tail -12
