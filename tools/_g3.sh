mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_tokenizer_gpu.py -x -q -s -k "flash or spatial" > gpurun_out/r4_flash_tests2.log 2>&1; echo "flash tests rc=$?"
grep -E "flash spatial|spatial attn|three-kernel|passed|failed|Error|error" gpurun_out/r4_flash_tests2.log | tail -15
for f in 1 0 1 0; do G3_TOK_FLASH_ATTN=$f timeout 300 python tools/bench_tokenizer.py 2>&1 | grep "pingpong=2" | sed "s/^/flash=$f /"; done | tee gpurun_out/r4_tok_flash_ab2.txt
