#!/bin/bash
# round 6, call 2: new norm kernels (tests), renderer fixture diagnostic, cross-attention variants, MLP-down cache-policy / prefetch A/B, CP emulation with the fused-QKV local_first path
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "qk_rmsnorm or qk_norm" > gpurun_out/r6_c2_norm.log 2>&1; echo "norm tests rc=$?"; tail -3 gpurun_out/r6_c2_norm.log
python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -x -k "qkv" > gpurun_out/r6_c2_fullsize.log 2>&1; echo "fullsize rc=$?"; tail -3 gpurun_out/r6_c2_fullsize.log
python -m pytest tests/test_dit_gpu.py tests/test_cp_gpu.py -m gpu -q -x > gpurun_out/r6_c2_dit.log 2>&1; echo "dit/cp rc=$?"; tail -3 gpurun_out/r6_c2_dit.log
python tools/render_fixture_diag.py > gpurun_out/r6_c2_render_diag.log 2>&1; echo "diag rc=$?"; tail -25 gpurun_out/r6_c2_render_diag.log
python -m pytest tests/test_reference_fixtures_gpu.py -m gpu -q -s > gpurun_out/r6_c2_fixtures.log 2>&1; echo "fixtures rc=$?"; grep "^\[" gpurun_out/r6_c2_fixtures.log; tail -3 gpurun_out/r6_c2_fixtures.log
python tools/cross_attn_probe.py > gpurun_out/r6_c2_cross_attn.log 2>&1; echo "cross rc=$?"; tail -8 gpurun_out/r6_c2_cross_attn.log
G3_W4E_AB_VARIANTS="ntF=-DG3_AB_GW4E_CPOL=1;ntS=-DG3_AB_GW4E_CPOL=2;pf3=-DG3_AB_GW4E_PF=3" G3_W4E_AB_SHAPES="w2:112640:4096:16384:2;out:112640:4096:4096:2;w2cp8:14080:4096:16384:2;outcp8:14080:4096:4096:2" python tools/gemm_w4e_ab.py > gpurun_out/r6_c2_mlp_down_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r6_c2_mlp_down_ab.log | tail -24
python tools/qkv_norm_probe.py > gpurun_out/r6_c2_qkv_norm_probe.log 2>&1; echo "probe rc=$?"; cat gpurun_out/r6_c2_qkv_norm_probe.log | tail
python tools/cp_rank_emulate.py --cps 1,8 --configs "4,auto,local_first;2,auto,local_first;4,auto,gather_first;2,auto,gather_first;1,auto,gather_first" --out gpurun_out/r6_cp_rank_shapes_c2.json > gpurun_out/r6_c2_cp_emulate.log 2>&1; echo "emulate rc=$?"; tail -16 gpurun_out/r6_c2_cp_emulate.log
