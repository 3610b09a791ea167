"""Cross-attention launch of the step (Sq = 56 320, Skv = 512 T5 tokens, B = 2, H = 32; 9.45e11 FLOP) on every attention kernel variant, interleaved in one process
(the default v3 kernel runs it at 0.27 of the bf16 MFMA peak). usage (GPU box): python tools/cross_attn_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
S, M, B, H = 56320, 512, 2, 32
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(S * B, H * 128, device=dev, generator=g).to(torch.bfloat16)
k = torch.randn(M * B, H * 128, device=dev, generator=g).to(torch.bfloat16)
k[64 * B:] = 0  # zero-padded T5 tokens (general_dit.py:407-410: they stay in the softmax denominator)
v = torch.randn(M * B, H * 128, device=dev, generator=g).to(torch.bfloat16)
vt = ops.transpose_v(v, M, B, H)
out = torch.empty_like(q)
flops = 4.0 * S * M * 128 * H * B
lib = _lib.load()
ref = None
variants = [int(a) for a in sys.argv[1:]] or [4, 3, 5, 9, 10, 11]
res = {}
for rep in range(3):
    for var in variants:
        name = lib.g3_flash_attn_kernel_name_ex(S, M, B, H, var).decode()
        try:
            ops.flash_attn(q, k, vt, S, M, B, H, out=out, variant=var)
            torch.cuda.synchronize()
            tm = ops.HipTimer()
            tm.start()
            for _ in range(5):
                ops.flash_attn(q, k, vt, S, M, B, H, out=out, variant=var)
            tm.stop()
            ms = tm.elapsed_ms() / 5
        except Exception as e:  # noqa: BLE001
            print(f"variant {var} ({name}): {e!r}")
            continue
        if ref is None:
            ref = out.clone()
        d = float((out.float() - ref.float()).norm() / ref.float().norm())
        res.setdefault(var, []).append(ms)
        print(f"variant {var:2d} {name:48s} {ms:.3f} ms  {flops / ms / 1e9:7.0f} TF/s = {flops / ms / 1e9 / 2500:.3f}   rel-L2 vs first variant {d:.2e}", flush=True)
