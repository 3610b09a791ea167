"""QKV projection at the benchmark shape (M = 56320 x B, N = 12288, K = 4096): plain GEMM vs the fused per-head RMSNorm + RoPE (+ V^T) epilogue vs the
separate passes, interleaved in one process (A/B numbers are only comparable within one run on one box)."""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
S, D, H = 56320, 4096, 32
for B in (1, 2):
    g = torch.Generator(device=dev).manual_seed(1)
    h = torch.randn(S * B, D, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(3 * D, D, device=dev, generator=g) / math.sqrt(D)).to(torch.bfloat16)
    nq = torch.ones(128, device=dev, dtype=torch.bfloat16)
    ang = torch.rand(S, 128, device=dev, generator=g)
    cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
    vt = torch.zeros(B, H, 128, S, device=dev, dtype=torch.bfloat16)
    out = torch.empty(S * B, 3 * D, device=dev, dtype=torch.bfloat16)
    flops = 2.0 * S * B * 3 * D * D

    def plain():
        ops.gemm_nt(h, w, out=out)

    def fused_vt():
        ops.gemm_qk_norm_rope(h, w, D, D, nq, nq, cos, sin, S, B, out=out, vt=vt)

    def fused_novt():
        ops.gemm_qk_norm_rope(h, w, D, D, nq, nq, cos, sin, S, B, out=out)

    def fused_norope():
        ops.gemm_qk_norm_rope(h, w, D, D, nq, nq, None, None, S, B, out=out, vt=vt)

    def separate():
        ops.gemm_nt(h, w, out=out)
        ops.qk_rmsnorm_rope(out[:, :D], nq, cos, sin, S, B, H, out=out[:, :D])
        ops.qk_rmsnorm_rope(out[:, D:2 * D], nq, cos, sin, S, B, H, out=out[:, D:2 * D])
        ops.transpose_v(out[:, 2 * D:], S, B, H, out=vt)

    cases = [("plain gemm", plain), ("fused norm+rope+vt", fused_vt), ("fused norm+rope (v plain)", fused_novt), ("fused norm+vt (no rope)", fused_norope),
             ("gemm + 2 norm passes + transpose", separate)]
    for _ in range(2):
        for name, fn in cases:
            fn()
    torch.cuda.synchronize()
    res = {n: [] for n, _ in cases}
    for rep in range(4):
        for name, fn in cases:
            tm = ops.HipTimer()
            tm.start()
            fn()
            tm.stop()
            res[name].append(tm.elapsed_ms())
    for name, _ in cases:
        ms = sorted(res[name])[len(res[name]) // 2]
        print(f"B={B} {name:36s} {ms:8.3f} ms  {flops / ms / 1e9:7.0f} TFLOP/s (GEMM flops only)", flush=True)

# cross-attention q projection (N = 4096, per-head RMSNorm only)
for B in (2,):
    g = torch.Generator(device=dev).manual_seed(2)
    h = torch.randn(S * B, D, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(D, D, device=dev, generator=g) / math.sqrt(D)).to(torch.bfloat16)
    nq = torch.ones(128, device=dev, dtype=torch.bfloat16)
    out = torch.empty(S * B, D, device=dev, dtype=torch.bfloat16)
    flops = 2.0 * S * B * D * D
    cases = [("ca_q plain gemm", lambda: ops.gemm_nt(h, w, out=out)),
             ("ca_q fused norm", lambda: ops.gemm_qk_norm_rope(h, w, D, 0, nq, None, None, None, S, B, out=out)),
             ("ca_q gemm + norm pass", lambda: (ops.gemm_nt(h, w, out=out), ops.qk_rmsnorm_rope(out, nq, None, None, S, B, H, out=out)))]
    for _ in range(2):
        for name, fn in cases:
            fn()
    torch.cuda.synchronize()
    res = {n: [] for n, _ in cases}
    for rep in range(4):
        for name, fn in cases:
            tm = ops.HipTimer()
            tm.start()
            fn()
            tm.stop()
            res[name].append(tm.elapsed_ms())
    for name, _ in cases:
        ms = sorted(res[name])[len(res[name]) // 2]
        print(f"B={B} {name:36s} {ms:8.3f} ms  {flops / ms / 1e9:7.0f} TFLOP/s (GEMM flops only)", flush=True)
