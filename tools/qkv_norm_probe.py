"""The q | k norm + RoPE passes and the V transpose behind the QKV projection at the benchmark shape (S = 56 320, B = 2, H = 32), old form (one 8-lane group per
(row, head), q and k in two launches) vs the round-6 octet form (one launch over q | k), interleaved. usage (GPU box): python tools/qkv_norm_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import _lib, ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
S, B, H = 56320, 2, 32
D = H * 128
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(S * B, 3 * D, device=dev, generator=g).to(torch.bfloat16)
wq = (torch.rand(128, device=dev, generator=g) + 0.5).to(torch.bfloat16)
wk = (torch.rand(128, device=dev, generator=g) + 0.5).to(torch.bfloat16)
ang = torch.rand(S, 64, device=dev, generator=g) * 6.28
ang = torch.cat([ang, ang], -1)
cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
vt = torch.zeros(B, H, 128, ops.ceil_to(S, 64), device=dev, dtype=torch.bfloat16)
lib = _lib.load()
payload = 2 * (S * B * 2 * D * 2)  # q | k read + written


def old():
    lib.g3_set_option(b"norm_octets", 0)
    ops.qk_rmsnorm_rope(qkv[:, :D], wq, cos, sin, S, B, H, out=qkv[:, :D])
    ops.qk_rmsnorm_rope(qkv[:, D:2 * D], wk, cos, sin, S, B, H, out=qkv[:, D:2 * D])
    lib.g3_set_option(b"norm_octets", 1)


def new2():
    ops.qk_rmsnorm_rope(qkv[:, :D], wq, cos, sin, S, B, H, out=qkv[:, :D])
    ops.qk_rmsnorm_rope(qkv[:, D:2 * D], wk, cos, sin, S, B, H, out=qkv[:, D:2 * D])


def pair():
    ops.qk_rmsnorm_rope_pair(qkv[:, :2 * D], wq, H, wk, H, cos, sin, S, B)


def crossq():  # the cross-attention Q norm (no RoPE), packed [S*B, D]
    ops.qk_rmsnorm_rope(qkv[:, :D], wq, None, None, S, B, H, out=qkv[:, :D])


def crossq_old():
    lib.g3_set_option(b"norm_octets", 0)
    ops.qk_rmsnorm_rope(qkv[:, :D], wq, None, None, S, B, H, out=qkv[:, :D])
    lib.g3_set_option(b"norm_octets", 1)


for rnd in range(3):
    line = []
    for name, fn, by in (("q,k general form (2 launches)", old, payload), ("q,k octet form (2 launches)", new2, payload), ("q|k octet form (1 launch)", pair, payload),
                         ("cross q general", crossq_old, payload // 2), ("cross q octet", crossq, payload // 2),
                         ("transpose_v", lambda: ops.transpose_v(qkv[:, 2 * D:], S, B, H, out=vt), payload // 2)):
        ms = timeit(fn, 5)
        line.append(f"[{name}: {ms:.3f} ms = {by / ms / 1e9:.2f} TB/s]")
    print("  ".join(line), flush=True)

# ---- the QKV projection itself: fused N = 3D + transpose pass  vs  q | k (N = 2D) + V^T by operand swap (one launch per batch item)
h = torch.randn(S * B, D, device=dev, generator=g).to(torch.bfloat16)
w = (torch.randn(3 * D, D, device=dev, generator=g) * 0.02).to(torch.bfloat16)
out3 = torch.empty(S * B, 3 * D, device=dev, dtype=torch.bfloat16)
out2 = torch.empty(S * B, 2 * D, device=dev, dtype=torch.bfloat16)
hv = h.view(S, B, D)


def fused3():
    ops.gemm_nt(h, w, out=out3)
    ops.transpose_v(out3[:, 2 * D:], S, B, H, out=vt)


def swap():
    ops.gemm_nt(h, w[:2 * D], out=out2)
    for b in range(B):
        ops.gemm_nt(w[2 * D:], hv[:, b], out=vt[b].view(D, -1)[:, :S])


def vt_only():
    for b in range(B):
        ops.gemm_nt(w[2 * D:], hv[:, b], out=vt[b].view(D, -1)[:, :S])


fl3 = 2.0 * S * B * 3 * D * D
for rnd in range(3):
    a, b_, c = timeit(fused3, 4), timeit(swap, 4), timeit(vt_only, 4)
    print(f"[QKV N=12288 + transpose_v: {a:.3f} ms]  [q|k N=8192 + 2 x V^T by operand swap: {b_:.3f} ms]  [the 2 V^T launches alone: {c:.3f} ms = {fl3 / 3 / c / 1e9:.0f} TF/s]", flush=True)
