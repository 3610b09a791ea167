"""A few launches of the two MFMA kernels at the benchmark shapes, for rocprofv3 --pmc passes.
(full-size self-attention: S=56320, 32 heads; the four block GEMMs)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
S, H = 56320, 32
q = torch.randn(S, H * 128, device=dev).to(torch.bfloat16)
k = torch.randn(S, H * 128, device=dev).to(torch.bfloat16)
v = torch.randn(S, H * 128, device=dev).to(torch.bfloat16)
vt = ops.transpose_v(v, S, 1, H)
out = torch.empty_like(q)
for _ in range(2):
    ops.flash_attn(q, k, vt, S, S, 1, H, out=out)
torch.cuda.synchronize()
del q, k, v, vt, out
for (M, N, K, epi) in [(56320, 12288, 4096, 0), (56320, 16384, 4096, 1), (56320, 4096, 16384, 2)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    gate = torch.randn(1, N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(gate=gate, residual=res) if epi == 2 else {}
    for _ in range(2):
        ops.gemm_nt(a, w, out=o, epilogue=epi, **kw)
    torch.cuda.synchronize()
    del a, w, gate, res, o
print("done")
