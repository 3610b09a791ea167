"""Per-rank kernel efficiency at context-parallel degree cp (one GPU emulates one rank's shapes): self-attention with
Sq = N/cp query rows against all N keys (per head group, as parallel.ContextParallelAttention launches it) and the block
GEMMs with M = N/cp rows.  usage (GPU box): python tools/cp_shapes_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
N, D = 56320, 4096
for cp in (1, 2, 4, 8):
    Sq = N // cp
    line = [f"cp={cp} Sq={Sq}:"]
    for H in (8, 32):
        q = torch.randn(Sq, H * 128, device=dev).to(torch.bfloat16)
        k = torch.randn(N, H * 128, device=dev).to(torch.bfloat16)
        v = torch.randn(N, H * 128, device=dev).to(torch.bfloat16)
        vt = ops.transpose_v(v, N, 1, H)
        out = torch.empty_like(q)
        ms = timeit(lambda: ops.flash_attn(q, k, vt, Sq, N, 1, H, out=out), 3)
        line.append(f"attn H={H} {ms:.3f} ms {4.0 * Sq * N * 128 * H / ms / 1e9:.0f} TF")
        del q, k, v, vt, out
    for name, Nn, K, epi in (("qkv", 12288, 4096, 0), ("out", 4096, 4096, 2), ("w1", 16384, 4096, 1), ("w2", 4096, 16384, 2)):
        a = torch.randn(Sq, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(Nn, K, device=dev) * 0.02).to(torch.bfloat16)
        gate = torch.randn(1, Nn, device=dev).to(torch.bfloat16)
        res = torch.randn(Sq, Nn, device=dev).to(torch.bfloat16)
        out = torch.empty(Sq, Nn, device=dev, dtype=torch.bfloat16)
        kw = dict(gate=gate, residual=res) if epi == 2 else {}
        ms = timeit(lambda: ops.gemm_nt(a, w, out=out, epilogue=epi, **kw), 5)
        line.append(f"{name} {ms:.3f} ms {2.0 * Sq * Nn * K / ms / 1e9:.0f} TF")
        del a, w, gate, res, out
    print("  ".join(line), flush=True)
