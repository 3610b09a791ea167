"""How much of a GEMM launch is epilogue: time the DiT output shapes at tiny K (main loop ~ nothing).
usage (GPU box): python tools/gemm_epilogue_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
for name, M, N, epi in [("qkv", 56320, 12288, 0), ("w1", 56320, 16384, 1), ("out/w2", 56320, 4096, 2)]:
    for K in (64, 4096):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        gate = torch.randn(1, N, device=dev).to(torch.bfloat16)
        res = torch.randn(M, N, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw = dict(gate=gate, residual=res) if epi == 2 else {}
        line = []
        for rnd in range(2):
            for variant in (0, 1):
                ops.set_option("gemm_wide_store", variant)
                ms = timeit(lambda: ops.gemm_nt(a, w, out=out, epilogue=epi, **kw), 5)
                line.append(f"wide={variant} {ms:.3f} ms")
        print(f"{name} N={N} K={K}: " + "  ".join(line) + f"   (C bytes {M * N * 2 / 1e9:.2f} GB)", flush=True)
        del a, w, gate, res, out
ops.set_option("gemm_wide_store", 1)
