"""Does a block GEMM run faster as two launches over halves of its weight rows? (round 6: the QKV projection as q|k (N = 8192) + V^T ran 8 % faster than as one
N = 12 288 launch, tools/qkv_norm_probe.py.) Same operand / output buffers, interleaved. usage (GPU box): python tools/gemm_split_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
M = 112640
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K, epi in (("mlp-up", 16384, 4096, 1), ("q|k", 8192, 4096, 0), ("qkv", 12288, 4096, 0), ("mlp-down", 4096, 16384, 2), ("out", 4096, 4096, 2)):
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    gate = torch.randn(2, N, device=dev, generator=g).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)

    def run(parts, mparts=1):
        n = N // parts
        m = M // mparts
        for j in range(mparts):
            for i in range(parts):
                kw = dict(gate=gate[:, i * n:(i + 1) * n], residual=res[j * m:(j + 1) * m, i * n:(i + 1) * n]) if epi == 2 else {}
                ops.gemm_nt(a[j * m:(j + 1) * m], w[i * n:(i + 1) * n], out=out[j * m:(j + 1) * m, i * n:(i + 1) * n], epilogue=epi, **kw)

    fl = 2.0 * M * N * K
    for rnd in range(3):
        line = []
        for label, parts, mparts in (("1 launch", 1, 1), ("2 x N/2", 2, 1), ("4 x N/4", 4, 1), ("2 x M/2", 1, 2)):
            if (N // parts) % 256 or (M // mparts) % 256:
                continue
            ms = timeit(lambda: run(parts, mparts), 4)
            line.append(f"[{label}: {ms:.3f} ms {fl / ms / 1e9:.0f} TF]")
        print(f"{name} {M}x{N}x{K} epi{epi}: " + "  ".join(line), flush=True)
    del a, w, out, gate, res
