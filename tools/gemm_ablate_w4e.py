"""Timing ablation of the deferred-epilogue GEMM (csrc/gemm_w4e.hpp): what does hiding the epilogue in the next tile's K loop still cost?
Build the ablated copies HERE before gpurun:   python tools/gemm_ablate_w4e.py --build
GPU box:                                       python tools/gemm_ablate_w4e.py
Bits (G3_AB_GW4E_ABLATE; results are garbage): 1 no epilogue arithmetic (the GELU / gate VALU in the MFMA gaps), 2 no stores, 4 no epilogue LDS traffic and residual
LDS-DMA pieces, 8 no drain. 15 = a bare persistent K stream: what the epilogue costs in total."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
CASES = [int(x) for x in (os.environ.get("G3_ABLATE_CASES") or "1,2,4,8,15").split(",")]

if "--build" in sys.argv:
    from gen3c_amd import build
    for bits in CASES:
        print(build.build(extra_flags=(f"-DG3_AB_GW4E_ABLATE={bits}",), suffix=f"_eabl{bits}", force=True), flush=True)
    sys.exit(0)

import torch  # noqa: E402
from gen3c_amd import _lib  # noqa: E402
from tools.microbench import timeit  # noqa: E402

libs = [("plain-w4", _lib.load(), 0), ("deferred", _lib.load(), 1)]
for bits in CASES:
    f = ROOT / "gen3c_amd" / "lib" / f"libgen3c_hip_eabl{bits}.so"
    if not f.exists():
        continue
    lib = C.CDLL(str(f))
    for name, argtypes in _lib.SIGNATURES.items():
        getattr(lib, name).argtypes = argtypes
    libs.append((f"ablate={bits}", lib, 1))
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = 2
for (nm, M, N, K, epi) in [("qkv", 56320 * B, 12288, 4096, 0), ("out", 56320 * B, 4096, 4096, 2), ("w1", 56320 * B, 16384, 4096, 1), ("w2", 56320 * B, 4096, 16384, 2)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    gate = torch.randn(B, N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for rnd in range(2):
        line = []
        for (label, lib, deferred) in libs:
            lib.g3_set_option(b"gemm_pingpong", 3)
            lib.g3_set_option(b"gemm_deferred", deferred)
            def run():
                rc = lib.g3_gemm_bf16_nt(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, epi, gate.data_ptr() if epi == 2 else None, B, N,
                                         res.data_ptr() if epi == 2 else None, N, st)
                assert rc == 0
            ms = timeit(run, 4)
            line.append(f"[{label} {ms:.3f}ms {fl / ms / 1e9:.0f}TF]")
        print(f"gemm {nm} {M}x{N}x{K} epi{epi}: " + "  ".join(line), flush=True)
    del a, w, gate, res, out
_lib.load().g3_set_option(b"gemm_deferred", 1)
