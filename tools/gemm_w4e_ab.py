"""A/B of build-time variants of the deferred-epilogue GEMM (csrc/gemm_w4e.hpp) at the benchmark shapes, interleaved in one process on one box.
Build HERE before gpurun:  python tools/gemm_w4e_ab.py --build      (copies of the library with -DG3_AB_GW4E_PF=0 / 1, -DG3_GW4E_PFD=8)
GPU box:                   python tools/gemm_w4e_ab.py
Shapes: G3_W4E_AB_SHAPES="name:M:N:K:epilogue;..." (default: the four block GEMMs at M = 2 x 56 320).
Variants (G3_W4E_AB_VARIANTS="name=flags;..."): product (no prefetch), pf1 (token slices by every workgroup), pf2 (token + weight slices), pf3 / pf4 (leaders only),
and the product library with gemm_deferred = 0 (the non-persistent one-wave kernel)."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
# pf3 / pf4: only the LEADER of the workgroups that share a slice prefetches it (token slices / token + weight slices)
VARIANTS = [(v.split("=")[0], tuple(v.split("=", 1)[1].split())) for v in (os.environ.get("G3_W4E_AB_VARIANTS") or "pf3=-DG3_AB_GW4E_PF=3").split(";") if v]

if "--build" in sys.argv:
    from gen3c_amd import build
    for suffix, flags in VARIANTS:
        print(build.build(extra_flags=flags, suffix="_" + suffix, force=True))
    sys.exit(0)

import torch  # noqa: E402
from gen3c_amd import _lib  # noqa: E402
from tools.microbench import timeit  # noqa: E402

libs = [("plain-w4", _lib.load(), 0, 0), ("product", _lib.load(), 1, 0), ("tokens-first", _lib.load(), 1, 1)]  # (label, library, gemm_deferred, gemm_tokens_first)
for suffix, _ in VARIANTS:
    f = ROOT / "gen3c_amd" / "lib" / f"libgen3c_hip_{suffix}.so"
    if not f.exists():
        continue
    lib = C.CDLL(str(f))
    for name, argtypes in _lib.SIGNATURES.items():
        getattr(lib, name).argtypes = argtypes
    libs.append((suffix, lib, 1, 0))
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = 2  # the benchmark's launches carry both CFG branches: M = 2 x 56 320
SHAPES = [("qkv", 56320 * B, 12288, 4096, 0), ("out", 56320 * B, 4096, 4096, 2), ("w1", 56320 * B, 16384, 4096, 1), ("w2", 56320 * B, 4096, 16384, 2)]
if os.environ.get("G3_W4E_AB_SHAPES"):  # "name:M:N:K:epilogue;..." (e.g. the per-rank shapes of cp = 8: M = 14 080)
    SHAPES = [(t.split(":")[0], *[int(x) for x in t.split(":")[1:]]) for t in os.environ["G3_W4E_AB_SHAPES"].split(";") if t]
for (nm, M, N, K, epi) in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    gate = torch.randn(B, N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    outs = {}
    fl = 2.0 * M * N * K
    for rnd in range(3):
        line = []
        for (label, lib, deferred, tf) in libs:
            lib.g3_set_option(b"gemm_pingpong", 3)
            lib.g3_set_option(b"gemm_deferred", deferred)
            lib.g3_set_option(b"gemm_tokens_first", tf)
            out = outs.setdefault(label, torch.empty(M, N, device=dev, dtype=torch.bfloat16))
            def run():
                rc = lib.g3_gemm_bf16_nt(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, epi, gate.data_ptr() if epi == 2 else None, B, N,
                                         res.data_ptr() if epi == 2 else None, N, st)
                assert rc == 0
            ms = timeit(run, 4)
            line.append(f"[{label} {ms:.3f}ms {fl / ms / 1e9:.0f}TF]")
        print(f"gemm {nm} {M}x{N}x{K} epi{epi}: " + "  ".join(line), flush=True)
    torch.cuda.synchronize()
    same = all(torch.equal(outs["plain-w4"], o) for o in outs.values())
    print(f"   outputs of all variants bitwise equal: {same}", flush=True)
    del a, w, gate, res, outs
_lib.load().g3_set_option(b"gemm_deferred", 1)
_lib.load().g3_set_option(b"gemm_tokens_first", 0)
