"""Timing ablation of the self-attention kernel: which instruction class bounds the 8-wave structure?
Build the ablated copies HERE before gpurun:   python tools/attn_ablate.py --build
GPU box:                                       python tools/attn_ablate.py          (variant 4 kernel through every copy)
Bits (G3_AB_ATTN_ABLATE, csrc/attention.hip): 1 no exp2, 2 no row-sum adds, 4 no row-max chain / rescale test, 8 no LDS fragment reads."""
import ctypes as C
import math
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
CASES = [int(x) for x in (os.environ.get('G3_ABLATE_CASES') or '3,4,7,8,15').split(',')]
VARIANT = int(os.environ.get('G3_ABLATE_VARIANT', '4'))

if "--build" in sys.argv:
    from gen3c_amd import build
    for bits in CASES:
        print(build.build(extra_flags=(f"-DG3_AB_ATTN_ABLATE={bits}",), suffix=f"_abl{bits}", force=True))
    sys.exit(0)

import torch  # noqa: E402
from gen3c_amd import _lib, ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

libs = [("product", _lib.load())]
for bits in CASES:
    lib = C.CDLL(str(ROOT / "gen3c_amd" / "lib" / f"libgen3c_hip_abl{bits}.so"))
    for name, argtypes in _lib.SIGNATURES.items():
        getattr(lib, name).argtypes = argtypes
    libs.append((f"ablate={bits}", lib))
dev = torch.device("cuda:0")
S, H = 56320, 8
q, k, v = (torch.randn(S, H * 128, device=dev).to(torch.bfloat16) for _ in range(3))
vt = ops.transpose_v(v, S, 1, H)
ld = vt.shape[-1]
out = torch.empty_like(q)
st = torch.cuda.current_stream().cuda_stream
fl = 4.0 * S * S * 128 * H
legend = {"product": "full kernel", "ablate=3": "no exp2, no row-sum adds", "ablate=4": "no row-max chain", "ablate=7": "no softmax VALU except cvt_pk",
          "ablate=8": "no LDS fragment reads", "ablate=15": "MFMA + cvt_pk + LDS-DMA + barriers only"}
for nm, lib in libs:
    lib.g3_set_option(b"attn_variant", VARIANT)
for rnd in range(2):
    for nm, lib in libs:
        def run():
            rc = lib.g3_flash_attn_fwd_bf16(q.data_ptr(), H * 128, H * 128, 128, k.data_ptr(), H * 128, H * 128, 128, vt.data_ptr(), ld, H * 128 * ld, 128 * ld,
                                            out.data_ptr(), H * 128, H * 128, 128, S, S, 1, H, 128, 1.0 / math.sqrt(128), st)
            assert rc == 0
        ms = timeit(run, 3)
        print(f"{nm:10s} {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF-equivalent   ({legend.get(nm, 'w4: 16 no pair units, 32 no fragment reads, 64 no tile barrier, 128 no in-stream LDS-DMA, 256 no row-max chains')})", flush=True)
