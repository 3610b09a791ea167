"""In-process A/B of compiler flags: the product library against a copy built with extra hipcc flags (built HERE, before
gpurun, with  python tools/ab_flags.py --build -fno-slp-vectorize ; the copy travels as gen3c_amd/lib/libgen3c_hip_ab.so).
usage (GPU box): python tools/ab_flags.py"""
import ctypes as C
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

if "--build" in sys.argv:
    from gen3c_amd import build
    flags = tuple(a for a in sys.argv[1:] if a != "--build")
    print(build.build(extra_flags=flags, suffix="_ab", force=True), flags)
    sys.exit(0)

import torch  # noqa: E402
from gen3c_amd import _lib, ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

base = _lib.load()
alt = C.CDLL(str(ROOT / "gen3c_amd" / "lib" / "libgen3c_hip_ab.so"))
for name, argtypes in _lib.SIGNATURES.items():
    getattr(alt, name).argtypes = argtypes
dev = torch.device("cuda:0")
S, H = 56320, 8
q = torch.randn(S, H * 128, device=dev).to(torch.bfloat16)
k = torch.randn(S, H * 128, device=dev).to(torch.bfloat16)
v = torch.randn(S, H * 128, device=dev).to(torch.bfloat16)
vt = ops.transpose_v(v, S, 1, H)
ld = vt.shape[-1]
outs = [torch.empty_like(q), torch.empty_like(q)]
st = torch.cuda.current_stream().cuda_stream


def run(lib, o):
    rc = lib.g3_flash_attn_fwd_bf16(q.data_ptr(), H * 128, H * 128, 128, k.data_ptr(), H * 128, H * 128, 128, vt.data_ptr(), ld, H * 128 * ld, 128 * ld,
                                    o.data_ptr(), H * 128, H * 128, 128, S, S, 1, H, 128, 1.0 / math.sqrt(128), st)
    assert rc == 0


fl = 4.0 * S * S * 128 * H
for rnd in range(3):
    for nm, lib, o in (("product", base, outs[0]), ("ab-flags", alt, outs[1])):
        ms = timeit(lambda: run(lib, o), 3)
        print(f"attention {nm:9s} {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF", flush=True)
# the QKV GEMM shape through both libraries
M, N, K = 56320, 12288, 4096
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
co = [torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16)]


def run_gemm(lib, o):
    rc = lib.g3_gemm_bf16_nt(a.data_ptr(), K, w.data_ptr(), K, o.data_ptr(), N, M, N, K, 0, None, 1, 0, None, 0, st)
    assert rc == 0


for rnd in range(3):
    for nm, lib, o in (("product", base, co[0]), ("ab-flags", alt, co[1])):
        ms = timeit(lambda: run_gemm(lib, o), 5)
        print(f"qkv gemm  {nm:9s} {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:6.0f} TF", flush=True)
print("gemm outputs equal:", bool(torch.equal(co[0], co[1])))
print("outputs equal:", bool(torch.equal(outs[0], outs[1])), " rel-l2:", float((outs[0].float() - outs[1].float()).norm() / outs[0].float().norm()))
