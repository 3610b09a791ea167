"""A/B of csrc/attention_d512.hip builds: the product library vs lib/libgen3c_hip_ab.so (built with `python -m gen3c_amd.build`-style extra flags, e.g.
`python tools/flash512_ab.py --build -DG3_AB_D512_TWO_CHAINS`), alternating launches at the benchmark shape (16 frames x 14 080 pixels, d = 512).
  build (here):  python tools/flash512_ab.py --build -D...      run (GPU box):  python tools/flash512_ab.py"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

if "--build" in sys.argv:
    from gen3c_amd import build
    sfx = next((a for a in sys.argv[1:] if a.startswith("_ab")), "_ab")
    print(build.build(extra_flags=tuple(a for a in sys.argv[1:] if a.startswith("-D")), suffix=sfx, force=False))
    sys.exit(0)

import torch  # noqa: E402
from gen3c_amd import _lib, ops  # noqa: E402

base = _lib.load()
libs = [("product", base)]
for pth in sorted((ROOT / "gen3c_amd" / "lib").glob("libgen3c_hip_ab*.so")):
    alt = C.CDLL(str(pth))
    for name, argtypes in _lib.SIGNATURES.items():
        getattr(alt, name).argtypes = argtypes
        getattr(alt, name).restype = _lib._RESTYPES.get(name, C.c_int)
    libs.append((pth.stem.replace("libgen3c_hip", ""), alt))
dev = torch.device("cuda:0")
T, HW, Cc = 16, 14080, 512
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (torch.randn(T, HW, Cc, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
vT = v.reshape(T * HW, Cc).t().contiguous()
outs = {}
st = torch.cuda.current_stream().cuda_stream
res = {n: [] for n, _ in libs}
for rep in range(6):
    for name, lib in libs:
        o = torch.empty_like(q)
        tm = ops.HipTimer()
        tm.start()
        rc = lib.g3_spatial_attn_d512_bf16(q.data_ptr(), k.data_ptr(), vT.data_ptr(), T * HW, HW, o.data_ptr(), T, HW, Cc ** -0.5, st)
        tm.stop()
        assert rc == 0
        if rep:
            res[name].append(tm.elapsed_ms())
        outs[name] = o
fl = 4.0 * T * HW * HW * Cc
for name, ms in res.items():
    m = sum(ms) / len(ms)
    print(f"{name:8s} {m:.3f} ms  {fl / m / 1e9:.0f} TFLOP/s   runs: " + " ".join(f"{x:.3f}" for x in ms))
for name in list(outs)[1:]:
    d = float((outs["product"].float() - outs[name].float()).norm() / outs["product"].float().norm())
    print(f"rel-L2 product vs {name}: {d:.3e}")
