"""A/B of the whole cache render (bench.py's roofline_render configuration: 32 items of 704x1280, foreground masking) between the product library and
lib/libgen3c_hip_ab.so (an older / variant build): the loader's handle is swapped between the timed blocks; outputs compared bitwise."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402
from gen3c_amd import _lib, ops, renderer  # noqa: E402
from bench_render import scene  # noqa: E402

base = _lib.load()
alt = C.CDLL(str(ROOT / "gen3c_amd" / "lib" / "libgen3c_hip_ab.so"))
for name, argtypes in _lib.SIGNATURES.items():
    if hasattr(alt, name):
        getattr(alt, name).argtypes = argtypes
        getattr(alt, name).restype = _lib._RESTYPES.get(name, C.c_int)
dev = torch.device("cuda:0")
h, w, F = 704, 1280, 32
depth, img, K = scene(h, w)
t = lambda a: torch.from_numpy(a).to(dev)
w2cs = torch.eye(4, device=dev).repeat(1, F, 1, 1)
w2cs[0, :, 0, 3] = torch.linspace(0, 0.3, F, device=dev)
Ks = t(K)[None, None].expand(1, F, 3, 3).contiguous()
outs, res = {}, {"product": [], "ab": []}
for rep in range(5):
    for name, lib in (("product", base), ("ab", alt)):
        _lib._lib = lib
        renderer._RENDER_WS.clear()  # the two builds lay their workspaces out differently
        cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=t(img)[None], input_depth=t(depth)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                        input_intrinsics=t(K)[None], filter_points_threshold=0.05, foreground_masking=True, input_format=["B", "C", "H", "W"])
        for _ in range(2):
            cache.render_cache(w2cs, Ks)
        torch.cuda.synchronize()
        tm = ops.HipTimer()
        tm.start()
        for _ in range(10):
            pix, msk = cache.render_cache(w2cs, Ks)
        tm.stop()
        res[name].append(tm.elapsed_ms() / 10 / F)
        outs[name] = (pix.clone(), msk.clone())
        del cache
_lib._lib = base
for name, ms in res.items():
    m = sum(ms[1:]) / len(ms[1:])
    print(f"{name:8s} {m:.4f} ms/item = {43.2e6 / (m * 1e-3) / 1e9:.0f} GB/s   runs: " + " ".join(f"{x:.4f}" for x in ms))
print("frames bitwise equal:", bool(torch.equal(outs["product"][0], outs["ab"][0])), " masks bitwise equal:", bool(torch.equal(outs["product"][1], outs["ab"][1])))
