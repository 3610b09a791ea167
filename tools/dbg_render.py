import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
from gen3c_amd import _lib
if os.environ.get('DBG_LIB'):
    _lib._LIB_PATH = Path(os.environ['DBG_LIB']).resolve()
from gen3c_amd import renderer
dev = torch.device("cuda:0")
z = dict(np.load(Path(__file__).resolve().parent.parent / "tests" / "golden" / "warp_small.npz"))
h, w = int(z["h"]), int(z["w"])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
b = 2
imgs = t(z["image"])[None].expand(b, 3, h, w).contiguous()
pts = t(z["points"])[None].expand(b, h, w, 3).contiguous()
mask = t(z["reliable"].astype(np.float32))[None, None].expand(b, 1, h, w).contiguous()
Ks = t(z["K"])[None].expand(b, 3, 3).contiguous()
outs = {}
for mode in (True, False):
    renderer._WINDOW_SPLAT = mode
    frame, m2, d2, flow = renderer.forward_warp(imgs, mask, None, None, t(z["w2cs"]), Ks, Ks, render_depth=True, world_points1=pts)
    torch.cuda.synchronize()
    outs[mode] = (frame.cpu().numpy(), d2.cpu().numpy())
print("h w", h, w)
ref = z["nofg_frame"]
for mode in (True, False):
    e = np.abs(outs[mode][0] - ref)
    print("mode", mode, "max err vs golden", e.max(), "bad frac", (e > 1e-4 + 1e-3 * np.abs(ref)).mean())
e = np.abs(outs[True][0] - outs[False][0])[0].max(0)
ys, xs = np.nonzero(e > 1e-3)
print("diff pixels", len(ys), "rows hist", np.bincount(ys % 8, minlength=8), "cols hist (mod 32)", np.bincount(xs % 32, minlength=32))
print("first few", list(zip(ys[:10], xs[:10])), e[ys[:10], xs[:10]])
fl = z["nofg_flow"]
print("flow sample", fl[0, :, 10, 10], fl[0, :, 10, 11])
