"""tests/golden/align_small.npz: outputs of the reference's camera_utils.align_depth (CPU, torch autograd) on a small seeded case.
Run in the build container (needs /root/reference); the GPU box only reads the committed .npz."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import ref_shims  # noqa: E402

ref_shims.install()
from cosmos_predict1.diffusion.inference.camera_utils import align_depth  # noqa: E402

torch.manual_seed(0)
rng = np.random.RandomState(7)
H, W = 40, 56
ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
# "cache" depth the new frame must agree with: a slanted plane with a near disc
target = 2.0 + 0.02 * xs + 0.01 * ys
target[((xs - 20) ** 2 + (ys - 18) ** 2) < 60] = 1.2
target = target.astype(np.float32)
# monocular prediction: affine in inverse depth + a smooth 4 % multiplicative error + a little noise
inv = 1.0 / target
src_inv = 0.6 * inv + 0.05
smooth = 1.0 + 0.04 * np.sin(xs / 9.0) * np.cos(ys / 7.0)
source = (1.0 / src_inv * smooth * (1.0 + 0.002 * rng.randn(H, W))).astype(np.float32)
mask = (rng.rand(H, W) < 0.8)
mask[:, :4] = False  # disoccluded band with no rendered depth
target_r = target.copy()
target_r[~mask] = 0.0  # the renderer leaves 0 where nothing landed
K = np.array([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]], np.float32)
ang = 0.1
c2w = np.eye(4, dtype=np.float32)
c2w[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
c2w[:3, 3] = [0.2, -0.1, 0.05]

t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
out = dict(source=source, target=target_r, mask=mask, K=K, c2w=c2w)
out["rigid"] = align_depth(t(source), t(target_r), t(mask)).numpy()
for iters in (1, 3, 100):
    with torch.enable_grad():
        r = align_depth(t(source), t(target_r), t(mask), k=t(K), c2w=t(c2w), alignment_method="non_rigid", num_iters=iters,
                        lambda_arap=0.1, smoothing_kernel_size=3)
    out[f"non_rigid_{iters}"] = r.detach().numpy()
np.savez_compressed(ROOT / "tests" / "golden" / "align_small.npz", **out)
for k, v in out.items():
    print(k, v.shape, v.dtype, float(np.asarray(v, np.float64).mean()))
