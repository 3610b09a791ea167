"""tests/golden/warp_*.npz: outputs of the REFERENCE's forward_warp / unproject_points / reliable_depth_mask_range_batch
(imported read-only from /root/reference, CPU tensors) on a small synthetic scene (plane + two discs).
For foreground_masking the reference's lazy Warp hook is pointed at oracle/warp_oracle.ray_triangle_depth (warp-lang is
not installable here), i.e. those cases pin everything except the Warp kernel itself."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"


def scene(h, w, seed=0):
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    depth = 4.0 + 0.004 * xs + 0.002 * ys
    for (cy, cx, r, z) in ((h * 0.4, w * 0.3, h * 0.22, 1.6), (h * 0.65, w * 0.7, h * 0.18, 2.4)):
        depth = np.where((ys - cy) ** 2 + (xs - cx) ** 2 < r * r, z + 0.001 * xs, depth)
    rs = np.random.RandomState(seed)
    img = np.stack([np.sin(xs * 0.21 + c) * np.cos(ys * 0.17 - c) for c in range(3)], 0).astype(np.float32)
    img = np.clip(img + 0.05 * rs.standard_normal(img.shape).astype(np.float32), -1, 1)
    f = 0.8 * w
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
    return depth.astype(np.float32), img, K


def look_left(dx, yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, :3] = R
    w2c[:3, 3] = np.array([dx, 0.01, 0.02], np.float32)
    return w2c


def gen_warp():
    import ref_shims
    ref_shims.install()
    from cosmos_predict1.diffusion.inference import forward_warp_utils_pytorch as fwu
    from oracle import warp_oracle

    def rt_hook(ray_origins, ray_directions, vertices, faces, device):
        tris = vertices.numpy()[faces.numpy()]
        return torch.from_numpy(warp_oracle.ray_triangle_depth(ray_directions.numpy(), tris))

    fwu._warp_initialized = True
    fwu._ray_triangle_intersection_func = rt_hook

    for name, (h, w) in {"warp_small": (48, 64), "warp_mid": (96, 160)}.items():
        depth, img, K = scene(h, w)
        depth_t = torch.from_numpy(depth)[None, None]
        K_t = torch.from_numpy(K)[None]
        src_w2c = torch.eye(4)[None]
        pts = fwu.unproject_points(depth_t, src_w2c, K_t)  # (1,h,w,3)
        rel = fwu.reliable_depth_mask_range_batch(depth_t, ratio_thresh=0.05)
        bnd = ~fwu.reliable_depth_mask_range_batch(depth_t)
        b = 2
        w2cs = torch.from_numpy(np.stack([look_left(0.15, 0.05), look_left(0.32, 0.11)]))
        Ks = K_t.expand(b, 3, 3).contiguous()
        imgs = torch.from_numpy(img)[None].expand(b, 3, h, w).contiguous()
        ptsb = pts.expand(b, h, w, 3).contiguous()
        maskb = rel.float().expand(b, 1, h, w).contiguous()
        out = dict(h=np.array(h), w=np.array(w), depth=depth, image=img, K=K, points=pts.numpy()[0], reliable=rel.numpy()[0, 0],
                   boundary=bnd.numpy()[0, 0], w2cs=w2cs.numpy())
        for fg in (False, True):
            wf, m2, d2, flow = fwu.forward_warp(imgs.clone(), mask1=maskb.clone(), depth1=None, transformation1=None,
                                                transformation2=w2cs, intrinsic1=Ks, intrinsic2=Ks, render_depth=True,
                                                world_points1=ptsb, foreground_masking=fg,
                                                boundary_mask=bnd[:, 0].expand(b, h, w).contiguous() if fg else None)
            tag = "fg" if fg else "nofg"
            out.update({f"{tag}_frame": wf.numpy(), f"{tag}_mask": m2.numpy(), f"{tag}_depth": d2.numpy(), f"{tag}_flow": flow.numpy()})
            print(name, tag, "mask coverage", float(m2.mean()), "occluded px", int((out["nofg_mask"] != m2.numpy()).sum()) if fg else 0)
        np.savez_compressed(GOLD / f"{name}.npz", **out)
        print(name, "file MB", (GOLD / f"{name}.npz").stat().st_size / 1e6)


if __name__ == "__main__":
    gen_warp()
