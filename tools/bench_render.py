"""Renderer throughput at the benchmark resolution (704x1280): items/s and algorithmic HBM GB/s.
Algorithmic bytes per item: points 10.8 MB + image 10.8 MB + mask 3.6 MB read, frame 10.8 MB + mask 3.6 MB
(+ depth 3.6 MB) written = 43.2 MB (46.8 with depth)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops, renderer  # noqa: E402
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))


def scene(h, w):
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    depth = 4.0 + 0.0004 * xs + 0.0002 * ys
    for (cy, cx, r, zz) in ((h * 0.4, w * 0.3, h * 0.22, 1.6), (h * 0.65, w * 0.7, h * 0.18, 2.4)):
        depth = np.where((ys - cy) ** 2 + (xs - cx) ** 2 < r * r, zz + 0.0001 * xs, depth)
    img = np.stack([np.sin(xs * 0.021 + c) * np.cos(ys * 0.017 - c) for c in range(3)], 0).astype(np.float32)
    K = np.array([[1000, 0, w / 2], [0, 1000, h / 2], [0, 0, 1]], np.float32)
    return depth.astype(np.float32), img, K


def main():
    dev = torch.device("cuda:0")
    h, w = 704, 1280
    depth, img, K = scene(h, w)
    t = lambda a: torch.from_numpy(a).to(dev)
    F = 32
    w2cs = torch.eye(4, device=dev).repeat(1, F, 1, 1)
    w2cs[0, :, 0, 3] = torch.linspace(0, 0.3, F, device=dev)  # "left" trajectory, movement_distance 0.3
    Ks = t(K)[None, None].expand(1, F, 3, 3).contiguous()
    for fg, tiled in ((False, 0), (False, 1), (False, 2), (True, 0), (True, 1), (True, 2)):
        ops.set_option("splat_tiled", min(tiled, 1))
        renderer._WINDOW_SPLAT = tiled == 2  # 2: window stores + destination-owned gather/resolve (default)
        cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=t(img)[None], input_depth=t(depth)[None, None],
                                        input_w2c=torch.eye(4, device=dev)[None], input_intrinsics=t(K)[None],
                                        filter_points_threshold=0.05, foreground_masking=fg, input_format=["B", "C", "H", "W"])
        cache.render_cache(w2cs, Ks)
        torch.cuda.synchronize()
        tm = ops.HipTimer()
        tm.start()
        reps = 3
        for _ in range(reps):
            pix, msk = cache.render_cache(w2cs, Ks)
        tm.stop()
        ms = tm.elapsed_ms() / reps
        per_item = ms / F
        gbs = 43.2e6 / (per_item * 1e-3) / 1e9
        print(f"render 704x1280 foreground_masking={fg} splat_tiled={tiled}: {per_item:.3f} ms/item ({F} items, {ms:.1f} ms)  algorithmic {gbs:.0f} GB/s "
              f"({gbs/8000*100:.1f}% of 8 TB/s); mask coverage {float(msk.mean()):.3f}", flush=True)


if __name__ == "__main__":
    main()
    ops.set_option("splat_tiled", 1)
    renderer._WINDOW_SPLAT = True
