"""Diagnostic for tests/test_reference_fixtures_gpu.py::test_render_cache_on_reference_image_vs_oracle: where do the colour outliers of the natural-image scene sit?
For the worst texels: product vs oracle colour, the oracle's total splat weight there, and the list of contributions (source pixel, corner weight, depth weight, colour).
usage (GPU box): python tools/render_fixture_diag.py"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import renderer  # noqa: E402
from oracle import warp_oracle as wo  # noqa: E402
from tests import ref_fixture_inputs as rf  # noqa: E402
from tests.test_reference_fixtures_gpu import _render_inputs, _t  # noqa: E402

F32 = np.float32
dev = torch.device("cuda:0")
h, w, img, depth, K = _render_inputs()
cache = renderer.Cache3D_Buffer(frame_buffer_max=2, noise_aug_strength=0, input_image=_t(img, dev)[None], input_depth=_t(depth, dev)[None, None],
                                input_w2c=torch.eye(4, device=dev)[None], input_intrinsics=_t(K, dev)[None], filter_points_threshold=0.05,
                                foreground_masking=False, input_format=["B", "C", "H", "W"])
w2cs = np.stack([np.eye(4, dtype=np.float32) for _ in range(2)])
w2cs[0, 0, 3] = -0.12
c, s = np.cos(0.06), np.sin(0.06)
w2cs[1, :3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
w2cs[1, :3, 3] = (-0.3, 0.02, -0.1)
pix, msk = cache.render_cache(_t(w2cs, dev)[None], _t(K, dev)[None, None].expand(1, 2, 3, 3))
torch.cuda.synchronize()
pts = wo.unproject_points(depth[None, None], np.eye(4, dtype=np.float32)[None], K[None])
rel = wo.reliable_depth_mask(depth[None, None], ratio_thresh=0.05).astype(np.float32)
# product's own cache-construction outputs vs the oracle's
b2 = lambda a: np.broadcast_to(a, (2,) + a.shape[1:])
fr, m2, _, flow, idx = wo.forward_warp(b2(img[None]), b2(rel), b2(pts), w2cs, b2(K[None]))
got = pix[0, :, 0].cpu().numpy()
err = np.abs(got - fr)
bad = err > (1e-4 + 1e-3 * np.abs(fr))
print("outliers", int(bad.sum()), "of", bad.size, "max", err.max())
# oracle weights
z = idx["z"][:, None]
logd = np.log1p(np.maximum(z, F32(0))).astype(F32)
expo = (logd / (logd.max() + F32(1e-7)) * F32(50)).astype(F32)
dw = (np.exp(np.minimum(expo, F32(80.0))) + F32(1e-7)).astype(F32)
mask1 = (b2(rel) * (idx["z"][:, None] > 0)).astype(F32)
wacc = np.zeros((2, h + 2, w + 2), np.float64)
cnt = np.zeros((2, h + 2, w + 2), np.int32)
wmax = np.zeros((2, h + 2, w + 2), np.float64)
bi = np.arange(2)[:, None, None]
for key, yy, xx in (("nw", "fy", "fx"), ("sw", "cy", "fx"), ("ne", "fy", "cx"), ("se", "cy", "cx")):
    wt = (idx[key] * mask1[:, 0] / dw[:, 0]).astype(F32)
    np.add.at(wacc, (bi, idx[yy], idx[xx]), wt.astype(np.float64))
    np.add.at(cnt, (bi, idx[yy], idx[xx]), (wt > 0).astype(np.int32))
    np.maximum.at(wmax, (bi, idx[yy], idx[xx]), wt.astype(np.float64))
wacc, cnt, wmax = wacc[:, 1:-1, 1:-1], cnt[:, 1:-1, 1:-1], wmax[:, 1:-1, 1:-1]
badpix = bad.any(axis=1)
print("outlier texels", int(badpix.sum()))
ws = wacc[badpix]
print("oracle total weight at outlier texels: min %.3e median %.3e max %.3e" % (ws.min(), np.median(ws), ws.max()))
print("contributions at outlier texels: min %d median %d max %d" % (cnt[badpix].min(), np.median(cnt[badpix]), cnt[badpix].max()))
print("share of the largest contribution at outlier texels: median %.4f" % np.median(wmax[badpix] / np.maximum(ws, 1e-300)))
allw = wacc[m2[:, 0] > 0]
print("all valid texels: weight quantiles 1e-4 / 1e-2 / 0.5:", np.quantile(allw, [1e-4, 1e-2, 0.5]))
order = np.argsort(-err.max(axis=1).reshape(-1))[:12]
for o in order:
    i, y, x = np.unravel_index(o, (2, h, w))
    print(f"item {i} texel ({y},{x}): got {got[i, :, y, x]} ref {fr[i, :, y, x]} wsum {wacc[i, y, x]:.3e} n {cnt[i, y, x]} wmax share {wmax[i, y, x] / max(wacc[i, y, x], 1e-300):.4f}")
# is the product's mask1 (reliable) the oracle's?
rel_p = renderer.reliable_depth_mask_range_batch(_t(depth, dev)[None, None], ratio_thresh=0.05)[0, 0].cpu().numpy()
print("reliable-mask px differing product vs oracle:", int((rel_p != (rel[0, 0] > 0)).sum()))
pts_p = renderer.unproject_points(_t(depth, dev)[None, None], torch.eye(4, device=dev)[None], _t(K, dev)[None])[0].cpu().numpy()
print("unprojected points max abs diff:", float(np.abs(pts_p - pts[0]).max()))

# ---- which path loses / changes a contribution? the same pair under every form of the renderer
from gen3c_amd import _lib  # noqa: E402
lib = _lib.load()


def run(label, opts, window=True):
    for k_, v_ in opts.items():
        assert lib.g3_set_option(k_.encode(), v_) == 0
    renderer._WINDOW_SPLAT = window
    try:
        c2 = renderer.Cache3D_Buffer(frame_buffer_max=2, noise_aug_strength=0, input_image=_t(img, dev)[None], input_depth=_t(depth, dev)[None, None],
                                     input_w2c=torch.eye(4, device=dev)[None], input_intrinsics=_t(K, dev)[None], filter_points_threshold=0.05,
                                     foreground_masking=False, input_format=["B", "C", "H", "W"])
        p2, m2_ = c2.render_cache(_t(w2cs, dev)[None], _t(K, dev)[None, None].expand(1, 2, 3, 3))
        torch.cuda.synchronize()
        g2 = p2[0, :, 0].cpu().numpy()
        e2 = np.abs(g2 - fr)
        b2_ = e2 > (1e-4 + 1e-3 * np.abs(fr))
        print(f"{label}: outliers {int(b2_.sum())} max {e2.max():.3e}; mask px differing {int((m2_[0, :, 0].cpu().numpy() != m2).sum())}; vs default product: "
              f"{int((np.abs(g2 - got) > 1e-4 + 1e-3 * np.abs(got)).sum())} values differ")
    finally:
        renderer._WINDOW_SPLAT = True
        for k_ in opts:
            lib.g3_set_option(k_.encode(), {"render_fused": 1, "render_full_extent": 1, "render_exclusive": 0, "splat_tiled": 1, "render_overlap": 1}[k_])


run("default", {})
run("render_fused=0", {"render_fused": 0})
run("render_full_extent=0", {"render_full_extent": 0})
run("render_exclusive=1", {"render_exclusive": 1})
run("two-call form (global accumulator atomics)", {}, window=False)
run("two-call form, splat_tiled=0", {"splat_tiled": 0}, window=False)
# the oracle with the product's fast-math depth weight (hardware log2 / exp2 / rcp: ~1e-5 relative) - does the weight's precision explain it?
wsum64 = np.zeros((2, h + 2, w + 2), np.float64)
csum64 = np.zeros((2, h + 2, w + 2, 3), np.float64)
frc = np.moveaxis(np.broadcast_to(img[None], (2, 3, h, w)), 1, -1).astype(np.float64)
for key, yy, xx in (("nw", "fy", "fx"), ("sw", "cy", "fx"), ("ne", "fy", "cx"), ("se", "cy", "cx")):
    wt = (idx[key].astype(np.float64) * mask1[:, 0] / dw[:, 0].astype(np.float64))
    np.add.at(wsum64, (bi, idx[yy], idx[xx]), wt)
    np.add.at(csum64, (bi, idx[yy], idx[xx]), frc * wt[..., None])
ref64 = np.moveaxis(csum64[:, 1:-1, 1:-1] / np.maximum(wsum64[:, 1:-1, 1:-1, None], 1e-300), -1, 1)
e64 = np.abs(fr - ref64) * (m2 > 0)
print("oracle fp32 vs the same sums in fp64: values beyond tolerance", int((e64 > 1e-4 + 1e-3 * np.abs(ref64)).sum()), "max", e64.max())
e64g = np.abs(got - ref64) * (m2 > 0)
print("product vs fp64 sums: values beyond tolerance", int((e64g > 1e-4 + 1e-3 * np.abs(ref64)).sum()), "max", e64g.max())
