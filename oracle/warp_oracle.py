"""CPU oracle for the GEN3C 3D-cache renderer (TEST INFRASTRUCTURE - never imported by gen3c_amd/).

Restates, with an explicit fp32 operation order, the reference's
  project_points / forward_warp / bilinear_splatting / points_to_mesh / get_camera_rays
  (cosmos_predict1/diffusion/inference/forward_warp_utils_pytorch.py:49-132, 151-168, 171-336, 462-486, 576-703),
  reliable_depth_mask_range_batch (:338-353), unproject_points (:410-460) and the Moller-Trumbore kernel
  (ray_triangle_intersection_warp.py:23-105).

Pinning: PINNED to the reference's own Python for everything except the NVIDIA-Warp kernel - tests/golden/warp_*.npz
hold outputs of the reference `forward_warp` (imported from /root/reference, CPU tensors) produced by
tools/gen_golden_warp.py; for the foreground_masking cases the reference's lazy hook
`_ray_triangle_intersection_func` was pointed at `ray_triangle_intersection` below, because warp-lang is absent.
That kernel is pinned since round 3 to the reference's own Warp SOURCE: tools/gen_golden_warp_kernel.py runs the reference's
ray_triangle_intersection_warp.py (kernel body + host wrapper, unmodified) under tools/wp_standin.py (fp32 scalars / vec3, one
rounding per operation) inside the reference's forward_warp; its outputs equal the warp_small goldens and ray_triangle_depth /
oracle/c/ray_tri.c reproduce the recorded depth maps bit for bit (tests/golden/warp_kernel_small.npz). What a stand-in cannot
pin is the instruction selection of a real Warp / NVRTC build (fma contraction inside dot / cross).

Integer / boolean products (pixel indices, masks) are exact; float products depend on atomics order in the reference
itself (index_put_ accumulate) and on libm (log1p/exp), and are compared with a tolerance.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _f(x):
    return np.asarray(x, dtype=F32)


def project_points(world_points: np.ndarray, w2c: np.ndarray, K: np.ndarray):
    """(b,h,w,3) world points -> projected (b,h,w,3) = K.(W.[p,1])[:3] and camera-space points (b,h,w,3).
    forward_warp_utils_pytorch.py:462-486. fp32, products summed left to right (no FMA)."""
    p = _f(world_points)
    W = _f(w2c)[:, None, None]  # (b,1,1,4,4)
    Km = _f(K)[:, None, None]
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    cam = []
    for i in range(3):
        cam.append(((W[..., i, 0] * x + W[..., i, 1] * y) + W[..., i, 2] * z) + W[..., i, 3] * F32(1.0))
    proj = []
    for i in range(3):
        proj.append((Km[..., i, 0] * cam[0] + Km[..., i, 1] * cam[1]) + Km[..., i, 2] * cam[2])
    return np.stack(proj, -1).astype(F32), np.stack(cam, -1).astype(F32)


def splat_indices(proj: np.ndarray, h: int, w: int):
    """Pixel indices + proximity weights of bilinear_splatting (forward_warp_utils_pytorch.py:244-250, 604-634).
    proj (b,h,w,3). Returns dict of int64 index arrays (b,h,w) and fp32 weights; all bit-exact by construction."""
    z = proj[..., 2]
    u = proj[..., 0] / (z + F32(1e-7))
    v = proj[..., 1] / (z + F32(1e-7))
    gx = np.arange(w, dtype=F32)[None, None, :]
    gy = np.arange(h, dtype=F32)[None, :, None]
    flow_x, flow_y = (u - gx).astype(F32), (v - gy).astype(F32)       # flow12 = trans_coordinates - grid
    tx, ty = (flow_x + gx).astype(F32), (flow_y + gy).astype(F32)     # trans_pos = flow12 + grid
    ox, oy = (tx + F32(1.0)).astype(F32), (ty + F32(1.0)).astype(F32)  # trans_pos_offset
    with np.errstate(invalid="ignore"):
        fx = np.clip(np.floor(ox).astype(np.int64), 0, w + 1)
        cx = np.clip(np.ceil(ox).astype(np.int64), 0, w + 1)
        fy = np.clip(np.floor(oy).astype(np.int64), 0, h + 1)
        cy = np.clip(np.ceil(oy).astype(np.int64), 0, h + 1)
    oxc = np.clip(ox, F32(0), F32(w + 1)).astype(F32)
    oyc = np.clip(oy, F32(0), F32(h + 1)).astype(F32)
    one = F32(1.0)
    wy_f = (one - (oyc - fy.astype(F32))).astype(F32)
    wy_c = (one - (cy.astype(F32) - oyc)).astype(F32)
    wx_f = (one - (oxc - fx.astype(F32))).astype(F32)
    wx_c = (one - (cx.astype(F32) - oxc)).astype(F32)
    return dict(fx=fx, cx=cx, fy=fy, cy=cy, nw=(wy_f * wx_f).astype(F32), sw=(wy_c * wx_f).astype(F32),
                ne=(wy_f * wx_c).astype(F32), se=(wy_c * wx_c).astype(F32), flow=np.stack([flow_x, flow_y], 1).astype(F32),
                z=z.astype(F32))


def bilinear_splatting(frame: np.ndarray, mask1: np.ndarray, idx: dict, is_image: bool):
    """frame (b,c,h,w), mask1 (b,1,h,w) -> warped (b,c,h,w), mask2 (b,1,h,w).
    forward_warp_utils_pytorch.py:636-695; the log-depth max is over the WHOLE call (all b items)."""
    b, c, h, w = frame.shape
    z = idx["z"][:, None]  # (b,1,h,w)
    logd = np.log1p(np.maximum(z, F32(0))).astype(F32)
    expo = (logd / (logd.max() + F32(1e-7)) * F32(50)).astype(F32)
    dw = (np.exp(np.minimum(expo, F32(80.0))) + F32(1e-7)).astype(F32)
    acc = np.zeros((b, h + 2, w + 2, c), F32)
    wacc = np.zeros((b, h + 2, w + 2, 1), F32)
    fr = np.moveaxis(_f(frame), 1, -1)  # (b,h,w,c)
    bi = np.arange(b)[:, None, None]
    for key, yy, xx in (("nw", "fy", "fx"), ("sw", "cy", "fx"), ("ne", "fy", "cx"), ("se", "cy", "cx")):
        wt = np.moveaxis((idx[key][:, None] * _f(mask1) * F32(1.0) / dw).astype(F32), 1, -1)  # (b,h,w,1)
        np.add.at(acc, (bi, idx[yy], idx[xx]), (fr * wt).astype(F32))
        np.add.at(wacc, (bi, idx[yy], idx[xx]), wt)
    acc = np.moveaxis(acc, -1, 1)[:, :, 1:-1, 1:-1]
    wts = np.moveaxis(wacc, -1, 1)[:, :, 1:-1, 1:-1]
    wts = np.where(np.isnan(wts), F32(1000.0), wts)
    m = wts > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(m, acc / wts, F32(-1.0 if is_image else 0.0)).astype(F32)
    if is_image:
        out = np.clip(out, F32(-1), F32(1))
    return out, m.astype(F32)


def camera_rays(h: int, w: int, K: np.ndarray) -> np.ndarray:
    """get_camera_rays for ONE intrinsic (3,3) -> (h,w,3) unit rays (forward_warp_utils_pytorch.py:151-168).
    K^-1 is computed in fp32 like torch.linalg.inv; the 3x3 product is summed left to right."""
    Ki = np.linalg.inv(_f(K)).astype(F32)
    xs = np.arange(w, dtype=F32)[None, :]
    ys = np.arange(h, dtype=F32)[:, None]
    r = [((Ki[i, 0] * xs + Ki[i, 1] * ys) + Ki[i, 2] * F32(1.0)).astype(F32) for i in range(3)]
    d = np.stack(r, -1)
    n = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(F32)
    n = np.where(n == 0, F32(1), n)
    return (d / n[..., None]).astype(F32)


def downsample_points_mask(cam_points: np.ndarray, mask: np.ndarray, factor: int = 4):
    """points_to_mesh's resize: bilinear(align_corners=False) on points, nearest on mask
    (forward_warp_utils_pytorch.py:65-77). For an integer factor f the bilinear sample sits at f*i + (f-1)/2."""
    h, w, _ = cam_points.shape
    nh, nw = h // factor, w // factor
    sy, sx = F32(h) / F32(nh), F32(w) / F32(nw)

    def src(n_out, scale, n_in):
        c = ((np.arange(n_out, dtype=F32) + F32(0.5)) * scale - F32(0.5)).astype(F32)
        c = np.maximum(c, F32(0))
        i0 = np.floor(c).astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (c - i0.astype(F32)).astype(F32)
        return i0, i1, (F32(1) - l1).astype(F32), l1

    y0, y1, wy0, wy1 = src(nh, sy, h)
    x0, x1, wx0, wx1 = src(nw, sx, w)
    p = _f(cam_points)
    top = p[y0][:, x0] * wx0[None, :, None] + p[y0][:, x1] * wx1[None, :, None]
    bot = p[y1][:, x0] * wx0[None, :, None] + p[y1][:, x1] * wx1[None, :, None]
    pts = (top * wy0[:, None, None] + bot * wy1[:, None, None]).astype(F32)
    my = np.floor(np.arange(nh, dtype=F32) * sy).astype(np.int64)
    mx = np.floor(np.arange(nw, dtype=F32) * sx).astype(np.int64)
    m = mask[my][:, mx].astype(bool)
    return pts, m


def mesh_triangles(pts: np.ndarray, m: np.ndarray) -> np.ndarray:
    """Triangles (M,3,3) of points_to_mesh (forward_warp_utils_pytorch.py:79-132): every 2x2 patch with at least one
    masked corner gives (tl,tr,bl) and (tr,br,bl). Vertex compaction (torch.unique) does not change the geometry."""
    valid = m[:-1, :-1] | m[:-1, 1:] | m[1:, :-1] | m[1:, 1:]
    ys, xs = np.nonzero(valid)
    if len(ys) == 0:
        return np.zeros((0, 3, 3), F32)
    tl, tr, bl, br = pts[ys, xs], pts[ys, xs + 1], pts[ys + 1, xs], pts[ys + 1, xs + 1]
    return np.concatenate([np.stack([tl, tr, bl], 1), np.stack([tr, br, bl], 1)], 0).astype(F32)


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1).astype(F32)


def _dot(a, b):
    return ((a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]).astype(F32)


def ray_triangle_depth(rays: np.ndarray, tris: np.ndarray, eps: float = 1e-8, chunk: int = 256) -> np.ndarray:
    """Moller-Trumbore, min t > eps over all triangles, 0 where no hit; ray origin = 0
    (ray_triangle_intersection_warp.py:23-105). rays (h,w,3), tris (M,3,3) -> (h,w) fp32."""
    h, w, _ = rays.shape
    d = rays.reshape(-1, 1, 3).astype(F32)
    best = np.full((h * w,), F32(1e10), F32)
    eps = F32(eps)
    for s in range(0, len(tris), chunk):
        t3 = tris[s:s + chunk]
        v0 = t3[None, :, 0]
        e1 = (t3[:, 1] - t3[:, 0])[None]
        e2 = (t3[:, 2] - t3[:, 0])[None]
        hh = _cross(np.broadcast_to(d, (d.shape[0], e2.shape[1], 3)), np.broadcast_to(e2, (d.shape[0], e2.shape[1], 3)))
        a = _dot(np.broadcast_to(e1, hh.shape), hh)
        ok = ~(np.abs(a) < eps)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            f = (F32(1.0) / a).astype(F32)
            sv = (F32(0) - v0).astype(F32)  # ray_origin - v0
            sv = np.broadcast_to(sv, hh.shape)
            u = (f * _dot(sv, hh)).astype(F32)
            ok &= ~((u < 0) | (u > 1))
            q = _cross(sv, np.broadcast_to(e1, hh.shape))
            v = (f * _dot(np.broadcast_to(d, hh.shape), q)).astype(F32)
            ok &= ~((v < 0) | ((u + v).astype(F32) > 1))
            t = (f * _dot(np.broadcast_to(e2, hh.shape), q)).astype(F32)
        ok &= (t > eps)
        t = np.where(ok, t, F32(1e10))
        best = np.minimum(best, t.min(axis=1))
    return np.where(best < F32(1e10), best, F32(0)).reshape(h, w).astype(F32)


_RAY_TRI_LIB = None


def ray_triangle_depth_c(rays: np.ndarray, tris: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """Same function as ray_triangle_depth, evaluated by the plain-C brute force of oracle/c/ray_tri.c (OpenMP over rays; identical
    fp32 operation order, bit-identical results - tests/test_warp_oracle_golden.py). Used at sizes where the numpy version needs
    minutes (704x1280 rays x thousands of triangles). Builds oracle/_build/libray_tri.so with gcc on first use if it is missing."""
    global _RAY_TRI_LIB
    import ctypes as C
    from pathlib import Path
    if _RAY_TRI_LIB is None:
        here = Path(__file__).resolve().parent
        so = here / "_build" / "libray_tri.so"
        if not so.exists():
            import subprocess
            subprocess.run(["make", "-C", str(here / "c")], check=True, capture_output=True)
        lib = C.CDLL(str(so))
        lib.g3o_ray_triangle_depth.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]
        lib.g3o_ray_triangle_depth.restype = None
        _RAY_TRI_LIB = lib
    h, w, _ = rays.shape
    r = np.ascontiguousarray(rays, dtype=F32).reshape(-1, 3)
    t3 = np.ascontiguousarray(tris, dtype=F32).reshape(-1, 9)
    out = np.empty((h * w,), F32)
    _RAY_TRI_LIB.g3o_ray_triangle_depth(r.ctypes.data, r.shape[0], t3.ctypes.data, t3.shape[0], float(eps), out.ctypes.data)
    return out.reshape(h, w)


def forward_warp(frame1, mask1, world_points1, w2c, K, render_depth=False, foreground_masking=False, boundary_mask=None,
                 ray_triangle_fn=None):
    """forward_warp for the cache path (depth1=None, world points given; cache_3d.py:202-214 calls it with
    intrinsic1 = intrinsic2 = target K). Returns (warped_frame2, mask2, warped_depth2 or None, flow12, idx)."""
    frame1 = _f(frame1)
    b, c, h, w = frame1.shape
    mask1 = np.ones((b, 1, h, w), F32) if mask1 is None else _f(mask1)
    proj, cam = project_points(world_points1, w2c, K)
    idx = splat_indices(proj, h, w)
    mask1 = (mask1 * (idx["z"][:, None] > 0)).astype(F32)
    warped, mask2 = bilinear_splatting(frame1, mask1, idx, is_image=True)
    depth2 = None
    if render_depth or foreground_masking:
        depth2 = bilinear_splatting(idx["z"][:, None], mask1, idx, is_image=False)[0][:, 0]
    if foreground_masking:
        for i in range(b):
            pts, m = downsample_points_mask(cam[i], np.asarray(boundary_mask[i]).astype(bool), 4)
            tris = mesh_triangles(pts, m)
            if len(tris) == 0:
                continue
            rays = camera_rays(h, w, K[i])
            t = (ray_triangle_fn or ray_triangle_depth)(rays, tris)
            mesh_z = (t * rays[..., 2]).astype(F32)
            closer = ((mesh_z + F32(0.02)).astype(F32) < depth2[i]) & (mesh_z > 0)
            keep = (~closer).astype(F32)
            mask2[i, 0] = mask2[i, 0] * keep
            warped[i] = ((warped[i] + F32(1)) * keep[None] - F32(1)).astype(F32)
            depth2[i] = depth2[i] * keep
    return warped, mask2, depth2, idx["flow"], idx


def reliable_depth_mask(depth: np.ndarray, window: int = 5, ratio_thresh: float = 0.05, eps: float = 1e-6) -> np.ndarray:
    """reliable_depth_mask_range_batch on (b,1,h,w): 5x5 max/min pooling (implicit -inf padding) and avg pooling
    (zero padding, count_include_pad) -> (max-min)/(mean+eps) < thr & depth > 0 (forward_warp_utils_pytorch.py:338-353)."""
    d = _f(depth)
    b, _, h, w = d.shape
    r = window // 2
    pad_inf = np.pad(d, ((0, 0), (0, 0), (r, r), (r, r)), constant_values=-np.inf)
    pad_ninf = np.pad(-d, ((0, 0), (0, 0), (r, r), (r, r)), constant_values=-np.inf)
    pad0 = np.pad(d, ((0, 0), (0, 0), (r, r), (r, r)), constant_values=0)
    mx = np.full_like(d, -np.inf)
    mn = np.full_like(d, -np.inf)
    sm = np.zeros_like(d)
    for dy in range(window):
        for dx in range(window):
            mx = np.maximum(mx, pad_inf[:, :, dy:dy + h, dx:dx + w])
            mn = np.maximum(mn, pad_ninf[:, :, dy:dy + h, dx:dx + w])
            sm = (sm + pad0[:, :, dy:dy + h, dx:dx + w]).astype(F32)
    mean = (sm / F32(window * window)).astype(F32)
    ratio = ((mx - (-mn)) / (mean + F32(eps))).astype(F32)
    return (ratio < F32(ratio_thresh)) & (d > 0)


def unproject_points(depth: np.ndarray, w2c: np.ndarray, K: np.ndarray) -> np.ndarray:
    """unproject_points(is_depth=True, mask = depth > 0) -> (b,h,w,3) world points, zeros where depth <= 0
    (forward_warp_utils_pytorch.py:410-460)."""
    d = _f(depth)
    b, _, h, w = d.shape
    out = np.zeros((b, h, w, 3), F32)
    xs = np.arange(w, dtype=F32)[None, :]
    ys = np.arange(h, dtype=F32)[:, None]
    for i in range(b):
        Ki = np.linalg.inv(_f(K[i])).astype(F32)
        c2w = np.linalg.inv(_f(w2c[i])).astype(F32)
        un = [((Ki[r, 0] * xs + Ki[r, 1] * ys) + Ki[r, 2] * F32(1.0)).astype(F32) for r in range(3)]
        camp = [(d[i, 0] * un[r]).astype(F32) for r in range(3)]
        for r in range(3):
            val = (((c2w[r, 0] * camp[0] + c2w[r, 1] * camp[1]) + c2w[r, 2] * camp[2]) + c2w[r, 3] * F32(1.0)).astype(F32)
            out[i, :, :, r] = np.where(d[i, 0] > 0, val, F32(0))
    return out
