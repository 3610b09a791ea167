"""GPU parity of the HIP 3D-cache renderer (through the C ABI) against (a) golden outputs of the reference's own
forward_warp (tests/golden/warp_*.npz) and (b) the numpy oracle at the benchmark resolution.

Bars: flow12 (hence every splat pixel index, which is floor/ceil of (flow+grid)+1) and the mask buffers are BIT-EXACT;
splatted colours / depths are sums of fp32 atomics (order-dependent in the reference too) with libm log1p/exp inside the
weights: rtol 1e-3 / atol 1e-4."""
import numpy as np
import pytest
import torch

from tests.golden_io import GOLD

pytestmark = pytest.mark.gpu


def _load(name):
    return dict(np.load(GOLD / f"{name}.npz"))


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("name", ["warp_small", "warp_mid"])
def test_cache_construction_kernels(name):
    from gen3c_amd import renderer
    dev = torch.device("cuda:0")
    z = _load(name)
    depth = _t(z["depth"], dev)[None, None]
    pts = renderer.unproject_points(depth, torch.eye(4, device=dev)[None], _t(z["K"], dev)[None])
    torch.testing.assert_close(pts[0].cpu(), torch.from_numpy(z["points"]), rtol=1e-6, atol=1e-6)
    rel = renderer.reliable_depth_mask_range_batch(depth, ratio_thresh=0.05)
    assert np.array_equal(rel[0, 0].cpu().numpy(), z["reliable"])
    bnd = ~renderer.reliable_depth_mask_range_batch(depth)
    assert np.array_equal(bnd[0, 0].cpu().numpy(), z["boundary"])


@pytest.mark.parametrize("name", ["warp_small", "warp_mid"])
@pytest.mark.parametrize("fg", [False, True])
def test_forward_warp_matches_reference_golden(name, fg):
    from gen3c_amd import renderer
    dev = torch.device("cuda:0")
    z = _load(name)
    h, w = int(z["h"]), int(z["w"])
    b = 2
    imgs = _t(z["image"], dev)[None].expand(b, 3, h, w).contiguous()
    pts = _t(z["points"], dev)[None].expand(b, h, w, 3).contiguous()
    mask = _t(z["reliable"].astype(np.float32), dev)[None, None].expand(b, 1, h, w).contiguous()
    Ks = _t(z["K"], dev)[None].expand(b, 3, 3).contiguous()
    bnd = _t(z["boundary"], dev)[None].expand(b, h, w).contiguous()
    frame, m2, d2, flow = renderer.forward_warp(imgs, mask, None, None, _t(z["w2cs"], dev), Ks, Ks, render_depth=True,
                                                world_points1=pts, foreground_masking=fg, boundary_mask=bnd if fg else None)
    torch.cuda.synchronize()
    tag = "fg" if fg else "nofg"
    flow_ref = z[f"{tag}_flow"]
    nbad = int((flow.cpu().numpy() != flow_ref).sum())
    assert nbad == 0, f"{nbad} flow values (=> splat indices) differ from the reference bit patterns"
    mdiff = int((m2.cpu().numpy() != z[f"{tag}_mask"]).sum())
    print(f"[{name} {tag}] mask px differing: {mdiff}; frame max err {np.abs(frame.cpu().numpy() - z[f'{tag}_frame']).max():.2e}")
    assert mdiff == 0
    np.testing.assert_allclose(frame.cpu().numpy(), z[f"{tag}_frame"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(d2.cpu().numpy(), z[f"{tag}_depth"], rtol=1e-3, atol=1e-4)


def _scene(h, w):
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    depth = 4.0 + 0.0004 * xs + 0.0002 * ys
    for (cy, cx, r, zz) in ((h * 0.4, w * 0.3, h * 0.22, 1.6), (h * 0.65, w * 0.7, h * 0.18, 2.4)):
        depth = np.where((ys - cy) ** 2 + (xs - cx) ** 2 < r * r, zz + 0.0001 * xs, depth)
    img = np.stack([np.sin(xs * 0.021 + c) * np.cos(ys * 0.017 - c) for c in range(3)], 0).astype(np.float32)
    K = np.array([[1000, 0, w / 2], [0, 1000, h / 2], [0, 0, 1]], np.float32)
    return depth.astype(np.float32), img, K


def test_render_cache_full_resolution_vs_oracle():
    """704x1280 (the benchmark resolution), 2 target frames of a 1-buffer cache = one reference pair, no mesh masking."""
    from gen3c_amd import renderer
    from oracle import warp_oracle
    dev = torch.device("cuda:0")
    h, w = 704, 1280
    depth, img, K = _scene(h, w)
    cache = renderer.Cache3D_Buffer(frame_buffer_max=2, noise_aug_strength=0, input_image=_t(img, dev)[None], input_depth=_t(depth, dev)[None, None],
                                    input_w2c=torch.eye(4, device=dev)[None], input_intrinsics=_t(K, dev)[None], filter_points_threshold=0.05,
                                    foreground_masking=False, input_format=["B", "C", "H", "W"])
    w2cs = np.stack([np.eye(4, dtype=np.float32) for _ in range(2)])
    w2cs[0, 0, 3], w2cs[1, 0, 3] = 0.1, 0.3
    pix, msk = cache.render_cache(_t(w2cs, dev)[None], _t(K, dev)[None, None].expand(1, 2, 3, 3))
    torch.cuda.synchronize()
    assert pix.shape == (1, 2, 1, 3, h, w) and msk.shape == (1, 2, 1, 1, h, w)
    pts = warp_oracle.unproject_points(depth[None, None], np.eye(4, dtype=np.float32)[None], K[None])
    rel = warp_oracle.reliable_depth_mask(depth[None, None], ratio_thresh=0.05).astype(np.float32)
    fr, m2, _, flow, _ = warp_oracle.forward_warp(np.broadcast_to(img[None], (2, 3, h, w)), np.broadcast_to(rel, (2, 1, h, w)),
                                                  np.broadcast_to(pts, (2, h, w, 3)), w2cs, np.broadcast_to(K[None], (2, 3, 3)))
    got_m = msk[0, :, 0].cpu().numpy()
    assert np.array_equal(got_m, m2), f"mask differs on {(got_m != m2).sum()} px"
    got = pix[0, :, 0].cpu().numpy()
    err = np.abs(got - fr)
    bad = err > (1e-4 + 1e-3 * np.abs(fr))
    # a handful of pixels whose total splat weight is ~1e-20 (far-background texels under exp(-50) depth weights) amplify
    # atomic-order / libm differences; the reference's own CUDA atomics have the same sensitivity there
    print(f"[render 704x1280] mask exact; colour outliers {int(bad.sum())}/{bad.size}, max abs err {err.max():.3e}")
    assert bad.mean() < 1e-5 and err.max() < 5e-2


def _cam(tx=0.0, ty=0.0, tz=0.0, yaw=0.0):
    M = np.eye(4, dtype=np.float32)
    c, s = np.cos(yaw), np.sin(yaw)
    M[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    M[:3, 3] = [tx, ty, tz]
    return M


def test_foreground_masking_full_resolution_vs_bruteforce_oracle():
    """BASELINE config 5 runs `--foreground_masking` at 704x1280: forward_warp(foreground_masking=True) on one reference pair
    (forward_warp_utils_pytorch.py:286-334) against the oracle, whose ray/triangle depth is the BRUTE FORCE over all
    901 120 rays x ~2 256 boundary triangles (oracle/c/ray_tri.c, the plain-C restatement of ray_triangle_intersection_warp.py:23-105)
    - no bounding boxes, so the kernel's conservative rasteriser is checked, not mirrored. Item 0: a large lateral move that drags
    the foreground disc's boundary mesh across the left image border; item 1: a camera plane that cuts through the 'skirt' of boundary
    triangles between the foreground disc (z 1.6) and the background (z 4): 780 triangles have a vertex behind the camera, 526 straddle
    z = 0, while 71 % of the frame stays valid. Masks bit-exact."""
    from gen3c_amd import renderer
    from oracle import warp_oracle as wo
    dev = torch.device("cuda:0")
    h, w = 704, 1280
    depth, img, K = _scene(h, w)
    pts = wo.unproject_points(depth[None, None], np.eye(4, dtype=np.float32)[None], K[None])
    rel = wo.reliable_depth_mask(depth[None, None], ratio_thresh=0.05).astype(np.float32)
    bnd = ~wo.reliable_depth_mask(depth[None, None])[0, 0]
    w2cs = np.stack([_cam(tx=-0.5), _cam(tx=-0.7, tz=-1.68)])
    Ks = np.broadcast_to(K[None], (2, 3, 3)).copy()
    # geometry of the case (so that a silent change of the scene cannot hollow the test out)
    _, camp = wo.project_points(np.broadcast_to(pts, (2, h, w, 3)), w2cs, Ks)
    tri_z = [wo.mesh_triangles(*wo.downsample_points_mask(camp[i], bnd, 4))[..., 2] for i in range(2)]
    n_behind = int((tri_z[1].min(1) <= 1e-4).sum())
    n_straddle = int(((tri_z[1].min(1) <= 1e-4) & (tri_z[1].max(1) > 1e-4)).sum())
    assert len(tri_z[0]) > 2000 and n_behind > 100 and n_straddle > 50, (len(tri_z[0]), n_behind, n_straddle)

    imgs = _t(np.broadcast_to(img[None], (2, 3, h, w)).copy(), dev)
    frame, m2, d2, flow = renderer.forward_warp(imgs, _t(np.broadcast_to(rel, (2, 1, h, w)).copy(), dev), None, None, _t(w2cs, dev), _t(Ks, dev), _t(Ks, dev),
                                                render_depth=True, world_points1=_t(np.broadcast_to(pts, (2, h, w, 3)).copy(), dev),
                                                foreground_masking=True, boundary_mask=_t(np.broadcast_to(bnd[None], (2, h, w)).copy(), dev))
    torch.cuda.synchronize()
    fr_o, m_o, d_o, flow_o, _ = wo.forward_warp(np.broadcast_to(img[None], (2, 3, h, w)), np.broadcast_to(rel, (2, 1, h, w)), np.broadcast_to(pts, (2, h, w, 3)),
                                               w2cs, Ks, render_depth=True, foreground_masking=True, boundary_mask=np.broadcast_to(bnd[None], (2, h, w)),
                                               ray_triangle_fn=wo.ray_triangle_depth_c)
    _, m_plain, _, _, _ = wo.forward_warp(np.broadcast_to(img[None], (2, 3, h, w)), np.broadcast_to(rel, (2, 1, h, w)), np.broadcast_to(pts, (2, h, w, 3)), w2cs, Ks)
    removed = (m_plain != m_o).sum(axis=(1, 2, 3))
    assert removed[0] > 500 and removed[1] > 500, f"mesh occlusion must actually remove pixels in both items: {removed}"
    assert int((flow.cpu().numpy() != flow_o).sum()) == 0
    got_m = m2.cpu().numpy()
    nd = int((got_m != m_o).sum())
    print(f"[fg 704x1280] tris {len(tri_z[0])}, behind camera {n_behind}, straddling {n_straddle}; occluded px {removed.tolist()}; mask px differing {nd}")
    assert nd == 0, f"mask differs on {nd} px"
    err = np.abs(frame.cpu().numpy() - fr_o)
    bad = err > (1e-4 + 1e-3 * np.abs(fr_o))
    assert bad.mean() < 1e-5 and err.max() < 5e-2, (float(bad.mean()), float(err.max()))
    derr = np.abs(d2.cpu().numpy() - d_o)
    assert (derr > (1e-4 + 1e-3 * np.abs(d_o))).mean() < 1e-5


def test_identity_camera_is_idempotent_on_valid_pixels():
    """Size-independent property: rendering the cache from its own camera returns the source image wherever the mask is 1."""
    from gen3c_amd import renderer
    dev = torch.device("cuda:0")
    h, w = 704, 1280
    depth, img, K = _scene(h, w)
    cache = renderer.Cache3D_Base(input_image=_t(img, dev)[None], input_depth=_t(depth, dev)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                  input_intrinsics=_t(K, dev)[None], input_format=["B", "C", "H", "W"])
    pix, msk = cache.render_cache(torch.eye(4, device=dev)[None, None], _t(K, dev)[None, None])
    torch.cuda.synchronize()
    m = msk[0, 0, 0, 0] > 0
    assert float(m.float().mean()) > 0.99
    err = (pix[0, 0, 0] - _t(img, dev)).abs()[:, m]
    assert float(err.max()) < 5e-2 and float(err.mean()) < 1e-3  # max sits on disc silhouettes (fore/background blend)


@pytest.mark.parametrize("zoom", [1.0, 0.25])
def test_window_splat_matches_atomic_splat(zoom):
    """The default splat (source tiles store their destination windows, a destination-owning pass sums them in tile order and resolves)
    against the two-call form that adds everything into the global accumulator with atomics: identical masks, colours within the
    atomics' own order sensitivity (both forms accumulate with fp32 LDS atomics, so neither is bit-reproducible). zoom 0.25 shrinks the image of the scene so
    that many source tiles land in one destination tile (long overlap lists) and windows cover far more than their own texels."""
    from gen3c_amd import renderer
    dev = torch.device("cuda:0")
    h, w = 704, 1280
    depth, img, K = _scene(h, w)
    cache = renderer.Cache3D_Buffer(frame_buffer_max=2, noise_aug_strength=0, input_image=_t(img, dev)[None], input_depth=_t(depth, dev)[None, None],
                                    input_w2c=torch.eye(4, device=dev)[None], input_intrinsics=_t(K, dev)[None], filter_points_threshold=0.05,
                                    foreground_masking=False, input_format=["B", "C", "H", "W"])
    w2cs = np.stack([np.eye(4, dtype=np.float32) for _ in range(4)])
    w2cs[:, 0, 3] = [0.05, 0.3, -0.4, 0.0]
    w2cs[3, 2, 3] = 6.0  # pull the camera back: the whole scene shrinks towards the image centre
    Kz = K.copy()
    Kz[0, 0] *= zoom
    Kz[1, 1] *= zoom
    Ks = _t(Kz, dev)[None, None].expand(1, 4, 3, 3).contiguous()
    outs = {}
    for mode in (True, False):
        renderer._WINDOW_SPLAT = mode
        try:
            pix, msk = cache.render_cache(_t(w2cs, dev)[None], Ks, render_depth=False)
            dep, _ = cache.render_cache(_t(w2cs, dev)[None], Ks, render_depth=True)
        finally:
            renderer._WINDOW_SPLAT = True
        torch.cuda.synchronize()
        outs.setdefault(mode, []).append((pix.clone(), msk.clone(), dep.clone()))
    (p1, m1, d1), = outs[True]
    (p2, m2, d2), = outs[False]
    assert torch.equal(m1, m2), f"masks differ on {int((m1 != m2).sum())} px"
    err = (p1 - p2).abs()
    bad = err > (1e-4 + 1e-3 * p2.abs())
    print(f"[window vs atomic splat zoom={zoom}] masks equal, coverage {float(m1.mean()):.3f}; colour outliers {int(bad.sum())}/{bad.numel()}, max abs {float(err.max()):.3e}")
    assert float(bad.float().mean()) < 1e-5 and float(err.max()) < 5e-2
    torch.testing.assert_close(d1, d2, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("fg", [False, True])
@pytest.mark.parametrize("n_buf", [1, 2])
def test_render_items_call_matches_the_expand_and_loop_form(fg, n_buf):
    """Cache3D.render_cache through g3_render_items_f32 (items name their source view, cached self-cleaning workspace, one preallocated output)
    against the reference-shaped path (sources expanded per item, forward_warp per chunk): masks (=> every splat index / occlusion decision)
    identical, colours / depths equal up to the summation order of the float atomics. Rendered TWICE through the cached workspace: the second
    render must not see anything the first left behind (the accumulator cleans itself) - with cameras that push corners out of the windows.
    With foreground masking the items call rasterises a thread-compacted patch list, sweeps camera-plane triangles in a separate pass and applies
    the occlusion inside the resolve pass; the expand form is the wave-per-patch rasteriser + mesh_apply_kernel that the 704x1280 test above
    holds against the brute force."""
    from gen3c_amd import renderer
    dev = torch.device("cuda:0")
    h, w, Fn = 96, 160, 7  # odd item count for N = 1: a ragged last pair / chunk
    depth, img, K = _scene(h, w)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=t(img)[None], input_depth=t(depth)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                    input_intrinsics=t(K)[None], filter_points_threshold=0.05, foreground_masking=fg, input_format=["B", "C", "H", "W"])
    if n_buf == 2:
        w2 = torch.eye(4, device=dev)
        w2[0, 3] = -0.15
        cache.update_cache(t(img[::-1].copy())[None], t(depth * 1.07)[None, None], w2[None], new_intrinsics=t(K)[None], depth_alignment=False)
    cams = [_cam(tx=0.05 * i, tz=-0.4 * (i % 3), yaw=0.03 * i) for i in range(Fn)]  # zooming in: corners leave windows
    cams[5] = _cam(tx=-0.7, tz=-1.68)  # camera plane through the boundary skirt: triangles behind the camera / straddling z = 0 (the heavy list of the items call)
    if fg:
        from oracle import warp_oracle as wo
        pts_o = wo.unproject_points(depth[None, None], np.eye(4, dtype=np.float32)[None], K[None])
        _, camp = wo.project_points(pts_o, cams[5][None], K[None])
        tz_ = wo.mesh_triangles(*wo.downsample_points_mask(camp[0], ~wo.reliable_depth_mask(depth[None, None])[0, 0], 4))[..., 2]
        assert int((tz_.min(1) <= 1e-4).sum()) > 10, "the case must contain triangles that touch the camera plane"
    w2cs = torch.stack([torch.from_numpy(c) for c in cams])[None].to(dev)
    Ks = t(K)[None, None].expand(1, Fn, 3, 3).contiguous()
    outs = {}
    for items in (True, False, True):
        renderer._ITEMS_CALL = items
        try:
            pix, msk = renderer.Cache3D_Base.render_cache(cache, w2cs, Ks, items_per_launch=4)
            dep, msk_d = renderer.Cache3D_Base.render_cache(cache, w2cs, Ks, render_depth=True, items_per_launch=4)
        finally:
            renderer._ITEMS_CALL = True
        torch.cuda.synchronize()
        assert torch.equal(msk, msk_d)
        outs.setdefault(items, []).append((pix.clone(), msk.clone(), dep.clone()))
    (p1, m1, d1), (p3, m3, d3) = outs[True]
    p2, m2, d2 = outs[False][0]
    assert torch.equal(m1, m2) and torch.equal(m1, m3), "masks differ between the items call and the expand-and-loop form"
    assert 0.2 < float(m1.mean()) < 1.0
    for a, b in ((p1, p2), (p3, p2), (d1, d2), (d3, d2)):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), f"max diff {float((a - b).abs().max()):.3e}"


def test_render_items_host_memo_follows_in_place_edits_of_the_callers_tensors():
    """Cache3D._render_items keeps the intrinsics' host inverse, the uint8 boundary mask and the item -> source indices between calls, keyed by
    tensor identity + in-place version. An in-place edit of the intrinsics tensor between two renders must be seen (a stale inverse would leave
    the occlusion rays of the first call), and so must an edit of the boundary mask."""
    from gen3c_amd import renderer
    dev = torch.device("cuda:0")
    h, w, Fn = 96, 160, 4
    depth, img, K = _scene(h, w)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mk = lambda: renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=t(img)[None], input_depth=t(depth)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                         input_intrinsics=t(K)[None], filter_points_threshold=0.05, foreground_masking=True, input_format=["B", "C", "H", "W"])
    cache = mk()
    w2cs = torch.stack([torch.from_numpy(_cam(tx=0.1 * (i + 1))) for i in range(Fn)])[None].to(dev)
    Ks = t(K)[None, None].expand(1, Fn, 3, 3).contiguous()
    p0, m0 = cache.render_cache(w2cs, Ks)
    p0b, m0b = cache.render_cache(w2cs, Ks)  # memo hit
    assert torch.equal(m0, m0b) and torch.allclose(p0, p0b, rtol=1e-4, atol=1e-5)
    Ks[:, :, 0, 0] *= 1.25  # in place: same tensor object, same storage
    Ks[:, :, 1, 1] *= 1.25
    p1, m1 = cache.render_cache(w2cs, Ks)
    p_ref, m_ref = mk().render_cache(w2cs, Ks.clone())
    torch.cuda.synchronize()
    assert not torch.equal(m1, m0), "the edit must change the render"
    assert torch.equal(m1, m_ref) and torch.allclose(p1, p_ref, rtol=1e-4, atol=1e-5), "stale intrinsics / inverse reused after an in-place edit"
    cache.boundary_mask.zero_()  # no boundary patches any more: nothing is occluded
    _, m2 = cache.render_cache(w2cs, Ks)
    fresh = mk()
    fresh.boundary_mask.zero_()
    _, m2_ref = fresh.render_cache(w2cs, Ks.clone())
    assert torch.equal(m2, m2_ref) and not torch.equal(m2, m1), "stale boundary mask reused after an in-place edit"


@pytest.mark.parametrize("case", ["all_masked", "behind_camera", "single_pixel", "camera_plane_points", "ragged_zoom_out"])
def test_render_items_edge_cases_vs_oracle(case):
    """Edge inputs through Cache3D.render_cache (g3_render_items_f32) at a frame size that is not a multiple of the 32 x 32 tiles (40 x 72), against
    the oracle: nothing valid (empty input mask / every point behind the target camera), ONE valid pixel, points ON the camera plane (z = 0: masked,
    their u = x / 1e-7 lands ~1e7 pixels outside and is clamped) and just in front of it (z = 1e-9: valid, clamped into the cropped border),
    and a zoom-out that folds many source pixels into few texels (many-to-one
    overlaps: the register merge must fall back to atomics without losing a contribution). Masks bit-exact, colours within the renderer's bound."""
    from gen3c_amd import renderer
    from oracle import warp_oracle as wo
    dev = torch.device("cuda:0")
    h, w = 40, 72
    depth, img, K = _scene(h, w)
    K = K.copy()
    K[0, 0] = K[1, 1] = 60.0
    cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=_t(img, dev)[None], input_depth=_t(depth, dev)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                    input_intrinsics=_t(K, dev)[None], filter_points_threshold=0.05, foreground_masking=False, input_format=["B", "C", "H", "W"])
    pts = wo.unproject_points(depth[None, None], np.eye(4, dtype=np.float32)[None], K[None])
    msk = wo.reliable_depth_mask(depth[None, None], ratio_thresh=0.05).astype(np.float32)
    cams = [_cam(tx=0.2), _cam(tx=-0.3, tz=-0.5)]
    if case == "all_masked":
        msk = np.zeros_like(msk)
    elif case == "behind_camera":
        cams = [_cam(tz=-30.0), _cam(tz=-50.0)]  # every point ends up at z < 0
    elif case == "single_pixel":
        msk = np.zeros_like(msk)
        msk[0, 0, 17, 33] = 1.0
    elif case == "camera_plane_points":
        pts = pts.copy()
        pts[0, 5:9, 10:30, 2] = 0.0
        pts[0, 20, 40:44, 2] = 1e-9
        cams = [_cam(tx=0.2), _cam(tx=-0.3)]  # no z translation: the edited points stay at z = 0 / 1e-9 in both target cameras
    elif case == "ragged_zoom_out":
        cams = [_cam(tz=6.0), _cam(tx=0.4, tz=9.0)]  # the scene shrinks to a fraction of the frame
    cache.input_points = _t(pts, dev).reshape(cache.input_points.shape).to(cache.input_points.dtype)
    cache.input_mask = _t(msk, dev).reshape(cache.input_mask.shape).to(cache.input_mask.dtype)
    w2cs = np.stack(cams)
    pix, m = cache.render_cache(_t(w2cs, dev)[None], _t(K, dev)[None, None].expand(1, 2, 3, 3))
    pix2, m2 = cache.render_cache(_t(w2cs, dev)[None], _t(K, dev)[None, None].expand(1, 2, 3, 3))  # cached workspace: must be clean again
    torch.cuda.synchronize()
    with np.errstate(all="ignore"):
        fr_o, m_o, _, _, _ = wo.forward_warp(np.broadcast_to(img[None], (2, 3, h, w)), np.broadcast_to(msk, (2, 1, h, w)), np.broadcast_to(pts, (2, h, w, 3)), w2cs,
                                             np.broadcast_to(K[None], (2, 3, 3)))
    got_m, got = m[0, :, 0].cpu().numpy(), pix[0, :, 0].cpu().numpy()
    assert np.array_equal(got_m, m_o), f"{case}: mask differs on {(got_m != m_o).sum()} px"
    assert torch.equal(m, m2) and torch.allclose(pix, pix2, rtol=1e-4, atol=1e-5), f"{case}: second render through the cached workspace differs"
    if case in ("all_masked", "behind_camera"):
        assert got_m.sum() == 0 and np.all(got == -1.0)
    if case == "single_pixel":
        assert 1 <= got_m.sum() <= 8
    if case == "ragged_zoom_out":
        assert 0 < got_m.mean() < 0.5
    err = np.abs(got - fr_o)
    bad = err > (1e-4 + 1e-3 * np.abs(fr_o))
    print(f"[render edge {case}] valid px {int(got_m.sum())}; colour outliers {int(bad.sum())}/{bad.size}, max abs err {np.nanmax(err):.3e}")
    assert bad.mean() < 1e-3 and np.nanmax(err) < 5e-2


@pytest.mark.parametrize("fg", [False, True])
def test_fused_projection_splat_matches_the_three_plane_form(fg):
    """Round 5: g3_render_items_f32 evaluates the projection INSIDE the splat (warp_splat_windows_kernel<true>, z-only pre-pass for the group maxima) instead
    of writing z / flow / validity planes and reading them back. Same arithmetic, operation for operation: masks (every splat index, every validity and
    occlusion decision) identical to the three-plane form (option render_fused = 0); colours / depths equal up to the order of the float atomics."""
    from gen3c_amd import ops, renderer
    dev = torch.device("cuda:0")
    h, w, Fn = 352, 640, 6
    depth, img, K = _scene(h, w)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=t(img)[None], input_depth=t(depth)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                    input_intrinsics=t(K)[None], filter_points_threshold=0.05, foreground_masking=fg, input_format=["B", "C", "H", "W"])
    cams = [_cam(tx=0.06 * i, tz=-0.3 * (i % 3), yaw=0.02 * i) for i in range(Fn)]
    w2cs = torch.stack([torch.from_numpy(c) for c in cams])[None].to(dev)
    Ks = t(K)[None, None].expand(1, Fn, 3, 3).contiguous()
    outs = {}
    try:
        for fused in (1, 0, 1):
            ops.set_option("render_fused", fused)
            pix, msk = cache.render_cache(w2cs, Ks)
            dep, msk_d = cache.render_cache(w2cs, Ks, render_depth=True)
            torch.cuda.synchronize()
            assert torch.equal(msk, msk_d)
            outs.setdefault(fused, []).append((pix.clone(), msk.clone(), dep.clone()))
    finally:
        ops.set_option("render_fused", 1)
    (p1, m1, d1), (p3, m3, d3) = outs[1]
    p0, m0, d0 = outs[0][0]
    assert torch.equal(m1, m0) and torch.equal(m3, m0), "masks differ between the fused and the three-plane form"
    assert 0.2 < float(m1.mean()) < 1.0
    for a, b in ((p1, p0), (p3, p0), (d1, d0), (d3, d0)):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), f"max diff {float((a - b).abs().max()):.3e}"


@pytest.mark.parametrize("fg", [False, True])
@pytest.mark.parametrize("hw", [(352, 640), (203, 333)])
def test_single_writer_splat_matches_the_window_round_trip(fg, hw):
    """Round 5, three forms of g3_render_items_f32's splat -> gather hand-over:
      o  (render_full_extent = 0) tiles publish rectangles clamped to their window; an item with ANY corner beyond a window is stamped dirty and the
         gather pass reads the dense accumulator for every pixel of it (round 4);
      f  (default) tiles publish their unclamped rectangle and the gather pass reads the accumulator for exactly the texels beyond a window;
      x  (render_exclusive = 1) a pre-pass (warp_extent_kernel) publishes every rectangle before any tile splats; the splat resolves the texels only
         ITS tile reaches straight into frame / mask / depth and the window workspace carries the shared texels only.
    The decisions (splat indices, validity, occlusion) are the same code on the same floats: masks identical; colours / depths equal up to the order of
    the LDS float atomics inside a window. The item list holds a 1.6x zoom (rectangles wider than the window everywhere) next to views with depth edges
    (wide rectangles along them), and an odd frame size with partial tiles."""
    from gen3c_amd import ops, renderer
    dev = torch.device("cuda:0")
    h, w = hw
    Fn = 6
    depth, img, K = _scene(h, w)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=t(img)[None], input_depth=t(depth)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                    input_intrinsics=t(K)[None], filter_points_threshold=0.05, foreground_masking=fg, input_format=["B", "C", "H", "W"])
    cams = [_cam(tx=0.06 * i, tz=-0.3 * (i % 3), yaw=0.02 * i) for i in range(Fn)]
    w2cs = torch.stack([torch.from_numpy(c) for c in cams])[None].to(dev)
    Ks = t(K)[None, None].expand(1, Fn, 3, 3).contiguous().clone()
    Ks[0, 2, :2, :2] *= 1.6  # zoomed view: destination rectangles of ~52 texels > the 40-texel window
    forms = {"x": (1, 1), "f": (0, 1), "o": (0, 0)}
    outs = {}
    try:
        for name in ("x", "f", "o", "x", "f"):
            ops.set_option("render_exclusive", forms[name][0])
            ops.set_option("render_full_extent", forms[name][1])
            pix, msk = cache.render_cache(w2cs, Ks)
            dep, msk_d = cache.render_cache(w2cs, Ks, render_depth=True)
            torch.cuda.synchronize()
            assert torch.equal(msk, msk_d)
            outs.setdefault(name, []).append((pix.clone(), msk.clone(), dep.clone()))
    finally:
        ops.set_option("render_exclusive", 0)
        ops.set_option("render_full_extent", 1)
    p0, m0, d0 = outs["o"][0]
    assert 0.2 < float(m0.mean()) < 1.0
    for name in ("x", "f"):
        for (p1, m1, d1) in outs[name]:
            assert torch.equal(m1, m0), f"form {name}: masks differ from the round-4 form"
            for a, b in ((p1, p0), (d1, d0)):
                assert torch.isfinite(a).all()
                assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), f"form {name}: max diff {float((a - b).abs().max()):.3e}"
        same = float((outs[name][0][0] == p0).float().mean())
        print(f"[hand-over form {name} {h}x{w} fg={fg}] coverage {float(m0.mean()):.3f}; colour values bitwise equal to the round-4 form: {same:.4f}")
        assert same > 0.98
