"""CPU: pins oracle/warp_oracle.py against outputs of the reference's own forward_warp (tests/golden/warp_*.npz)."""
import numpy as np
import pytest

from oracle import warp_oracle
from tests.golden_io import GOLD


def _load(name):
    return dict(np.load(GOLD / f"{name}.npz"))


@pytest.mark.parametrize("name", ["warp_small", "warp_mid"])
def test_unproject_and_reliability_mask(name):
    z = _load(name)
    h, w = int(z["h"]), int(z["w"])
    depth = z["depth"][None, None]
    pts = warp_oracle.unproject_points(depth, np.eye(4, dtype=np.float32)[None], z["K"][None])
    np.testing.assert_allclose(pts[0], z["points"], rtol=1e-6, atol=1e-6)
    rel = warp_oracle.reliable_depth_mask(depth, ratio_thresh=0.05)
    assert np.array_equal(rel[0, 0], z["reliable"])
    assert np.array_equal(~warp_oracle.reliable_depth_mask(depth)[0, 0], z["boundary"])


@pytest.mark.parametrize("name", ["warp_small", "warp_mid"])
@pytest.mark.parametrize("fg", [False, True])
def test_forward_warp_matches_reference(name, fg):
    z = _load(name)
    h, w = int(z["h"]), int(z["w"])
    b = 2
    imgs = np.broadcast_to(z["image"][None], (b, 3, h, w)).copy()
    pts = np.broadcast_to(z["points"][None], (b, h, w, 3)).copy()
    mask = np.broadcast_to(z["reliable"][None, None].astype(np.float32), (b, 1, h, w)).copy()
    Ks = np.broadcast_to(z["K"][None], (b, 3, 3)).copy()
    bnd = np.broadcast_to(z["boundary"][None], (b, h, w)).copy()
    frame, m2, d2, flow, idx = warp_oracle.forward_warp(imgs, mask, pts, z["w2cs"], Ks, render_depth=True,
                                                        foreground_masking=fg, boundary_mask=bnd if fg else None)
    tag = "fg" if fg else "nofg"
    # flow12 is an OUTPUT of the reference; the splat indices are floor/ceil((flow+grid)+1): bit-exact flow => bit-exact indices
    nbad = int((flow != z[f"{tag}_flow"]).sum())
    assert nbad == 0, f"{nbad} flow values differ from the reference bit patterns"
    assert np.array_equal(m2, z[f"{tag}_mask"]), f"mask differs on {(m2 != z[f'{tag}_mask']).sum()} px"
    np.testing.assert_allclose(frame, z[f"{tag}_frame"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(d2, z[f"{tag}_depth"], rtol=2e-4, atol=2e-5)


def test_c_bruteforce_ray_triangle_is_bit_identical_to_the_numpy_restatement():
    """oracle/c/ray_tri.c (used by the full-resolution mesh-occlusion GPU test) against warp_oracle.ray_triangle_depth on (a) the
    boundary mesh of the golden scene and (b) random triangles around the camera, incl. behind it / through the origin plane /
    degenerate (zero-area) ones."""
    z = _load("warp_mid")
    h, w = int(z["h"]), int(z["w"])
    _, cam = warp_oracle.project_points(z["points"][None], z["w2cs"][:1], z["K"][None])
    pts, m = warp_oracle.downsample_points_mask(cam[0], z["boundary"], 4)
    tris = warp_oracle.mesh_triangles(pts, m)
    rays = warp_oracle.camera_rays(h, w, z["K"])
    assert len(tris) > 50
    a, b = warp_oracle.ray_triangle_depth(rays, tris), warp_oracle.ray_triangle_depth_c(rays, tris)
    assert (a > 0).sum() > 100 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    rs = np.random.RandomState(4)
    tr = (rs.standard_normal((300, 3, 3)) * np.array([1.5, 1.5, 2.0]) + np.array([0, 0, 1.0])).astype(np.float32)
    tr[:10, 2] = tr[:10, 1]  # degenerate
    tr[10:20, :, 2] = 0.0    # in the camera plane
    a, b = warp_oracle.ray_triangle_depth(rays[::3, ::3], tr), warp_oracle.ray_triangle_depth_c(rays[::3, ::3], tr)
    assert (a > 0).mean() > 0.3 and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_ray_triangle_restatements_match_the_reference_warp_kernel_source():
    """tests/golden/warp_kernel_small.npz = the inputs / depth maps of the reference's own `ray_triangle_intersection_warp`
    (ray_triangle_intersection_warp.py:23-105, 194-292) as called by its forward_warp(foreground_masking=True) on the warp_small scene, the
    `@wp.kernel` body executed under tools/wp_standin.py (fp32 scalars / vec3, one rounding per operation; tools/gen_golden_warp_kernel.py).
    The numpy restatement and the C brute force must reproduce the depth maps BIT FOR BIT. (What stays unpinned: the instruction
    selection of a real Warp / NVRTC build, e.g. fma contraction - warp-lang is not installable here.)"""
    z = _load("warp_kernel_small")
    n = int(z["n_calls"])
    assert n == 2
    for i in range(n):
        dirs, verts, faces, depth = z[f"c{i}_dirs"], z[f"c{i}_vertices"], z[f"c{i}_faces"], z[f"c{i}_depth"]
        assert not z[f"c{i}_origins"].any()  # camera-space rays start at the origin (forward_warp_utils_pytorch.py:705-721): the restatements assume it
        tris = verts[faces]
        assert tris.shape[1:] == (3, 3) and len(tris) > 100 and (depth > 0).sum() > 500
        a = warp_oracle.ray_triangle_depth(dirs, tris)
        assert np.array_equal(a.view(np.uint32), depth.view(np.uint32)), f"call {i}: numpy restatement differs on {(a != depth).sum()} rays"
        b = warp_oracle.ray_triangle_depth_c(dirs, tris)
        assert np.array_equal(b.view(np.uint32), depth.view(np.uint32)), f"call {i}: C restatement differs on {(b != depth).sum()} rays"
