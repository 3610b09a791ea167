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
