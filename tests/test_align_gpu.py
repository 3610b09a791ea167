"""GPU: csrc/align.hip (through gen3c_amd.camera_utils.align_depth) against the oracle and the reference golden."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(Path(__file__).parent / "golden" / "align_small.npz")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_rigid_matches_reference_golden():
    from gen3c_amd.camera_utils import align_depth
    r = align_depth(_t(G["source"]), _t(G["target"]), _t(G["mask"])).cpu().numpy()
    rel = np.abs(r / G["rigid"] - 1).max()
    print(f"[align rigid] max rel vs reference {rel:.2e}")
    assert rel < 2e-6


def test_rigid_without_mask_and_with_invalid_pixels():
    from gen3c_amd.camera_utils import align_depth
    from oracle import align_oracle as ao
    src, tgt = G["source"].copy(), G["target"].copy()
    src[3, 5] = 0.0  # 1/0 = inf is a valid (largest) inverse depth for torch.quantile; must not poison the fit
    ref = ao.align_depth(src, tgt, None)
    with np.errstate(divide="ignore"):
        r = align_depth(_t(src), _t(tgt), None).cpu().numpy()
    assert np.abs(r / ref - 1)[np.isfinite(ref) & (ref != 0)].max() < 2e-6


@pytest.mark.parametrize("iters,tol_max,tol_mean", [(1, 2e-6, 1e-6), (3, 2e-3, 1e-5), (100, 8e-3, 3e-4)])
def test_non_rigid_matches_reference_golden(iters, tol_max, tol_mean):
    # same tolerances as the oracle-vs-reference CPU test: sign()-driven Adam steps agree to O(lr = 1e-3)
    from gen3c_amd.camera_utils import align_depth
    r = align_depth(_t(G["source"]), _t(G["target"]), _t(G["mask"]), k=_t(G["K"]), c2w=_t(G["c2w"]), alignment_method="non_rigid",
                    num_iters=iters).cpu().numpy()
    rel = np.abs(r / G[f"non_rigid_{iters}"] - 1)
    print(f"[align non_rigid {iters}] max rel {rel.max():.2e} mean rel {rel.mean():.2e}")
    assert rel.max() < tol_max and rel.mean() < tol_mean


def test_non_rigid_full_resolution_against_oracle():
    """704 x 1280 (BASELINE size): quantile selection over 0.9 M keys and 20 Adam steps against the numpy oracle."""
    from gen3c_amd.camera_utils import align_depth
    from oracle import align_oracle as ao
    rng = np.random.RandomState(3)
    H, W = 704, 1280
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    tgt = (2.0 + 0.001 * xs + 0.0015 * ys).astype(np.float32)
    tgt[((xs - 600) ** 2 + (ys - 300) ** 2) < 150 ** 2] = 1.3
    src = (1.0 / (0.7 / tgt + 0.04) * (1.0 + 0.03 * np.sin(xs / 90.0) * np.cos(ys / 70.0)) * (1 + 0.001 * rng.randn(H, W))).astype(np.float32)
    mask = rng.rand(H, W) < 0.85
    tgt[~mask] = 0
    K = np.array([[1000.0, 0, 640], [0, 1000.0, 352], [0, 0, 1]], np.float32)
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 3] = [0.3, 0.0, 0.1]
    ref_rigid = ao.align_depth(src, tgt, mask)
    r = align_depth(_t(src), _t(tgt), _t(mask)).cpu().numpy()
    assert np.abs(r / ref_rigid - 1).max() < 5e-6
    ref = ao.align_depth(src, tgt, mask, k=K, c2w=c2w, alignment_method="non_rigid", num_iters=20)
    r = align_depth(_t(src), _t(tgt), _t(mask), k=_t(K), c2w=_t(c2w), alignment_method="non_rigid", num_iters=20).cpu().numpy()
    rel = np.abs(r / ref - 1)
    print(f"[align 704x1280 x20] max rel {rel.max():.2e} mean rel {rel.mean():.2e} moved {np.abs(ref / ref_rigid - 1).max():.3f}")
    assert rel.max() < 5e-3 and rel.mean() < 1e-5
