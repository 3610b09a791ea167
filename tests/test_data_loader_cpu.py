"""CPU: input loaders of the dynamic entry point (gen3c_amd/data_loader_utils.py vs data_loader_utils.py:137-193 of the reference):
both on-disk formats yield the same five tensors with the reference's shapes, dtypes and value ranges."""
import numpy as np
import pytest
import torch

from gen3c_amd import data_loader_utils as dl


def _scene(F=5, H=16, W=24):
    rng = np.random.RandomState(0)
    rgb = rng.randint(0, 256, size=(F, H, W, 3)).astype(np.uint8)
    depth = (1 + rng.rand(F, H, W)).astype(np.float32)
    mask = (rng.rand(F, H, W) < 0.8).astype(np.float32)
    w2c = np.repeat(np.eye(4, dtype=np.float32)[None], F, 0)
    K = np.repeat(np.array([[20, 0, W / 2], [0, 20, H / 2], [0, 0, 1]], np.float32)[None], F, 0)
    return rgb, depth, mask, w2c, K


def test_distributed_and_packaged_formats_agree(tmp_path):
    rgb, depth, mask, w2c, K = _scene()
    d = tmp_path / "clip"
    d.mkdir()
    np.savez(d / "rgb.npz", rgb=rgb)
    np.savez(d / "depth.npz", depth=depth)
    np.savez(d / "mask.npz", mask=mask)
    np.savez(d / "camera.npz", w2c=w2c, intrinsics=K)
    img, dep, msk, cam, intr = dl.load_data_auto_detect(d)
    assert img.shape == (5, 3, 16, 24) and img.dtype == torch.float32 and float(img.min()) >= -1 and float(img.max()) <= 1
    assert torch.equal(img, torch.from_numpy(rgb).permute(0, 3, 1, 2).float() / 127.5 - 1.0)  # [0,255] -> [-1,1] (data_loader_utils.py:150)
    assert dep.shape == (5, 1, 16, 24) and msk.shape == (5, 1, 16, 24) and cam.shape == (5, 4, 4) and intr.shape == (5, 3, 3)
    torch.save((img, dep, msk, cam, intr), tmp_path / "clip.pt")
    packed = dl.load_data_auto_detect(tmp_path / "clip.pt")
    assert all(torch.equal(a, b) for a, b in zip(packed, (img, dep, msk, cam, intr)))


def test_loader_errors_are_loud(tmp_path):
    with pytest.raises(ValueError):
        dl.load_data_auto_detect(tmp_path / "missing.bin")
    d = tmp_path / "empty"
    d.mkdir()
    with pytest.raises(FileNotFoundError):
        dl.load_data_auto_detect(d)
    torch.save((torch.zeros(1), torch.zeros(1)), tmp_path / "short.pt")
    with pytest.raises(ValueError):
        dl.load_data_auto_detect(tmp_path / "short.pt")
