"""CPU: gen3c_amd.camera_utils against the reference's generate_camera_trajectory outputs (tests/golden/camera_paths.npz)."""
import numpy as np
import torch

from gen3c_amd.camera_utils import generate_camera_trajectory
from tests.golden_io import GOLD


def test_all_trajectories_match_reference():
    z = np.load(GOLD / "camera_paths.npz")
    w2c, K = torch.from_numpy(z["w2c"]), torch.from_numpy(z["K"])
    n = 0
    for key in z.files:
        if ":" not in key:
            continue
        traj, rot = key.split(":")
        w, k = generate_camera_trajectory(traj, w2c, K, 25, 0.3, rot, center_depth=2.5, device="cpu")
        assert w.shape == (1, 25, 4, 4) and k.shape == (1, 25, 3, 3)
        np.testing.assert_array_equal(w.numpy(), z[key])
        n += 1
    assert n == 24
