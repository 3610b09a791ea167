"""tests/golden/camera_paths.npz: outputs of the reference's generate_camera_trajectory (CPU) for every trajectory type."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import ref_shims  # noqa: E402

ref_shims.install()
from cosmos_predict1.diffusion.inference.camera_utils import generate_camera_trajectory  # noqa: E402

out = {}
w2c = torch.eye(4)
w2c[:3, 3] = torch.tensor([0.1, -0.05, 0.2])
K = torch.tensor([[1000.0, 0, 640], [0, 1000, 352], [0, 0, 1]])
for traj in ("left", "right", "up", "down", "zoom_in", "zoom_out", "clockwise", "counterclockwise"):
    for rot in ("center_facing", "no_rotation", "trajectory_aligned"):
        w, k = generate_camera_trajectory(traj, w2c, K, 25, 0.3, rot, center_depth=2.5, device="cpu")
        out[f"{traj}:{rot}"] = w.numpy()
out["w2c"] = w2c.numpy(); out["K"] = K.numpy()
np.savez_compressed(ROOT / "tests" / "golden" / "camera_paths.npz", **out)
print(len(out), "entries")
