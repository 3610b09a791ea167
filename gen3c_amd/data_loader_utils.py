"""Input loaders of the dynamic (video-to-video) entry point: counterpart of
cosmos_predict1/diffusion/inference/data_loader_utils.py:137-193.

Formats (same tensors as the reference returns: image [F,3,H,W] in [-1,1], depth [F,1,H,W], mask [F,1,H,W], w2c [F,4,4],
intrinsics [F,3,3]):
  * packaged   `<file>.pt`  - torch.save of the 5 tensors (load_data_packaged_format, :171-184)
  * distributed `<dir>/`    - depth.npz['depth'], mask.npz['mask'], camera.npz['w2c','intrinsics'] and the RGB frames as
                              rgb.npz['rgb'] uint8 [F,H,W,3] (this image has no video decoder; rgb.mp4 is read only when
                              OpenCV is importable, as load_data_distributed_format :137-168 does)
ViPE folders (vipe_utils.py:172-270) are read by gen3c_amd/vipe_utils.py.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch


def load_data_packaged_format(pt_path):
    data = torch.load(pt_path, map_location="cpu", weights_only=True)
    if len(data) != 5:
        raise ValueError(f"Expected 5 tensors in pt file, got {len(data)}")
    return tuple(data)


def _read_rgb(data_path: Path) -> np.ndarray:
    if (data_path / "rgb.npz").exists():
        return np.asarray(np.load(data_path / "rgb.npz")["rgb"], dtype=np.uint8)
    mp4 = data_path / "rgb.mp4"
    if mp4.exists():
        try:
            import cv2
        except ImportError as e:  # no silent substitute for the frames
            raise RuntimeError(f"{mp4} needs OpenCV to decode; provide rgb.npz['rgb'] uint8 [F,H,W,3] instead") from e
        cap, frames = cv2.VideoCapture(str(mp4)), []
        while True:
            ret, frame = cap.read()
            if not ret:
                break
            frames.append(cv2.cvtColor(frame, cv2.COLOR_BGR2RGB))
        cap.release()
        return np.stack(frames, axis=0)
    raise FileNotFoundError(f"neither rgb.npz nor rgb.mp4 in {data_path}")


def load_data_distributed_format(data_dir):
    data_path = Path(data_dir)
    frames = _read_rgb(data_path)
    image = torch.from_numpy(frames).permute(0, 3, 1, 2).float() / 127.5 - 1.0  # [0,255] -> [-1,1] (:150)
    depth = torch.from_numpy(np.load(data_path / "depth.npz")["depth"].astype(np.float32)).unsqueeze(1)
    mask = torch.from_numpy(np.load(data_path / "mask.npz")["mask"].astype(np.float32)).unsqueeze(1)
    cam = np.load(data_path / "camera.npz")
    return image, depth, mask, torch.from_numpy(cam["w2c"]).float(), torch.from_numpy(cam["intrinsics"]).float()


def load_data_auto_detect(input_path):
    input_path = Path(input_path)
    if input_path.is_file() and input_path.suffix == ".pt":
        return load_data_packaged_format(input_path)
    if input_path.is_dir():
        return load_data_distributed_format(input_path)
    raise ValueError(f"Invalid input path: {input_path}")
