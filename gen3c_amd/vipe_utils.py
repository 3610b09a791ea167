"""ViPE clip loader of the dynamic entry point: counterpart of cosmos_predict1/diffusion/inference/vipe_utils.py:172-270.

A ViPE result folder holds, per clip `<base>`:  rgb/<base>.mp4, depth/<base>.zip (one half-float `Z` EXR per frame, `%05d.exr`),
pose/<base>.npz and intrinsics/<base>.npz (`inds` = frame numbers, `data` = c2w 4x4 / K 3x3 or fx,fy,cx,cy), optionally
mask/<base>.zip (`%05d.png`). `load_vipe_data` returns what the reference returns: frames [T,3,704,1280] in [-1,1], depth
[T,1,704,1280], mask [T,1,704,1280], w2c [T,4,4], K [T,3,3] - resize to 720x1280 (bilinear; mask nearest), centre crop to 704x1280,
intrinsics rescaled and shifted accordingly, the last available frame repeated when the clip is shorter than `num_frames`.

Decoders: the reference needs `decord` (mp4) and `OpenEXR` (depth); both are used here when importable. Neither ships in this
image, so each stream also has a decoder-free sidecar that is read FIRST when present (written by whoever exported the clip):
    rgb/<base>.npz    `rgb`   uint8 [F,H,W,3]     (instead of decoding rgb/<base>.mp4)
    depth/<base>.npz  `depth` float [F,H,W]       (instead of depth/<base>.zip)
    mask/<base>.npz   `mask`  [F,H,W]             (instead of mask/<base>.zip)
Everything after decoding - index clamping / repetition, pose inversion, intrinsics adjustment, resize, crop, value ranges - is the
reference's arithmetic and is pinned to it (tests/golden/vipe_small.npz, tools/gen_golden_vipe.py).
"""
from __future__ import annotations

import os
import zipfile
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _center_crop(t: torch.Tensor, crop_h: int, crop_w: int) -> torch.Tensor:
    h, w = t.shape[-2:]
    top, left = max((h - crop_h) // 2, 0), max((w - crop_w) // 2, 0)
    return t[..., top:top + crop_h, left:left + crop_w]


def adjust_intrinsics_for_resize_and_crop(K: np.ndarray, src_hw, resize_hw, crop_hw) -> np.ndarray:
    """vipe_utils.py:32-58: scale fx,cx by resize_w/src_w and fy,cy by resize_h/src_h, then subtract the crop offsets."""
    (sh, sw), (rh, rw), (ch, cw) = src_hw, resize_hw, crop_hw
    out = K.copy()
    sx, sy = rw / float(sw), rh / float(sh)
    out[0, 0] *= sx
    out[1, 1] *= sy
    out[0, 2] *= sx
    out[1, 2] *= sy
    out[0, 2] -= max((rw - cw) // 2, 0)
    out[1, 2] -= max((rh - ch) // 2, 0)
    return out


def _indexed_entry(npz_path: str, frame_idx: int, what: str) -> np.ndarray:
    """`inds` (sorted frame numbers) / `data` lookup shared by the pose and intrinsics files (vipe_utils.py:67-102)."""
    data = np.load(npz_path)
    inds, arr = data["inds"], data["data"]
    pos = int(np.searchsorted(inds, frame_idx))
    if not (0 <= pos < len(inds)) or int(inds[pos]) != int(frame_idx):
        raise FileNotFoundError(f"{what} for frame {frame_idx} not found in {npz_path}")
    return arr[pos]


def load_pose_matrix_for_frame(pose_npz_path: str, frame_idx: int) -> np.ndarray:
    mat = _indexed_entry(pose_npz_path, frame_idx, "Pose")
    if mat.shape == (16,):
        mat = mat.reshape(4, 4)
    assert mat.shape == (4, 4)
    return mat.astype(np.float32)


def load_intrinsics_for_frame(intrinsics_npz_path: str, frame_idx: int) -> np.ndarray:
    item = _indexed_entry(intrinsics_npz_path, frame_idx, "Intrinsics")
    if item.shape == (3, 3):
        return item.astype(np.float32)
    if item.shape[-1] == 4:
        fx, fy, cx, cy = (float(v) for v in item)
        return np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]], dtype=np.float32)
    raise ValueError(f"Unsupported intrinsics format {item.shape} in {intrinsics_npz_path}")


def find_clip_paths(vipe_root_or_mp4: str, video_idx: int = 0):
    """vipe_utils.py:143-169 (+ the sidecar: an rgb/<base>.npz counts as the clip when no mp4 is there)."""
    if vipe_root_or_mp4.endswith((".mp4", ".npz")):
        rgb_path = vipe_root_or_mp4
        root = os.path.dirname(os.path.dirname(rgb_path))
    else:
        rgb_dir = os.path.join(vipe_root_or_mp4, "rgb")
        names = sorted(os.listdir(rgb_dir))
        clips = [f for f in names if f.endswith(".mp4")] or [f for f in names if f.endswith(".npz")]
        if not clips:
            raise FileNotFoundError(f"No mp4 found under {rgb_dir}")
        rgb_path = os.path.join(rgb_dir, clips[video_idx])
        root = vipe_root_or_mp4
    base = os.path.splitext(os.path.basename(rgb_path))[0]
    return dict(rgb=rgb_path, base=base, root=root, depth_zip=os.path.join(root, "depth", f"{base}.zip"),
                depth_npz=os.path.join(root, "depth", f"{base}.npz"), pose=os.path.join(root, "pose", f"{base}.npz"),
                intrinsics=os.path.join(root, "intrinsics", f"{base}.npz"), mask_zip=os.path.join(root, "mask", f"{base}.zip"),
                mask_npz=os.path.join(root, "mask", f"{base}.npz"))


class _Frames:
    """Random access to the clip's RGB frames: sidecar npz, else decord, else OpenCV."""

    def __init__(self, paths: dict):
        side = os.path.join(paths["root"], "rgb", paths["base"] + ".npz")
        self._arr = self._vr = None
        if os.path.exists(side):
            self._arr = np.load(side)["rgb"]
            return
        try:
            from decord import VideoReader
            self._vr = VideoReader(paths["rgb"], num_threads=4)
            return
        except ImportError:
            pass
        try:
            import cv2
        except ImportError as e:
            raise ImportError(f"reading {paths['rgb']} needs decord (as the reference) or OpenCV; neither is importable and there is no "
                              f"decoder-free sidecar {side} (`rgb` uint8 [F,H,W,3])") from e
        cap, frames = cv2.VideoCapture(paths["rgb"]), []
        while True:
            ok, fr = cap.read()
            if not ok:
                break
            frames.append(cv2.cvtColor(fr, cv2.COLOR_BGR2RGB))
        cap.release()
        self._arr = np.stack(frames, 0)

    def __len__(self) -> int:
        return len(self._arr) if self._arr is not None else len(self._vr)

    def get_batch(self, indices: List[int]) -> np.ndarray:
        if self._arr is not None:
            return np.asarray(self._arr[indices])
        batch = self._vr.get_batch(indices)
        return batch.asnumpy() if hasattr(batch, "asnumpy") else batch.numpy()


def _read_depth(paths: dict, frame_idx: int) -> np.ndarray:
    if os.path.exists(paths["depth_npz"]):
        return np.asarray(np.load(paths["depth_npz"])["depth"][frame_idx], dtype=np.float32)
    try:
        import OpenEXR
    except ImportError as e:
        raise ImportError(f"reading {paths['depth_zip']} needs OpenEXR (as the reference); it is not importable and there is no "
                          f"decoder-free sidecar {paths['depth_npz']} (`depth` [F,H,W])") from e
    with zipfile.ZipFile(paths["depth_zip"], "r") as zf, zf.open(f"{frame_idx:05d}.exr", "r") as f:
        exr = OpenEXR.InputFile(f)
        dw = exr.header()["dataWindow"]
        h, w = dw.max.y - dw.min.y + 1, dw.max.x - dw.min.x + 1
        return np.frombuffer(exr.channel("Z"), np.float16).astype(np.float32).reshape(h, w)


def _read_mask(paths: dict, frame_idx: int) -> Optional[np.ndarray]:
    if os.path.exists(paths["mask_npz"]):
        return (np.asarray(np.load(paths["mask_npz"])["mask"][frame_idx]) > 0).astype(np.float32)
    if not os.path.exists(paths["mask_zip"]):
        return None
    from PIL import Image
    with zipfile.ZipFile(paths["mask_zip"], "r") as zf:
        try:
            with zf.open(f"{frame_idx:05d}.png", "r") as f:
                img = np.asarray(Image.open(f))
        except KeyError:
            return None
    if img.ndim == 3:
        img = img[..., 0]
    return (img > 0).astype(np.float32)


def load_vipe_data(vipe_root_or_mp4: str, starting_frame_idx: int, resize_hw: Tuple[int, int] = (720, 1280),
                   crop_hw: Tuple[int, int] = (704, 1280), num_frames: int = 121, read_mask: bool = False, video_idx: int = 0):
    """vipe_utils.py:172-270, same arguments and return tuple."""
    paths = find_clip_paths(vipe_root_or_mp4, video_idx=video_idx)
    frames = _Frames(paths)
    total = len(frames)
    if starting_frame_idx >= total:  # beyond the clip: clamp to its last frame
        starting_frame_idx = max(0, total - 1)
    idx = list(range(starting_frame_idx, min(starting_frame_idx + num_frames, total)))
    idx += [total - 1] * (num_frames - len(idx))  # short clip: repeat the last available frame
    rgb = frames.get_batch(idx).astype(np.float32) / 255.0
    src_h, src_w = rgb.shape[1], rgb.shape[2]

    w2cs = np.stack([np.linalg.inv(load_pose_matrix_for_frame(paths["pose"], f)).astype(np.float32) for f in idx], 0)
    Ks = np.stack([adjust_intrinsics_for_resize_and_crop(load_intrinsics_for_frame(paths["intrinsics"], f), (src_h, src_w), resize_hw, crop_hw)
                   for f in idx], 0)
    depth = np.stack([_read_depth(paths, f) for f in idx], 0)
    masks = []
    for f in idx:
        m = _read_mask(paths, f) if read_mask else None
        masks.append(np.ones((src_h, src_w), np.float32) if m is None else m.astype(np.float32))

    frames_t = torch.from_numpy(rgb).permute(0, 3, 1, 2).contiguous()
    depth_t = torch.from_numpy(depth).unsqueeze(1).contiguous()
    mask_t = torch.from_numpy(np.stack(masks, 0)).unsqueeze(1).contiguous()
    frames_t = _center_crop(F.interpolate(frames_t, size=resize_hw, mode="bilinear", align_corners=False), *crop_hw)
    depth_t = _center_crop(F.interpolate(depth_t, size=resize_hw, mode="bilinear", align_corners=False), *crop_hw)
    mask_t = _center_crop(F.interpolate(mask_t, size=resize_hw, mode="nearest"), *crop_hw)
    return frames_t * 2.0 - 1.0, depth_t, mask_t, torch.from_numpy(w2cs).contiguous(), torch.from_numpy(Ks).contiguous()
