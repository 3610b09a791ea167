"""Inputs built from the reference's own data files (tests/golden/ref_inputs/, see its README): shared by the CPU and the GPU halves of the
reference-fixture parity tests. PIL + numpy only; nothing here reads /root/reference."""
from pathlib import Path

import numpy as np

REF_IN = Path(__file__).resolve().parent / "golden" / "ref_inputs"
SHA256 = {
    "diffusion_000000.png": "b7e6eab7548c2ede900f8b504a5cef981e0cd0ec38af90dbea3f0db860e002c3",
    "tokenizer_image.png": "27f2261a585eea38a0c9ec16f2ea81a2295b49c5ad6a3e39fc7cfdd1aa39f53b",
}


def load_rgb(name: str, size=None) -> np.ndarray:
    """uint8 [H, W, 3]; size = (W, H) resizes with PIL's LANCZOS filter (what the reference's image loader does before the depth model,
    inference_utils.py `load_image`-style resizing)."""
    from PIL import Image
    im = Image.open(REF_IN / name).convert("RGB")
    if size is not None:
        im = im.resize(size, Image.LANCZOS)
    return np.asarray(im).copy()


def to_unit(a: np.ndarray) -> np.ndarray:
    """uint8 HWC -> float32 CHW in [-1, 1] (the reference reads uint8 / 127.5 - 1)."""
    return (a.astype(np.float32) / 127.5 - 1.0).transpose(2, 0, 1).copy()


def _box_blur(a: np.ndarray, r: int) -> np.ndarray:
    k = 2 * r + 1
    c = np.cumsum(np.pad(a, ((r + 1, r), (0, 0)), mode="edge"), axis=0)
    a = (c[k:] - c[:-k]) / k
    c = np.cumsum(np.pad(a, ((0, 0), (r + 1, r)), mode="edge"), axis=1)
    return ((c[:, k:] - c[:, :-k]) / k).astype(np.float32)


def pseudo_depth(rgb_u8: np.ndarray) -> np.ndarray:
    """A depth map with the image's OWN irregular structure (there is no depth network here): the luminance, blurred, cut into three depth layers
    along its level sets (foreground 1.6-1.9, middle 2.6-2.9, background 4.2-4.5) plus a smooth in-layer term - depth edges follow natural object
    outlines (ragged, with islands) instead of the analytic discs of the synthetic scenes."""
    luma = (rgb_u8.astype(np.float32) @ np.array([0.299, 0.587, 0.114], np.float32)) / 255.0
    s = _box_blur(_box_blur(luma, 12), 12)
    lo, hi = np.quantile(s, 0.35), np.quantile(s, 0.7)
    layer = np.where(s > hi, 1.6, np.where(s > lo, 2.6, 4.2)).astype(np.float32)
    return (layer + 0.3 * _box_blur(luma, 4)).astype(np.float32)


def pan_clip(rgb_u8: np.ndarray, T: int, H: int, W: int, step: int = 4) -> np.ndarray:
    """A camera pan over a still: frame t = the H x W window whose left edge moves `step` pixels per frame (a natural video with real motion
    and natural spatial statistics). float32 [3, T, H, W] in [-1, 1]."""
    h, w = rgb_u8.shape[:2]
    top = (h - H) // 2
    left0 = (w - W - step * (T - 1)) // 2
    assert top >= 0 and left0 >= 0, "image too small for this pan"
    return np.stack([to_unit(rgb_u8[top:top + H, left0 + step * t:left0 + step * t + W]) for t in range(T)], axis=1)


def centre_crop(rgb_u8: np.ndarray, H: int, W: int) -> np.ndarray:
    h, w = rgb_u8.shape[:2]
    t, l = (h - H) // 2, (w - W) // 2
    return rgb_u8[t:t + H, l:l + W]
