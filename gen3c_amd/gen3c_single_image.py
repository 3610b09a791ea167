"""Single-image -> 121-frame GEN3C video on the MI355X path: counterpart of
cosmos_predict1/diffusion/inference/gen3c_single_image.py (flags of inference_utils.py:53-170 + :40-102 that concern
this path keep their names: --checkpoint_dir --num_gpus --guidance --num_steps --num_video_frames --height --width --fps
--seed --trajectory --camera_rotation --movement_distance --noise_aug_strength --filter_points_threshold
--foreground_masking --video_save_name --video_save_folder --input_image_path).

What differs (the models around the path are out of scope here): MoGe depth and T5 embeddings are INPUTS:
  --depth_path   .npz with `depth` [H,W] (metres, invalid = 0/NaN/inf) and optional `intrinsics` [3,3] (pixels)
  --t5_embedding_path / --negative_t5_embedding_path   .pt tensors [1,512,1024] (all-zero embeddings if omitted, i.e. the
                 reference's DummyT5TextEncoder behaviour)
The video is written as <folder>/<name>.npz (uint8 [T,H,W,3]) plus first/last-frame PNGs (no mp4 writer in this image).
--num_video_frames N*120+1 runs the autoregressive loop (:378-419): the last generated frame is pushed into the cache with its
depth aligned to the cache's own rendering (Cache3D_Buffer.update_cache); --ar_depth supplies that frame's depth.
Multi-GPU: torchrun --nproc-per-node N ... --num_gpus N (context parallel).
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from gen3c_amd.cli_common import Session, add_common_args


def create_parser() -> argparse.ArgumentParser:
    p = add_common_args(argparse.ArgumentParser(description="GEN3C single image -> video on MI355X"))
    p.add_argument("--input_image_path", type=str, required=True)
    p.add_argument("--depth_path", type=str, required=True)
    p.add_argument("--trajectory", type=str, default="left",
                   choices=["left", "right", "up", "down", "zoom_in", "zoom_out", "clockwise", "counterclockwise", "none"])
    p.add_argument("--camera_rotation", type=str, default="center_facing", choices=["center_facing", "no_rotation", "trajectory_aligned"])
    p.add_argument("--movement_distance", type=float, default=0.3)
    p.add_argument("--noise_aug_strength", type=float, default=0.0)
    p.add_argument("--ar_depth", type=str, default="cache",
                   help="depth for the last frame of each autoregressive chunk (the reference runs MoGe, which is not available "
                        "offline): 'cache' = depth rendered from the 3D cache at that camera with holes filled by the median, or "
                        "'module:function' naming a callable image[3,H,W] in [0,1] -> (depth[1,1,H,W], mask[1,1,H,W] or None)")
    return p


def _read_rgb(path: str) -> np.ndarray:
    """uint8 [h,w,3]; RGBA is blended onto a white background (inference_utils.py:621-633)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode == "RGBA":
        a = np.asarray(im).astype(np.float64)
        alpha = a[..., 3:4] / 255.0
        return (a[..., :3] * alpha + 255.0 * (1 - alpha)).astype(np.uint8)
    return np.asarray(im.convert("RGB"))


def load_condition_image(path: str, H: int, W: int) -> torch.Tensor:
    """The seeding frame exactly as read_video_or_image_into_frames_BCTHW prepares it (inference_utils.py:597-660):
    uint8 / 128 - 1 -> bf16 -> torchvision resize(BICUBIC, antialias=True) (= F.interpolate in fp32, cast back) -> [1,3,1,H,W] bf16."""
    x = torch.from_numpy(_read_rgb(path) / 128.0 - 1.0).permute(2, 0, 1)[None].to(torch.bfloat16)
    if tuple(x.shape[-2:]) != (H, W):
        x = torch.nn.functional.interpolate(x.float(), size=(H, W), mode="bicubic", antialias=True, align_corners=False).to(torch.bfloat16)
    return x[:, :, None]


def load_cache_image(path: str, H: int, W: int) -> torch.Tensor:
    """The image the 3D cache is built from, as _predict_moge_depth prepares it (gen3c_single_image.py:118-180): OpenCV bilinear
    resize to 1280x720 on uint8, / 255, bilinear resize to (H, W), * 2 - 1 -> [1,3,H,W] fp32. (cv2.INTER_LINEAR is the
    half-pixel, non-antialiased bilinear of F.interpolate(align_corners=False); its uint8 output rounding is reproduced.)"""
    x = torch.from_numpy(_read_rgb(path).astype(np.float32)).permute(2, 0, 1)[None]
    x = torch.nn.functional.interpolate(x, size=(720, 1280), mode="bilinear", align_corners=False).round().clamp(0, 255) / 255.0
    x = torch.nn.functional.interpolate(x, size=(H, W), mode="bilinear", align_corners=False)
    return x * 2 - 1


def _resolve_depth_fn(spec: str, cache):
    """-> callable(image[3,H,W] in [0,1], w2c[1,4,4], K[1,3,3]) -> (depth[1,1,H,W], mask or None). Stands in for
    _predict_moge_depth_from_tensor (gen3c_single_image.py:183-229); user callables take the image only, like MoGe."""
    if spec == "cache":
        def from_cache(_image, w2c, K):
            d, m = cache.render_cache(w2c[:, None], K[:, None], render_depth=True)
            d, m = d[:, 0, 0], m[:, 0, 0, 0] > 0                                   # newest buffer: [1,H,W]
            fill = d[m].median() if bool(m.any()) else d.new_tensor(1.0)
            return torch.where(m, d, fill)[:, None], None
        return from_cache
    mod, _, fn = spec.partition(":")
    import importlib
    user = getattr(importlib.import_module(mod), fn)
    return lambda image, _w2c, _K: user(image)


def demo(args) -> np.ndarray:
    from gen3c_amd import renderer
    from gen3c_amd.camera_utils import generate_camera_trajectory

    ses = Session(args)
    dev, H, W = ses.dev, args.height, args.width

    # ---- inputs: image, depth (MoGe stand-in), intrinsics
    image = load_cache_image(args.input_image_path, H, W).to(dev)                 # renders use x/255*2-1 (gen3c_single_image.py:135,176)
    z = np.load(args.depth_path)
    depth = torch.from_numpy(np.asarray(z["depth"], dtype=np.float32)).to(dev)
    depth = torch.where(torch.isfinite(depth) & (depth > 0), depth, torch.full_like(depth, 1000.0))  # invalid -> 1000 (:141), then
    if depth.shape != (H, W):                                                                            # bilinear to the target (:153-158)
        depth = torch.nn.functional.interpolate(depth[None, None], size=(H, W), mode="bilinear", align_corners=False)[0, 0]
    if "intrinsics" in z.files:
        K = torch.from_numpy(np.asarray(z["intrinsics"], dtype=np.float32)).to(dev)
    else:
        f = 0.9 * W
        K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], device=dev)
    w2c0 = torch.eye(4, device=dev)
    cache = renderer.Cache3D_Buffer(frame_buffer_max=ses.model.frame_buffer_max, noise_aug_strength=args.noise_aug_strength,
                                    generator=torch.Generator(device=dev).manual_seed(args.seed), input_image=image, input_depth=depth[None, None],
                                    input_w2c=w2c0[None], input_intrinsics=K[None], filter_points_threshold=args.filter_points_threshold,
                                    foreground_masking=args.foreground_masking, input_format=["B", "C", "H", "W"])
    center_depth = 1.0  # the reference passes this constant (gen3c_single_image.py:340-349)
    traj = "left" if args.trajectory == "none" else args.trajectory
    dist_ = 0.0 if args.trajectory == "none" else args.movement_distance
    w2cs, Ks = generate_camera_trajectory(traj, w2c0, K, args.num_video_frames, dist_, args.camera_rotation, center_depth=center_depth, device=dev)
    depth_fn = _resolve_depth_fn(args.ar_depth, cache)

    def render(start: int, last01):
        # autoregressive chunks (gen3c_single_image.py:378-419): last frame -> depth -> aligned cache update -> next 121 frames
        if last01 is not None:
            pred_depth, _pred_mask = depth_fn(last01, w2cs[:, start], Ks[:, start])
            cache.update_cache(new_image=last01[None] * 2 - 1, new_depth=pred_depth, new_w2c=w2cs[:, start], new_intrinsics=Ks[:, start])
        return cache.render_cache(w2cs[:, start:start + ses.chunk], Ks[:, start:start + ses.chunk])

    cond_image = load_condition_image(args.input_image_path, H, W)          # condition image uses x/128-1 (inference_utils.py:648)
    video = ses.finalize(ses.run_chunks(cond_image, render))
    ses.save(video)
    return video


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    demo(create_parser().parse_args())
