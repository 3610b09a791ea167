"""Single-image -> 121-frame GEN3C video on the MI355X path: counterpart of
cosmos_predict1/diffusion/inference/gen3c_single_image.py. Every flag of the reference (inference_utils.py:53-170 +
gen3c_single_image.py:40-102) is accepted with its name, type and default, so the reference's command lines run unchanged
(`cosmos_predict1/diffusion/inference/gen3c_single_image.py` in this repo forwards here):

  torchrun --nproc_per_node=8 cosmos_predict1/diffusion/inference/gen3c_single_image.py --checkpoint_dir checkpoints \
      --input_image_path assets/diffusion/000000.png --video_save_name test --num_gpus 8 --guidance 1 --foreground_masking

The models AROUND the path stay outside it and are used when present, replaced by inputs when not:
  * depth: MoGe is run exactly as _predict_moge_depth does (:118-180) when the `moge` package is importable; otherwise
    --depth_path <npz with `depth` [H,W] (metres, invalid = 0/NaN/inf) and optional `intrinsics` [3,3] (pixels)> is required;
  * text: see cli_common.TextEmbedder (--t5_embedding_path, a T5-11B checkpoint, or the reference's dummy zero embeddings).
Output: <video_save_folder>/<video_save_name>.mp4 (.npz when no mp4 encoder is importable) + first/last-frame PNGs; with
--batch_input_path one video per JSONL record, named by its index (:462-465).
--num_video_frames N*120+1 runs the autoregressive loop (:378-419): the last generated frame is pushed into the cache with its
depth aligned to the cache's own rendering (Cache3D_Buffer.update_cache); its depth comes from MoGe when available, else --ar_depth.
Multi-GPU: torchrun --nproc_per_node N ... --num_gpus N (context parallel over the latent frames, RCCL).
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from gen3c_amd.cli_common import Session, add_common_args, read_prompts_from_file


TRAJECTORIES = ["left", "right", "up", "down", "zoom_in", "zoom_out", "clockwise", "counterclockwise", "none"]


def create_parser() -> argparse.ArgumentParser:
    p = add_common_args(argparse.ArgumentParser(description="GEN3C single image -> video on MI355X"))
    p.add_argument("--input_image_path", type=str, default=None, help="Input image path for generating a single video")
    p.add_argument("--trajectory", type=str, default="left", choices=TRAJECTORIES)
    p.add_argument("--camera_rotation", type=str, default="center_facing", choices=["center_facing", "no_rotation", "trajectory_aligned"])
    p.add_argument("--movement_distance", type=float, default=0.3)
    p.add_argument("--noise_aug_strength", type=float, default=0.0)
    p.add_argument("--depth_path", type=str, default=None,
                   help=".npz with `depth` [H,W] and optional `intrinsics` [3,3]: required only when the `moge` package is not importable")
    p.add_argument("--ar_depth", type=str, default="cache",
                   help="without MoGe: depth for the last frame of each autoregressive chunk - 'cache' = depth rendered from the 3D cache at "
                        "that camera with holes filled by the median, or 'module:function' naming a callable image[3,H,W] in [0,1] -> "
                        "(depth[1,1,H,W], mask[1,1,H,W] or None)")
    return p


def validate_args(args) -> None:
    """gen3c_single_image.py:110-112 (+ the two inputs this path cannot invent)."""
    if args.num_video_frames is not None and not getattr(args, "tiny", False):
        assert (args.num_video_frames - 1) % 120 == 0, "num_video_frames must be 121, 241, 361, ... (N*120+1)"
    if not args.batch_input_path and not args.input_image_path:
        raise SystemExit("gen3c_single_image: --input_image_path (or --batch_input_path) is required")


def load_moge():
    """MoGeModel.from_pretrained("Ruicheng/moge-vitl") (gen3c_single_image.py:283) when the package is importable, else None."""
    try:
        from moge.model.v1 import MoGeModel
    except ImportError:
        return None
    return MoGeModel.from_pretrained("Ruicheng/moge-vitl")


def predict_moge_depth(image_rgb_u8: np.ndarray, H: int, W: int, device, moge_model):
    """_predict_moge_depth (gen3c_single_image.py:115-180) for a uint8 [h,w,3] image: bilinear resize to 1280x720, MoGe, invalid -> 1000,
    normalised intrinsics -> pixels, depth bilinear / image bilinear to (H, W). -> (image [1,3,H,W] in [-1,1], depth [1,1,H,W], K [3,3])."""
    F = torch.nn.functional
    x = torch.from_numpy(image_rgb_u8.astype(np.float32)).permute(2, 0, 1)[None]
    x = F.interpolate(x, size=(720, 1280), mode="bilinear", align_corners=False).round().clamp(0, 255)[0].to(device) / 255.0  # cv2.resize on uint8
    out = moge_model.infer(x)
    depth = torch.where(out["mask"] == 0, torch.tensor(1000.0, device=out["depth"].device), out["depth"])
    K = out["intrinsics"].clone()
    K[0, 0] *= 1280; K[0, 2] *= 1280; K[1, 1] *= 720; K[1, 2] *= 720
    depth = F.interpolate(depth[None, None], size=(H, W), mode="bilinear", align_corners=False)
    image = F.interpolate(x[None], size=(H, W), mode="bilinear", align_corners=False) * 2 - 1
    K[1, 1] *= H / 720; K[1, 2] *= H / 720; K[0, 0] *= W / 1280; K[0, 2] *= W / 1280
    depth = torch.clamp(torch.nan_to_num(depth, nan=1e4), min=0, max=1e4)
    return image.float(), depth.float(), K.float()


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


def _depth_inputs(args, image_path: str, depth_path, H: int, W: int, dev, moge_model):
    """-> (cache image [1,3,H,W] in [-1,1], depth [1,1,H,W], K [3,3]) from MoGe when it is available, else from the depth file."""
    if moge_model is not None and depth_path is None:
        return predict_moge_depth(_read_rgb(image_path), H, W, dev, moge_model)
    if depth_path is None:
        raise SystemExit("gen3c_single_image: the `moge` package is not importable here, so the depth MoGe would predict must be given: "
                         "--depth_path <npz with `depth` [H,W] and optional `intrinsics` [3,3]>")
    image = load_cache_image(image_path, H, W).to(dev)                             # renders use x/255*2-1 (gen3c_single_image.py:135,176)
    z = np.load(depth_path)
    depth = torch.from_numpy(np.asarray(z["depth"], dtype=np.float32)).to(dev)
    depth = torch.where(torch.isfinite(depth) & (depth > 0), depth, torch.full_like(depth, 1000.0))  # invalid -> 1000 (:141), then
    if depth.shape != (H, W):                                                                            # bilinear to the target (:153-158)
        depth = torch.nn.functional.interpolate(depth[None, None], size=(H, W), mode="bilinear", align_corners=False)[0, 0]
    if "intrinsics" in z.files:
        K = torch.from_numpy(np.asarray(z["intrinsics"], dtype=np.float32)).to(dev)
    else:
        f = 0.9 * W
        K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], device=dev)
    return image, depth[None, None], K


def generate_one(ses: Session, args, image_path: str, depth_path, moge_model) -> np.ndarray:
    """One input image -> one video (body of the reference's per-prompt loop, gen3c_single_image.py:299-460)."""
    from gen3c_amd import renderer
    from gen3c_amd.camera_utils import generate_camera_trajectory

    dev, H, W = ses.dev, args.height, args.width
    image, depth, K = _depth_inputs(args, image_path, depth_path, H, W, dev, moge_model)
    w2c0 = torch.eye(4, device=dev)
    cache = renderer.Cache3D_Buffer(frame_buffer_max=ses.model.frame_buffer_max, noise_aug_strength=args.noise_aug_strength,
                                    generator=torch.Generator(device=dev).manual_seed(args.seed), input_image=image, input_depth=depth,
                                    input_w2c=w2c0[None], input_intrinsics=K[None], filter_points_threshold=args.filter_points_threshold,
                                    foreground_masking=args.foreground_masking, input_format=["B", "C", "H", "W"])
    cache.shard_group = ses.cp_group  # multi-GPU: every rank renders its share of the item pairs
    center_depth = 1.0  # the reference passes this constant (gen3c_single_image.py:340-349)
    traj = "left" if args.trajectory == "none" else args.trajectory
    dist_ = 0.0 if args.trajectory == "none" else args.movement_distance
    w2cs, Ks = generate_camera_trajectory(traj, w2c0, K, args.num_video_frames, dist_, args.camera_rotation, center_depth=center_depth, device=dev)
    if moge_model is not None:  # _predict_moge_depth_from_tensor (:183-197)
        def depth_fn(image01, _w2c, _K):
            out = moge_model.infer(image01)
            d = torch.clamp(torch.nan_to_num(out["depth"][None, None], nan=1e4), min=0, max=1e4)
            return torch.where(out["mask"][None, None] == 0, torch.tensor(1000.0, device=d.device), d), out["mask"][None, None]
    else:
        depth_fn = _resolve_depth_fn(args.ar_depth, cache)
    ses.rendered_warps.clear()

    def render(start: int, last01):
        # autoregressive chunks (gen3c_single_image.py:378-419): last frame -> depth -> aligned cache update -> next 121 frames
        if last01 is not None:
            pred_depth, _pred_mask = depth_fn(last01, w2cs[:, start], Ks[:, start])
            cache.update_cache(new_image=last01[None] * 2 - 1, new_depth=pred_depth, new_w2c=w2cs[:, start], new_intrinsics=Ks[:, start])
        return cache.render_cache(w2cs[:, start:start + ses.chunk], Ks[:, start:start + ses.chunk])

    cond_image = load_condition_image(image_path, H, W)          # condition image uses x/128-1 (inference_utils.py:648)
    return ses.finalize(ses.run_chunks(cond_image, render))


def demo(args) -> np.ndarray:
    validate_args(args)
    ses = Session(args)
    moge_model = load_moge()
    if moge_model is not None:
        moge_model = moge_model.to(ses.dev)
    if args.batch_input_path:   # one JSON record per line: {"prompt": ..., "visual_input": ...[, "depth_path": ...]} (utils/io.py:21-37)
        records = read_prompts_from_file(args.batch_input_path)
    else:
        records = [{"prompt": args.prompt, "visual_input": args.input_image_path}]
    video = None
    for i, rec in enumerate(records):
        image_path = rec.get("visual_input")
        if image_path is None or not os.path.exists(image_path):
            print(f"[gen3c_amd] record {i}: visual input {image_path!r} is missing, skipping world generation")  # :311-320
            continue
        ses.set_prompt(rec.get("prompt"))
        video = generate_one(ses, args, image_path, rec.get("depth_path", args.depth_path), moge_model)
        ses.save(video, name=str(i) if args.batch_input_path else None)
    ses.close()
    return video


def main(argv=None) -> None:
    torch.set_grad_enabled(False)  # gen3c_single_image.py:33
    args = create_parser().parse_args(argv)
    if args.prompt is None:
        args.prompt = ""
    demo(args)


if __name__ == "__main__":
    main()
