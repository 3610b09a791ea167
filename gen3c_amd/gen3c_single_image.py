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
Autoregressive extension (--num_video_frames > 121) needs Cache3D_Buffer.update_cache with depth alignment, the next row
of the scope table, and is refused. Multi-GPU: torchrun --nproc-per-node N ... --num_gpus N (context parallel).
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch


def create_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="GEN3C single-image video generation on MI355X")
    p.add_argument("--checkpoint_dir", type=str, default="checkpoints")
    p.add_argument("--input_image_path", type=str, required=True)
    p.add_argument("--depth_path", type=str, required=True)
    p.add_argument("--t5_embedding_path", type=str, default=None)
    p.add_argument("--negative_t5_embedding_path", type=str, default=None)
    p.add_argument("--video_save_name", type=str, default="output")
    p.add_argument("--video_save_folder", type=str, default="outputs/")
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--guidance", type=float, default=1.0)
    p.add_argument("--num_steps", type=int, default=35)
    p.add_argument("--num_video_frames", type=int, default=None, help="N*120+1 (default 121); N > 1 runs autoregressive chunks")
    p.add_argument("--height", type=int, default=704)
    p.add_argument("--width", type=int, default=1280)
    p.add_argument("--fps", type=int, default=24)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--trajectory", type=str, default="left",
                   choices=["left", "right", "up", "down", "zoom_in", "zoom_out", "clockwise", "counterclockwise", "none"])
    p.add_argument("--camera_rotation", type=str, default="center_facing", choices=["center_facing", "no_rotation", "trajectory_aligned"])
    p.add_argument("--movement_distance", type=float, default=0.3)
    p.add_argument("--noise_aug_strength", type=float, default=0.0)
    p.add_argument("--filter_points_threshold", type=float, default=0.05)
    p.add_argument("--foreground_masking", action="store_true")
    p.add_argument("--ar_depth", type=str, default="cache",
                   help="depth for the last frame of each autoregressive chunk (the reference runs MoGe, which is not available "
                        "offline): 'cache' = depth rendered from the 3D cache at that camera with holes filled by the median, or "
                        "'module:function' naming a callable image[3,H,W] in [0,1] -> (depth[1,1,H,W], mask[1,1,H,W] or None)")
    p.add_argument("--random_init", action="store_true", help="random weights instead of checkpoints (plumbing tests)")
    p.add_argument("--tiny", action="store_true", help="with --random_init: a small DiT/tokenizer and a 9-frame chunk (plumbing tests)")
    return p


def _load_image(path: str, H: int, W: int) -> torch.Tensor:
    from PIL import Image
    img = Image.open(path).convert("RGB").resize((W, H), Image.BICUBIC)
    return torch.from_numpy(np.asarray(img).astype(np.float32))  # [H,W,3] 0..255


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
    from gen3c_amd.dit import VideoExtendGeneralDIT
    from gen3c_amd.parallel import init_distributed, parallel_state
    from gen3c_amd.pipeline import DiffusionGen3CModel, Gen3cPipeline
    from gen3c_amd.tokenizer import VideoTokenizer

    tiny = args.tiny and args.random_init
    step_frames = 8 if tiny else 120
    if args.num_video_frames is None:
        args.num_video_frames = step_frames + 1
    assert (args.num_video_frames - 1) % step_frames == 0, \
        f"num_video_frames must be N*{step_frames}+1"   # gen3c_single_image.py:112
    local = 0
    if args.num_gpus > 1:
        local = init_distributed("nccl")
        parallel_state.initialize_model_parallel(context_parallel_size=args.num_gpus)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    H, W = args.height, args.width
    chunk = step_frames + 1

    # ---- models
    if args.random_init and args.tiny:
        net = VideoExtendGeneralDIT(max_img_h=240, max_img_w=240, max_frames=16, in_channels=81, model_channels=256, num_blocks=2, num_heads=2,
                                    adaln_lora_dim=32, rope_t_extrapolation_ratio=2.0, device=dev, init_weights=False)
        tk = VideoTokenizer(pixel_chunk_duration=chunk, channels=16, device=dev)
    else:
        net = VideoExtendGeneralDIT(in_channels=16 + 16 * 4 + 1, rope_t_extrapolation_ratio=2.0, device=dev, init_weights=False)
        tk = VideoTokenizer(pixel_chunk_duration=chunk, device=dev)
    if args.random_init:
        net.initialize_weights(randomize_adaln=True, seed=args.seed)
        tk.net.init_random(seed=args.seed)
        tk.register_mean_std(torch.zeros(16, 32), torch.ones(16, 32))
    else:
        sd = torch.load(os.path.join(args.checkpoint_dir, "Gen3C-Cosmos-7B", "model.pt"), map_location="cpu", weights_only=True)
        sd = sd.get("model", sd)
        net.load_state_dict({k[len("net."):]: v for k, v in sd.items() if k.startswith("net.")}, strict=True)
        tk.load_weights(os.path.join(args.checkpoint_dir, "Cosmos-Tokenize1-CV8x8x8-720p"))
    if args.num_gpus > 1:
        net.enable_context_parallel(parallel_state.get_context_parallel_group())
    lat_T = tk.get_latent_num_frames(chunk)
    model = DiffusionGen3CModel(net, tk, latent_shape=(16, lat_T, H // 8, W // 8))
    pipe = Gen3cPipeline(model, guidance=args.guidance, num_steps=args.num_steps, height=H, width=W, fps=args.fps, num_video_frames=chunk, seed=args.seed)

    # ---- inputs: image, depth (MoGe stand-in), intrinsics
    img255 = _load_image(args.input_image_path, H, W).to(dev)
    image = (img255.permute(2, 0, 1) / 255.0 * 2 - 1)[None]                       # renders use x/255*2-1 (gen3c_single_image.py:135)
    z = np.load(args.depth_path)
    depth = torch.from_numpy(np.asarray(z["depth"], dtype=np.float32)).to(dev)
    if depth.shape != (H, W):
        depth = torch.nn.functional.interpolate(depth[None, None], size=(H, W), mode="nearest")[0, 0]
    depth = torch.where(torch.isfinite(depth) & (depth > 0), depth, torch.full_like(depth, 1000.0))  # invalid -> 1000 (:141)
    if "intrinsics" in z.files:
        K = torch.from_numpy(np.asarray(z["intrinsics"], dtype=np.float32)).to(dev)
    else:
        f = 0.9 * W
        K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], device=dev)
    w2c0 = torch.eye(4, device=dev)
    cache = renderer.Cache3D_Buffer(frame_buffer_max=model.frame_buffer_max, noise_aug_strength=args.noise_aug_strength,
                                    generator=torch.Generator(device=dev).manual_seed(args.seed), input_image=image, input_depth=depth[None, None],
                                    input_w2c=w2c0[None], input_intrinsics=K[None], filter_points_threshold=args.filter_points_threshold,
                                    foreground_masking=args.foreground_masking, input_format=["B", "C", "H", "W"])
    valid = depth[depth < 100]
    center_depth = float(torch.quantile(valid.flatten()[:: max(1, valid.numel() // 100000)], 0.5)) if valid.numel() else 1.0
    traj = "left" if args.trajectory == "none" else args.trajectory
    dist_ = 0.0 if args.trajectory == "none" else args.movement_distance
    w2cs, Ks = generate_camera_trajectory(traj, w2c0, K, args.num_video_frames, dist_, args.camera_rotation, center_depth=center_depth, device=dev)
    renders, masks = cache.render_cache(w2cs[:, :chunk], Ks[:, :chunk])

    def emb(path):
        if path is None:
            return torch.zeros(1, 512, net.crossattn_emb_channels, dtype=torch.bfloat16)
        return torch.load(path, map_location="cpu", weights_only=True).to(torch.bfloat16).reshape(1, -1, net.crossattn_emb_channels)

    cond_image = (img255.permute(2, 0, 1) / 128.0 - 1.0)[None, :, None].to(torch.bfloat16)   # condition image uses x/128-1 (inference_utils.py:648)
    neg = emb(args.negative_t5_embedding_path) if args.negative_t5_embedding_path else None
    video = pipe.generate(emb(args.t5_embedding_path), cond_image, renders, masks, negative_prompt_embedding=neg)

    # ---- autoregressive chunks (gen3c_single_image.py:378-419): last frame -> depth -> aligned cache update -> next 121 frames
    depth_fn = _resolve_depth_fn(args.ar_depth, cache)
    for it in range(1, (args.num_video_frames - 1) // (chunk - 1)):
        start = it * (chunk - 1)  # chunks overlap by one frame
        pred01 = torch.from_numpy(video[-1]).to(dev).permute(2, 0, 1).to(torch.float32) / 255.0
        pred_depth, pred_mask = depth_fn(pred01, w2cs[:, start], Ks[:, start])
        cache.update_cache(new_image=pred01[None] * 2 - 1, new_depth=pred_depth, new_w2c=w2cs[:, start], new_intrinsics=Ks[:, start])
        renders, masks = cache.render_cache(w2cs[:, start:start + chunk], Ks[:, start:start + chunk])
        cond = (pred01[None, :, None] * 2 - 1).to(torch.bfloat16)
        video_new = pipe.generate(emb(args.t5_embedding_path), cond, renders, masks, negative_prompt_embedding=neg)
        video = np.concatenate([video, video_new[1:]], axis=0)

    rank = int(os.environ.get("RANK", "0"))
    if rank == 0:  # the reference lets every rank write the same file (gen3c_single_image.py:469-476); one writer suffices
        os.makedirs(args.video_save_folder, exist_ok=True)
        base = os.path.join(args.video_save_folder, args.video_save_name)
        np.savez_compressed(base + ".npz", video=video, fps=args.fps)
        from PIL import Image
        Image.fromarray(video[0]).save(base + "_first.png")
        Image.fromarray(video[-1]).save(base + "_last.png")
    return video


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    demo(create_parser().parse_args())
