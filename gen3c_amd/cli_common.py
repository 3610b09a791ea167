"""Shared plumbing of the three inference entry points (gen3c_single_image / gen3c_multiview / gen3c_dynamic): the flags of
cosmos_predict1/diffusion/inference/inference_utils.py:53-170 that concern this path, model construction (checkpoints or
random weights), the chunk loop's bookkeeping, --save_buffer stacking and the output writer."""
from __future__ import annotations

import argparse
import os
from typing import Callable, List, Optional

import numpy as np
import torch


def add_common_args(p: argparse.ArgumentParser) -> argparse.ArgumentParser:
    p.add_argument("--checkpoint_dir", type=str, default="checkpoints")
    p.add_argument("--t5_embedding_path", type=str, default=None, help=".pt tensor [1,512,1024]; all-zero embedding if omitted")
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
    p.add_argument("--filter_points_threshold", type=float, default=0.05)
    p.add_argument("--foreground_masking", action="store_true")
    p.add_argument("--save_buffer", action="store_true", help="prepend the rendered warp buffers to every frame (inference_utils.py:160-164)")
    p.add_argument("--random_init", action="store_true", help="random weights instead of checkpoints (plumbing tests)")
    p.add_argument("--tiny", action="store_true", help="a small DiT/tokenizer configuration and a 9-frame chunk (plumbing tests; with --random_init or a matching checkpoint_dir)")
    return p


class Session:
    """Everything the entry points share after argument parsing."""

    def __init__(self, args):
        from .dit import VideoExtendGeneralDIT
        from .parallel import init_distributed, parallel_state
        from .pipeline import DiffusionGen3CModel, Gen3cPipeline
        from .tokenizer import VideoTokenizer

        self.args = args
        tiny = args.tiny
        self.step_frames = 8 if tiny else 120
        self.chunk = self.step_frames + 1
        if args.num_video_frames is None:
            args.num_video_frames = self.chunk
        assert (args.num_video_frames - 1) % self.step_frames == 0, f"num_video_frames must be N*{self.step_frames}+1"  # gen3c_single_image.py:112
        local = 0
        if args.num_gpus > 1:
            local = init_distributed("nccl")
            parallel_state.initialize_model_parallel(context_parallel_size=args.num_gpus)
        self.dev = dev = torch.device(f"cuda:{local}")
        torch.cuda.set_device(dev)
        H, W = args.height, args.width
        if tiny:
            net = VideoExtendGeneralDIT(max_img_h=240, max_img_w=240, max_frames=16, in_channels=81, model_channels=256, num_blocks=2, num_heads=2,
                                        adaln_lora_dim=32, rope_t_extrapolation_ratio=2.0, device=dev, init_weights=False)
            tk = VideoTokenizer(pixel_chunk_duration=self.chunk, channels=16, device=dev)
        else:
            net = VideoExtendGeneralDIT(in_channels=16 + 16 * 4 + 1, rope_t_extrapolation_ratio=2.0, device=dev, init_weights=False)
            tk = VideoTokenizer(pixel_chunk_duration=self.chunk, device=dev)
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
        self.net, self.tokenizer = net, tk
        self.model = DiffusionGen3CModel(net, tk, latent_shape=(16, tk.get_latent_num_frames(self.chunk), H // 8, W // 8))
        self.pipe = Gen3cPipeline(self.model, guidance=args.guidance, num_steps=args.num_steps, height=H, width=W, fps=args.fps,
                                  num_video_frames=self.chunk, seed=args.seed)
        self._emb = self._load_emb(args.t5_embedding_path)
        self._neg = self._load_emb(args.negative_t5_embedding_path) if args.negative_t5_embedding_path else None
        self.rendered_warps: List[torch.Tensor] = []

    def _load_emb(self, path: Optional[str]) -> torch.Tensor:
        c = self.net.crossattn_emb_channels
        if path is None:
            return torch.zeros(1, 512, c, dtype=torch.bfloat16)
        return torch.load(path, map_location="cpu", weights_only=True).to(torch.bfloat16).reshape(1, -1, c)

    @property
    def num_chunks(self) -> int:
        return (self.args.num_video_frames - 1) // (self.chunk - 1)

    def generate_chunk(self, cond_image: torch.Tensor, renders: torch.Tensor, masks: torch.Tensor, first: bool) -> np.ndarray:
        """One 121-frame chunk. cond_image [1,3,1,H,W] in [-1,1]. Keeps the buffers for --save_buffer (chunks after the first
        drop their overlapping frame, gen3c_single_image.py:404-405)."""
        if self.args.save_buffer:
            self.rendered_warps.append((renders if first else renders[:, 1:]).clone().cpu())
        return self.pipe.generate(self._emb, cond_image.to(torch.bfloat16), renders, masks, negative_prompt_embedding=self._neg)

    def run_chunks(self, cond_image: torch.Tensor, render_fn: Callable[[int, Optional[torch.Tensor]], tuple]) -> np.ndarray:
        """The chunk loop shared by all entry points. render_fn(start_frame, last_frame01 or None) -> (renders, masks) for frames
        [start, start + chunk); it may update the cache from the last generated frame first (single-image AR)."""
        renders, masks = render_fn(0, None)
        video = self.generate_chunk(cond_image, renders, masks, True)
        for it in range(1, self.num_chunks):
            start = it * (self.chunk - 1)  # chunks overlap by one frame
            pred01 = torch.from_numpy(video[-1]).to(self.dev).permute(2, 0, 1).to(torch.float32) / 255.0
            renders, masks = render_fn(start, pred01)
            video_new = self.generate_chunk(pred01[None, :, None] * 2 - 1, renders, masks, False)
            video = np.concatenate([video, video_new[1:]], axis=0)
        return video

    def finalize(self, video: np.ndarray, save_buffer: Optional[bool] = None) -> np.ndarray:
        """--save_buffer stacking (gen3c_single_image.py:421-460): buffers side by side, left of the generated frame."""
        if (self.args.save_buffer if save_buffer is None else save_buffer) and self.rendered_warps:
            sq = [t.squeeze(0) for t in self.rendered_warps]  # (T_chunk, n_i, C, H, W)
            n_max = max(t.shape[1] for t in sq)
            sq = [torch.nn.functional.pad(t, (0, 0, 0, 0, 0, 0, 0, n_max - t.shape[1]), value=-1.0) for t in sq]
            full = torch.cat(sq, dim=0)
            T, _, C, H, W = full.shape
            stacked = full.permute(0, 2, 3, 1, 4).contiguous().view(T, C, H, n_max * W)
            stacked = ((stacked * 0.5 + 0.5) * 255.0).numpy().astype(np.uint8)
            video = np.concatenate([np.transpose(stacked, (0, 2, 3, 1)), video], axis=2)
        return video

    def save(self, video: np.ndarray, name: Optional[str] = None) -> None:
        if int(os.environ.get("RANK", "0")) != 0:  # the reference lets every rank write the same file (:469-476); one writer suffices
            return
        from PIL import Image
        os.makedirs(self.args.video_save_folder, exist_ok=True)
        base = os.path.join(self.args.video_save_folder, name or self.args.video_save_name)
        np.savez_compressed(base + ".npz", video=video, fps=self.args.fps)
        Image.fromarray(video[0]).save(base + "_first.png")
        Image.fromarray(video[-1]).save(base + "_last.png")
