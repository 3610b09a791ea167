"""Shared plumbing of the three inference entry points (gen3c_single_image / gen3c_multiview / gen3c_dynamic): the flags of
cosmos_predict1/diffusion/inference/inference_utils.py:53-170 that concern this path, model construction (checkpoints or
random weights), the chunk loop's bookkeeping, --save_buffer stacking and the output writer."""
from __future__ import annotations

import argparse
import os
from typing import Callable, List, Optional

import numpy as np
import torch


NEGATIVE_PROMPT_DEFAULT = (
    "The video captures a series of frames showing ugly scenes, static with no motion, motion blur, over-saturation, shaky footage, "
    "low resolution, grainy texture, pixelated images, poorly lit areas, underexposed and overexposed scenes, poor color balance, "
    "washed out colors, choppy sequences, jerky movements, low frame rate, artifacting, color banding, unnatural transitions, "
    "outdated special effects, fake elements, unconvincing visuals, poorly edited content, jump cuts, visual noise, and flickering. "
    "Overall, the video is of poor quality.")

# reference flags that have no counterpart on this path; accepted so that the reference's command lines run unchanged
_NO_COUNTERPART = {
    "offload_diffusion_transformer": "nothing is offloaded: weights + activations of the whole chunk fit the 288 GB of one MI355X",
    "offload_tokenizer": "nothing is offloaded (288 GB HBM)",
    "offload_text_encoder_model": "nothing is offloaded (288 GB HBM)",
    "offload_prompt_upsampler": "the prompt up-sampler is outside this path and never loaded",
    "offload_guardrail_models": "guardrail models are outside this path and never loaded",
    "disable_guardrail": "guardrail models are outside this path: no guardrail runs with or without this flag",
    "disable_prompt_upsampler": "the prompt up-sampler (Pixtral-12B) is outside this path: prompts are never up-sampled",
}


def add_common_args(p: argparse.ArgumentParser) -> argparse.ArgumentParser:
    """Every flag of the reference's `add_common_arguments` (inference_utils.py:53-170), same names / types / defaults, plus the
    inputs this path takes instead of running T5 itself (--t5_embedding_path ...) and two plumbing switches (--random_init, --tiny)."""
    p.add_argument("--checkpoint_dir", type=str, default="checkpoints", help="Base directory containing model checkpoints")
    p.add_argument("--tokenizer_dir", type=str, default="Cosmos-Tokenize1-CV8x8x8-720p", help="Tokenizer weights directory relative to checkpoint_dir")
    p.add_argument("--video_save_name", type=str, default="output")
    p.add_argument("--video_save_folder", type=str, default="outputs/")
    p.add_argument("--prompt", type=str, default=None, help="Text prompt; encoded by T5-11B when checkpoints/google-t5/t5-11b and `transformers` "
                   "are available, otherwise (or with --disable_prompt_encoder) all-zero embeddings like the reference's DummyT5TextEncoder")
    p.add_argument("--batch_input_path", type=str, default=None, help="JSONL file of {prompt, visual_input} records (utils/io.py:21-37)")
    p.add_argument("--negative_prompt", type=str, default=NEGATIVE_PROMPT_DEFAULT)
    p.add_argument("--num_steps", type=int, default=35)
    p.add_argument("--guidance", type=float, default=1)
    p.add_argument("--num_video_frames", type=int, default=None, help="N*120+1 (default 121); N > 1 runs autoregressive chunks")
    p.add_argument("--height", type=int, default=704)
    p.add_argument("--width", type=int, default=1280)
    p.add_argument("--fps", type=int, default=24)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--num_gpus", type=int, default=1, help="context-parallel ranks (launch with torchrun --nproc_per_node=N)")
    for flag, why in _NO_COUNTERPART.items():
        p.add_argument(f"--{flag}", action="store_true", help=f"accepted for command-line compatibility; {why}")
    p.add_argument("--disable_prompt_encoder", action="store_true", help="all-zero text embeddings (DummyT5TextEncoder, t5_text_encoder.py:111-132)")
    p.add_argument("--prompt_upsampler_dir", type=str, default="Pixtral-12B", help="accepted for command-line compatibility (up-sampler not run)")
    # shared by the three GEN3C entry points (gen3c_single_image.py:79-101)
    p.add_argument("--save_buffer", action="store_true", help="prepend the rendered warp buffers to every frame (gen3c_single_image.py:421-460)")
    p.add_argument("--filter_points_threshold", type=float, default=0.05)
    p.add_argument("--foreground_masking", action="store_true")
    # inputs of this path that the reference computes with models outside it
    p.add_argument("--t5_embedding_path", type=str, default=None, help=".pt tensor [1,512,1024]: the prompt's T5 embedding (overrides --prompt)")
    p.add_argument("--negative_t5_embedding_path", type=str, default=None, help=".pt tensor [1,512,1024]: the negative prompt's T5 embedding")
    p.add_argument("--random_init", action="store_true", help="random weights instead of checkpoints (plumbing tests)")
    p.add_argument("--tiny", action="store_true", help="a small DiT/tokenizer configuration and a 9-frame chunk (plumbing tests; with --random_init or a matching checkpoint_dir)")
    return p


def log_ignored_flags(args, log=print) -> List[str]:
    """One line per reference flag that was given but has no counterpart here. Returns the flag names (tests)."""
    given = [f for f in _NO_COUNTERPART if getattr(args, f, False)]
    for f in given:
        log(f"[gen3c_amd] --{f}: {_NO_COUNTERPART[f]}")
    return given


def read_prompts_from_file(path: str) -> List[dict]:
    """utils/io.py:21-37: one JSON object per line."""
    import json
    with open(path, "r") as f:
        return [json.loads(line) for line in f if line.strip()]


def write_video(video: np.ndarray, fps: int, path_base: str, quality: int = 5, log=print) -> str:
    """Writer chain for the result (uint8 [T,H,W,3]): `<base>.mp4` through imageio exactly as the reference's save_video does
    (utils/io.py:41-60) when imageio is importable, else through OpenCV's VideoWriter, else `<base>.npz` (video, fps) - this image
    ships neither encoder. Returns the path written."""
    T, H, W, _ = video.shape
    try:
        import imageio
        imageio.mimsave(path_base + ".mp4", video, "mp4", fps=fps, quality=quality, macro_block_size=1, ffmpeg_params=["-s", f"{W}x{H}"],
                        output_params=["-f", "mp4"])
        return path_base + ".mp4"
    except ImportError:
        pass
    try:
        import cv2
        vw = cv2.VideoWriter(path_base + ".mp4", cv2.VideoWriter_fourcc(*"mp4v"), float(fps), (W, H))
        for fr in video:
            vw.write(cv2.cvtColor(fr, cv2.COLOR_RGB2BGR))
        vw.release()
        return path_base + ".mp4"
    except ImportError:
        pass
    log(f"[gen3c_amd] no mp4 encoder importable (imageio / cv2): writing {path_base}.npz (uint8 video [T,H,W,3] + fps)")
    np.savez(path_base + ".npz", video=video, fps=fps)  # uncompressed: zlib over a 121x704x1280 video costs 18 s, as much as five denoise steps
    return path_base + ".npz"


class TextEmbedder:
    """Where the cross-attention context comes from (world_generation_pipeline.py:188-231 + t5_text_encoder.py). Order:
    an explicit embedding file; else T5-11B through `transformers` when `<checkpoint_dir>/google-t5/t5-11b` exists and
    --disable_prompt_encoder is not set (tokenise to 512, last hidden state, positions past the prompt length zeroed,
    t5_text_encoder.py:80-109); else all-zero embeddings (DummyT5TextEncoder)."""

    def __init__(self, args, channels: int, device, log=print):
        self.channels, self.device, self.log = channels, device, log
        self._enc = None
        t5_dir = os.path.join(args.checkpoint_dir, "google-t5", "t5-11b")
        if not getattr(args, "disable_prompt_encoder", False) and os.path.isdir(t5_dir):
            try:
                from transformers import T5EncoderModel, T5TokenizerFast
                self._tok = T5TokenizerFast.from_pretrained(t5_dir)
                self._enc = T5EncoderModel.from_pretrained(t5_dir).to(device).eval()
            except Exception as e:  # missing package / incomplete directory: say so, fall back like --disable_prompt_encoder
                log(f"[gen3c_amd] T5 encoder unavailable ({e!r}): using all-zero text embeddings")
        self.source = "t5" if self._enc is not None else "dummy"

    def __call__(self, prompt: Optional[str], embedding_path: Optional[str] = None) -> torch.Tensor:
        if embedding_path:
            return torch.load(embedding_path, map_location="cpu", weights_only=True).to(torch.bfloat16).reshape(1, -1, self.channels)
        if self._enc is not None and prompt is not None:
            be = self._tok.batch_encode_plus([prompt], return_tensors="pt", truncation=True, padding="max_length", max_length=512)
            ids, am = be.input_ids.to(self.device), be.attention_mask.to(self.device)
            with torch.inference_mode():
                emb = self._enc(input_ids=ids, attention_mask=am).last_hidden_state
            emb = emb.clone()
            emb[0, int(am.sum()):] = 0
            return emb.to(torch.bfloat16).cpu()
        if prompt:
            self.log("[gen3c_amd] --prompt given but no T5 source (no --t5_embedding_path, no google-t5/t5-11b checkpoint): all-zero text "
                     "embeddings, as the reference's --disable_prompt_encoder")
        return torch.zeros(1, 512, self.channels, dtype=torch.bfloat16)


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
            tk.load_weights(os.path.join(args.checkpoint_dir, getattr(args, "tokenizer_dir", "Cosmos-Tokenize1-CV8x8x8-720p")))
        self.cp_group = None  # also shards the renderer's item pairs and the tokenizer encodes of a chunk
        if args.num_gpus > 1:
            self.cp_group = parallel_state.get_context_parallel_group()
            net.enable_context_parallel(self.cp_group)
        self.net, self.tokenizer = net, tk
        self.model = DiffusionGen3CModel(net, tk, latent_shape=(16, tk.get_latent_num_frames(self.chunk), H // 8, W // 8))
        self.pipe = Gen3cPipeline(self.model, guidance=args.guidance, num_steps=args.num_steps, height=H, width=W, fps=args.fps,
                                  num_video_frames=self.chunk, seed=args.seed)
        log_ignored_flags(args)
        self.text = TextEmbedder(args, net.crossattn_emb_channels, dev)
        self.pipe.text_encoder = self.text  # Gen3cPipeline.generate(prompt="...") - the reference's keyword form - embeds through it
        self.set_prompt(getattr(args, "prompt", None))
        self.rendered_warps: List[torch.Tensor] = []

    def set_prompt(self, prompt: Optional[str]) -> None:
        """Text conditions of the next generation (world_generation_pipeline.py:274-283): the prompt's embedding, and the negative
        prompt's only when it can actually be embedded (a T5 source or an explicit file) - without one the unconditional branch keeps
        the positive text, exactly what the reference's conditioner does when `neg_t5_text_embeddings` is absent."""
        a = self.args
        self._emb = self.text(prompt, a.t5_embedding_path)
        if a.negative_t5_embedding_path:
            self._neg = self.text(None, a.negative_t5_embedding_path)
        elif self.text.source == "t5" and getattr(a, "negative_prompt", None):
            self._neg = self.text(a.negative_prompt)
        else:
            self._neg = None
            if a.t5_embedding_path or self.text.source == "t5":
                # classifier-free guidance is c + g (c - u) even at guidance 1: with u = c the text term cancels. The reference always embeds its
                # default negative prompt; here that needs a T5 source (--negative_t5_embedding_path or the google-t5/t5-11b checkpoint).
                self.text.log("[gen3c_amd] WARNING: a text embedding is used but no negative-prompt embedding is available "
                         "(--negative_t5_embedding_path / --negative_prompt with a T5 checkpoint): the unconditional branch reuses the positive "
                         "text, so the text contribution to classifier-free guidance vanishes")

    @property
    def num_chunks(self) -> int:
        return (self.args.num_video_frames - 1) // (self.chunk - 1)

    def generate_chunk(self, cond_image: torch.Tensor, renders: torch.Tensor, masks: torch.Tensor, first: bool) -> np.ndarray:
        """One 121-frame chunk. cond_image [1,3,1,H,W] in [-1,1]. Keeps the buffers for --save_buffer (chunks after the first
        drop their overlapping frame, gen3c_single_image.py:404-405)."""
        if self.args.save_buffer:
            self.rendered_warps.append((renders if first else renders[:, 1:]).clone().cpu())
        return self.pipe.generate_from_embeddings(self._emb, cond_image.to(torch.bfloat16), renders, masks, negative_prompt_embedding=self._neg)

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

    def close(self) -> None:
        """gen3c_single_image.py:478-484: tear the process groups down at the end of a multi-GPU run."""
        if self.args.num_gpus > 1:
            import torch.distributed as dist
            from .parallel import parallel_state
            parallel_state.destroy_model_parallel()
            if dist.is_initialized():
                dist.destroy_process_group()

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
        self.saved_path = write_video(video, self.args.fps, base)  # <folder>/<name>.mp4 (gen3c_single_image.py:462-476), .npz without an encoder
        Image.fromarray(video[0]).save(base + "_first.png")
        Image.fromarray(video[-1]).save(base + "_last.png")
