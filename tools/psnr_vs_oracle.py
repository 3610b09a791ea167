"""Checkpoint parity harness (north_star: "per-frame PSNR within 0.1 dB of reference"): given a checkpoint directory in the
reference's layout (checkpoints/Gen3C-Cosmos-7B/model.pt + checkpoints/Cosmos-Tokenize1-CV8x8x8-720p/{encoder,decoder}.jit, mean_std.pt) -
or `--random_init` weights - generate ONE chunk three ways from the same rendered buffers, text embeddings and injected initial noise:

  hip      the product (gen3c_amd.pipeline.Gen3cPipeline.generate_from_embeddings: HIP kernels, bf16)
  ref      oracle/chain_oracle.py with the networks in bf16 = the reference's own precision (config/base/model.py:29), torch kernels of the device
  fp32     the same oracle chain in fp32 = the ground truth both are measured against

and report, per frame, PSNR(hip, fp32), PSNR(ref, fp32) and their difference. Exit status 1 if any frame of the HIP video is more than
`--threshold_db` (0.1) dB WORSE than the reference-precision chain's frame (being closer to fp32 than the reference's arithmetic is never a failure).
The rendered buffers are shared by the three chains: the renderer's masks / indices are bit-exact against the reference on their own
(tests/test_render_gpu.py), so this isolates the tokenizer + DiT + sampler numerics the 0.1 dB is about.

  python tools/psnr_vs_oracle.py --checkpoint_dir checkpoints [--num_steps 35] [--json out.json]
  python tools/psnr_vs_oracle.py --random_init --tiny --height 64 --width 96 --num_steps 3        # what tests/test_cli_gpu.py runs

Cost at full size (measured, profiles/r4_psnr_fullsize.json: random weights, 1 step): each oracle chain - three tokenizer encodes, two 28-block DiT forwards at 56 320
tokens with the attention scores materialised per head, one decode - takes ~65 s on an MI355X's torch, i.e. `--num_steps 35` is ~1 h for both chains and `--num_steps 3` ~5 min.
Test infrastructure: imports oracle/, never imported by gen3c_amd/."""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def synthetic_scene(H, W):
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = 3.0 + 0.6 * xs / W + 0.3 * ys / H
    depth = np.where((ys - 0.45 * H) ** 2 + (xs - 0.4 * W) ** 2 < (0.22 * H) ** 2, 1.5 + 0.1 * xs / W, depth).astype(np.float32)
    img = np.stack([np.sin(xs * 12.0 / W + c) * np.cos(ys * 9.0 / H - c) for c in range(3)], 0).astype(np.float32)
    K = np.array([[0.8 * W, 0, W / 2], [0, 0.8 * W, H / 2], [0, 0, 1]], np.float32)
    return depth, img, K


def psnr_per_frame(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    mse = ((a.astype(np.float64) - b.astype(np.float64)) ** 2).reshape(a.shape[0], -1).mean(axis=1)
    return 10 * np.log10(1.0 / np.maximum(mse, 1e-12))


def tokenizer_state_dict(args, tk):
    if args.random_init:
        return {k: v.float() for k, v in tk.net.init_random(seed=args.seed).items()}  # (re-seeds the same weights)
    sd = {}
    vae_dir = os.path.join(args.checkpoint_dir, getattr(args, "tokenizer_dir", "Cosmos-Tokenize1-CV8x8x8-720p"))
    for part in ("encoder", "decoder"):
        sd.update(torch.jit.load(os.path.join(vae_dir, f"{part}.jit"), map_location="cpu").state_dict())
    return {k: v.float() for k, v in sd.items() if k in tk.net.expected_keys()}


def main(argv=None) -> int:
    from gen3c_amd import renderer
    from gen3c_amd.camera_utils import generate_camera_trajectory
    from gen3c_amd.cli_common import Session, add_common_args
    from oracle import chain_oracle

    ap = add_common_args(argparse.ArgumentParser(description=__doc__.split("\n\n")[0]))
    ap.add_argument("--threshold_db", type=float, default=0.1)
    ap.add_argument("--json", type=str, default=None, help="write the per-frame table here")
    ap.add_argument("--trajectory", type=str, default="left")
    ap.add_argument("--movement_distance", type=float, default=0.3)
    args = ap.parse_args(argv)
    ses = Session(args)
    dev, H, W, T = ses.dev, args.height, args.width, ses.chunk
    depth, img, K = synthetic_scene(H, W)
    t = lambda a: torch.from_numpy(a).to(dev)
    cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=t(img)[None], input_depth=t(depth)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                    input_intrinsics=t(K)[None], filter_points_threshold=args.filter_points_threshold, foreground_masking=args.foreground_masking,
                                    input_format=["B", "C", "H", "W"])
    w2cs, Ks = generate_camera_trajectory(args.trajectory, torch.eye(4, device=dev), t(K), T, args.movement_distance, "center_facing", center_depth=3.0, device=dev)
    renders, masks = cache.render_cache(w2cs, Ks)
    image = t(img)[None, :, None]
    net, tk = ses.net, ses.tokenizer
    g = torch.Generator().manual_seed(args.seed)
    prompt = ses._emb if ses._emb.abs().sum() > 0 else (0.2 * torch.randn(1, 512, net.crossattn_emb_channels, generator=g)).to(torch.bfloat16)
    if ses._emb.abs().sum() == 0:
        prompt[:, 64:] = 0  # a 64-token prompt, zero-padded like T5's output (t5_text_encoder.py:102-106)
    negp = ses._neg
    ses.model.scheduler.set_timesteps(args.num_steps)
    lat_shape = ses.model.state_shape
    xt = (torch.randn(1, *lat_shape, generator=g) * ses.model.scheduler.init_noise_sigma).to(torch.bfloat16)

    ses.pipe.num_steps, ses.pipe.guidance = args.num_steps, args.guidance
    hip = ses.pipe.generate_from_embeddings(prompt, image.to(torch.bfloat16), renders, masks, negative_prompt_embedding=negp, xt=xt.to(dev)).astype(np.float32) / 255.0

    dit_sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    tok_sd = tokenizer_state_dict(args, tk)
    tok_sd = {k: v.to(torch.bfloat16).float() for k, v in tok_sd.items()}  # the product holds bf16 weights: every chain sees the same values
    dit_sd = {k: (v if k == "pos_embedder.seq" else v.to(torch.bfloat16).float()) for k, v in dit_sd.items()}
    mean, std = tk.latent_mean.float(), tk.latent_std.float()
    common = dict(num_steps=args.num_steps, guidance=args.guidance, num_blocks=net.num_blocks, num_heads=net.num_heads, seed=args.seed, fps=float(args.fps))
    bfr = lambda x: x.to(torch.bfloat16).float()
    videos = {}
    from oracle import tokenizer_oracle
    import time
    if H * W >= 352 * 640:  # the vendor's fp32 conv3d falls back to a naive kernel at these sizes: the oracle's per-tap matmul form (equal: tests/test_tokenizer_oracle_golden.py)
        tokenizer_oracle.CONV_IMPL = "taps"
    with torch.no_grad():
        for name, dt in (("fp32", torch.float32), ("ref", torch.bfloat16)):
            torch.cuda.empty_cache()
            t_chain = time.perf_counter()
            v = chain_oracle.generate_chunk(dit_sd, tok_sd, mean, std, bfr(image), bfr(renders), bfr(masks), prompt.float(), None if negp is None else negp.float(),
                                            xt.float().to(dev), net_dtype=dt, **common)
            videos[name] = np.round(v.clamp(0, 1).cpu().numpy() * 255.0) / 255.0 if name == "ref" else v.cpu().numpy()  # the reference writes uint8 frames too
            del v
            print(f"[psnr_vs_oracle] oracle chain '{name}' took {time.perf_counter() - t_chain:.1f} s", flush=True)
    tokenizer_oracle.CONV_IMPL = "torch"
    p_hip, p_ref = psnr_per_frame(hip, videos["fp32"]), psnr_per_frame(videos["ref"], videos["fp32"])
    delta = p_hip - p_ref
    worst = float(delta.min())
    print(f"[psnr_vs_oracle] {T} frames {H}x{W}, {args.num_steps} steps, guidance {args.guidance}, weights: {'random' if args.random_init else args.checkpoint_dir}")
    print("  frame  PSNR(hip, fp32)  PSNR(reference precision, fp32)  hip - ref [dB]")
    for f in range(T):
        print(f"  {f:5d}  {p_hip[f]:15.2f}  {p_ref[f]:31.2f}  {delta[f]:+14.2f}")
    print(f"  worst frame: hip is {worst:+.2f} dB relative to the reference-precision chain (threshold -{args.threshold_db} dB); "
          f"PSNR(hip, ref) min {float(psnr_per_frame(hip, videos['ref']).min()):.2f} dB")
    ok = worst >= -args.threshold_db
    if args.json:
        Path(args.json).write_text(json.dumps(dict(frames=T, height=H, width=W, num_steps=args.num_steps, guidance=args.guidance, weights="random" if args.random_init else args.checkpoint_dir,
                                                   psnr_hip_vs_fp32=p_hip.round(3).tolist(), psnr_ref_vs_fp32=p_ref.round(3).tolist(), worst_delta_db=worst,
                                                   threshold_db=args.threshold_db, passed=bool(ok)), indent=1))
    ses.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
