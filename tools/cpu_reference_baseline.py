"""Reference-side CPU baseline (SURVEY.md 8d "CPU baseline"). Build container only: times the REFERENCE's own Python
(imported read-only from /root/reference behind tools/ref_shims.py) on this container's host cores and writes
profiles/r4_cpu_reference.json. Three stages, each on a stated, bounded sample:

  dit        cosmos_predict1.diffusion.networks.general_dit_video_conditioned.VideoExtendGeneralDIT at the 7B width (D = 4096, 32 heads,
             MLP 16 384, context 512 x 1024) on BASELINE.json configs[0]'s latent 16 x 64 x 64 = 16 384 tokens, one forward; `--blocks n` of the
             28 blocks are instantiated (every block has the same shapes; per-block additivity: profiles/r3_cpu_baseline_linearity.txt);
             bf16 parameters / activations as the reference runs it (config/base/model.py:29) and fp32.
             Third-party arithmetic behind the shims: TE RMSNorm / DotProductAttention -> torch (F.scaled_dot_product_attention).
  render     forward_warp_utils_pytorch.forward_warp (:171-336) on CPU tensors, 704 x 1280, pairs of items as Cache3D_Base.render_cache calls it
             (cache_3d.py:163-214), without foreground masking; with foreground masking the reference needs NVIDIA Warp on a CUDA device
             (ray_triangle_intersection_warp.py) - its hook is pointed at the oracle's C restatement (oracle/c/ray_tri.c, OpenMP over the rays)
             and that leg is labelled "reference + port".
  tokenizer  TokenizerModels.CV (CV8x8x8_720p, channels = 128: EncoderFactorized / DecoderFactorized, layers3d.py:669-949) encoder_jit /
             decoder_jit on one clip (`--clip T,H,W`, default the full 121 x 704 x 1280), fp32 and bf16.

  python tools/cpu_reference_baseline.py [dit] [render] [tokenizer] [--threads 8] [--blocks 2] [--clip 121,704,1280]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT))
OUT = ROOT / "profiles" / "r4_cpu_reference.json"


def dit_forward_flops(N, D=4096, M=512, Dctx=1024, L=28, patch_dim=328, out_dim=64):
    return L * (28 * N * D * D + 4 * N * N * D + 4 * N * M * D + 4 * M * Dctx * D) + 2 * N * patch_dim * D + 2 * N * D * out_dim


def cpu_info(threads):
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(cores=os.cpu_count(), threads_used=threads, cpu=model, torch=torch.__version__)


def time_dit(blocks: int, dtype: torch.dtype):
    from cosmos_predict1.diffusion.networks.general_dit_video_conditioned import VideoExtendGeneralDIT
    torch.manual_seed(0)
    net = VideoExtendGeneralDIT(
        max_img_h=240, max_img_w=240, max_frames=128, in_channels=16 + 16 * 4 + 1, out_channels=16, patch_spatial=2, patch_temporal=1,
        model_channels=4096, block_config="FA-CA-MLP", num_blocks=blocks, num_heads=32, concat_padding_mask=True, pos_emb_cls="rope3d",
        pos_emb_learnable=False, pos_emb_interpolation="crop", block_x_format="THWBD", affline_emb_norm=True, use_adaln_lora=True,
        adaln_lora_dim=256, crossattn_emb_channels=1024, rope_h_extrapolation_ratio=1.0, rope_w_extrapolation_ratio=1.0, rope_t_extrapolation_ratio=2.0,
    ).eval()
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("adaLN_modulation.2.weight"):
                p.normal_(0.0, 0.02)
    net = net.to(dtype)
    B, T, H, W, M = 1, 16, 64, 64, 512
    x = torch.randn(B, 16, T, H, W).to(dtype)
    pose = (0.5 * torch.randn(B, 64, T, H, W)).to(dtype)
    mask = torch.zeros(B, 1, T, H, W, dtype=dtype)
    mask[:, :, :1] = 1
    ctx = (0.2 * torch.randn(B, M, 1024)).to(dtype)
    ctx[:, 64:] = 0
    kw = dict(x=x, timesteps=torch.tensor([0.3], dtype=dtype), crossattn_emb=ctx, crossattn_mask=None, fps=torch.tensor([24.0]), image_size=None,
              padding_mask=torch.zeros(B, 1, 8 * H, 8 * W, dtype=dtype), scalar_feature=None, condition_video_indicator=mask[:, :, :, :1, :1],
              condition_video_input_mask=mask, condition_video_augment_sigma=None, condition_video_pose=pose)
    t0 = time.perf_counter()
    with torch.no_grad():
        y = net(**kw)
    dt = time.perf_counter() - t0
    assert torch.isfinite(y.float()).all()
    N = T * (H // 2) * (W // 2)
    fl = dit_forward_flops(N, L=blocks)
    full = dit_forward_flops(N, L=28)
    return dict(dtype=str(dtype).replace("torch.", ""), blocks=blocks, tokens=N, seconds=round(dt, 2), tflops=round(fl / dt / 1e12, 4),
                forward_seconds_28_blocks=round(dt * full / fl, 1), forwards_per_sec=round(fl / dt / full, 6),
                # the headline step: 2 forwards at 56 320 tokens (4.419 PFLOP), extrapolated by FLOPs
                steps_per_sec_at_configs1=fl / dt / (2 * dit_forward_flops(56320)))


def bench_scene(h, w):
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    depth = 4.0 + 0.0004 * xs + 0.0002 * ys
    for (cy, cx, r, zz) in ((h * 0.4, w * 0.3, h * 0.22, 1.6), (h * 0.65, w * 0.7, h * 0.18, 2.4)):
        depth = np.where((ys - cy) ** 2 + (xs - cx) ** 2 < r * r, zz + 0.0001 * xs, depth)
    img = np.stack([np.sin(xs * 0.021 + c) * np.cos(ys * 0.017 - c) for c in range(3)], 0).astype(np.float32)
    K = np.array([[1000.0, 0, w / 2], [0, 1000.0, h / 2], [0, 0, 1]], np.float32)
    return depth.astype(np.float32), img, K


def time_render(pairs: int):
    """bench.py's scene (plane + 2 discs, K = 1000 / (640, 352), camera sliding 0.3 to the left over 32 frames)."""
    from cosmos_predict1.diffusion.inference import forward_warp_utils_pytorch as fwu
    from oracle import warp_oracle
    h, w, F = 704, 1280, 32
    depth, img, K = bench_scene(h, w)
    depth_t, K_t = torch.from_numpy(depth)[None, None], torch.from_numpy(K)[None]
    pts = fwu.unproject_points(depth_t, torch.eye(4)[None], K_t)
    rel = fwu.reliable_depth_mask_range_batch(depth_t, ratio_thresh=0.05)
    bnd = ~fwu.reliable_depth_mask_range_batch(depth_t)

    def rt_hook(ray_origins, ray_directions, vertices, faces, device):
        tris = vertices.numpy()[faces.numpy()]
        return torch.from_numpy(warp_oracle.ray_triangle_depth_c(ray_directions.numpy(), tris))

    fwu._warp_initialized = True
    fwu._ray_triangle_intersection_func = rt_hook
    out = {}
    for fg in (False, True):
        n_pairs = pairs if not fg else max(1, pairs // 2)
        t0 = time.perf_counter()
        for j in range(n_pairs):
            w2cs = torch.eye(4).repeat(2, 1, 1)
            w2cs[:, 0, 3] = torch.tensor([0.3 * (2 * j) / (F - 1), 0.3 * (2 * j + 1) / (F - 1)])
            Ks = K_t.expand(2, 3, 3).contiguous()
            wf, m2, d2, _ = fwu.forward_warp(torch.from_numpy(img)[None].expand(2, 3, h, w).contiguous(), mask1=rel.float().expand(2, 1, h, w).contiguous(),
                                             depth1=None, transformation1=None, transformation2=w2cs, intrinsic1=Ks, intrinsic2=Ks, render_depth=False,
                                             world_points1=pts.expand(2, h, w, 3).contiguous(), foreground_masking=fg,
                                             boundary_mask=bnd[:, 0].expand(2, h, w).contiguous() if fg else None)
        dt = time.perf_counter() - t0
        per_item = dt / (2 * n_pairs)
        out["foreground_masking" if fg else "plain"] = dict(
            items=2 * n_pairs, seconds=round(dt, 2), ms_per_item=round(per_item * 1e3, 1), gb_per_s_algorithmic=round(43.2e6 / per_item / 1e9, 4),
            kind="reference" if not fg else "reference + port (the Warp ray x triangle kernel replaced by oracle/c/ray_tri.c: Warp needs a CUDA device)",
            mask_coverage=round(float(m2.mean()), 4))
    return out


def time_tokenizer(clip, dtype: torch.dtype):
    from cosmos_predict1.tokenizer.networks import TokenizerConfigs, TokenizerModels
    torch.manual_seed(7)
    cfg = dict(TokenizerConfigs.CV8x8x8_720p.value)
    model = TokenizerModels.CV.value(**cfg).eval().to(dtype)
    T, H, W = clip
    x = (torch.rand(1, 3, T, H, W) * 2 - 1).to(dtype)
    out = dict(dtype=str(dtype).replace("torch.", ""), clip=list(clip), channels=cfg["channels"])
    full = dict(encode=35.7, decode=61.3)  # TFLOP at 121 x 704 x 1280
    scale = (1 + (T - 1) / 8) * H * W / (16 * 704 * 1280)  # ~ latent volume (attention is a sixth of the work: slightly optimistic for small clips)
    with torch.no_grad():
        enc, dec = model.encoder_jit(), model.decoder_jit()
        t0 = time.perf_counter()
        z = enc(x)
        z = z[0] if isinstance(z, tuple) else z
        te = time.perf_counter() - t0
        t0 = time.perf_counter()
        y = dec(z)
        td = time.perf_counter() - t0
    assert torch.isfinite(z.float()).all() and torch.isfinite(y.float()).all()
    for name, dt in (("encode", te), ("decode", td)):
        out[name] = dict(seconds=round(dt, 2), tflops=round(full[name] * scale / dt, 4))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["dit", "render", "tokenizer"])
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--blocks", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--clip", type=str, default="121,704,1280")
    ap.add_argument("--dtypes", type=str, default="bfloat16,float32")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    import ref_shims
    ref_shims.install()
    res = json.loads(OUT.read_text()) if OUT.exists() else {}
    res["host"] = cpu_info(args.threads)
    res["source"] = "the reference's own Python from /root/reference behind tools/ref_shims.py (kind = reference); tools/cpu_reference_baseline.py"
    dts = [getattr(torch, d) for d in args.dtypes.split(",")]
    if "dit" in args.what:
        res["dit"] = [time_dit(args.blocks, dt) for dt in dts]
        print(json.dumps(res["dit"]), flush=True)
        OUT.write_text(json.dumps(res, indent=1))
    if "render" in args.what:
        res["render"] = time_render(args.pairs)
        print(json.dumps(res["render"]), flush=True)
        OUT.write_text(json.dumps(res, indent=1))
    if "tokenizer" in args.what:
        res["tokenizer"] = []
        for dt in dts:
            try:
                res["tokenizer"].append(time_tokenizer(tuple(int(v) for v in args.clip.split(",")), dt))
            except NotImplementedError as e:  # torch 2.10 CPU: "avg_pool3d_out_frame" not implemented for 'BFloat16' (CausalHybridDownsample3d, layers3d.py:222)
                res["tokenizer"].append(dict(dtype=str(dt).replace("torch.", ""), error=f"the reference's modules do not run in this dtype on CPU: {e}"))
        print(json.dumps(res["tokenizer"]), flush=True)
    OUT.write_text(json.dumps(res, indent=1))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
