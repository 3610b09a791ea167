"""Generates tests/golden/*.npz by running the REFERENCE's own Python (imported read-only from /root/reference
behind tools/ref_shims.py) on seeded inputs. Run in the build container only:

    python tools/gen_golden.py [dit] [warp]

Weights and inputs are rounded to bf16-representable values and stored as uint16 bit patterns (half the bytes; the
fp32 oracle, the reference and the bf16 HIP path then all consume bit-identical operands). Reference outputs are
computed in fp32 and stored as fp32.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
GOLD = ROOT / "tests" / "golden"


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.bfloat16).view(torch.int16).numpy().astype(np.uint16)


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.bfloat16).float()


def gen_dit():
    import ref_shims
    ref_shims.install()
    from cosmos_predict1.diffusion.networks.general_dit_video_conditioned import VideoExtendGeneralDIT

    cases = {
        # name: (D, heads, blocks, ctx_dim, lora, latent T,H,W, M, B)
        "dit_tiny": dict(D=128, heads=1, blocks=2, ctx=64, lora=32, T=2, H=8, W=12, M=24, B=1),
        "dit_small": dict(D=256, heads=2, blocks=1, ctx=128, lora=32, T=3, H=12, W=20, M=40, B=1),
    }
    for name, c in cases.items():
        torch.manual_seed(1234)
        net = VideoExtendGeneralDIT(
            max_img_h=48, max_img_w=48, max_frames=16, in_channels=16 + 16 * 4 + 1, out_channels=16, patch_spatial=2,
            patch_temporal=1, model_channels=c["D"], block_config="FA-CA-MLP", num_blocks=c["blocks"],
            num_heads=c["heads"], concat_padding_mask=True, pos_emb_cls="rope3d", pos_emb_learnable=False,
            pos_emb_interpolation="crop", block_x_format="THWBD", affline_emb_norm=True, use_adaln_lora=True,
            adaln_lora_dim=c["lora"], crossattn_emb_channels=c["ctx"], rope_h_extrapolation_ratio=1.0,
            rope_w_extrapolation_ratio=1.0, rope_t_extrapolation_ratio=2.0,
        ).float().eval()
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n.endswith("adaLN_modulation.2.weight"):
                    p.normal_(0.0, 0.05)  # zero-init would make every block an identity
                if n.endswith("to_q.1.weight") or n.endswith("to_k.1.weight") or n == "affline_norm.weight":
                    p.uniform_(0.5, 1.5)  # exercise the learned RMSNorm weights
                p.copy_(bf16_round(p))
        B, T, H, W, M = c["B"], c["T"], c["H"], c["W"], c["M"]
        x = bf16_round(torch.randn(B, 16, T, H, W))
        pose = bf16_round(0.5 * torch.randn(B, 64, T, H, W))
        mask = torch.zeros(B, 1, T, H, W)
        mask[:, :, :1] = 1.0
        ctx = bf16_round(0.2 * torch.randn(B, M, c["ctx"]))
        ctx[:, M - M // 3:] = 0.0  # zero-padded T5 tail (unmasked in attention)
        timesteps = bf16_round(torch.tensor([0.25 * np.log(3.7)] * B, dtype=torch.float32))
        fps = torch.tensor([24.0] * B)
        padding_mask = torch.zeros(B, 1, 4 * H, 4 * W)
        with torch.no_grad():
            y = net(x=x, timesteps=timesteps, crossattn_emb=ctx, crossattn_mask=None, fps=fps, image_size=None,
                    padding_mask=padding_mask, scalar_feature=None, condition_video_indicator=mask[:, :, :, :1, :1],
                    condition_video_input_mask=mask, condition_video_augment_sigma=None, condition_video_pose=pose)
        out = {"cfg_" + k: np.array(v) for k, v in c.items()}
        for n, p in net.state_dict().items():
            if n.endswith("_extra_state"):
                continue
            if n == "pos_embedder.seq":
                out["w:" + n] = p.numpy().astype(np.float32)
            else:
                out["w:" + n] = bf16_bits(p)
        out.update(x=bf16_bits(x), pose=bf16_bits(pose), mask=mask.numpy().astype(np.float32), ctx=bf16_bits(ctx),
                   timesteps=bf16_bits(timesteps), fps=fps.numpy(), padding_mask=padding_mask.numpy(),
                   y_ref=y.numpy().astype(np.float32))
        GOLD.mkdir(parents=True, exist_ok=True)
        np.savez_compressed(GOLD / f"{name}.npz", **out)
        print(name, "y_ref", tuple(y.shape), "abs mean", float(y.abs().mean()), "file MB",
              (GOLD / f"{name}.npz").stat().st_size / 1e6)


if __name__ == "__main__":
    what = sys.argv[1:] or ["dit"]
    if "dit" in what:
        gen_dit()
    if "warp" in what:
        from gen_golden_warp import gen_warp
        gen_warp()
