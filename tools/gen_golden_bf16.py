"""tests/golden/dit_bf16_ref.npz: the reference's OWN VideoExtendGeneralDIT run the way the reference runs it - parameters and activations
in bf16 (`precision="bfloat16"`, config/base/model.py:29) - on the inputs and weights of tests/golden/dit_{tiny,small}.npz (whose y_ref is the same
class in fp32). CPU kernels of this container's torch; TransformerEngine's RMSNorm / attention behind tools/ref_shims.py (fp32 statistics, bf16
in / out; F.scaled_dot_product_attention in bf16). tests/test_reference_precision.py puts the HIP path's distance to fp32 next to this one's."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT))


def main():
    import ref_shims
    ref_shims.install()
    from cosmos_predict1.diffusion.networks.general_dit_video_conditioned import VideoExtendGeneralDIT
    from tests.golden_io import load_dit_case
    out = {}
    for name in ("dit_tiny", "dit_small"):
        c, sd, inp, y_ref = load_dit_case(name)
        net = VideoExtendGeneralDIT(
            max_img_h=48, max_img_w=48, max_frames=16, in_channels=16 + 16 * 4 + 1, out_channels=16, patch_spatial=2, patch_temporal=1,
            model_channels=c["D"], block_config="FA-CA-MLP", num_blocks=c["blocks"], num_heads=c["heads"], concat_padding_mask=True,
            pos_emb_cls="rope3d", pos_emb_learnable=False, pos_emb_interpolation="crop", block_x_format="THWBD", affline_emb_norm=True,
            use_adaln_lora=True, adaln_lora_dim=c["lora"], crossattn_emb_channels=c["ctx"], rope_h_extrapolation_ratio=1.0,
            rope_w_extrapolation_ratio=1.0, rope_t_extrapolation_ratio=2.0).eval()
        missing = net.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
        assert not [k for k in missing.missing_keys if not k.endswith("_extra_state")], missing
        B, T, H, W = inp["x"].shape[0], inp["x"].shape[2], inp["x"].shape[3], inp["x"].shape[4]
        mask = inp["mask"]

        def run(dtype):
            m = net.to(dtype)
            with torch.no_grad():
                return m(x=inp["x"].to(dtype), timesteps=inp["timesteps"].to(dtype), crossattn_emb=inp["ctx"].to(dtype), crossattn_mask=None, fps=inp["fps"],
                         image_size=None, padding_mask=inp["padding_mask"].to(dtype), scalar_feature=None,
                         condition_video_indicator=mask[:, :, :, :1, :1].to(dtype), condition_video_input_mask=mask.to(dtype),
                         condition_video_augment_sigma=None, condition_video_pose=inp["pose"].to(dtype)).float()

        y32 = run(torch.float32)
        assert torch.allclose(y32, y_ref, rtol=1e-5, atol=1e-5), "fp32 re-run does not reproduce the committed golden"
        y16 = run(torch.bfloat16)
        rel = float((y16 - y_ref).norm() / y_ref.norm())
        print(name, "reference class in bf16 vs the same class in fp32: rel-L2", f"{rel:.3e}", "max-abs / max|y|", float((y16 - y_ref).abs().max() / y_ref.abs().max()))
        out[f"{name}_y_bf16"] = y16.numpy().astype(np.float32)
    np.savez_compressed(ROOT / "tests" / "golden" / "dit_bf16_ref.npz", **out)


if __name__ == "__main__":
    main()
