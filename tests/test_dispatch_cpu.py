"""CPU checks of host-side dispatch logic added in round 2 (no GPU needed: the library loads and these entry points do not launch)."""
import torch

from gen3c_amd import _lib
from gen3c_amd.sampler import Gen3CDenoiser, VideoExtendCondition


def _name(Sq, Skv, B, H):
    return _lib.load().g3_flash_attn_kernel_name(Sq, Skv, B, H).decode()


def test_attention_kernel_choice_follows_the_fill_rule():
    lib = _lib.load()
    lib.g3_set_option(b"attn_variant", 0)
    # full-size self-attention, one launch over all heads: 220 x 32 workgroups of 256 rows = 27.5 rounds of the 256 CUs -> one-wave kernel
    assert _name(56320, 56320, 1, 32) == "flash_attn_fwd_w4b_kernel<true>"
    assert _name(56320, 56320, 2, 32) == "flash_attn_fwd_w4b_kernel<true>"
    # a context-parallel head group on its own (8 heads): 3.4 / 1.7 / 0.9 rounds -> the 8-wave kernel (128-row workgroups fill better)
    for sq in (28160, 14080, 7040):
        assert _name(sq, 56320, 1, 8).startswith("flash_attn_fwd_v3_kernel<0"), sq
    # ragged context (not a whole number of 64-key tiles) and short contexts never take the one-wave kernel
    assert _name(56320, 56321, 1, 32).startswith("flash_attn_fwd_v3_kernel<0")
    assert _name(56320, 512, 1, 32).startswith("flash_attn_fwd_v3_kernel<1")
    # explicit variants are honoured (A/B runs), 10 / 11 fall back to w4 on ragged contexts
    lib.g3_set_option(b"attn_variant", 11)
    try:
        assert _name(7040, 56320, 1, 8) == "flash_attn_fwd_w4b_kernel<true>"
        assert _name(7040, 56321, 1, 8) == "flash_attn_fwd_w4_kernel<0>"
    finally:
        lib.g3_set_option(b"attn_variant", 0)


def _cond(ctx, pose, flag=True):
    return VideoExtendCondition(crossattn_emb=ctx, padding_mask=torch.zeros(1, 1, 16, 16), fps=torch.tensor([24.0]), video_cond_bool=flag,
                                condition_video_pose=pose, condition_video_input_mask=torch.ones(1, 1, 2, 2, 2),
                                condition_video_indicator=torch.ones(1, 1, 2, 1, 1), gt_latent=torch.zeros(1, 4, 2, 2, 2))


def test_fused_cond_uncond_arguments():
    c = _cond(torch.randn(1, 5, 8), torch.randn(1, 3, 2, 2, 2))
    u = _cond(torch.randn(1, 5, 8), torch.zeros(1, 3, 2, 2, 2))
    f = Gen3CDenoiser._fused_cond_uncond_kwargs(c, u, 1)
    assert f is not None
    assert f["crossattn_emb"].shape == (2, 5, 8) and torch.equal(f["crossattn_emb"][0], c.crossattn_emb[0]) and torch.equal(f["crossattn_emb"][1], u.crossattn_emb[0])
    assert f["condition_video_pose"].shape == (2, 3, 2, 2, 2) and f["fps"].shape == (2,) and f["padding_mask"].shape == (2, 1, 16, 16)
    assert f["video_cond_bool"] is True and f["crossattn_mask"] is None and f["gt_latent"] is c.gt_latent
    # conditions that cannot share one call: different flags / shapes -> None (the sampler then makes two calls)
    assert Gen3CDenoiser._fused_cond_uncond_kwargs(c, _cond(torch.randn(1, 5, 8), torch.zeros(1, 3, 2, 2, 2), flag=False), 1) is None
    assert Gen3CDenoiser._fused_cond_uncond_kwargs(c, _cond(torch.randn(1, 6, 8), torch.zeros(1, 3, 2, 2, 2)), 1) is None


def test_cache_keys_work_on_inference_tensors():
    """the reference wraps its pipeline entry points in torch.inference_mode() (world_generation_pipeline.py:1225); inference
    tensors raise on `_version`. The cache keys must not."""
    from gen3c_amd.dit import VideoExtendGeneralDIT, cacheable, tensor_version
    with torch.inference_mode():
        t = torch.zeros(3)
    assert t.is_inference() and tensor_version(t) is None  # = do not cache on this tensor
    u = torch.zeros(3)
    assert cacheable(u, None, 3.0) and not cacheable(u, t)
    v0 = tensor_version(u)
    u.add_(1)
    assert tensor_version(u) == v0 + 1
    with torch.inference_mode():
        net = VideoExtendGeneralDIT(max_img_h=16, max_img_w=16, max_frames=8, in_channels=81, model_channels=128, num_blocks=1, num_heads=1,
                                    adaln_lora_dim=8, crossattn_emb_channels=16, device="cpu", init_weights=True)
        k1 = net._weights_key()  # parameters created under inference mode
    assert k1 == net._weights_key()
