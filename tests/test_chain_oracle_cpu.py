"""CPU: oracle/chain_oracle.py (the one-chunk oracle behind tools/psnr_vs_oracle.py) assembled from the pinned component oracles runs end to end on the golden
weights (tests/golden/dit_tiny.npz + tokenizer_small.npz, both produced by the reference's own classes): shapes, determinism, the order-based cond / uncond
dispatch, and the relation the harness reports - the reference-precision (bf16) chain stays close to the fp32 chain."""
import numpy as np
import torch

from oracle import chain_oracle, sampler_oracle
from tests.golden_io import load_dit_case, load_tokenizer_case


def _inputs():
    cfg, dsd, inp, _ = load_dit_case("dit_tiny")
    tsd, *_ = load_tokenizer_case()
    g = torch.Generator().manual_seed(3)
    T, H, W = 9, 32, 48
    image = (torch.rand(1, 3, 1, H, W, generator=g) * 2 - 1).to(torch.bfloat16).float()
    renders = (torch.rand(1, T, 1, 3, H, W, generator=g) * 2 - 1).to(torch.bfloat16).float()
    masks = (torch.rand(1, T, 1, 1, H, W, generator=g) > 0.3).float()
    prompt = (0.2 * torch.randn(1, 24, cfg["ctx"], generator=g)).to(torch.bfloat16).float()
    negp = (0.2 * torch.randn(1, 24, cfg["ctx"], generator=g)).to(torch.bfloat16).float()
    xt = (torch.randn(1, 16, 2, H // 8, W // 8, generator=g) * float((sampler_oracle.SIGMA_MAX ** 2 + 1) ** 0.5)).to(torch.bfloat16).float()
    mean, std = torch.zeros(1, 16, 2, 1, 1), torch.ones(1, 16, 2, 1, 1)
    sd = {k: v.float() for k, v in dsd.items()}
    tk = {k: v.float() for k, v in tsd.items()}
    kw = dict(num_steps=3, guidance=1.5, num_blocks=cfg["blocks"], num_heads=cfg["heads"])
    return sd, tk, mean, std, image, renders, masks, prompt, negp, xt, kw


def test_chain_oracle_runs_on_the_golden_weights_and_bf16_stays_near_fp32():
    sd, tk, mean, std, image, renders, masks, prompt, negp, xt, kw = _inputs()
    with torch.no_grad():
        v32 = chain_oracle.generate_chunk(sd, tk, mean, std, image, renders, masks, prompt, negp, xt, **kw)
        v32b = chain_oracle.generate_chunk(sd, tk, mean, std, image, renders, masks, prompt, negp, xt, **kw)
        v16 = chain_oracle.generate_chunk(sd, tk, mean, std, image, renders, masks, prompt, negp, xt, net_dtype=torch.bfloat16, **kw)
        lat = chain_oracle.generate_chunk(sd, tk, mean, std, image, renders, masks, prompt, negp, xt, return_latent=True, **kw)
    assert v32.shape == (9, 32, 48, 3) and torch.isfinite(v32).all() and float(v32.min()) >= 0 and float(v32.max()) <= 1
    assert torch.equal(v32, v32b) and lat.shape == (1, 16, 2, 4, 6)
    mse = ((v16 - v32) ** 2).reshape(9, -1).mean(dim=1)
    psnr = 10 * torch.log10(1.0 / mse.clamp_min(1e-12))
    print("[chain oracle, golden weights] reference-precision chain vs fp32 chain, per-frame PSNR:", " ".join(f"{p:.1f}" for p in psnr.tolist()))
    assert float(psnr.min()) > 25.0


def test_chain_oracle_uses_the_negative_prompt_for_the_unconditional_branch_only():
    sd, tk, mean, std, image, renders, masks, prompt, negp, xt, kw = _inputs()
    with torch.no_grad():
        a = chain_oracle.generate_chunk(sd, tk, mean, std, image, renders, masks, prompt, negp, xt, return_latent=True, **kw)
        b = chain_oracle.generate_chunk(sd, tk, mean, std, image, renders, masks, prompt, None, xt, return_latent=True, **kw)  # no negative prompt: the positive text
        c = chain_oracle.generate_chunk(sd, tk, mean, std, image, renders, masks, prompt, prompt, xt, return_latent=True, **kw)
    assert torch.equal(b, c) and not torch.equal(a, b)
    # zero renders: the conditional pose is all zero too - the dispatch must still be by call order, not by the pose's content
    z = torch.zeros_like(renders)
    zm = torch.zeros_like(masks)
    with torch.no_grad():
        d = chain_oracle.generate_chunk(sd, tk, mean, std, image, z, zm, prompt, negp, xt, return_latent=True, **kw)
        e = chain_oracle.generate_chunk(sd, tk, mean, std, image, z, zm, prompt, prompt, xt, return_latent=True, **kw)
    assert not torch.equal(d, e)
