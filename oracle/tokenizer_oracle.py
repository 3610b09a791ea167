"""CPU oracle for the causal video tokenizer Cosmos-Tokenize1-CV8x8x8 (TEST INFRASTRUCTURE - never imported by gen3c_amd/).

Functional PyTorch restatement (channels-first, dtype of the given tensors) of the architecture the reference traces into
encoder.jit / decoder.jit (`torch.jit.trace(model.encoder_jit())`, tokenizer/training/jit_cli.py:98-101):
  CausalContinuousVideoTokenizer.encoder_jit / decoder_jit   tokenizer/networks/continuous_video.py:56-75
  EncoderFactorized / DecoderFactorized                      tokenizer/modules/layers3d.py:669-949
  CausalConv3d, CausalHybrid{Down,Up}sample3d, CausalResnetBlockFactorized3d, CausalAttnBlock,
  CausalTemporalAttnBlock                                    tokenizer/modules/layers3d.py:50-97, 135-234, 276-427
  Patcher3D / UnPatcher3D (Haar)                             tokenizer/modules/patching.py:111-175, 250-311
  CausalNormalize                                            tokenizer/modules/utils.py:66-83
and of the latent normalisation of BasePretrainedVideoTokenizer / JITVAE (diffusion/module/pretrained_vae.py:126-152, 342-359).

Pinning: PINNED - tests/golden/tokenizer_small.npz was produced by tools/gen_golden_tokenizer.py running the reference's
own modules (imported from /root/reference) with seeded weights; tests/test_tokenizer_oracle_golden.py replays it.
State-dict keys are the reference's (`encoder.*`, `quant_conv.*`, `post_quant_conv.*`, `decoder.*`).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def causal_conv3d(x: torch.Tensor, sd: SD, pre: str, stride=(1, 1, 1), spatial_pad: int = 0) -> torch.Tensor:
    """CausalConv3d.forward: replicate the first frame `time_pad` times in front, zero-pad H/W, conv3d
    (layers3d.py:50-97). time_pad = (kt - 1) + (1 - time_stride)."""
    w, b = sd[f"{pre}.conv3d.weight"], sd.get(f"{pre}.conv3d.bias")
    kt = w.shape[2]
    time_pad = (kt - 1) + (1 - stride[0])
    if time_pad > 0:
        x = torch.cat([x[:, :, :1].repeat(1, 1, time_pad, 1, 1), x], dim=2)
    if spatial_pad:
        x = F.pad(x, (spatial_pad,) * 4 + (0, 0))
    if CONV_IMPL == "taps":
        return _conv3d_taps(x, w, b, stride)
    return F.conv3d(x, w, b, stride=stride)


def causal_normalize(x: torch.Tensor, sd: SD, pre: str) -> torch.Tensor:
    """GroupNorm(1 group, eps 1e-6, affine) applied per frame (utils.py:66-83, num_groups == 1)."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.group_norm(y, 1, sd[f"{pre}.norm.weight"], sd[f"{pre}.norm.bias"], eps=1e-6)
    return y.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)


def swish(x):
    return x * torch.sigmoid(x)


def res_block(x: torch.Tensor, sd: SD, pre: str) -> torch.Tensor:
    """CausalResnetBlockFactorized3d (layers3d.py:276-342): norm-swish-(1,3,3)-(3,1,1) twice + optional 1x1x1 shortcut."""
    h = swish(causal_normalize(x, sd, f"{pre}.norm1"))
    h = causal_conv3d(h, sd, f"{pre}.conv1.0", spatial_pad=1)
    h = causal_conv3d(h, sd, f"{pre}.conv1.1")
    h = swish(causal_normalize(h, sd, f"{pre}.norm2"))
    h = causal_conv3d(h, sd, f"{pre}.conv2.0", spatial_pad=1)
    h = causal_conv3d(h, sd, f"{pre}.conv2.1")
    if f"{pre}.nin_shortcut.conv3d.weight" in sd:
        x = causal_conv3d(x, sd, f"{pre}.nin_shortcut")
    return x + h


def spatial_attn(x: torch.Tensor, sd: SD, pre: str) -> torch.Tensor:
    """CausalAttnBlock (layers3d.py:345-383): per-frame single-head attention over H*W, scale C^-0.5."""
    h_ = causal_normalize(x, sd, f"{pre}.norm")
    q, k, v = (causal_conv3d(h_, sd, f"{pre}.{n}") for n in ("q", "k", "v"))
    b, c, t, hh, ww = q.shape
    fl = lambda z: z.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh * ww)
    q, k, v = fl(q), fl(k), fl(v)
    w_ = torch.bmm(q.permute(0, 2, 1), k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    o = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)
    return x + causal_conv3d(o, sd, f"{pre}.proj_out")


def temporal_attn(x: torch.Tensor, sd: SD, pre: str) -> torch.Tensor:
    """CausalTemporalAttnBlock (layers3d.py:386-427): per-pixel causal attention over T."""
    h_ = causal_normalize(x, sd, f"{pre}.norm")
    q, k, v = (causal_conv3d(h_, sd, f"{pre}.{n}") for n in ("q", "k", "v"))
    b, c, t, hh, ww = q.shape
    fl = lambda z: z.permute(0, 3, 4, 2, 1).reshape(b * hh * ww, t, c)
    q, k, v = fl(q), fl(k), fl(v)
    w_ = torch.bmm(q, k.permute(0, 2, 1)) * (int(c) ** (-0.5))
    mask = torch.tril(torch.ones_like(w_))
    w_ = F.softmax(w_.masked_fill(mask == 0, float("-inf")), dim=2)
    o = torch.bmm(w_, v).reshape(b, hh, ww, t, c).permute(0, 4, 3, 1, 2)
    return x + causal_conv3d(o, sd, f"{pre}.proj_out")


def hybrid_downsample(x: torch.Tensor, sd: SD, pre: str) -> torch.Tensor:
    """CausalHybridDownsample3d(spatial_down=True, temporal_down=True) (layers3d.py:185-234)."""
    # (pooling evaluated in fp32 and cast back: identical for fp32 inputs; for bf16 inputs it is what a bf16 kernel computes - fp32 accumulation, one
    # rounding - and torch's CPU build has no bf16 avg_pool3d at all)
    pool = lambda v, k: F.avg_pool3d(v.float(), k, k).to(v.dtype)
    x = F.pad(x, (0, 1, 0, 1, 0, 0))
    x = causal_conv3d(x, sd, f"{pre}.conv1", stride=(1, 2, 2)) + pool(x, (1, 2, 2))
    x = torch.cat([x[:, :, :1], x], dim=2)  # replication_pad
    x = causal_conv3d(x, sd, f"{pre}.conv2", stride=(2, 1, 1)) + pool(x, (2, 1, 1))
    return causal_conv3d(x, sd, f"{pre}.conv3")


def hybrid_upsample(x: torch.Tensor, sd: SD, pre: str) -> torch.Tensor:
    """CausalHybridUpsample3d(spatial_up=True, temporal_up=True) (layers3d.py:135-182)."""
    tf = 2 if x.shape[2] > 1 else 1
    x = x.repeat_interleave(tf, dim=2)[:, :, tf - 1:]
    x = causal_conv3d(x, sd, f"{pre}.conv1") + x
    x = x.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
    x = causal_conv3d(x, sd, f"{pre}.conv2", spatial_pad=1) + x
    return causal_conv3d(x, sd, f"{pre}.conv3")


def haar_patch3d(x: torch.Tensor, patch_size: int = 4) -> torch.Tensor:
    """Patcher3D._haar (patching.py:111-175): first frame repeated patch_size times, then log2(patch_size) levels of
    the 2x2x2 Haar analysis, sub-bands concatenated [lll,llh,lhl,lhh,hll,hlh,hhl,hhh] (letters = t,h,w), each level
    divided by 2*sqrt(2)."""
    x = torch.cat([x[:, :, :1].repeat_interleave(patch_size, dim=2), x[:, :, 1:]], dim=2)
    s = 0.7071067811865476
    for _ in range(int(math.log2(patch_size))):
        a, bb = x[:, :, 0::2], x[:, :, 1::2]
        t_l, t_h = (a + bb) * s, (a - bb) * s
        outs = []
        for xt in (t_l, t_h):
            a, bb = xt[:, :, :, 0::2], xt[:, :, :, 1::2]
            for xh in ((a + bb) * s, (a - bb) * s):
                a2, b2 = xh[..., 0::2], xh[..., 1::2]
                outs += [(a2 + b2) * s, (a2 - b2) * s]
        x = torch.cat(outs, dim=1) / (2 * math.sqrt(2.0))
    return x


def haar_unpatch3d(x: torch.Tensor, patch_size: int = 4) -> torch.Tensor:
    """UnPatcher3D._ihaar (patching.py:250-311): inverse levels (w, then h, then t), times 2*sqrt(2), drop the first
    patch_size-1 frames."""
    s = 0.7071067811865476
    for _ in range(int(math.log2(patch_size))):
        bands = torch.chunk(x, 8, dim=1)  # lll llh lhl lhh hll hlh hhl hhh

        def merge(lo, hi, dim):
            even, odd = (lo + hi) * s, (lo - hi) * s
            out = torch.stack([even, odd], dim=dim + 1)
            shape = list(even.shape)
            shape[dim] *= 2
            return out.reshape(shape)

        ll, lh = merge(bands[0], bands[1], 4), merge(bands[2], bands[3], 4)
        hl, hh = merge(bands[4], bands[5], 4), merge(bands[6], bands[7], 4)
        lo_t, hi_t = merge(ll, lh, 3), merge(hl, hh, 3)
        x = merge(lo_t, hi_t, 2) * (2 * math.sqrt(2.0))
    return x[:, :, patch_size - 1:]


def encoder(sd: SD, x: torch.Tensor, num_res_blocks: int = 2, num_levels: int = 3) -> torch.Tensor:
    """encoder_jit = EncoderFactorized -> quant_conv -> identity 'AE' distribution (continuous_video.py:56-65).
    x: [B,3,T,H,W] -> [B,16,1+(T-1)/8,H/8,W/8]."""
    h = haar_patch3d(x)
    h = causal_conv3d(h, sd, "encoder.conv_in.0", spatial_pad=1)
    h = causal_conv3d(h, sd, "encoder.conv_in.1")
    for lvl in range(num_levels):
        for j in range(num_res_blocks):
            h = res_block(h, sd, f"encoder.down.{lvl}.block.{j}")
        if f"encoder.down.{lvl}.downsample.conv1.conv3d.weight" in sd:
            h = hybrid_downsample(h, sd, f"encoder.down.{lvl}.downsample")
    h = res_block(h, sd, "encoder.mid.block_1")
    h = spatial_attn(h, sd, "encoder.mid.attn_1.0")
    h = temporal_attn(h, sd, "encoder.mid.attn_1.1")
    h = res_block(h, sd, "encoder.mid.block_2")
    h = swish(causal_normalize(h, sd, "encoder.norm_out"))
    h = causal_conv3d(h, sd, "encoder.conv_out.0", spatial_pad=1)
    h = causal_conv3d(h, sd, "encoder.conv_out.1")
    return causal_conv3d(h, sd, "quant_conv")


def decoder(sd: SD, z: torch.Tensor, num_res_blocks: int = 2, num_levels: int = 3) -> torch.Tensor:
    """decoder_jit = post_quant_conv -> DecoderFactorized (continuous_video.py:67-75). z: [B,16,t,h,w] -> [B,3,1+8(t-1),8h,8w]."""
    h = causal_conv3d(z, sd, "post_quant_conv")
    h = causal_conv3d(h, sd, "decoder.conv_in.0", spatial_pad=1)
    h = causal_conv3d(h, sd, "decoder.conv_in.1")
    h = res_block(h, sd, "decoder.mid.block_1")
    h = spatial_attn(h, sd, "decoder.mid.attn_1.0")
    h = temporal_attn(h, sd, "decoder.mid.attn_1.1")
    h = res_block(h, sd, "decoder.mid.block_2")
    for lvl in reversed(range(num_levels)):
        for j in range(num_res_blocks + 1):
            h = res_block(h, sd, f"decoder.up.{lvl}.block.{j}")
        if f"decoder.up.{lvl}.upsample.conv1.conv3d.weight" in sd:
            h = hybrid_upsample(h, sd, f"decoder.up.{lvl}.upsample")
    h = swish(causal_normalize(h, sd, "decoder.norm_out"))
    h = causal_conv3d(h, sd, "decoder.conv_out.0", spatial_pad=1)
    h = causal_conv3d(h, sd, "decoder.conv_out.1")
    return haar_unpatch3d(h)


def encode(sd: SD, state: torch.Tensor, latent_mean: torch.Tensor, latent_std: torch.Tensor) -> torch.Tensor:
    """VideoJITTokenizer.encode for one chunk: (encoder(state) - mean) / std (pretrained_vae.py:126-142)."""
    return (encoder(sd, state) - latent_mean) / latent_std


def decode(sd: SD, latent: torch.Tensor, latent_mean: torch.Tensor, latent_std: torch.Tensor) -> torch.Tensor:
    """VideoJITTokenizer.decode for one chunk (pretrained_vae.py:144-152)."""
    return decoder(sd, latent * latent_std + latent_mean)


# ---- convolution back end of the oracle ---------------------------------------------------------------------------------------------
# "torch": F.conv3d (the form the golden fixtures pin). "taps": the same convolution written as one matmul per kernel tap over shifted
# views of the padded input - for the full-size on-device evaluation (tests/test_fullsize_gpu.py), where the vendor's fp32 conv3d may
# fall back to a naive kernel. tests/test_tokenizer_oracle_golden.py holds "taps" against "torch" on every geometry of the network.
CONV_IMPL = "torch"


def _conv3d_taps(x: torch.Tensor, w: torch.Tensor, b, stride) -> torch.Tensor:
    """x [B,Cin,T,H,W] (already padded), w [Cout,Cin,kt,kh,kw] -> [B,Cout,To,Ho,Wo]; out = sum over taps of x_shift @ w_tap^T."""
    B, Cin, T, H, W = x.shape
    Cout, _, kt, kh, kw = w.shape
    st, sh, sw = stride
    To, Ho, Wo = (T - kt) // st + 1, (H - kh) // sh + 1, (W - kw) // sw + 1
    xl = x.permute(0, 2, 3, 4, 1).contiguous()  # channels last
    out = None
    for a in range(kt):
        for i in range(kh):
            for j in range(kw):
                v = xl[:, a:a + (To - 1) * st + 1:st, i:i + (Ho - 1) * sh + 1:sh, j:j + (Wo - 1) * sw + 1:sw]
                y = torch.matmul(v.reshape(-1, Cin), w[:, :, a, i, j].t())
                out = y if out is None else out.add_(y)
    if b is not None:
        out = out + b
    return out.view(B, To, Ho, Wo, Cout).permute(0, 4, 1, 2, 3)
