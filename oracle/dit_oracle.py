"""CPU oracle for the GEN3C DiT denoiser (TEST INFRASTRUCTURE - never imported by gen3c_amd/).

A plain-PyTorch, functional restatement of `VideoExtendGeneralDIT.forward` over a reference-named state dict.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinning status: PINNED against the reference's own Python for everything that lives under /root/reference
(tests/golden/dit_*.npz were produced by tools/gen_golden.py importing cosmos_predict1.* from /root/reference with
shims for the absent third-party packages, tests/test_oracle_golden.py replays them). The arithmetic that lives in
third-party packages which are NOT vendored in the reference is restated from their published semantics and is
"parity unpinned" (SURVEY.md 8c):
  * transformer-engine 1.12.0  RMSNorm / DotProductAttention  (INSTALL.md:20)
  * diffusers 0.32.2 EDMEulerScheduler (oracle/sampler_oracle.py)
TE's apply_rotary_pos_emb IS pinned since round 3: te_rope_fused reproduces bit for bit the reference's own copy of it
(autoregressive/modules/embedding.py:46-85; tests/golden/te_rope.npz, tests/test_te_rope_golden.py).

Every function cites the reference lines it follows. All math runs in the dtype of the given tensors (use fp32
weights/inputs for the parity oracle; bf16 reproduces the reference's rounding points for the CPU baseline).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


def timestep_sinusoid(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """Timesteps.forward - cos|sin halves, fp32 internally (module/blocks.py:38-51)."""
    half = dim // 2
    expo = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / (half - 0.0)
    ang = timesteps[:, None].float() * torch.exp(expo)[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).to(timesteps.dtype)


def te_rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """transformer_engine.pytorch.RMSNorm: fp32 math, weight multiply in fp32, cast back (attention.py:130-131)."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * weight.float()).to(x.dtype)


def te_rope_fused(t: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """apply_rotary_pos_emb(t [s,b,h,d], freqs [s,1,1,d], 'sbhd', fused=True): fp32 t*cos + rotate_half(t)*sin,
    rotate_half = cat(-t[d/2:], t[:d/2]) (attention.py:277-279; corroborated in-repo by
    autoregressive/modules/embedding.py:46-85)."""
    tf = t.float()
    half = tf.shape[-1] // 2
    rot = torch.cat([-tf[..., half:], tf[..., :half]], dim=-1)
    return (tf * torch.cos(freqs) + rot * torch.sin(freqs)).to(t.dtype)


def rope_freqs(seq: torch.Tensor, head_dim: int, T: int, H: int, W: int, fps: Optional[torch.Tensor],
               h_ratio: float, w_ratio: float, t_ratio: float, base_fps: int = 24) -> torch.Tensor:
    """VideoRopePosition3DEmb.generate_embeddings -> [(T H W), 1, 1, head_dim] fp32 (position_embedding.py:106-187)."""
    dim_h = head_dim // 6 * 2
    dim_w = dim_h
    dim_t = head_dim - 2 * dim_h
    rs = torch.arange(0, dim_h, 2, device=seq.device)[: dim_h // 2].float() / dim_h
    rt = torch.arange(0, dim_t, 2, device=seq.device)[: dim_t // 2].float() / dim_t
    h_theta = 10000.0 * h_ratio ** (dim_h / (dim_h - 2))
    w_theta = 10000.0 * w_ratio ** (dim_w / (dim_w - 2))
    t_theta = 10000.0 * t_ratio ** (dim_t / (dim_t - 2))
    fh, fw, ft = 1.0 / (h_theta ** rs), 1.0 / (w_theta ** rs), 1.0 / (t_theta ** rt)
    eh = torch.outer(seq[:H].float(), fh)
    ew = torch.outer(seq[:W].float(), fw)
    if fps is None:
        assert T == 1
        et = torch.outer(seq[:T].float(), ft)
    else:
        et = torch.outer(seq[:T].float() / fps[:1].float() * base_fps, ft)
    half = torch.cat([et[:, None, None, :].expand(T, H, W, -1), eh[None, :, None, :].expand(T, H, W, -1),
                      ew[None, None, :, :].expand(T, H, W, -1)], dim=-1)
    return torch.cat([half, half], dim=-1).reshape(T * H * W, 1, 1, head_dim).float()


def abs_pos_emb(pos_t: torch.Tensor, pos_h: torch.Tensor, pos_w: torch.Tensor, B: int, T: int, H: int, W: int) -> torch.Tensor:
    """LearnablePosEmbAxis.generate_embeddings + normalize(dim=-1, eps=1e-6) -> [B,T,H,W,D]
    (position_embedding.py:218-233; attention.py:108-124)."""
    emb = (pos_t[:T][None, :, None, None, :] + pos_h[:H][None, None, :, None, :]) + pos_w[:W][None, None, None, :, :]
    emb = emb.expand(B, T, H, W, -1)
    norm = torch.linalg.vector_norm(emb, dim=-1, keepdim=True, dtype=torch.float32)
    norm = torch.add(1e-6, norm, alpha=math.sqrt(norm.numel() / emb.numel()))
    return emb / norm.to(emb.dtype)


def attention_sbhd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """DotProductAttention(no mask, scale 1/sqrt(d)) on [s,b,h,d] -> [s,b,h*d], fp32 softmax (attention.py:228-238,288)."""
    d = q.shape[-1]
    qf, kf, vf = (t.permute(1, 2, 0, 3).float() for t in (q, k, v))  # b h s d
    # heads are independent: long sequences go through in head groups so the score matrix stays below ~8 GB (same arithmetic per head)
    hg = max(1, min(qf.shape[1], int(2e9 // max(1, qf.shape[0] * qf.shape[2] * kf.shape[2]))))
    outs = []
    for h0 in range(0, qf.shape[1], hg):
        scores = torch.matmul(qf[:, h0:h0 + hg], kf[:, h0:h0 + hg].transpose(-1, -2)) / math.sqrt(d)
        p = torch.softmax(scores, dim=-1)
        outs.append(torch.matmul(p.to(v.dtype).float(), vf[:, h0:h0 + hg]))
    o = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
    return o.permute(2, 0, 1, 3).reshape(q.shape[0], q.shape[1], -1).to(q.dtype)


def _attn_module(sd: Dict[str, torch.Tensor], pre: str, x: torch.Tensor, ctx: Optional[torch.Tensor], heads: int,
                 rope: Optional[torch.Tensor]) -> torch.Tensor:
    """Attention.forward = cal_qkv + cal_attn (attention.py:247-313): per-head RMSNorm on q,k ("RRI"), RoPE on
    self-attention only, unmasked attention, output projection."""
    src = x if ctx is None else ctx
    q = F.linear(x, sd[f"{pre}.to_q.0.weight"])
    k = F.linear(src, sd[f"{pre}.to_k.0.weight"])
    v = F.linear(src, sd[f"{pre}.to_v.0.weight"])
    hd = q.shape[-1] // heads
    q, k, v = (t.reshape(t.shape[0], t.shape[1], heads, hd) for t in (q, k, v))
    q = te_rmsnorm(q, sd[f"{pre}.to_q.1.weight"])
    k = te_rmsnorm(k, sd[f"{pre}.to_k.1.weight"])
    if ctx is None and rope is not None:
        q = te_rope_fused(q, rope)
        k = te_rope_fused(k, rope)
    return F.linear(attention_sbhd(q, k, v), sd[f"{pre}.to_out.0.weight"])


def _layernorm(x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)


def dit_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, timesteps: torch.Tensor, crossattn_emb: torch.Tensor,
                condition_video_input_mask: torch.Tensor, condition_video_pose: Optional[torch.Tensor],
                padding_mask: torch.Tensor, fps: Optional[torch.Tensor], *, num_blocks: int, num_heads: int,
                patch_spatial: int = 2, patch_temporal: int = 1, rope_ratios=(1.0, 1.0, 2.0),
                return_intermediates: bool = False):
    """VideoExtendGeneralDIT.forward for the GEN3C configuration (general_dit_video_conditioned.py:58-217,
    general_dit.py:272-358, 439-522; blocks.py:442-471, 537-558). `sd` uses the reference's parameter names without
    the `net.` prefix. rope_ratios = (h, w, t) extrapolation ratios."""
    B, C, T, H, W = x.shape
    parts = [x, condition_video_input_mask]
    if condition_video_pose is not None:
        parts.append(condition_video_pose)
    x = torch.cat(parts, dim=1)
    pm = F.interpolate(padding_mask.float(), size=(H, W), mode="nearest").to(x.dtype)  # torchvision NEAREST
    x = torch.cat([x, pm.unsqueeze(1).repeat(1, 1, T, 1, 1)], dim=1)

    ps, pt = patch_spatial, patch_temporal
    Tp, Hp, Wp = T // pt, H // ps, W // ps
    xp = x.reshape(B, -1, Tp, pt, Hp, ps, Wp, ps).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, Tp, Hp, Wp, -1)
    x_B_T_H_W_D = F.linear(xp, sd["x_embedder.proj.1.weight"])
    D = x_B_T_H_W_D.shape[-1]

    pos = abs_pos_emb(sd["extra_pos_embedder.pos_emb_t"], sd["extra_pos_embedder.pos_emb_h"],
                      sd["extra_pos_embedder.pos_emb_w"], B, Tp, Hp, Wp)
    rope = rope_freqs(sd["pos_embedder.seq"], D // num_heads, Tp, Hp, Wp, fps, *rope_ratios)

    t_sin = timestep_sinusoid(timesteps.flatten(), D)
    h1 = F.linear(t_sin, sd["t_embedder.1.linear_1.weight"])
    adaln_lora = F.linear(F.silu(h1), sd["t_embedder.1.linear_2.weight"])
    emb = te_rmsnorm(t_sin, sd["affline_norm.weight"])

    xs = x_B_T_H_W_D.permute(1, 2, 3, 0, 4)      # T H W B D
    pos = pos.permute(1, 2, 3, 0, 4)
    ctx = crossattn_emb.permute(1, 0, 2)          # M B D
    inter = {}

    def modulation(pre: str, lora: torch.Tensor):
        m = F.linear(F.linear(F.silu(emb), sd[f"{pre}.adaLN_modulation.1.weight"]), sd[f"{pre}.adaLN_modulation.2.weight"])
        return m + lora

    for i in range(num_blocks):
        pre = f"blocks.block{i}.blocks"
        xs = xs + pos
        for j in range(3):
            shift, scale, gate = modulation(f"{pre}.{j}", adaln_lora).chunk(3, dim=1)
            h = _layernorm(xs) * (1 + scale) + shift           # broadcasting over T,H,W: [B,D]
            hf = h.reshape(-1, B, D)
            if j == 0:
                y = _attn_module(sd, f"{pre}.0.block.attn", hf, None, num_heads, rope)
            elif j == 1:
                y = _attn_module(sd, f"{pre}.1.block.attn", hf, ctx, num_heads, None)
            else:
                y = F.linear(F.gelu(F.linear(hf, sd[f"{pre}.2.block.layer1.weight"])), sd[f"{pre}.2.block.layer2.weight"])
            xs = xs + gate * y.reshape(xs.shape)
        if return_intermediates:
            inter[f"block{i}"] = xs.clone()

    xb = xs.permute(3, 0, 1, 2, 4).reshape(B * Tp, Hp * Wp, D)
    m2 = F.linear(F.linear(F.silu(emb), sd["final_layer.adaLN_modulation.1.weight"]), sd["final_layer.adaLN_modulation.2.weight"])
    shift, scale = (m2 + adaln_lora[:, : 2 * D]).chunk(2, dim=1)
    shift = shift.repeat_interleave(Tp, dim=0)
    scale = scale.repeat_interleave(Tp, dim=0)
    y = _layernorm(xb) * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)
    y = F.linear(y, sd["final_layer.linear.weight"])
    Co = y.shape[-1] // (ps * ps * pt)
    y = y.reshape(B, Tp, Hp, Wp, ps, ps, pt, Co).permute(0, 7, 1, 6, 2, 4, 3, 5).reshape(B, Co, Tp * pt, Hp * ps, Wp * ps)
    if return_intermediates:
        return y, inter
    return y
