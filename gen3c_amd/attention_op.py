"""Operator-level attention seam (SURVEY.md 8b "Attention operator").

The reference's `Attention` module takes its core attention as a plug-in: `Attention(..., attn_op: BaseAttentionOp = None,
backend="transformer_engine")` (cosmos_predict1/diffusion/module/attention.py:172-242) and calls it as
`attn_op(q, k, v, core_attention_bias_type="no_bias", core_attention_bias=None)` on `sbhd` tensors, expecting `[S, B, H*d]` back
(attention.py:282-297); context parallelism reaches it through `attn_op.set_context_parallel_group(cp_group, cp_ranks, stream)`
(general_dit.py:536-541). `HipDotProductAttention` is that operator on the MI355X kernels: a maintainer constructs the reference's
module with `attn_op=HipDotProductAttention(heads, dim_head)` and keeps everything else (projections, TE norms, RoPE) as it is.

Nothing is computed in PyTorch: q / k are handed to g3_flash_attn_fwd_ex_bf16 as strided views, V goes through g3_transpose_v_bf16
(the kernels read V^T), and with a context-parallel group the call becomes ContextParallelAttention (all-gather of K / V^T over RCCL in
head groups instead of TE's P2P ring; same result as attention over the gathered sequence).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from .parallel import ContextParallelAttention


class HipDotProductAttention(torch.nn.Module):
    """Drop-in for transformer_engine.pytorch.DotProductAttention as the reference constructs it (attention.py:228-238):
    qkv_format "sbhd", attn_mask_type "no_mask", no dropout, softmax scale 1/sqrt(dim_head). head_dim must be 128 (the kernels')."""

    def __init__(self, num_attention_heads: int, kv_channels: int, num_gqa_groups: Optional[int] = None, attention_dropout: float = 0.0,
                 qkv_format: str = "sbhd", attn_mask_type: str = "no_mask", tp_size: int = 1, tp_group=None, sequence_parallel: bool = False,
                 softmax_scale: Optional[float] = None, **_unused):
        super().__init__()
        if kv_channels != 128:
            raise NotImplementedError(f"HipDotProductAttention: head dim {kv_channels} (the HIP attention kernels are built for 128)")
        if qkv_format != "sbhd" or attn_mask_type != "no_mask" or attention_dropout != 0.0:
            raise NotImplementedError("HipDotProductAttention: only qkv_format='sbhd', attn_mask_type='no_mask', dropout 0 (what GEN3C's DiT uses)")
        if num_gqa_groups not in (None, num_attention_heads) or tp_size != 1 or sequence_parallel:
            raise NotImplementedError("HipDotProductAttention: no grouped-query attention / tensor parallelism on this path (attention.py:205)")
        self.heads, self.dim_head, self.softmax_scale = num_attention_heads, kv_channels, softmax_scale
        self.cp_group = None
        self.cp_ranks: Optional[List[int]] = None
        self.cp_stream = None
        self._cp: Optional[ContextParallelAttention] = None

    def set_context_parallel_group(self, cp_group, cp_global_ranks=None, cp_stream=None, cp_comm_type: str = "all_gather"):
        """TE's hook (general_dit.py:541 passes (cp_group, cp_ranks, torch.cuda.Stream())). cp_group=None switches context parallelism off.
        The stream TE wants for its ring is kept as this operator's second launch stream (head groups alternate between it and the caller's)."""
        self.cp_group, self.cp_ranks, self.cp_stream = cp_group, (list(cp_global_ranks) if cp_global_ranks is not None else None), cp_stream
        self._cp = None
        if cp_group is not None:
            self._cp = ContextParallelAttention(cp_group)
            if cp_stream is not None:
                self._cp._side = cp_stream

    @torch.no_grad()
    def forward(self, query_layer: torch.Tensor, key_layer: torch.Tensor, value_layer: torch.Tensor, attention_mask=None,
                core_attention_bias_type: str = "no_bias", core_attention_bias=None, **_unused) -> torch.Tensor:
        """q [Sq, B, H, 128], k / v [Skv, B, H, 128] (bf16, last dim contiguous) -> [Sq, B, H*128]."""
        if core_attention_bias_type != "no_bias" or core_attention_bias is not None or attention_mask is not None:
            raise NotImplementedError("HipDotProductAttention: no bias / mask (the reference calls it with 'no_bias', None: attention.py:288)")
        Sq, B, H, d = query_layer.shape
        Skv = key_layer.shape[0]
        assert (H, d) == (self.heads, self.dim_head) and key_layer.shape[1:] == (B, H, d) and value_layer.shape == key_layer.shape
        as_rows = lambda t, S: t.reshape(S * B, H * d) if t.is_contiguous() else t.contiguous().reshape(S * B, H * d)  # rows (s, b), b fastest
        q, k, v = as_rows(query_layer.to(torch.bfloat16), Sq), as_rows(key_layer.to(torch.bfloat16), Skv), as_rows(value_layer.to(torch.bfloat16), Skv)
        if self._cp is not None:
            assert Sq == Skv, "context parallelism shards self-attention (cross-attention K / V are replicated: general_dit.py:536-539)"
            out = self._cp(q, k, v, Sq, B, H)
        else:
            out = ops.flash_attn(q, k, ops.transpose_v(v, Skv, B, H), Sq, Skv, B, H, softmax_scale=self.softmax_scale)
        return out.reshape(Sq, B, H * d)
