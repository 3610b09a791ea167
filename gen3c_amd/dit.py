"""MI355X-native stand-in for the reference's `VideoExtendGeneralDIT` (the GEN3C-Cosmos-7B denoiser network).

Boundary (SURVEY.md 8b "Net plugin"): same constructor keywords, same `forward(x, timesteps, crossattn_emb, ...,
condition_video_pose, **kwargs)` signature and return shape, same `enable_context_parallel / disable_context_parallel /
is_context_parallel_enabled / cp_group` surface and the SAME state-dict keys as
cosmos_predict1/diffusion/networks/general_dit_video_conditioned.py:29-217 + general_dit.py:41-569, so the
`net.*` entries of checkpoints/Gen3C-Cosmos-7B/model.pt load unchanged (TE `*_extra_state` keys are ignored exactly
like inference_utils.py:240-242 does).

Nothing here computes the network in PyTorch: forward() drives the HIP kernels of libgen3c_hip.so through
gen3c_amd.ops (GEMM/attention/norm/embedding kernels) on torch-allocated HBM buffers. What torch does do is plumbing:
allocation, context-parallel slicing of the inputs, and building the input-independent tables (RoPE cos/sin, normalised
absolute position embedding) once per (shape, fps).
"""
from __future__ import annotations

import math
from enum import Enum
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import ops
from .parallel import ContextParallelAttention, split_inputs_cp

# 1: run the per-head RMSNorm + RoPE and the V transpose in the QKV projection's epilogue (g3_gemm_qk_norm_rope_bf16) instead of as separate
# HBM-bound passes behind a plain GEMM. Measured in one process at the benchmark shape (tools/qkv_probe.py, profiles/r3_qkv_probe.txt):
# B = 2: plain GEMM 8.23 ms; GEMM + 2 norm passes + transpose 9.21 ms; fused 11.85 ms - at one wave per SIMD nothing overlaps the
# epilogue's VALU work and table reads, so the separate passes (0.98 ms at the HBM rate) are the default since round 3 (+2 % on the step).
_FUSE_QKV_EPILOGUE = __import__("os").environ.get("G3_FUSE_QKV_EPILOGUE", "0") != "0"


# 1 (default, round 6): the V third of the self-attention QKV projection is computed with the operands SWAPPED - V^T[b] = W_v . h[:, b]^T, one GEMM per batch item whose
# "token" operand is the weight and whose output rows are the 128 H value features - so it lands directly in the V^T [B, H, 128, S] layout the attention kernel reads and
# the separate transpose pass (g3_transpose_v_bf16: 10.7 ms of a 3.3 s step) disappears. Same products, same K order per element: bitwise equal to the
# transpose of the fused projection's v columns (tests/test_kernels_gpu.py). 0: fused [S*B, 3D] projection + transpose (A/B).
_V_OPERAND_SWAP = __import__("os").environ.get("G3_V_OPERAND_SWAP", "1") != "0"


# 1 (default, round 6): the cross-attention's Q goes from the projection GEMM straight into the attention kernel, which applies to_q[1]'s per-head RMSNorm while
# it loads the rows (g3_cross_attn_fwd_bf16) - the separate norm pass (one read + one write of the [S*B, 4096] Q per block, ~10 ms of a 3.3 s step) is gone.
# Same rounding points as the separate pass; the 128 squares are summed in another order (tests/test_kernels_gpu.py). 0: separate pass (A/B).
_CROSS_Q_NORM_IN_ATTENTION = __import__("os").environ.get("G3_CROSS_Q_NORM_IN_ATTENTION", "1") != "0"


def _project_norm_rope(h, w, n_q, n_k, norm_q, norm_k, cos, sin, S, B, nH_total):
    """a @ w^T with per-head RMSNorm (+ RoPE) on the first n_q (weight norm_q) and the next n_k (norm_k) output features; the rest plain.
    Same rounding points either way (tested): the fused GEMM epilogue, or the plain GEMM followed by the in-place norm passes."""
    if _FUSE_QKV_EPILOGUE:
        return ops.gemm_qk_norm_rope(h, w, n_q, n_k, norm_q, norm_k, cos, sin, S, B)
    y = ops.gemm_nt(h, w)
    if n_q and n_k:
        ops.qk_rmsnorm_rope_pair(y[:, :n_q + n_k], norm_q, n_q // 128, norm_k, n_k // 128, cos, sin, S, B)
    elif n_q:
        ops.qk_rmsnorm_rope(y[:, :n_q], norm_q, cos, sin, S, B, n_q // 128, out=y[:, :n_q])
    elif n_k:
        ops.qk_rmsnorm_rope(y[:, n_q:n_q + n_k], norm_k, cos, sin, S, B, n_k // 128, out=y[:, n_q:n_q + n_k])
    return y


class DataType(Enum):
    """Mirror of cosmos_predict1/diffusion/conditioner.py DataType (IMAGE/VIDEO)."""
    IMAGE = "image"
    VIDEO = "video"


def tensor_version(t: torch.Tensor) -> Optional[int]:
    """In-place version counter of `t` for cache keys - or None: do not cache on this tensor. Inference tensors (created under
    torch.inference_mode(), which the reference's pipeline entry points use: world_generation_pipeline.py:1225) track no version
    (reading `_version` raises), yet inside inference mode they CAN be edited in place (crossattn_emb.copy_(new_prompt)): a key of
    storage address + identity alone would serve stale results, so callers skip their cache for them (a cross-attention K / V rebuild
    is 28 small GEMMs, a condition re-concatenation ~30 MB - far below 1 % of a forward)."""
    return None if t.is_inference() else t._version


def cacheable(*tensors) -> bool:
    """True when every tensor has a version counter (see tensor_version)."""
    return all(not (isinstance(t, torch.Tensor) and t.is_inference()) for t in tensors)


def _is_video(data_type) -> bool:
    # accept our enum, the reference's enum, or a plain string
    v = getattr(data_type, "value", data_type)
    return str(v).lower().endswith("video")


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's parameter names (e.g. `to_q.0.weight`)."""


def _register(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool = False):
    *path, leaf = dotted.split(".")
    mod = root
    for p in path:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(leaf, tensor, persistent=True)
    else:
        mod.register_parameter(leaf, nn.Parameter(tensor, requires_grad=False))


def _xavier_uniform_(t: torch.Tensor):
    fan_out, fan_in = t.shape
    a = math.sqrt(6.0 / (fan_in + fan_out))
    return t.uniform_(-a, a)


class VideoExtendGeneralDIT(nn.Module):
    def __init__(
        self,
        max_img_h: int = 240,
        max_img_w: int = 240,
        max_frames: int = 128,
        in_channels: int = 16 + 1,
        out_channels: int = 16,
        patch_spatial: int = 2,
        patch_temporal: int = 1,
        concat_padding_mask: bool = True,
        block_config: str = "FA-CA-MLP",
        model_channels: int = 4096,
        num_blocks: int = 28,
        num_heads: int = 32,
        mlp_ratio: float = 4.0,
        block_x_format: str = "THWBD",
        crossattn_emb_channels: int = 1024,
        use_cross_attn_mask: bool = False,
        pos_emb_cls: str = "rope3d",
        pos_emb_learnable: bool = False,
        pos_emb_interpolation: str = "crop",
        affline_emb_norm: bool = True,
        use_adaln_lora: bool = True,
        adaln_lora_dim: int = 256,
        rope_h_extrapolation_ratio: float = 1.0,
        rope_w_extrapolation_ratio: float = 1.0,
        rope_t_extrapolation_ratio: float = 1.0,
        extra_per_block_abs_pos_emb: bool = True,
        extra_per_block_abs_pos_emb_type: str = "learnable",
        extra_h_extrapolation_ratio: float = 1.0,
        extra_w_extrapolation_ratio: float = 1.0,
        extra_t_extrapolation_ratio: float = 1.0,
        add_augment_sigma_embedding: bool = False,
        device: Optional[torch.device | str] = None,
        dtype: torch.dtype = torch.bfloat16,
        init_weights: bool = True,
    ) -> None:
        super().__init__()
        # the GEN3C-Cosmos-7B configuration space (config/base/net.py:23-43 + cosmos-1-diffusion-gen3c.py:38-43);
        # anything else the reference class supports but GEN3C never instantiates is refused loudly.
        if block_config.upper() != "FA-CA-MLP":
            raise NotImplementedError(f"block_config {block_config!r}: only 'FA-CA-MLP' (GEN3C-Cosmos-7B) is built")
        if block_x_format != "THWBD" or pos_emb_cls != "rope3d" or not use_adaln_lora or not affline_emb_norm:
            raise NotImplementedError("only the faditv2 configuration (THWBD, rope3d, AdaLN-LoRA, affine emb norm) is built")
        if not extra_per_block_abs_pos_emb or extra_per_block_abs_pos_emb_type.lower() != "learnable":
            raise NotImplementedError("extra_per_block_abs_pos_emb must be the learnable per-axis embedding")
        if use_cross_attn_mask or add_augment_sigma_embedding or pos_emb_learnable:
            raise NotImplementedError("use_cross_attn_mask / add_augment_sigma_embedding / learnable rope are not used by GEN3C")
        if model_channels // num_heads != 128:
            raise NotImplementedError("head_dim must be 128 (HIP attention kernel)")
        if dtype != torch.bfloat16:
            raise NotImplementedError("the HIP path computes in bf16 (reference precision='bfloat16', config/base/model.py:29)")

        self.max_img_h, self.max_img_w, self.max_frames = max_img_h, max_img_w, max_frames
        self.in_channels, self.out_channels = in_channels, out_channels
        self.patch_spatial, self.patch_temporal = patch_spatial, patch_temporal
        self.concat_padding_mask = concat_padding_mask
        self.model_channels, self.num_blocks, self.num_heads = model_channels, num_blocks, num_heads
        self.mlp_hidden = int(model_channels * mlp_ratio)
        self.crossattn_emb_channels = crossattn_emb_channels
        self.adaln_lora_dim = adaln_lora_dim
        self.block_x_format = block_x_format
        self.head_dim = 128
        self.base_fps = 24
        self.cp_group = None
        self.cp_size = None
        self._cp_attn: Optional[ContextParallelAttention] = None
        # True (default): the cross-attention runs its tile loop over the non-zero context tokens only and takes the zero-padded tail in closed form
        # (_cross_attention_kv); False: every context token goes through the loop (bench.py's timed region: the dense workload of BASELINE.json)
        self.cross_attention_skip_zero_context = True
        self._tables: Dict[tuple, tuple] = {}
        self._packed = None
        self._tune_blocks: Optional[int] = None  # bench.py's context-parallel autotune: run only the first n blocks (not a model option)

        D, Hd = model_channels, self.head_dim
        kw = dict(device=device, dtype=dtype)
        E = lambda *shape: torch.empty(*shape, **kw)
        emb_in = (in_channels + (1 if concat_padding_mask else 0)) * patch_spatial * patch_spatial * patch_temporal
        self.patch_dim = emb_in

        # ---- parameters, under the reference's names
        _register(self, "x_embedder.proj.1.weight", E(D, emb_in))
        len_h, len_w, len_t = max_img_h // patch_spatial, max_img_w // patch_spatial, max_frames // patch_temporal
        _register(self, "pos_embedder.seq", torch.arange(max(len_h, len_w, len_t), dtype=torch.float, device=device), buffer=True)
        _register(self, "extra_pos_embedder.pos_emb_h", E(len_h, D))
        _register(self, "extra_pos_embedder.pos_emb_w", E(len_w, D))
        _register(self, "extra_pos_embedder.pos_emb_t", E(len_t, D))
        _register(self, "t_embedder.1.linear_1.weight", E(D, D))
        _register(self, "t_embedder.1.linear_2.weight", E(3 * D, D))
        for i in range(num_blocks):
            pre = f"blocks.block{i}.blocks"
            for j, ctx in ((0, D), (1, crossattn_emb_channels)):  # 0 = FA (self), 1 = CA (cross)
                a = f"{pre}.{j}.block.attn"
                _register(self, f"{a}.to_q.0.weight", E(D, D))
                _register(self, f"{a}.to_q.1.weight", E(Hd))
                _register(self, f"{a}.to_k.0.weight", E(D, ctx))
                _register(self, f"{a}.to_k.1.weight", E(Hd))
                _register(self, f"{a}.to_v.0.weight", E(D, ctx))
                _register(self, f"{a}.to_out.0.weight", E(D, D))
            _register(self, f"{pre}.2.block.layer1.weight", E(self.mlp_hidden, D))
            _register(self, f"{pre}.2.block.layer2.weight", E(D, self.mlp_hidden))
            for j in range(3):
                _register(self, f"{pre}.{j}.adaLN_modulation.1.weight", E(adaln_lora_dim, D))
                _register(self, f"{pre}.{j}.adaLN_modulation.2.weight", E(3 * D, adaln_lora_dim))
        _register(self, "final_layer.linear.weight", E(patch_spatial * patch_spatial * patch_temporal * out_channels, D))
        _register(self, "final_layer.adaLN_modulation.1.weight", E(adaln_lora_dim, D))
        _register(self, "final_layer.adaLN_modulation.2.weight", E(2 * D, adaln_lora_dim))
        _register(self, "affline_norm.weight", E(D))

        # RoPE dimension split (position_embedding.py:106-124)
        dim_h = Hd // 6 * 2
        dim_t = Hd - 2 * dim_h
        self._rope_dims = (dim_t, dim_h, dim_h)
        self.h_ntk_factor = rope_h_extrapolation_ratio ** (dim_h / (dim_h - 2))
        self.w_ntk_factor = rope_w_extrapolation_ratio ** (dim_h / (dim_h - 2))
        self.t_ntk_factor = rope_t_extrapolation_ratio ** (dim_t / (dim_t - 2))

        if init_weights:
            self.initialize_weights()

    # ------------------------------------------------------------------------------------------------ init / load
    @torch.no_grad()
    def initialize_weights(self, randomize_adaln: bool = False, seed: Optional[int] = None):
        """Same distributions as GeneralDIT.initialize_weights (general_dit.py:180-203) and LearnablePosEmbAxis
        (position_embedding.py:210-216). `randomize_adaln=True` replaces the zero-init of the last AdaLN layers by
        N(0, 0.02^2) so that gates are non-zero on random weights - used by tests and bench only."""
        gen = None
        if seed is not None:
            gen = torch.Generator(device=self.affline_norm.weight.device)
            gen.manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith(".to_q.1.weight") or name.endswith(".to_k.1.weight") or name == "affline_norm.weight":
                p.fill_(1.0)
            elif name.startswith("extra_pos_embedder."):
                tmp = torch.empty(p.shape, dtype=torch.float32, device=p.device)
                nn.init.trunc_normal_(tmp, std=0.02, a=-2.0, b=2.0, generator=gen)  # timm trunc_normal_: cut at +-2 (absolute)
                p.copy_(tmp)
            elif name.startswith("t_embedder."):
                p.normal_(0.0, 0.02, generator=gen)
            elif name.endswith("adaLN_modulation.2.weight") and not name.startswith("final_layer."):
                if randomize_adaln:
                    p.normal_(0.0, 0.02, generator=gen)
                else:
                    p.zero_()
            else:  # every other nn.Linear: xavier uniform
                fan_out, fan_in = p.shape
                a = math.sqrt(6.0 / (fan_in + fan_out))
                p.uniform_(-a, a, generator=gen)
        self._packed = None
        self._tables.clear()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        # TE modules carry `_extra_state` blobs in the reference checkpoint; drop them like non_strict_load_model does
        # (cosmos_predict1/diffusion/inference/inference_utils.py:240-242).
        sd = {k: v for k, v in state_dict.items() if not k.endswith("_extra_state")}
        out = super().load_state_dict(sd, strict=strict, assign=assign)
        self._packed = None
        self._tables.clear()
        return out

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .bfloat16() move or re-create the parameters: the fused weight copies and the cached tables would be
        # stale (or on the old device)
        out = super()._apply(fn, *args, **kwargs)
        self._packed = None
        self._tables.clear()
        return out

    def _weights_key(self) -> tuple:
        """Identity of the current weight set: (storage address, in-place version counter) of every parameter. Any in-place
        update (p.copy_, weight swapping) or re-assignment (load_state_dict(assign=True) on a sub-module) changes it."""
        return tuple((p.data_ptr(), tensor_version(p)) for p in self.parameters())

    # ------------------------------------------------------------------------------------------------ weight packing
    def _pack(self):
        """Fuse per-layer projection weights that share an input into one GEMM operand (done once per weight set; rebuilt when
        a parameter was replaced or modified in place since)."""
        key = self._weights_key()
        # Inference tensors (a model built or loaded under torch.inference_mode()) carry no version counter - their key is the storage address
        # alone, and an in-place p.copy_(new) inside inference mode would leave the fused QKV / K-V copies stale. Such a weight set is re-fused on
        # every call (two concatenations per block: ~2 ms of a 1.7 s forward) instead of trusting the address.
        if self._packed is not None and self._packed["key"] == key and self._packed["versioned"]:
            return self._packed
        if self._packed is not None and self._packed["key"] != key:
            self._tables.clear()  # the position tables derive from the pos-emb parameters (an unversioned re-fuse of the SAME storages keeps them)
        P = dict(self.named_parameters())
        blocks = []
        for i in range(self.num_blocks):
            pre = f"blocks.block{i}.blocks"
            fa, ca, mlp = f"{pre}.0", f"{pre}.1", f"{pre}.2"
            blocks.append(dict(
                fa_qkv=torch.cat([P[f"{fa}.block.attn.to_q.0.weight"], P[f"{fa}.block.attn.to_k.0.weight"],
                                  P[f"{fa}.block.attn.to_v.0.weight"]], dim=0).contiguous(),
                fa_qn=P[f"{fa}.block.attn.to_q.1.weight"], fa_kn=P[f"{fa}.block.attn.to_k.1.weight"],
                fa_out=P[f"{fa}.block.attn.to_out.0.weight"],
                ca_q=P[f"{ca}.block.attn.to_q.0.weight"],
                ca_kv=torch.cat([P[f"{ca}.block.attn.to_k.0.weight"], P[f"{ca}.block.attn.to_v.0.weight"]], dim=0).contiguous(),
                ca_qn=P[f"{ca}.block.attn.to_q.1.weight"], ca_kn=P[f"{ca}.block.attn.to_k.1.weight"],
                ca_out=P[f"{ca}.block.attn.to_out.0.weight"],
                w1=P[f"{mlp}.block.layer1.weight"], w2=P[f"{mlp}.block.layer2.weight"],
                ada=[(P[f"{pre}.{j}.adaLN_modulation.1.weight"], P[f"{pre}.{j}.adaLN_modulation.2.weight"]) for j in range(3)],
            ))
        self._packed = dict(blocks=blocks, P=P, key=key, versioned=cacheable(*P.values()))
        return self._packed

    # ------------------------------------------------------------------------------------------------ context parallel
    def enable_context_parallel(self, cp_group):
        """general_dit.py:524-543. Self-attention K/V are exchanged by RCCL all-gather (see parallel.py)."""
        import torch.distributed as dist
        self.cp_group = cp_group
        self.cp_size = dist.get_world_size(cp_group)
        # default schedule of the DiT: local_first, 4 head groups - on one GPU playing one rank (tools/cp_rank_emulate.py, profiles/r6_cp_rank_shapes.txt) it is the
        # fastest or within 2 % of the fastest at cp = 2 / 4 / 8, it never waits for the first exchange, and with it the QKV projection stays ONE launch
        # (forward()). The class default stays gather_first (bitwise equal to the single-rank attention).
        self._cp_attn = ContextParallelAttention(cp_group, head_groups=4, schedule="local_first")
        # G3_CP_CONFIG="<head groups>,<auto|w4b|wave8>,<gather_first|local_first>": the configuration `python bench.py --gpus N` measured fastest on
        # this node (its `cp.chosen`), for the entry points that do not tune themselves (gen3c_single_image.py --num_gpus N, ...)
        cfg = __import__("os").environ.get("G3_CP_CONFIG")
        if cfg:
            g_, kern_, sched_ = (c.strip() for c in cfg.split(","))
            self._cp_attn.configure(head_groups=int(g_), kernel=kern_, schedule=sched_)
        self._tables.clear()

    def disable_context_parallel(self):
        self.cp_group = None
        self.cp_size = None
        self._cp_attn = None
        self._tables.clear()

    @property
    def is_context_parallel_enabled(self) -> bool:
        return self.cp_group is not None

    # ------------------------------------------------------------------------------------------------ tables
    @torch.no_grad()
    def _position_tables(self, B: int, T: int, Hp: int, Wp: int, fps: Optional[torch.Tensor], device) -> tuple:
        """RoPE cos/sin [S,128] f32 (position_embedding.py:126-187) and the unit-RMS absolute position embedding
        [S*B, D] bf16 (position_embedding.py:218-233 + attention.py:108-124). Input-independent: cached per
        (B, T, H, W, fps, cp). With CP the tables are generated for the GLOBAL T and this rank's frames are sliced
        out (position_embedding.py:66-79), so positions stay absolute."""
        fps_val = None if fps is None else float(fps.flatten()[0])
        cp = self.cp_size or 1
        rank = 0
        if self.cp_group is not None:
            import torch.distributed as dist
            rank = dist.get_rank(self.cp_group)
        key = (B, T, Hp, Wp, fps_val, cp, rank, str(device))
        if key in self._tables:
            return self._tables[key]
        Tg = T * cp
        dim_t, dim_h, dim_w = self._rope_dims
        seq = self.pos_embedder.seq.to(device=device, dtype=torch.float32)
        rng_s = torch.arange(0, dim_h, 2, device=device)[: dim_h // 2].float() / dim_h
        rng_t = torch.arange(0, dim_t, 2, device=device)[: dim_t // 2].float() / dim_t
        h_freqs = 1.0 / ((10000.0 * self.h_ntk_factor) ** rng_s)
        w_freqs = 1.0 / ((10000.0 * self.w_ntk_factor) ** rng_s)
        t_freqs = 1.0 / ((10000.0 * self.t_ntk_factor) ** rng_t)
        assert Hp <= self.max_img_h // self.patch_spatial and Wp <= self.max_img_w // self.patch_spatial
        half_h = torch.outer(seq[:Hp], h_freqs)
        half_w = torch.outer(seq[:Wp], w_freqs)
        if fps_val is None:
            assert Tg == 1, "T should be 1 for image batch."
            half_t = torch.outer(seq[:Tg], t_freqs)
        else:
            half_t = torch.outer(seq[:Tg] / fps_val * self.base_fps, t_freqs)
        half = torch.cat([
            half_t[:, None, None, :].expand(Tg, Hp, Wp, -1),
            half_h[None, :, None, :].expand(Tg, Hp, Wp, -1),
            half_w[None, None, :, :].expand(Tg, Hp, Wp, -1),
        ], dim=-1)
        freqs = torch.cat([half, half], dim=-1)  # [Tg,Hp,Wp,128]
        freqs = freqs[rank * T:(rank + 1) * T].reshape(T * Hp * Wp, 128).float()
        cos, sin = torch.cos(freqs).contiguous(), torch.sin(freqs).contiguous()

        # absolute position embedding: only the per-token normaliser is precomputed (bf16, [S]); the embedding itself is rebuilt from
        # the three per-axis tables inside the fused kernel (g3_posemb_layernorm_modulate_bf16), with the same bf16 rounding points as here
        pe_t = self.extra_pos_embedder.pos_emb_t[:Tg][rank * T:(rank + 1) * T].contiguous()
        pe_h = self.extra_pos_embedder.pos_emb_h[:Hp].contiguous()
        pe_w = self.extra_pos_embedder.pos_emb_w[:Wp].contiguous()
        emb = (pe_t[:, None, None, :] + pe_h[None, :, None, :]) + pe_w[None, None, :, :]  # bf16 adds, reference order
        norm = torch.linalg.vector_norm(emb, dim=-1, keepdim=True, dtype=torch.float32)
        norm = torch.add(1e-6, norm, alpha=math.sqrt(1.0 / emb.shape[-1]))
        pos = dict(pe_t=pe_t, pe_h=pe_h, pe_w=pe_w, norm=norm.to(emb.dtype).reshape(T * Hp * Wp).contiguous())
        del emb
        self._tables[key] = (cos, sin, pos)
        return self._tables[key]

    def position_embedding_rows(self, B: int, T: int, Hp: int, Wp: int, fps, device) -> torch.Tensor:
        """The materialised [S*B, D] per-block absolute position embedding (tests / debugging; the forward pass never builds it)."""
        pos = self._position_tables(B, T, Hp, Wp, fps, device)[2]
        emb = (pos["pe_t"][:, None, None, :] + pos["pe_h"][None, :, None, :]) + pos["pe_w"][None, None, :, :]
        emb = emb / pos["norm"].reshape(T, Hp, Wp, 1)
        D = emb.shape[-1]
        return emb.reshape(T * Hp * Wp, 1, D).expand(-1, B, -1).reshape(T * Hp * Wp * B, D).contiguous()

    # ------------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(
        self,
        x: torch.Tensor,
        timesteps: torch.Tensor,
        crossattn_emb: torch.Tensor,
        crossattn_mask: Optional[torch.Tensor] = None,
        fps: Optional[torch.Tensor] = None,
        image_size: Optional[torch.Tensor] = None,
        padding_mask: Optional[torch.Tensor] = None,
        scalar_feature: Optional[torch.Tensor] = None,
        data_type=DataType.VIDEO,
        video_cond_bool: Optional[torch.Tensor] = None,
        condition_video_indicator: Optional[torch.Tensor] = None,
        condition_video_input_mask: Optional[torch.Tensor] = None,
        condition_video_augment_sigma: Optional[torch.Tensor] = None,
        condition_video_pose: Optional[torch.Tensor] = None,
        **kwargs,
    ) -> torch.Tensor:
        """x: [B, 16, T_local, H, W] bf16 -> [B, 16, T_local, H, W] bf16 (general_dit_video_conditioned.py:58-132)."""
        if scalar_feature is not None:
            raise NotImplementedError("Scalar feature is not implemented yet.")  # same as the reference
        B, C, T, H, W = x.shape
        dev = x.device
        sources = [(x.to(torch.bfloat16).contiguous(), True)]
        if _is_video(data_type):
            assert condition_video_input_mask is not None, "condition_video_input_mask is required for video data type"
            if self.cp_group is not None:
                condition_video_input_mask = split_inputs_cp(condition_video_input_mask, 2, self.cp_group)
                if condition_video_indicator is not None:
                    condition_video_indicator = split_inputs_cp(condition_video_indicator, 2, self.cp_group)
                if condition_video_pose is not None:
                    condition_video_pose = split_inputs_cp(condition_video_pose, 2, self.cp_group)
            sources.append((condition_video_input_mask.to(torch.bfloat16).contiguous(), True))
            if condition_video_pose is not None:
                sources.append((condition_video_pose.to(torch.bfloat16).contiguous(), True))
        if self.concat_padding_mask:
            pm = _nearest_resize(padding_mask, (H, W)).to(torch.bfloat16)  # torchvision NEAREST resize (general_dit.py:305-307)
            sources.append((pm.reshape(B, 1, H, W).contiguous(), False))  # broadcast over T by the gather kernel
        c_in = sum(t.shape[1] for t, _ in sources)
        assert c_in * self.patch_spatial ** 2 * self.patch_temporal == self.patch_dim, f"channel mismatch: got {c_in} input channels"

        ps, pt = self.patch_spatial, self.patch_temporal
        Tp, Hp, Wp = T // pt, H // ps, W // ps
        S = Tp * Hp * Wp
        D = self.model_channels
        # channel concat (general_dit_video_conditioned.py:77-101) + patch gather "b c (t r) (h m) (w n) -> (t h w) b (c r m n)"
        # (blocks.py:154-159 + THWBD order) in one HIP kernel
        patches = ops.dit_patchify(sources, B, T, H, W, pt, ps)

        pk = self._pack()
        P = pk["P"]
        cos, sin, pos = self._position_tables(B, Tp, Hp, Wp, fps, dev)

        xs = ops.gemm_nt(patches, P["x_embedder.proj.1.weight"])  # [S*B, D]

        # ---- timestep embedding (blocks.py:38-80) + affine RMSNorm (general_dit.py:173-177)
        ts = timesteps.flatten().to(torch.float32)
        if ts.shape[0] != B:
            ts = ts.expand(B)
        t_sin, emb = ops.timestep_embedding(ts.contiguous(), P["affline_norm.weight"], D)
        h1 = ops.gemv(t_sin, P["t_embedder.1.linear_1.weight"])
        adaln_lora = ops.gemv(h1, P["t_embedder.1.linear_2.weight"], act_in=1)  # [B, 3D]

        # ---- context
        M = crossattn_emb.shape[1]
        nH = self.num_heads
        ca_kv, ca_dense = self._cross_attention_kv(pk, crossattn_emb)
        if not self.cross_attention_skip_zero_context:
            ca_dense = 0
        for bi, blk in enumerate(pk["blocks"][: self._tune_blocks] if self._tune_blocks else pk["blocks"]):
            # -- self attention; "x = x + extra_per_block_pos_emb" (blocks.py:547-548) rides in the same pass over x as the LayerNorm
            shift, scale, gate = self._modulation(emb, blk["ada"][0], adaln_lora, 3)
            if "full" not in pos:  # the finished embedding [S, D], built once per shape with the reference's bf16 rounding points (bf16 tensor ops)
                pe_sum = (pos["pe_t"][:, None, None, :] + pos["pe_h"][None, :, None, :]) + pos["pe_w"][None, None, :, :]
                pos["full"] = (pe_sum / pos["norm"].reshape(Tp, Hp, Wp, 1)).reshape(S, D).contiguous()
                del pe_sum
            h = ops.posemb_layernorm_modulate(xs, pos["full"], None, None, None, Tp, Hp, Wp, B, shift, scale)
            if self._cp_attn is not None and self._cp_attn.schedule == "local_first":
                # local_first starts every head group on this rank's OWN K / V shard, so nothing waits for the exchange at first: one fused QKV
                # projection + one norm / RoPE pass over q | k, then the exchange goes out under the local attention. At the cp = 8 shape
                # (M = 14 080) a separate N = 4096 Q projection is 3.44 rounds of 256 x 256 tiles on 256 workgroups and ran at 77 % of its cp = 1
                # rate (profiles/r6_cp_rank_shapes.txt); inside the N = 12 288 projection the same tiles are part of 10.3 rounds.
                qkv = _project_norm_rope(h, blk["fa_qkv"], D, D, blk["fa_qn"], blk["fa_kn"], cos, sin, S, B, nH)
                pending = self._cp_attn.start(qkv[:, D:2 * D], qkv[:, 2 * D:], S, B, nH)
                o = self._cp_attn.finish(qkv[:, :D], pending)
            elif self._cp_attn is not None:
                # K / V first, so their exchange is in flight while Q is still being projected (same fused weight, sliced;
                # every output element sees the same K order, so this is bit-identical to the single fused GEMM)
                # the per-head RMSNorm + RoPE of q and k (attention.py:262-280) run in the projections' epilogues
                kv = _project_norm_rope(h, blk["fa_qkv"][D:], 0, D, None, blk["fa_kn"], cos, sin, S, B, nH)  # [S*B, 2D]: k normalised + rotated, v plain
                pending = self._cp_attn.start(kv[:, :D], kv[:, D:], S, B, nH)
                q = _project_norm_rope(h, blk["fa_qkv"][:D], D, 0, blk["fa_qn"], None, cos, sin, S, B, nH)
                o = self._cp_attn.finish(q, pending)
            else:
                if _FUSE_QKV_EPILOGUE:
                    # [S*B, 3D]: q and k normalised + rotated; the v heads go straight into V^T (their columns of qkv stay unwritten)
                    vt = self._vt_buffer(S, B, nH, dev)
                    qkv = ops.gemm_qk_norm_rope(h, blk["fa_qkv"], D, D, blk["fa_qn"], blk["fa_kn"], cos, sin, S, B, vt=vt)
                    q, k = qkv[:, :D], qkv[:, D:2 * D]
                elif _V_OPERAND_SWAP:
                    # q | k: plain GEMM, then normalised + rotated IN PLACE by one pass; v: projected straight into V^T (operands swapped, see _V_OPERAND_SWAP)
                    qk = ops.gemm_nt(h, blk["fa_qkv"][:2 * D])
                    ops.qk_rmsnorm_rope_pair(qk, blk["fa_qn"], nH, blk["fa_kn"], nH, cos, sin, S, B)
                    q, k = qk[:, :D], qk[:, D:]
                    vt = self._vt_buffer(S, B, nH, dev)
                    hv = h.view(S, B, D)
                    for b_ in range(B):
                        ops.gemm_nt(blk["fa_qkv"][2 * D:], hv[:, b_], out=vt[b_].view(D, -1)[:, :S])
                else:  # plain GEMM, then q / k normalised + rotated IN PLACE in the fused buffer and v transposed
                    qkv = ops.gemm_nt(h, blk["fa_qkv"])
                    ops.qk_rmsnorm_rope_pair(qkv[:, :2 * D], blk["fa_qn"], nH, blk["fa_kn"], nH, cos, sin, S, B)  # one pass over q | k
                    q, k = qkv[:, :D], qkv[:, D:2 * D]
                    vt = ops.transpose_v(qkv[:, 2 * D:], S, B, nH, out=self._vt_buffer(S, B, nH, dev))
                o = ops.flash_attn(q, k, vt, S, S, B, nH)
            ops.gemm_nt(o, blk["fa_out"], out=xs, epilogue=ops.EPI_GATED_RESIDUAL, gate=gate, residual=xs)
            # -- cross attention (unmasked over all M context tokens, general_dit.py:407-410)
            shift, scale, gate = self._modulation(emb, blk["ada"][1], adaln_lora, 3)
            h = ops.layernorm_modulate(xs, shift, scale)
            k, vt = ca_kv[bi]
            if _CROSS_Q_NORM_IN_ATTENTION:  # plain projection; to_q[1]'s per-head RMSNorm runs in the attention kernel's Q load (no RoPE in cross-attention)
                o = ops.flash_attn(ops.gemm_nt(h, blk["ca_q"]), k, vt, S, M, B, nH, kv_dense=ca_dense, q_norm_weight=blk["ca_qn"])
            else:
                q = _project_norm_rope(h, blk["ca_q"], D, 0, blk["ca_qn"], None, None, None, S, B, nH)
                o = ops.flash_attn(q, k, vt, S, M, B, nH, kv_dense=ca_dense)
            ops.gemm_nt(o, blk["ca_out"], out=xs, epilogue=ops.EPI_GATED_RESIDUAL, gate=gate, residual=xs)
            # -- MLP
            shift, scale, gate = self._modulation(emb, blk["ada"][2], adaln_lora, 3)
            h = ops.layernorm_modulate(xs, shift, scale)
            u = ops.gemm_nt(h, blk["w1"], epilogue=ops.EPI_GELU)
            ops.gemm_nt(u, blk["w2"], out=xs, epilogue=ops.EPI_GATED_RESIDUAL, gate=gate, residual=xs)

        # ---- final layer (blocks.py:222-242) + unpatchify (general_dit.py:348-357)
        fl = (P["final_layer.adaLN_modulation.1.weight"], P["final_layer.adaLN_modulation.2.weight"])
        shift, scale = self._modulation(emb, fl, adaln_lora[:, : 2 * D], 2)
        h = ops.layernorm_modulate(xs, shift, scale)
        y = ops.gemm_nt(h, P["final_layer.linear.weight"])  # [S*B, p1*p2*t*C]
        return ops.dit_unpatchify(y, B, self.out_channels, T, H, W, pt, ps)

    def _vt_buffer(self, S: int, B: int, H: int, dev) -> torch.Tensor:
        """V^T [B, H, 128, ceil64(S)] shared by all blocks of all forwards of this shape (each block overwrites positions [0, S); the zero
        tail the attention kernel relies on is written once, here)."""
        key = (S, B, H, str(dev))
        buf = getattr(self, "_vt_cache", None)
        if buf is None or buf[0] != key:
            buf = self._vt_cache = (key, torch.zeros((B, H, 128, ops.ceil_to(S, 64)), dtype=torch.bfloat16, device=dev))
        return buf[1]

    def _cross_attention_kv(self, pk, crossattn_emb: torch.Tensor):
        """Per block: K = RMSNorm(to_k(context)) and V^T of the cross-attention (attention.py:247-280 with the T5 context as k/v
        input). They depend on the weights and the context only - not on x or the timestep - so they are computed once per
        (weight set, context tensor) and reused by the 2 x 35 forwards of a chunk (28 x 2 x 4.3 MB per context). The cache key is
        the context tensor's storage address + in-place version counter + shape / dtype; a new prompt tensor or an in-place edit
        rebuilds the entry."""
        # inference tensors: no version counter -> recompute (see tensor_version); the same holds for the WEIGHTS: an unversioned weight set
        # (pk["versioned"] False) can have to_k / to_v edited in place behind an unchanged pk["key"]
        use_cache = cacheable(crossattn_emb) and pk["versioned"]
        key = (crossattn_emb.data_ptr(), tensor_version(crossattn_emb), tuple(crossattn_emb.shape), crossattn_emb.dtype, pk["key"])
        cache = self.__dict__.setdefault("_ca_kv_cache", {})
        hit = cache.get(key) if use_cache else None
        if hit is not None and hit[0] is crossattn_emb:
            return hit[1], hit[2]
        B, M = crossattn_emb.shape[:2]
        D, nH = self.model_channels, self.num_heads
        ctx = crossattn_emb.to(torch.bfloat16).permute(1, 0, 2).reshape(M * B, -1).contiguous()  # rows (m, b)
        per_block = []
        for blk in pk["blocks"]:
            kv = ops.gemm_nt(ctx, blk["ca_kv"])  # [M*B, 2D]
            per_block.append((ops.qk_rmsnorm_rope(kv[:, :D], blk["ca_kn"], None, None, M, B, nH), ops.transpose_v(kv[:, D:], M, B, nH)))
        if not use_cache:
            return per_block, 0  # (no zero-tail detection without a cache entry to keep it in: it costs a host synchronisation)
        # Zero-padded context (text_encoder pads the T5 embedding with zero rows to 512 tokens): to_k / to_v have no bias and RMSNorm(0) = 0, so the padding's K
        # rows and V^T columns are exactly zero in every block. How many leading tokens carry anything is read ONCE per cached context (one host
        # synchronisation, with the K / V^T tails VERIFIED to be zero on the device) and handed to the attention launch, which then runs its tile loop over
        # those keys only and adds the tail in closed form (g3_cross_attn_fwd_bf16: the padded tokens stay in the softmax denominator exactly as
        # general_dit.py:407-410 has them). A context without a zero tail, or any non-zero found in a tail, gives 0 = every key through the loop.
        dense = 0
        live = (crossattn_emb != 0).any(dim=-1).any(dim=0)  # [M]: token m is non-zero for some batch item
        n_live = int(live.nonzero().max()) + 1 if bool(live.any()) else 1
        cand = min(M, ops.ceil_to(n_live, 64))
        if cand < M:
            dirty = torch.zeros((), dtype=torch.bool, device=crossattn_emb.device)
            for k_, vt_ in per_block:
                dirty |= (k_[cand * B:] != 0).any() | (vt_[..., cand:M] != 0).any()
            if not bool(dirty):
                dense = cand
        while len(cache) >= 4:  # cond / uncond (+ one spare pair): bounded, oldest first
            cache.pop(next(iter(cache)))
        cache[key] = (crossattn_emb, per_block, dense)  # holding the tensor keeps its storage (hence the address in the key) alive
        return per_block, dense

    def _modulation(self, emb, ada, lora, n):
        """(shift, scale[, gate]) = chunk_n( W2 . (W1 . SiLU(emb)) + adaln_lora )   (blocks.py:442-447)"""
        w1, w2 = ada
        mid = ops.gemv(emb, w1, act_in=1)
        mod = ops.gemv(mid, w2, add=lora)  # [B, n*D]
        D = self.model_channels
        return tuple(mod[:, i * D:(i + 1) * D] for i in range(n))


def _nearest_resize(mask: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """torchvision.transforms.functional.resize(..., NEAREST) on a [B,1,h,w] / [B,h,w] mask == F.interpolate nearest."""
    m = mask
    squeeze = False
    if m.dim() == 3:
        m = m.unsqueeze(1)
        squeeze = True
    if tuple(m.shape[-2:]) != tuple(size):
        m = torch.nn.functional.interpolate(m.float(), size=size, mode="nearest").to(mask.dtype)
    # reference keeps [B, 1, H, W] then unsqueeze(1) -> here return [B, H, W]-like squeezed channel
    return m[:, 0] if (m.dim() == 4) else m
