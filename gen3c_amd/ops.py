"""Thin torch-tensor front end over the C ABI (include/gen3c_hip.h).

torch is used for what it is good at here - device memory and streams. Every function hands raw device pointers,
shapes and the current HIP stream to libgen3c_hip.so; none of them computes anything in PyTorch.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib

EPI_NONE, EPI_GELU, EPI_GATED_RESIDUAL, EPI_BIAS = 0, 1, 2, 3

# bench.py switches this to a list to collect (kernel, shape, HipTimer) triples for the dominant kernel: hipEvents are
# recorded on the launch stream right around the launch (a handful of events per step - no measurable overhead).
_KERNEL_TIMERS = None


def set_option(name: str, value: int):
    """Runtime A/B switch of the C library (g3_set_option)."""
    _lib.check(_lib.load().g3_set_option(name.encode(), int(value)), "g3_set_option")


def enable_kernel_timers(on: bool = True):
    global _KERNEL_TIMERS
    _KERNEL_TIMERS = [] if on else None


def collected_kernel_timers():
    return list(_KERNEL_TIMERS or [])


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, dtype=torch.bfloat16) -> int:
    if not t.is_cuda:
        raise _lib.Gen3cHipError(f"{name}: expected a GPU (HIP) tensor, got device {t.device}; gen3c_amd has no CPU path")
    if t.dtype != dtype:
        raise _lib.Gen3cHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t.data_ptr()


def _rowmajor2d(t: torch.Tensor, name: str):
    if t.dim() != 2 or t.stride(1) != 1:
        raise _lib.Gen3cHipError(f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


def gemm_nt(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE,
            gate: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = epi(a[M,K] @ w[N,K]^T).  gate: [rows, N] (row = m % rows); residual: [M, N]."""
    M, K, lda = _rowmajor2d(a, "a")
    N, Kw, ldw = _rowmajor2d(w, "w")
    if K != Kw:
        raise _lib.Gen3cHipError(f"gemm_nt: K mismatch {K} vs {Kw}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    Mo, No, ldc = _rowmajor2d(out, "out")
    assert (Mo, No) == (M, N)
    gp, grows, ldg, rp, ldr = 0, 1, 0, 0, 0
    if gate is not None:
        grows, gn, ldg = _rowmajor2d(gate, "gate")
        assert gn == N
        gp = _dev(gate, "gate")
    if residual is not None:
        rm, rn, ldr = _rowmajor2d(residual, "residual")
        assert (rm, rn) == (M, N)
        rp = _dev(residual, "residual")
    lib = _lib.load()
    timer = None
    if _KERNEL_TIMERS is not None and M >= 4096:  # the block GEMMs; tiny ones (context K/V, embeddings of a few rows) are not worth an event pair
        timer = HipTimer()
        timer.start()
    _lib.check(lib.g3_gemm_bf16_nt(_dev(a, "a"), lda, _dev(w, "w"), ldw, _dev(out, "out"), ldc, M, N, K, epilogue, gp, grows,
                                   ldg, rp, ldr, _stream()), "g3_gemm_bf16_nt")
    if timer is not None:
        timer.stop()
        _KERNEL_TIMERS.append(("gemm_nt", dict(M=M, N=N, K=K, epilogue=epilogue), timer))
    return out


def gemv(a: torch.Tensor, w: torch.Tensor, add: Optional[torch.Tensor] = None, act_in: int = 0,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M<=8, N] = act_in(a) @ w^T (+ add)."""
    M, K, lda = _rowmajor2d(a, "a")
    N, Kw, ldw = _rowmajor2d(w, "w")
    assert K == Kw
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    ap, ldadd = 0, 0
    if add is not None:
        am, an, ldadd = _rowmajor2d(add, "add")
        assert (am, an) == (M, N)
        ap = _dev(add, "add")
    lib = _lib.load()
    _lib.check(lib.g3_gemv_bf16(_dev(a, "a"), lda, _dev(w, "w"), ldw, ap, ldadd, _dev(out, "out"), out.stride(0), M, N, K,
                                act_in, _stream()), "g3_gemv_bf16")
    return out


def layernorm_modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, out: Optional[torch.Tensor] = None,
                       eps: float = 1e-6) -> torch.Tensor:
    """x: [rows, D] (rows = (s, b), b fastest); shift/scale: [B, D]."""
    rows, D, ldx = _rowmajor2d(x, "x")
    B, Ds, ldmod = _rowmajor2d(shift, "shift")
    assert Ds == D and scale.shape == shift.shape and scale.stride(0) == ldmod
    if out is None:
        out = torch.empty((rows, D), dtype=torch.bfloat16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.g3_layernorm_modulate_bf16(_dev(x, "x"), ldx, _dev(shift, "shift"), _dev(scale, "scale"), ldmod, B,
                                              _dev(out, "out"), out.stride(0), rows, D, eps, _stream()),
               "g3_layernorm_modulate_bf16")
    return out


def posemb_layernorm_modulate(x: torch.Tensor, pe_t: torch.Tensor, pe_h: Optional[torch.Tensor], pe_w: Optional[torch.Tensor],
                              pos_norm: Optional[torch.Tensor], T: int, Hp: int, Wp: int, B: int, shift: torch.Tensor, scale: torch.Tensor,
                              out: Optional[torch.Tensor] = None, eps: float = 1e-6) -> torch.Tensor:
    """x [T*Hp*Wp*B, D] += per-block absolute position embedding (IN PLACE), returns LayerNorm(x)*(1+scale)+shift.
    Either the three axis tables + norm (the kernel rebuilds every row), or pe_h = pe_w = pos_norm = None and pe_t the finished embedding
    [T*Hp*Wp, D] (one table read per row). See g3_posemb_layernorm_modulate_bf16."""
    rows, D, ldx = _rowmajor2d(x, "x")
    assert rows == T * Hp * Wp * B
    if pe_h is None:
        assert pe_w is None and pos_norm is None and pe_t.shape == (T * Hp * Wp, D) and pe_t.is_contiguous()
    else:
        for t, n in ((pe_t, T), (pe_h, Hp), (pe_w, Wp)):
            assert t.dim() == 2 and t.shape[0] >= n and t.shape[1] == D and t.is_contiguous()
        assert pos_norm.numel() == T * Hp * Wp and pos_norm.is_contiguous()
    Bm, Ds, ldmod = _rowmajor2d(shift, "shift")
    assert Ds == D and scale.shape == shift.shape and scale.stride(0) == ldmod
    if out is None:
        out = torch.empty((rows, D), dtype=torch.bfloat16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.g3_posemb_layernorm_modulate_bf16(_dev(x, "x"), ldx, _dev(pe_t, "pe_t"), _dev(pe_h, "pe_h") if pe_h is not None else 0,
                                                     _dev(pe_w, "pe_w") if pe_w is not None else 0, _dev(pos_norm, "pos_norm") if pos_norm is not None else 0,
                                                     T, Hp, Wp, B, _dev(shift, "shift"), _dev(scale, "scale"), ldmod, Bm,
                                                     _dev(out, "out"), out.stride(0), D, eps, _stream()), "g3_posemb_layernorm_modulate_bf16")
    return out


def qk_rmsnorm_rope(x: torch.Tensor, weight: torch.Tensor, cos: Optional[torch.Tensor], sin: Optional[torch.Tensor],
                    S: int, B: int, H: int, out: Optional[torch.Tensor] = None, eps: float = 1e-6) -> torch.Tensor:
    """x: [S*B, >=H*128] view (row stride arbitrary), per-head RMSNorm + optional RoPE -> out [S*B, H*128]."""
    rows, width, ld_in = _rowmajor2d(x, "x")
    assert rows == S * B and width == H * 128
    if out is None:
        out = torch.empty((rows, H * 128), dtype=torch.bfloat16, device=x.device)
    cp = sp = 0
    if cos is not None:
        assert cos.shape == (S, 128) and sin.shape == (S, 128) and cos.is_contiguous() and sin.is_contiguous()
        cp, sp = _dev(cos, "cos", torch.float32), _dev(sin, "sin", torch.float32)
    lib = _lib.load()
    _lib.check(lib.g3_qk_rmsnorm_rope_bf16(_dev(x, "x"), ld_in, _dev(weight, "weight"), cp, sp, _dev(out, "out"),
                                           out.stride(0), S, B, H, 128, eps, _stream()), "g3_qk_rmsnorm_rope_bf16")
    return out


def qk_rmsnorm_rope_pair(x: torch.Tensor, weight_q: torch.Tensor, H_q: int, weight_k: torch.Tensor, H_k: int, cos: Optional[torch.Tensor],
                         sin: Optional[torch.Tensor], S: int, B: int, out: Optional[torch.Tensor] = None, eps: float = 1e-6) -> torch.Tensor:
    """x: [S*B, (H_q + H_k)*128] view: heads [0, H_q) normalised with weight_q, the next H_k with weight_k, RoPE on all of them - q | k of the fused QKV
    buffer in ONE launch (g3_qk_rmsnorm_rope_pair_bf16). out=None: in place."""
    rows, width, ld_in = _rowmajor2d(x, "x")
    assert rows == S * B and width == (H_q + H_k) * 128
    if out is None:
        out = x
    cp = sp = 0
    if cos is not None:
        assert cos.shape == (S, 128) and sin.shape == (S, 128) and cos.is_contiguous() and sin.is_contiguous()
        cp, sp = _dev(cos, "cos", torch.float32), _dev(sin, "sin", torch.float32)
    lib = _lib.load()
    _lib.check(lib.g3_qk_rmsnorm_rope_pair_bf16(_dev(x, "x"), ld_in, _dev(weight_q, "weight_q"), H_q, _dev(weight_k, "weight_k"), H_k, cp, sp, _dev(out, "out"),
                                                out.stride(0), S, B, 128, eps, _stream()), "g3_qk_rmsnorm_rope_pair_bf16")
    return out


def gemm_qk_norm_rope(a: torch.Tensor, w: torch.Tensor, n_q: int, n_k: int, norm_q: Optional[torch.Tensor], norm_k: Optional[torch.Tensor],
                      cos: Optional[torch.Tensor], sin: Optional[torch.Tensor], S: int, B: int, out: Optional[torch.Tensor] = None,
                      eps: float = 1e-6, vt: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = a @ w^T with per-head RMSNorm (+ RoPE) applied to the feature ranges [0, n_q) (weight norm_q) and [n_q, n_q+n_k) (norm_k)
    in the GEMM epilogue; the remaining features (v) are stored as they are - or, with `vt` ([B, H_v, 128, ld] bf16, ld >= S, tail zero),
    written transposed into it instead (their columns of `out` then stay unwritten). Replaces gemm_nt + qk_rmsnorm_rope (+ transpose_v)."""
    M, K, lda = _rowmajor2d(a, "a")
    N, Kw, ldw = _rowmajor2d(w, "w")
    assert K == Kw and M == S * B
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    Mo, No, ldc = _rowmajor2d(out, "out")
    assert (Mo, No) == (M, N)
    cp = sp = 0
    if cos is not None:
        assert cos.shape == (S, 128) and sin.shape == (S, 128) and cos.is_contiguous() and sin.is_contiguous()
        cp, sp = _dev(cos, "cos", torch.float32), _dev(sin, "sin", torch.float32)
    lib = _lib.load()
    timer = None
    if _KERNEL_TIMERS is not None and M >= 4096:
        timer = HipTimer()
        timer.start()
    vtp, vt_ld = 0, 0
    if vt is not None:
        n_v = N - n_q - n_k
        assert n_v > 0 and vt.dim() == 4 and vt.shape[:3] == (B, n_v // 128, 128) and vt.shape[3] >= S and vt.is_contiguous()
        vtp, vt_ld = _dev(vt, "vt"), vt.shape[3]
    _lib.check(lib.g3_gemm_qk_norm_rope_bf16(_dev(a, "a"), lda, _dev(w, "w"), ldw, _dev(out, "out"), ldc, M, N, K, n_q, n_k,
                                             _dev(norm_q, "norm_q") if n_q else 0, _dev(norm_k, "norm_k") if n_k else 0, cp, sp, B, eps, vtp, vt_ld,
                                             _stream()),
               "g3_gemm_qk_norm_rope_bf16")
    if timer is not None:
        timer.stop()
        _KERNEL_TIMERS.append(("gemm_nt", dict(M=M, N=N, K=K, epilogue=5), timer))
    return out


def ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def transpose_v(v: torch.Tensor, S: int, B: int, H: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """v: [S*B, H*128] view -> V^T [B, H, 128, ceil64(S)] with a zero tail."""
    rows, width, ld_in = _rowmajor2d(v, "v")
    assert rows == S * B and width == H * 128
    ldvt = ceil_to(S, 64)
    if out is None:
        out = torch.empty((B, H, 128, ldvt), dtype=torch.bfloat16, device=v.device)
    assert out.shape == (B, H, 128, ldvt) and out.is_contiguous()
    lib = _lib.load()
    _lib.check(lib.g3_transpose_v_bf16(_dev(v, "v"), ld_in, _dev(out, "out"), ldvt, S, B, H, 128, _stream()),
               "g3_transpose_v_bf16")
    return out


def flash_attn(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, Sq: int, Skv: int, B: int, H: int,
               out: Optional[torch.Tensor] = None, softmax_scale: Optional[float] = None, variant: int = 0, partial: bool = False, kv_dense: int = 0,
               q_norm_weight: Optional[torch.Tensor] = None, q_norm_eps: float = 1e-6):
    """q: [Sq*B, H*128], k: [Skv*B, H*128] (rows (s,b), b fastest), vt: [B, H, 128, ldvt] -> out [Sq*B, H*128].
    vt may also be 5-D [n_seg, B, H, 128, ld_seg]: V^T in key segments of Skv / n_seg keys each (a rank-major all-gather of
    per-rank V^T shards, see g3_flash_attn_fwd_kvseg_bf16).
    variant: per-call kernel choice (0 = library option / automatic; 4 = 8-wave kernel, 11 = one-wave-per-SIMD kernel) - no global state.
    partial=True: the keys are only PART of the rows' keys - returns (o_part fp32 [Sq*B, H*128], lse fp32 [B, H, Sq]) for attn_merge.
    kv_dense (0 < kv_dense < Skv): the CALLER guarantees all-zero K rows and V^T columns for keys [kv_dense, Skv) (zero-padded context tokens): they enter the
    softmax in closed form instead of through the tile loop; q_norm_weight ([128] bf16): q is the raw projection and its per-head RMSNorm runs inside the
    kernel's Q load (both: g3_cross_attn_fwd_bf16)."""
    qr, qw, ldq = _rowmajor2d(q, "q")
    kr, kw, ldk = _rowmajor2d(k, "k")
    assert qr == Sq * B and kr == Skv * B and qw == H * 128 and kw == H * 128
    assert vt.is_contiguous() and vt.dim() in (4, 5) and tuple(vt.shape[-4:-1]) == (B, H, 128)
    ldvt = vt.shape[-1]
    o_part = lse = None
    if partial:
        assert out is None
        o_part = torch.empty((Sq * B, H * 128), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
        ldo = o_part.stride(0)
    else:
        if out is None:
            out = torch.empty((Sq * B, H * 128), dtype=torch.bfloat16, device=q.device)
        ldo = out.stride(0)
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(128)
    lib = _lib.load()
    timer = None
    if _KERNEL_TIMERS is not None:
        timer = HipTimer()
        timer.start()
    n_seg = vt.shape[0] if vt.dim() == 5 else 0
    if n_seg:
        assert Skv % n_seg == 0
    if 0 < kv_dense < Skv or q_norm_weight is not None:
        assert not partial and not n_seg and not variant, "the cross-attention form takes plain V^T, a bf16 output and the automatic kernel choice"
        kd = int(kv_dense) if 0 < kv_dense < Skv else 0
        _lib.check(lib.g3_cross_attn_fwd_bf16(_dev(q, "q"), ldq * B, ldq, 128, _dev(q_norm_weight, "q_norm_weight") if q_norm_weight is not None else 0, float(q_norm_eps),
                                              _dev(k, "k"), ldk * B, ldk, 128, _dev(vt, "vt"), ldvt, H * 128 * ldvt, 128 * ldvt,
                                              _dev(out, "out"), ldo * B, ldo, 128, Sq, Skv, kd, B, H, 128, float(softmax_scale), _stream()),
                   "g3_cross_attn_fwd_bf16")
        if timer is not None:
            timer.stop()
            _KERNEL_TIMERS.append(("flash_attn_fwd", dict(Sq=Sq, Skv=Skv, B=B, H=H, partial=False, kv_dense=kd, q_norm=q_norm_weight is not None,
                                                          kernel=lib.g3_flash_attn_kernel_name_ex(Sq, Skv, B, H, 4).decode()), timer))
        return out
    _lib.check(lib.g3_flash_attn_fwd_ex_bf16(_dev(q, "q"), ldq * B, ldq, 128, _dev(k, "k"), ldk * B, ldk, 128, _dev(vt, "vt"), ldvt, H * 128 * ldvt, 128 * ldvt,
                                             Skv // n_seg if n_seg else 0, B * H * 128 * ldvt if n_seg else 0,
                                             0 if partial else _dev(out, "out"), _dev(o_part, "o_part", torch.float32) if partial else 0,
                                             _dev(lse, "lse", torch.float32) if partial else 0, ldo * B, ldo, 128, Sq, Skv, B, H, 128,
                                             float(softmax_scale), int(variant), _stream()), "g3_flash_attn_fwd_ex_bf16")
    if timer is not None:
        timer.stop()
        # the kernel the launcher picked for THIS launch (per-call variant, else the library option in force)
        _KERNEL_TIMERS.append(("flash_attn_fwd", dict(Sq=Sq, Skv=Skv, B=B, H=H, partial=partial,
                                                      kernel=lib.g3_flash_attn_kernel_name_ex(Sq, Skv, B, H, int(variant)).decode()), timer))
    return (o_part, lse) if partial else out


def attn_merge(parts, Sq: int, B: int, H: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """parts: list of (o_part fp32 [Sq*B, H*128], lse fp32 [B, H, Sq]) from flash_attn(partial=True) over disjoint key sets of the same
    queries -> bf16 [Sq*B, H*128] = attention over the union of the keys (g3_attn_merge_partials_bf16). `out` may be a column view."""
    import ctypes as C
    n = len(parts)
    assert 1 <= n <= 8
    ld = parts[0][0].stride(0)
    for o, l in parts:
        assert o.shape == (Sq * B, H * 128) and o.stride(1) == 1 and o.stride(0) == ld and l.shape == (B, H, Sq) and l.is_contiguous()
    if out is None:
        out = torch.empty((Sq * B, H * 128), dtype=torch.bfloat16, device=parts[0][0].device)
    orows, ow, ldo = _rowmajor2d(out, "out")
    assert (orows, ow) == (Sq * B, H * 128)
    op = (C.c_void_p * n)(*[_dev(o, "o_part", torch.float32) for o, _ in parts])
    lp = (C.c_void_p * n)(*[_dev(l, "lse", torch.float32) for _, l in parts])
    _lib.check(_lib.load().g3_attn_merge_partials_bf16(op, lp, n, ld * B, ld, 128, _dev(out, "out"), ldo * B, ldo, 128, Sq, B, H, 128, _stream()),
               "g3_attn_merge_partials_bf16")
    return out


def add_inplace(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    lib = _lib.load()
    _lib.check(lib.g3_add_inplace_bf16(_dev(x, "x"), _dev(y, "y"), x.numel(), _stream()), "g3_add_inplace_bf16")
    return x


def dit_patchify(sources, B: int, T: int, H: int, W: int, patch_t: int, patch_s: int) -> torch.Tensor:
    """sources: list of (tensor, has_t) - bf16 contiguous [B,C_i,T,H,W] (has_t) or [B,C_i,H,W] broadcast over T.
    -> patches [(T/pt)*(H/ps)*(W/ps)*B, sum(C_i)*pt*ps*ps] bf16, rows (t h w b), columns (c r m n). See g3_dit_patchify_bf16."""
    import ctypes as C
    assert 1 <= len(sources) <= 4
    ptrs, chans, has_t = [], [], []
    for i, (t, ht) in enumerate(sources):
        assert t.is_contiguous() and t.shape[0] == B and tuple(t.shape[-2:]) == (H, W) and (t.dim() == 5 and t.shape[2] == T if ht else t.dim() == 4)
        ptrs.append(_dev(t, f"source{i}"))
        chans.append(int(t.shape[1]))
        has_t.append(1 if ht else 0)
    n = len(sources)
    ctot = sum(chans)
    out = torch.empty(((T // patch_t) * (H // patch_s) * (W // patch_s) * B, ctot * patch_t * patch_s * patch_s), dtype=torch.bfloat16,
                      device=sources[0][0].device)
    lib = _lib.load()
    _lib.check(lib.g3_dit_patchify_bf16((C.c_void_p * n)(*ptrs), (C.c_int * n)(*chans), (C.c_int * n)(*has_t), n, out.data_ptr(), B, T, H, W,
                                        patch_t, patch_s, _stream()), "g3_dit_patchify_bf16")
    return out


def dit_unpatchify(y: torch.Tensor, B: int, C_out: int, T: int, H: int, W: int, patch_t: int, patch_s: int) -> torch.Tensor:
    """y [(T/pt)(H/ps)(W/ps) B, ps*ps*pt*C_out] -> [B, C_out, T, H, W]. See g3_dit_unpatchify_bf16."""
    rows, cols, ldy = _rowmajor2d(y, "y")
    assert rows == (T // patch_t) * (H // patch_s) * (W // patch_s) * B and cols == patch_s * patch_s * patch_t * C_out
    out = torch.empty((B, C_out, T, H, W), dtype=torch.bfloat16, device=y.device)
    lib = _lib.load()
    _lib.check(lib.g3_dit_unpatchify_bf16(_dev(y, "y"), ldy, out.data_ptr(), B, C_out, T, H, W, patch_t, patch_s, _stream()), "g3_dit_unpatchify_bf16")
    return out


def timestep_embedding(timesteps: torch.Tensor, norm_weight: torch.Tensor, D: int):
    """timesteps f32 [B] -> (t_sin bf16 [B,D], emb bf16 [B,D]). See g3_timestep_embedding_bf16."""
    assert timesteps.dim() == 1 and timesteps.is_contiguous() and norm_weight.numel() == D
    Bn = timesteps.shape[0]
    t_sin = torch.empty((Bn, D), dtype=torch.bfloat16, device=timesteps.device)
    emb = torch.empty_like(t_sin)
    lib = _lib.load()
    _lib.check(lib.g3_timestep_embedding_bf16(_dev(timesteps, "timesteps", torch.float32), _dev(norm_weight, "norm_weight"), t_sin.data_ptr(),
                                              emb.data_ptr(), Bn, D, _stream()), "g3_timestep_embedding_bf16")
    return t_sin, emb


def edm_prepare_input(xt, gt_latent, noise, indicator, T: int, hw: int, augment_sigma: float, c_in_aug: float,
                      c_in_bf16: float, c_in_step: float):
    """-> (new_xt, new_xt_scaled), both bf16 like xt. See g3_edm_prepare_input_bf16."""
    assert xt.is_contiguous() and gt_latent.is_contiguous() and noise.is_contiguous() and indicator.is_contiguous()
    assert xt.shape == gt_latent.shape == noise.shape and indicator.numel() == T
    new_xt, new_xt_scaled = torch.empty_like(xt), torch.empty_like(xt)
    lib = _lib.load()
    _lib.check(lib.g3_edm_prepare_input_bf16(_dev(xt, "xt"), _dev(gt_latent, "gt_latent"), _dev(noise, "noise", torch.float32),
                                             _dev(indicator, "indicator", torch.float32), _dev(new_xt, "new_xt"),
                                             _dev(new_xt_scaled, "new_xt_scaled"), xt.numel(), T, hw, augment_sigma, c_in_aug,
                                             c_in_bf16, c_in_step, _stream()), "g3_edm_prepare_input_bf16")
    return new_xt, new_xt_scaled


def edm_cfg_euler_step(out_cond, out_uncond, new_xt, gt_latent, indicator, T: int, hw: int, guidance: float,
                       c_skip_bf16: float, c_out_bf16: float, c_skip: float, c_out: float, sigma: float, sigma_next: float):
    """-> xt_next (bf16). See g3_edm_cfg_euler_step_bf16."""
    for t in (out_cond, out_uncond, new_xt, gt_latent, indicator):
        assert t.is_contiguous()
    assert out_cond.shape == out_uncond.shape == new_xt.shape == gt_latent.shape
    xt_next = torch.empty_like(new_xt)
    lib = _lib.load()
    _lib.check(lib.g3_edm_cfg_euler_step_bf16(_dev(out_cond, "out_cond"), _dev(out_uncond, "out_uncond"), _dev(new_xt, "new_xt"),
                                              _dev(gt_latent, "gt_latent"), _dev(indicator, "indicator", torch.float32),
                                              _dev(xt_next, "xt_next"), new_xt.numel(), T, hw, guidance, c_skip_bf16, c_out_bf16,
                                              c_skip, c_out, sigma, sigma_next, _stream()), "g3_edm_cfg_euler_step_bf16")
    return xt_next


class HipTimer:
    """hipEvent pair recorded on torch's current stream through the C ABI (used by bench.py for kernel timing)."""

    def __init__(self):
        import ctypes as C
        self._lib = _lib.load()
        self._a, self._b = C.c_void_p(), C.c_void_p()
        _lib.check(self._lib.g3_event_create(C.byref(self._a)))
        _lib.check(self._lib.g3_event_create(C.byref(self._b)))

    def start(self):
        _lib.check(self._lib.g3_event_record(self._a, _stream()))

    def stop(self):
        _lib.check(self._lib.g3_event_record(self._b, _stream()))

    def elapsed_ms(self) -> float:
        import ctypes as C
        ms = C.c_float(0)
        _lib.check(self._lib.g3_event_elapsed_ms(self._a, self._b, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            self._lib.g3_event_destroy(self._a)
            self._lib.g3_event_destroy(self._b)
        except Exception:
            pass
