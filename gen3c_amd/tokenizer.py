"""MI355X-native causal video tokenizer (Cosmos-Tokenize1-CV8x8x8-720p) behind the reference's tokenizer plug-in
interface (SURVEY.md 8b "Tokenizer plugin"; cosmos_predict1/diffusion/module/pretrained_vae.py:271-611).

`CausalVideoTokenizerNet` holds the weights under the SAME state-dict keys as the modules the reference traces into
encoder.jit / decoder.jit (`encoder.*`, `quant_conv.*`, `post_quant_conv.*`, `decoder.*`;
tokenizer/networks/continuous_video.py:28-75), so `torch.jit.load("encoder.jit").state_dict()` drops in
(tokenizer/inference/utils.py:50-92 shows the same extraction). Its encoder()/decoder() run every convolution and both
attention products on the MFMA implicit-GEMM kernel (g3_conv3d_cl_bf16 / g3_gemm_bf16_nt) over channels-last bf16
activations, and everything else (GroupNorm+swish, Haar (un)patching, pooling / repeat up-sampling, softmax, temporal
attention) on the HBM-bound kernels of csrc/tokenizer.hip. PyTorch allocates; it computes nothing on this path except
the 16-channel latent's final layout change and mean/std scaling.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from . import _lib, ops

bf16 = torch.bfloat16
# 0: GroupNorm statistics always by their own pass over the tensor (A/B switch; default: produced in the epilogue of the convolution that writes it)
_FUSE_GN_STATS = os.environ.get("G3_FUSE_GN_STATS", "1") != "0"
# streams the spatial attention's frames are spread over (1 = all on the caller's stream)
# 0: the spatial attention as three kernels per frame (scores GEMM -> row softmax -> P.V GEMM; rounds 1-3, kept for A/B and for widths other than 512)
_FLASH_SPATIAL_ATTN = os.environ.get("G3_TOK_FLASH_ATTN", "1") != "0"
_ATTN_STREAMS = max(1, int(os.environ.get("G3_TOK_ATTN_STREAMS", "2")))  # measured 1: 47.0 / 73.0 ms, 2: 45.0 / 70.4, 3: 45.0 / 70.7, 4: 45.4 / 71.3, 6: 45.8 / 71.1 (encode / decode, one run)


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> int:
    if t is None:
        return 0
    if not t.is_cuda or not t.is_contiguous():
        raise _lib.Gen3cHipError("tokenizer: expected a contiguous GPU tensor (no CPU path)")
    return t.data_ptr()


# conv geometries: (kt,kh,kw, st,sh,sw, ot,oh,ow)
_GEOM = {
    "s3": (1, 3, 3, 1, 1, 1, 0, -1, -1),     # (1,3,3), zero pad 1
    "t3": (3, 1, 1, 1, 1, 1, -2, 0, 0),      # (3,1,1), causal (first frame replicated twice)
    "p1": (1, 1, 1, 1, 1, 1, 0, 0, 0),
    "s3s2": (1, 3, 3, 1, 2, 2, 0, 0, 0),     # stride-2 on the right/bottom zero-padded input (layers3d.py:216-218)
    "t3s2": (3, 1, 1, 2, 1, 1, -2, 0, 0),    # time-stride 2 on the front-replicated input (layers3d.py:224-225)
}


class CausalVideoTokenizerNet(torch.nn.Module):
    def __init__(self, channels: int = 128, channels_mult=(2, 4, 4), num_res_blocks: int = 2, z_channels: int = 16,
                 latent_channels: int = 16, in_channels: int = 3, patch_size: int = 4, device=None):
        super().__init__()
        if patch_size != 4 or in_channels != 3 or len(channels_mult) != 3:
            raise NotImplementedError("only the CV8x8x8 configuration (Haar patch 4, 3 levels, one spatial+temporal down) is built")
        self.channels, self.channels_mult, self.num_res_blocks = channels, tuple(channels_mult), num_res_blocks
        self.z_channels, self.latent_channels = z_channels, latent_channels
        self.dev = torch.device(device) if device is not None else torch.device("cuda")
        self._w: Dict[str, torch.Tensor] = {}   # packed tap-major conv weights [taps][N][K] and 1-D params, bf16 on device
        self._shapes: Dict[str, tuple] = {}

    # ------------------------------------------------------------------------------------------------ weights
    def expected_keys(self):
        c, m = self.channels, self.channels_mult
        keys = {}

        def conv(name, cout, cin, k):
            keys[f"{name}.conv3d.weight"] = (cout, cin, *k)
            keys[f"{name}.conv3d.bias"] = (cout,)

        def norm(name, ch):
            keys[f"{name}.norm.weight"] = (ch,)
            keys[f"{name}.norm.bias"] = (ch,)

        def res(name, cin, cout):
            norm(f"{name}.norm1", cin)
            conv(f"{name}.conv1.0", cout, cin, (1, 3, 3)); conv(f"{name}.conv1.1", cout, cout, (3, 1, 1))
            norm(f"{name}.norm2", cout)
            conv(f"{name}.conv2.0", cout, cout, (1, 3, 3)); conv(f"{name}.conv2.1", cout, cout, (3, 1, 1))
            if cin != cout:
                conv(f"{name}.nin_shortcut", cout, cin, (1, 1, 1))

        def attn(name, ch):
            for j in (0, 1):
                norm(f"{name}.{j}.norm", ch)
                for n in ("q", "k", "v", "proj_out"):
                    conv(f"{name}.{j}.{n}", ch, ch, (1, 1, 1))

        pin = 3 * 64
        conv("encoder.conv_in.0", c, pin, (1, 3, 3)); conv("encoder.conv_in.1", c, c, (3, 1, 1))
        cin = c
        for lvl in range(3):
            cout = c * m[lvl]
            for j in range(self.num_res_blocks):
                res(f"encoder.down.{lvl}.block.{j}", cin, cout)
                cin = cout
            if lvl == 0:
                conv("encoder.down.0.downsample.conv1", cin, cin, (1, 3, 3)); conv("encoder.down.0.downsample.conv2", cin, cin, (3, 1, 1))
                conv("encoder.down.0.downsample.conv3", cin, cin, (1, 1, 1))
        res("encoder.mid.block_1", cin, cin); attn("encoder.mid.attn_1", cin); res("encoder.mid.block_2", cin, cin)
        norm("encoder.norm_out", cin)
        conv("encoder.conv_out.0", self.z_channels, cin, (1, 3, 3)); conv("encoder.conv_out.1", self.z_channels, self.z_channels, (3, 1, 1))
        conv("quant_conv", self.latent_channels, self.z_channels, (1, 1, 1))
        conv("post_quant_conv", self.z_channels, self.latent_channels, (1, 1, 1))
        cin = c * m[2]
        conv("decoder.conv_in.0", cin, self.z_channels, (1, 3, 3)); conv("decoder.conv_in.1", cin, cin, (3, 1, 1))
        res("decoder.mid.block_1", cin, cin); attn("decoder.mid.attn_1", cin); res("decoder.mid.block_2", cin, cin)
        for lvl in (2, 1, 0):
            cout = c * m[lvl]
            for j in range(self.num_res_blocks + 1):
                res(f"decoder.up.{lvl}.block.{j}", cin, cout)
                cin = cout
            if lvl == 1:
                conv("decoder.up.1.upsample.conv1", cin, cin, (3, 1, 1)); conv("decoder.up.1.upsample.conv2", cin, cin, (1, 3, 3))
                conv("decoder.up.1.upsample.conv3", cin, cin, (1, 1, 1))
        norm("decoder.norm_out", cin)
        conv("decoder.conv_out.0", pin, cin, (1, 3, 3)); conv("decoder.conv_out.1", pin, pin, (3, 1, 1))
        return keys

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        exp = self.expected_keys()
        skip = ("wavelets", "_arange", "patch_size_buffer")
        extra = [k for k in state_dict if k not in exp and not any(s in k for s in skip)]
        missing = [k for k in exp if k not in state_dict]
        if strict and (extra or missing):
            raise RuntimeError(f"tokenizer state dict mismatch: missing {missing[:5]}..., unexpected {extra[:5]}...")
        self._w.clear()
        for k, shape in exp.items():
            if k not in state_dict:
                continue
            t = state_dict[k]
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"{k}: expected shape {shape}, got {tuple(t.shape)}")
            t = t.to(self.dev, bf16)
            if t.dim() == 5:  # [N,K,kt,kh,kw] -> tap-major [kt*kh*kw][N][K]
                n, kk = t.shape[:2]
                t = t.permute(2, 3, 4, 0, 1).reshape(-1, n, kk)
            self._w[k] = t.contiguous()
        return torch.nn.modules.module._IncompatibleKeys(missing, extra)

    def init_random(self, seed: int = 0):
        """Seeded random weights with the reference's shapes (tests / benchmarks without checkpoints)."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shape in self.expected_keys().items():
            if k.endswith("norm.weight"):
                sd[k] = torch.rand(shape, generator=g) + 0.5
            elif k.endswith(".bias"):
                sd[k] = torch.randn(shape, generator=g) * 0.05
            else:
                fan_in = shape[1] * shape[2] * shape[3] * shape[4]
                sd[k] = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
        self.load_state_dict(sd)
        return sd

    # ------------------------------------------------------------------------------------------------ primitive calls
    def _conv(self, x: torch.Tensor, name: str, kind: str, residual: Optional[torch.Tensor] = None, stats: bool = False) -> torch.Tensor:
        """stats=True: the output is the input of a CausalNormalize - its per-frame GroupNorm statistics are produced with the convolution
        (g3_conv3d_cl_gnstats_bf16: in the kernel's epilogue) and kept for the _gn call that consumes this very tensor."""
        w, b = self._w[f"{name}.conv3d.weight"], self._w[f"{name}.conv3d.bias"]
        kt, kh, kw, st, sh, sw, ot, oh, ow = _GEOM[kind]
        assert w.shape[0] == kt * kh * kw, f"{name}: weight taps {w.shape[0]} vs geometry {kind}"
        Ti, Hi, Wi, K = x.shape
        N = w.shape[1]
        assert w.shape[2] == K, f"{name}: Cin {w.shape[2]} vs activation channels {K}"
        To, Ho, Wo = Ti, Hi, Wi
        if kind == "s3s2":
            Ho, Wo = (Hi + 1 - 3) // 2 + 1, (Wi + 1 - 3) // 2 + 1
        elif kind == "t3s2":
            To = (Ti + 2 - 3) // 2 + 1
        out = torch.empty((To, Ho, Wo, N), dtype=bf16, device=x.device)
        if residual is not None:
            assert residual.shape == out.shape and residual.is_contiguous()
        lib = _lib.load()
        geom = (K, N, Ti, Hi, Wi, To, Ho, Wo, kt, kh, kw, st, sh, sw, ot, oh, ow)
        if stats and _FUSE_GN_STATS:
            st64 = torch.zeros((To, 2), dtype=torch.float64, device=x.device)
            _lib.check(lib.g3_conv3d_cl_gnstats_bf16(_ptr(x), K, _ptr(w), K, _ptr(b), _ptr(residual), N, _ptr(out), N, *geom, _ptr(st64), Ho * Wo, _st()),
                       f"g3_conv3d_cl_gnstats_bf16({name})")
            self._pending_stats = (out, st64)  # consumed by the next _gn of `out` (held by reference: the address cannot be recycled meanwhile)
        else:
            _lib.check(lib.g3_conv3d_cl_bf16(_ptr(x), K, _ptr(w), K, _ptr(b), _ptr(residual), N, _ptr(out), N, *geom, _st()), f"g3_conv3d_cl_bf16({name})")
        return out

    def _gn(self, x: torch.Tensor, name: str, swish: bool) -> torch.Tensor:
        T, H, W, C = x.shape
        out = torch.empty_like(x)
        lib = _lib.load()
        gamma, beta = self._w[f"{name}.norm.weight"], self._w[f"{name}.norm.bias"]
        pend = getattr(self, "_pending_stats", None)
        if pend is not None and pend[0] is x:  # the producing convolution already delivered the statistics: one pass over x
            self._pending_stats = None
            _lib.check(lib.g3_groupnorm_apply_cl_bf16(_ptr(x), C, _ptr(gamma), _ptr(beta), _ptr(pend[1]), _ptr(out), C, T, H * W, C, 1e-6,
                                                      1 if swish else 0, _st()), f"g3_groupnorm_apply_cl_bf16({name})")
            return out
        stats = torch.empty((T, 2), dtype=torch.float64, device=x.device)
        _lib.check(lib.g3_groupnorm_swish_cl_bf16(_ptr(x), C, _ptr(gamma), _ptr(beta), _ptr(stats), _ptr(out), C, T, H * W, C, 1e-6,
                                                  1 if swish else 0, _st()), f"g3_groupnorm_swish_cl_bf16({name})")
        return out

    def _resample(self, x: torch.Tensor, mode: int) -> torch.Tensor:
        T, H, W, C = x.shape
        To, Ho, Wo = T, H, W
        if mode == 0:
            Ho, Wo = (H + 1) // 2, (W + 1) // 2
        elif mode == 1:
            To = (T + 1) // 2
        elif mode == 2:
            To = 2 * T - 1 if T > 1 else 1
        elif mode == 3:
            Ho, Wo = 2 * H, 2 * W
        out = torch.empty((To, Ho, Wo, C), dtype=bf16, device=x.device)
        _lib.check(_lib.load().g3_resample_cl_bf16(_ptr(x), _ptr(out), T, H, W, C, mode, _st()), "g3_resample_cl_bf16")
        return out

    def _res_block(self, x: torch.Tensor, name: str, next_is_norm: bool = True) -> torch.Tensor:
        """next_is_norm: the block's output feeds a CausalNormalize (the next res block's norm1, an attention norm, norm_out) - its GroupNorm
        statistics are then produced by the last convolution's epilogue. False for the two blocks in front of a re-sampling stage
        (encoder.down.0.block.1 -> downsample, decoder.up.1.block.2 -> upsample): nobody would read them."""
        h = self._gn(x, f"{name}.norm1", True)
        h = self._conv(h, f"{name}.conv1.0", "s3")
        h = self._conv(h, f"{name}.conv1.1", "t3", stats=True)
        h = self._gn(h, f"{name}.norm2", True)
        h = self._conv(h, f"{name}.conv2.0", "s3")
        skip = self._conv(x, f"{name}.nin_shortcut", "p1") if f"{name}.nin_shortcut.conv3d.weight" in self._w else x
        return self._conv(h, f"{name}.conv2.1", "t3", residual=skip, stats=next_is_norm)

    def _spatial_attn(self, x: torch.Tensor, name: str) -> torch.Tensor:
        T, H, W, C = x.shape
        HW = H * W
        hn = self._gn(x, f"{name}.norm", False)
        q = self._conv(hn, f"{name}.q", "p1").view(T, HW, C)
        k = self._conv(hn, f"{name}.k", "p1").view(T, HW, C)
        v = self._conv(hn, f"{name}.v", "p1").view(T, HW, C)
        o = torch.empty((T, HW, C), dtype=bf16, device=x.device)
        lib = _lib.load()
        if _FLASH_SPATIAL_ATTN and C == 512 and HW % 64 == 0 and x.is_cuda:
            # One flash-style pass per frame (csrc/attention_d512.hip): no 14 080 x 14 080 score matrix in HBM, no separate softmax launch. V^T of
            # ALL frames by one transpose of the [T HW, C] matrix: frame f's V^T = columns [f HW, (f + 1) HW) of the [C, T HW] result.
            vT = torch.empty((C, T * HW), dtype=bf16, device=x.device)
            _lib.check(lib.g3_transpose2d_bf16(_ptr(v), C, _ptr(vT), T * HW, T * HW, C, _st()), "g3_transpose2d_bf16")
            rc = lib.g3_spatial_attn_d512_bf16(_ptr(q), _ptr(k), _ptr(vT), T * HW, HW, _ptr(o), T, HW, float(C) ** -0.5, _st())
            if rc == 0:
                return self._conv(o.view(T, H, W, C), f"{name}.proj_out", "p1", residual=x, stats=True)
            if rc not in (_lib.G3_ERR_ARG, _lib.G3_ERR_RESOURCE):
                _lib.check(rc, "g3_spatial_attn_d512_bf16")  # a launch failure / sticky device error is not a refusal: raise here, under its own name
            # (148 KB of dynamic LDS refused, a clip beyond the kernel's 32-bit offsets): the three-kernel HIP path below computes the same
            # attention; say so once - a silent 2x slowdown of this stage would otherwise go unnoticed
            if not getattr(self, "_flash_warned", False):
                self._flash_warned = True
                import warnings
                warnings.warn(f"g3_spatial_attn_d512_bf16 refused ({_lib.last_error()}); using the score-matrix path")
            del vT
        ldp = ops.ceil_to(HW, 8)
        # Frames are independent (time2batch, layers3d.py:362-364). One frame's products do not fill the chip - scores = q k^T is 8 K tiles per
        # 256 x 256 output tile (its launches wait on their stores), p v has (HW / 256) x 2 = 110 tiles for 256 CUs at 704 x 1280 - so the
        # frames go round-robin over a few streams, each with its own score / V^T buffers: the tiles of one frame's launch fill the CUs another
        # frame's leaves idle (measured: profiles/r3_*_tokenizer_breakdown.txt).
        ns = min(_ATTN_STREAMS, T) if x.is_cuda else 1
        main = torch.cuda.current_stream(x.device)
        streams = [main] + [self._side_stream(i, x.device) for i in range(ns - 1)]
        bufs = [(torch.zeros((HW, ldp), dtype=bf16, device=x.device), torch.zeros((C, ldp), dtype=bf16, device=x.device)) for _ in range(ns)]
        ready = main.record_event()  # q, k, v, o and the zeroed buffers exist
        for f in range(T):
            st = streams[f % ns]
            scores, vT = bufs[f % ns]
            with torch.cuda.stream(st):
                if f < ns and st is not main:
                    st.wait_event(ready)
                ops.gemm_nt(q[f], k[f], out=scores[:, :HW])
                _lib.check(lib.g3_softmax_rows_bf16(_ptr(scores), ldp, HW, HW, float(C) ** -0.5, _st()), "g3_softmax_rows_bf16")
                _lib.check(lib.g3_transpose2d_bf16(_ptr(v[f]), C, _ptr(vT), ldp, HW, C, _st()), "g3_transpose2d_bf16")
                ops.gemm_nt(scores, vT, out=o[f])  # K = ldp (zero padded columns contribute nothing)
        for st in streams[1:]:
            main.wait_stream(st)  # the projection below (and the release of the buffers) follows every frame
        return self._conv(o.view(T, H, W, C), f"{name}.proj_out", "p1", residual=x, stats=True)

    def _side_stream(self, i: int, device) -> "torch.cuda.Stream":
        pool = self.__dict__.setdefault("_streams", {})
        key = (i, str(device))
        if key not in pool:
            pool[key] = torch.cuda.Stream(device=device)
        return pool[key]

    def _temporal_attn(self, x: torch.Tensor, name: str) -> torch.Tensor:
        T, H, W, C = x.shape
        hn = self._gn(x, f"{name}.norm", False)
        q = self._conv(hn, f"{name}.q", "p1")
        k = self._conv(hn, f"{name}.k", "p1")
        v = self._conv(hn, f"{name}.v", "p1")
        o = torch.empty_like(q)
        _lib.check(_lib.load().g3_temporal_attn_cl_bf16(_ptr(q), _ptr(k), _ptr(v), _ptr(o), T, H * W, C, float(C) ** -0.5, _st()),
                   "g3_temporal_attn_cl_bf16")
        return self._conv(o, f"{name}.proj_out", "p1", residual=x, stats=True)

    # ------------------------------------------------------------------------------------------------ networks
    @torch.no_grad()
    def encoder(self, video: torch.Tensor) -> torch.Tensor:
        """encoder_jit: [1,3,T,H,W] (T = 1+8k, H,W % 8 == 0) -> [1,16,1+k,H/8,W/8]  (layers3d.py:793-818)."""
        assert video.dim() == 5 and video.shape[0] == 1 and video.shape[1] == 3, "one clip at a time (the reference encodes B=1 chunks)"
        _, _, T, H, W = video.shape
        self._pending_stats = None
        vid = video[0].to(bf16).contiguous()
        Tp, Hp, Wp = (T + 3) // 4, H // 4, W // 4
        h = torch.empty((Tp, Hp, Wp, 192), dtype=bf16, device=vid.device)
        _lib.check(_lib.load().g3_haar3d_patch_bf16(_ptr(vid), _ptr(h), T, H, W, _st()), "g3_haar3d_patch_bf16")
        h = self._conv(h, "encoder.conv_in.0", "s3")
        h = self._conv(h, "encoder.conv_in.1", "t3", stats=True)
        for lvl in range(3):
            for j in range(self.num_res_blocks):
                h = self._res_block(h, f"encoder.down.{lvl}.block.{j}", next_is_norm=not (lvl == 0 and j == self.num_res_blocks - 1))
            if lvl == 0:
                d = "encoder.down.0.downsample"
                h = self._conv(h, f"{d}.conv1", "s3s2", residual=self._resample(h, 0))
                h = self._conv(h, f"{d}.conv2", "t3s2", residual=self._resample(h, 1))
                h = self._conv(h, f"{d}.conv3", "p1", stats=True)
        h = self._res_block(h, "encoder.mid.block_1")
        h = self._spatial_attn(h, "encoder.mid.attn_1.0")
        h = self._temporal_attn(h, "encoder.mid.attn_1.1")
        h = self._res_block(h, "encoder.mid.block_2")
        h = self._gn(h, "encoder.norm_out", True)
        h = self._conv(h, "encoder.conv_out.0", "s3")
        h = self._conv(h, "encoder.conv_out.1", "t3")
        h = self._conv(h, "quant_conv", "p1")
        self._pending_stats = None  # (do not keep the last producer's output alive)
        return h.permute(3, 0, 1, 2).unsqueeze(0).contiguous()

    @torch.no_grad()
    def decoder(self, z: torch.Tensor) -> torch.Tensor:
        """decoder_jit: [1,16,t,h,w] -> [1,3,1+8(t-1),8h,8w]  (layers3d.py:930-949)."""
        assert z.dim() == 5 and z.shape[0] == 1
        self._pending_stats = None
        h = z[0].to(bf16).permute(1, 2, 3, 0).contiguous()
        h = self._conv(h, "post_quant_conv", "p1")
        h = self._conv(h, "decoder.conv_in.0", "s3")
        h = self._conv(h, "decoder.conv_in.1", "t3", stats=True)
        h = self._res_block(h, "decoder.mid.block_1")
        h = self._spatial_attn(h, "decoder.mid.attn_1.0")
        h = self._temporal_attn(h, "decoder.mid.attn_1.1")
        h = self._res_block(h, "decoder.mid.block_2")
        for lvl in (2, 1, 0):
            for j in range(self.num_res_blocks + 1):
                h = self._res_block(h, f"decoder.up.{lvl}.block.{j}", next_is_norm=not (lvl == 1 and j == self.num_res_blocks))
            if lvl == 1:
                u = "decoder.up.1.upsample"
                hu = self._resample(h, 2)
                h = self._conv(hu, f"{u}.conv1", "t3", residual=hu)
                hu = self._resample(h, 3)
                h = self._conv(hu, f"{u}.conv2", "s3", residual=hu)
                h = self._conv(h, f"{u}.conv3", "p1", stats=True)
        h = self._gn(h, "decoder.norm_out", True)
        h = self._conv(h, "decoder.conv_out.0", "s3")
        h = self._conv(h, "decoder.conv_out.1", "t3")
        Tp, Hp, Wp, _ = h.shape
        vid = torch.empty((3, 4 * Tp - 3, 4 * Hp, 4 * Wp), dtype=bf16, device=h.device)
        _lib.check(_lib.load().g3_haar3d_unpatch_bf16(_ptr(h), 192, _ptr(vid), Tp, Hp, Wp, _st()), "g3_haar3d_unpatch_bf16")
        self._pending_stats = None
        return vid.unsqueeze(0)


class VideoTokenizer:
    """Plug-in surface of `VideoJITTokenizer` / `JointImageVideoSharedJITTokenizer` for the video path
    (pretrained_vae.py:24-76, 314-611): load_weights, encode, decode (latent mean/std normalisation, chunking by
    pixel_chunk_duration), reset_dtype, channel, compression factors, frame-count helpers."""

    def __init__(self, name: str = "cosmos_diffusion_tokenizer_comp8x8x8", latent_ch: int = 16, is_bf16: bool = True,
                 spatial_compression_factor: int = 8, temporal_compression_factor: int = 8, pixel_chunk_duration: int = 121,
                 spatial_resolution: str = "720", channels: int = 128, device=None, **_ignored):
        self.name, self.latent_ch = name, latent_ch
        self.dtype = bf16 if is_bf16 else torch.float32
        self._spatial, self._temporal, self._pixel_chunk = spatial_compression_factor, temporal_compression_factor, pixel_chunk_duration
        self.spatial_resolution = spatial_resolution
        self.net = CausalVideoTokenizerNet(channels=channels, latent_channels=latent_ch, device=device)
        self.latent_mean = self.latent_std = None
        self.image_latent_mean = self.image_latent_std = None

    # -- weights
    def load_weights(self, vae_dir: str):
        """checkpoints/Cosmos-Tokenize1-CV8x8x8-720p/{encoder.jit,decoder.jit,mean_std.pt} (pretrained_vae.py:194-214, 342-359)."""
        sd = {}
        for part in ("encoder", "decoder"):
            jit = torch.jit.load(os.path.join(vae_dir, f"{part}.jit"), map_location="cpu")
            sd.update(jit.state_dict())
        self.net.load_state_dict(sd, strict=True)
        mean, std = torch.load(os.path.join(vae_dir, "mean_std.pt"), weights_only=True)
        self.register_mean_std(mean, std)
        img_stats = os.path.join(vae_dir, "image_mean_std.pt")  # the image half of the joint tokenizer shares encoder / decoder, not the statistics
        if os.path.exists(img_stats):
            self.register_image_mean_std(*torch.load(img_stats, weights_only=True))

    def register_mean_std(self, latent_mean: torch.Tensor, latent_std: torch.Tensor):
        t = self.latent_chunk_duration
        shape = [1, self.latent_ch, t, 1, 1]
        self.latent_mean = latent_mean.view(self.latent_ch, -1)[:, :t].to(self.dtype).reshape(shape).to(self.net.dev)
        self.latent_std = latent_std.view(self.latent_ch, -1)[:, :t].to(self.dtype).reshape(shape).to(self.net.dev)

    def register_image_mean_std(self, latent_mean: torch.Tensor, latent_std: torch.Tensor):
        """Per-channel latent statistics of the IMAGE branch (`image_mean_std.pt`, BasePretrainedImageVAE.register_mean_std,
        pretrained_vae.py:110-124): JointImageVideoSharedJITTokenizer routes T == 1 inputs through the same encoder / decoder with these
        (pretrained_vae.py:520-545, 588-611). GEN3C's entry points never encode a single frame; the branch exists for the plug-in's completeness."""
        shape = [1, self.latent_ch, 1, 1, 1]
        self.image_latent_mean = latent_mean.to(self.dtype).reshape(shape).to(self.net.dev)
        self.image_latent_std = latent_std.to(self.dtype).reshape(shape).to(self.net.dev)

    def reset_dtype(self, *a, **k):
        return None  # weights are bf16 by construction

    # -- interface
    # The reference's joint tokenizer exposes its video half as `.video_vae` and helpers outside the class reach through it
    # (`model.tokenizer.video_vae.pixel_chunk_duration / latent_chunk_duration / is_casual`: inference_utils.py:677-691, 768-782;
    # JointImageVideoTokenizer, pretrained_vae.py:500-611). This class IS the video tokenizer, so the attribute is an alias of itself; the
    # image half (`image_vae`) is never reached from GEN3C's entry points and is not provided.
    is_casual = True  # (sic - the reference's spelling, pretrained_vae.py; "causal")

    @property
    def video_vae(self) -> "VideoTokenizer":
        return self

    @property
    def channel(self) -> int:
        return self.latent_ch

    @property
    def spatial_compression_factor(self) -> int:
        return self._spatial

    @property
    def temporal_compression_factor(self) -> int:
        return self._temporal

    @property
    def pixel_chunk_duration(self) -> int:
        return self._pixel_chunk

    @property
    def latent_chunk_duration(self) -> int:
        assert (self._pixel_chunk - 1) % self._temporal == 0
        return (self._pixel_chunk - 1) // self._temporal + 1

    def get_latent_num_frames(self, num_pixel_frames: int) -> int:
        if num_pixel_frames == 1:
            return 1
        assert num_pixel_frames % self._pixel_chunk == 0
        return num_pixel_frames // self._pixel_chunk * self.latent_chunk_duration

    def get_pixel_num_frames(self, num_latent_frames: int) -> int:
        if num_latent_frames == 1:
            return 1
        assert num_latent_frames % self.latent_chunk_duration == 0
        return num_latent_frames // self.latent_chunk_duration * self._pixel_chunk

    @torch.no_grad()
    def encode(self, state: torch.Tensor) -> torch.Tensor:
        """[B,3,T,H,W] in [-1,1] -> [B,16,T_lat,H/8,W/8], (z - mean) / std per (channel, latent frame)."""
        B, C, T, H, W = state.shape
        in_dtype = state.dtype
        if T == 1:  # image branch of the joint tokenizer (JointImageVideoTokenizer.encode, pretrained_vae.py:531-537)
            assert getattr(self, "image_latent_mean", None) is not None, "image branch: register_image_mean_std / image_mean_std.pt missing"
            z = torch.cat([self.net.encoder(state[b:b + 1].to(self.net.dev)) for b in range(B)], dim=0)
            return (z.to(in_dtype) - self.image_latent_mean.to(in_dtype)) / self.image_latent_std.to(in_dtype)
        assert T % self._pixel_chunk == 0, f"Temporal dimension {T} is not divisible by chunk_length {self._pixel_chunk}"
        outs = []
        for b in range(B):
            chunks = []
            for n in range(T // self._pixel_chunk):
                clip = state[b:b + 1, :, n * self._pixel_chunk:(n + 1) * self._pixel_chunk]
                z = self.net.encoder(clip.to(self.net.dev))
                chunks.append((z.to(in_dtype) - self.latent_mean.to(in_dtype)) / self.latent_std.to(in_dtype))
            outs.append(torch.cat(chunks, dim=2))
        return torch.cat(outs, dim=0)

    @torch.no_grad()
    def decode(self, latent: torch.Tensor) -> torch.Tensor:
        B, _, T, _, _ = latent.shape
        tl = self.latent_chunk_duration
        in_dtype = latent.dtype
        if T == 1 and tl != 1:  # image branch (JointImageVideoTokenizer.decode, pretrained_vae.py:539-544)
            assert getattr(self, "image_latent_mean", None) is not None, "image branch: register_image_mean_std / image_mean_std.pt missing"
            z = latent.to(self.net.dev) * self.image_latent_std.to(in_dtype) + self.image_latent_mean.to(in_dtype)
            return torch.cat([self.net.decoder(z[b:b + 1].to(self.dtype)) for b in range(B)], dim=0).to(in_dtype)
        assert T % tl == 0, f"Temporal dimension {T} is not divisible by chunk_length {tl}"
        outs = []
        for b in range(B):
            chunks = []
            for n in range(T // tl):
                z = latent[b:b + 1, :, n * tl:(n + 1) * tl].to(self.net.dev)
                z = z * self.latent_std.to(in_dtype) + self.latent_mean.to(in_dtype)
                chunks.append(self.net.decoder(z.to(self.dtype)).to(in_dtype))
            outs.append(torch.cat(chunks, dim=2))
        return torch.cat(outs, dim=0)
