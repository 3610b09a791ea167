"""MI355X renderer of the GEN3C 3D cache: same Python surface as the reference's
`forward_warp`, `unproject_points`, `reliable_depth_mask_range_batch` (forward_warp_utils_pytorch.py:171-187, 338-353,
410-460) and `Cache3D_Base / Cache3D_Buffer` (cache_3d.py:26-343), driving the HIP kernels of csrc/render.hip.

Differences in mechanics, not in results:
  * the cache (images, world points, masks) lives in HBM for its whole life - the reference keeps it on the CPU and
    pays an H2D copy + `torch.cuda.empty_cache()` for every 2 items (cache_3d.py:183-223);
  * all 121*N (frame, buffer) items of a render are processed in large batches; the reference's per-call coupling
    (log-depth max over the warp_chunk_size=2 items of one forward_warp call) is reproduced with `group_size=2`;
  * 4x4 / 3x3 inverses are taken on the host in fp32 (LAPACK, like the reference's torch.linalg.inv on CPU tensors).
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

import torch

from . import _lib

f32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor], name: str, dtype=f32) -> int:
    if t is None:
        return 0
    if not t.is_cuda:
        raise _lib.Gen3cHipError(f"{name}: expected a GPU (HIP) tensor, got {t.device}; the renderer has no CPU path")
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.Gen3cHipError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def _tensor_version(t: torch.Tensor):
    """In-place version counter for cache keys, or None = do not memoise on this tensor: inference tensors track no version but can still be
    edited in place inside torch.inference_mode() (the reference's pipeline entry points run under it, world_generation_pipeline.py:1225)."""
    return None if t.is_inference() else t._version


def _host_inverse(m: torch.Tensor) -> torch.Tensor:
    """inverse_with_conversion (forward_warp_utils_pytorch.py:147-148) on the host, result back on m's device."""
    return torch.linalg.inv(m.detach().to("cpu", f32)).to(m.device)


def unproject_points(depth: torch.Tensor, w2c: torch.Tensor, intrinsic: torch.Tensor, is_depth: bool = True,
                     mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(b,1,h,w) depth -> (b,h,w,3) world points, zeros where depth <= 0 (forward_warp_utils_pytorch.py:410-460)."""
    if not is_depth or mask is not None:
        raise NotImplementedError("only is_depth=True with the default mask (depth > 0) is used by the GEN3C cache")
    b, _, h, w = depth.shape
    d = depth.to(f32).reshape(b, h, w).contiguous()
    c2w = _host_inverse(w2c.reshape(b, 4, 4)).contiguous()
    kinv = _host_inverse(intrinsic.reshape(b, 3, 3)).contiguous()
    out = torch.empty((b, h, w, 3), dtype=f32, device=depth.device)
    lib = _lib.load()
    _lib.check(lib.g3_unproject_points_f32(_p(d, "depth"), _p(c2w, "c2w"), _p(kinv, "Kinv"), _p(out, "points"), b, h, w, _stream()),
               "g3_unproject_points_f32")
    return out


def reliable_depth_mask_range_batch(depth: torch.Tensor, window_size: int = 5, ratio_thresh: float = 0.05,
                                    eps: float = 1e-6) -> torch.Tensor:
    """-> bool (b,1,h,w): (local max - local min) / (local mean + eps) < thr and depth > 0 (:338-353)."""
    assert window_size % 2 == 1, "Window size must be odd."
    d4 = depth.unsqueeze(1) if depth.dim() == 3 else depth
    b, _, h, w = d4.shape
    d = d4.to(f32).reshape(b, h, w).contiguous()
    out = torch.empty((b, h, w), dtype=torch.uint8, device=depth.device)
    lib = _lib.load()
    _lib.check(lib.g3_reliable_depth_mask_f32(_p(d, "depth"), _p(out, "out", torch.uint8), b, h, w, window_size, ratio_thresh, eps,
                                              _stream()), "g3_reliable_depth_mask_f32")
    return out.bool().reshape(b, 1, h, w)


_ITEMS_CALL = True   # False: Cache3D.render_cache expands the sources per item and loops forward_warp (the reference's structure; A/B and tests)
_RENDER_WS: dict = {}  # insertion-ordered: least recently used first
_RENDER_WS_MAX_BYTES = int(os.environ.get("G3_RENDER_WS_MAX_BYTES", str(24 << 30)))  # ~1 GB per 32-item workspace at 704 x 1280; of 288 GB


def _render_workspace(lib, n: int, h: int, w: int, group_size: int, dev, stream) -> torch.Tensor:
    """Workspace of g3_render_items_f32, cached per (n, h, w, group_size, device, stream) and prepared once (g3_render_workspace_init): the
    kernels keep its accumulator part zero themselves, so a render never clears anything. Launches that share a workspace are ordered on its
    stream (renders issued from two streams get two workspaces). The cache is least-recently-used and bounded by BYTES, not entries: one
    two-stream render already holds four shapes (chunk and tail halves on two streams), a second resolution or an uneven multi-GPU shard
    must not push those out and re-run the init (a memset of the accumulator) on every call."""
    key = (n, h, w, group_size, str(dev), int(stream or 0))
    t = _RENDER_WS.pop(key, None)
    if t is None:
        nbytes = int(lib.g3_render_workspace_bytes(n, h, w, group_size))
        held = sum(v.numel() for v in _RENDER_WS.values())
        while _RENDER_WS and held + nbytes > _RENDER_WS_MAX_BYTES:
            held -= _RENDER_WS.pop(next(iter(_RENDER_WS))).numel()
        t = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        off = (-t.data_ptr()) % 256
        t = t[off:off + nbytes]
        _lib.check(lib.g3_render_workspace_init(t.data_ptr(), n, h, w, group_size, stream), "g3_render_workspace_init")
    _RENDER_WS[key] = t  # (re-)inserted at the most-recently-used end
    return t


def _drop_render_workspace(t: torch.Tensor):
    """A failed g3_render_items_f32 may leave its workspace's accumulator / tmin planes non-clean: never reuse it."""
    for k in [k for k, v in _RENDER_WS.items() if v is t]:
        del _RENDER_WS[k]


_TWO_STREAM_CHUNKS = os.environ.get("G3_RENDER_TWO_STREAMS", "1") != "0"
_SIDE_STREAMS: dict = {}


def _render_side_stream(dev) -> "torch.cuda.Stream":
    s = _SIDE_STREAMS.get(str(dev))
    if s is None:
        s = _SIDE_STREAMS[str(dev)] = torch.cuda.Stream(device=dev)
    return s


_WINDOW_SPLAT = True  # False: the two-call form (splat into the global accumulator with atomics, then resolve) - kept for A/B and tests
_WS_CACHE: dict = {}


def _window_workspace(nbytes: int, dev) -> torch.Tensor:
    """Scratch of the window splat (one 40x40x5-float window = 32 KiB per 32x32 source tile and item; sized by g3_warp_windows_workspace_bytes), cached per device: every call of a render reuses it - the
    launches are ordered on the stream, and the buffer is never read before it is rewritten."""
    t = _WS_CACHE.get(dev)
    if t is None or t.numel() < nbytes:
        t = _WS_CACHE[dev] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return t


def forward_warp(
    frame1: torch.Tensor,
    mask1: Optional[torch.Tensor],
    depth1: Optional[torch.Tensor],
    transformation1: Optional[torch.Tensor],
    transformation2: torch.Tensor,
    intrinsic1: Optional[torch.Tensor],
    intrinsic2: Optional[torch.Tensor],
    is_image=True,
    conditioned_normal1=None,
    cameraray_filtering=False,
    is_depth=True,
    render_depth=False,
    world_points1=None,
    foreground_masking=False,
    boundary_mask=None,
    group_size: Optional[int] = None,
) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], torch.Tensor]:
    """Same contract as the reference (forward_warp_utils_pytorch.py:171-336) for the cache path: `depth1=None`,
    `world_points1` (b,h,w,3) given. Returns (warped_frame2 (b,3,h,w), mask2 (b,1,h,w), warped_depth2 (b,h,w) or None,
    flow12 (b,2,h,w)). `group_size` (default: the whole batch, i.e. one reference call) sets which items share the
    log-depth maximum of the splat weights."""
    if depth1 is not None or conditioned_normal1 is not None or cameraray_filtering or not is_image:
        raise NotImplementedError("HIP forward_warp implements the GEN3C cache path (depth1=None, world_points1 given, images)")
    b, c, h, w = frame1.shape
    assert c == 3 and world_points1.shape == (b, h, w, 3)
    if intrinsic2 is None:
        assert intrinsic1 is not None, "intrinsic2 cannot be derived if intrinsic1 is None and intrinsic2 is None"
        intrinsic2 = intrinsic1
    dev = frame1.device
    gs = b if group_size is None else group_size
    img = frame1.to(f32).contiguous()
    pts = world_points1.to(f32).contiguous()
    w2c = transformation2.to(f32).reshape(b, 16).contiguous()
    K = intrinsic2.to(f32).reshape(b, 9).contiguous()
    m1 = None if mask1 is None else mask1.to(f32).reshape(b, h, w).contiguous()
    need_depth = bool(render_depth or foreground_masking)

    z = torch.empty((b, h, w), dtype=f32, device=dev)
    flow = torch.empty((b, 2, h, w), dtype=f32, device=dev)
    maskz = torch.empty((b, h, w), dtype=f32, device=dev)
    cam = torch.empty((b, h, w, 3), dtype=f32, device=dev) if foreground_masking else None
    ngroups = (b + gs - 1) // gs
    gmax = torch.zeros((ngroups,), dtype=torch.int32, device=dev)
    accum = torch.zeros((b, h + 2, w + 2, 5), dtype=f32, device=dev)
    frame = torch.empty((b, 3, h, w), dtype=f32, device=dev)
    mask2 = torch.empty((b, h, w), dtype=f32, device=dev)
    depth2 = torch.empty((b, h, w), dtype=f32, device=dev) if need_depth else None

    lib = _lib.load()
    st = _stream()
    _lib.check(lib.g3_warp_project_f32(_p(pts, "points"), _p(w2c, "w2c"), _p(K, "K"), _p(m1, "mask1"), _p(z, "z"), _p(flow, "flow"),
                                       _p(cam, "cam"), _p(maskz, "maskz"), _p(gmax, "gmax", torch.int32), b, h, w, gs, st),
               "g3_warp_project_f32")
    if _WINDOW_SPLAT:
        # splat + resolve without global atomics: source tiles store their destination windows, a destination-owning pass sums and resolves
        ws = _window_workspace(int(lib.g3_warp_windows_workspace_bytes(b, h, w)), dev)
        _lib.check(lib.g3_warp_splat_resolve_f32(_p(img, "image"), _p(z, "z"), _p(flow, "flow"), _p(maskz, "maskz"), _p(gmax, "gmax", torch.int32),
                                                 _p(accum, "accum"), ws.data_ptr(), _p(frame, "frame"), _p(mask2, "mask"), _p(depth2, "depth"), b, h, w, gs, st),
                   "g3_warp_splat_resolve_f32")
    else:
        _lib.check(lib.g3_warp_splat_f32(_p(img, "image"), _p(z, "z"), _p(flow, "flow"), _p(maskz, "maskz"), _p(gmax, "gmax", torch.int32),
                                         _p(accum, "accum"), b, h, w, gs, st), "g3_warp_splat_f32")
        _lib.check(lib.g3_warp_resolve_f32(_p(accum, "accum"), _p(frame, "frame"), _p(mask2, "mask"), _p(depth2, "depth"), b, h, w, st),
                   "g3_warp_resolve_f32")
    if foreground_masking:
        assert boundary_mask is not None
        bm = boundary_mask.reshape(b, h, w).to(torch.uint8).contiguous()
        kinv = _host_inverse(intrinsic2.reshape(b, 3, 3)).reshape(b, 9).contiguous()
        factor = 4  # mesh_downsample_factor (forward_warp_utils_pytorch.py:290)
        nh, nw = h // factor, w // factor
        pts_ds = torch.empty((b, nh, nw, 3), dtype=f32, device=dev)
        m_ds = torch.empty((b, nh, nw), dtype=torch.uint8, device=dev)
        tmin = torch.empty((b, h, w), dtype=torch.int32, device=dev)
        _lib.check(lib.g3_mesh_occlusion_f32(_p(cam, "cam"), _p(bm, "boundary", torch.uint8), _p(K, "K"), _p(kinv, "Kinv"),
                                             _p(pts_ds, "pts_ds"), _p(m_ds, "m_ds", torch.uint8), _p(tmin, "tmin", torch.int32),
                                             _p(frame, "frame"), _p(mask2, "mask"), _p(depth2, "depth"), b, h, w, factor, st),
                   "g3_mesh_occlusion_f32")
    out_dtype = frame1.dtype
    return (frame.to(out_dtype), mask2.reshape(b, 1, h, w).to(out_dtype), None if depth2 is None else depth2.to(out_dtype),
            flow.to(out_dtype))


class Cache3D_Base:
    """GPU-resident 3D cache with the reference's constructor keywords (cache_3d.py:26-134) for the layouts GEN3C
    uses: `input_format` a permutation/subset of B,F,N,V,C,H,W with V == 1."""

    def __init__(self, input_image, input_depth, input_w2c, input_intrinsics, input_mask=None, input_format=None,
                 input_points=None, weight_dtype=torch.float32, is_depth=True, device="cuda", filter_points_threshold=1.0,
                 foreground_masking=False):
        if weight_dtype != torch.float32:
            raise NotImplementedError("the HIP renderer computes in fp32 (the reference's default weight_dtype)")
        self.weight_dtype, self.is_depth, self.device = weight_dtype, is_depth, torch.device(device)
        self.filter_points_threshold, self.foreground_masking = filter_points_threshold, foreground_masking
        if input_format is None:
            assert input_image.dim() == 4
            input_format = ["B", "C", "H", "W"]
        pos = {d: i for i, d in enumerate(input_format)}
        shape = input_image.shape
        dim = lambda k: shape[pos[k]] if k in pos else 1
        B, F, N, V, H, W = dim("B"), dim("F"), dim("N"), dim("V"), dim("H"), dim("W")
        if V != 1:
            raise NotImplementedError("multi-view aggregation (V > 1) is not implemented by the reference either (cache_3d.py:229-230)")

        def canon(t, channels):  # -> (B,F,N,V,C,H,W)
            order = [pos[d] for d in ("B", "F", "N", "V", "C", "H", "W") if d in pos]
            t = t.permute(*order)
            for i, d in enumerate(("B", "F", "N", "V", "C", "H", "W")):
                if d not in pos:
                    t = t.unsqueeze(i)
            return t

        self.input_image = canon(input_image, 3).to(self.device, f32).contiguous()
        self.input_mask = None if input_mask is None else canon(input_mask, 1).to(self.device, f32).contiguous()
        if input_points is not None:
            self.input_points = input_points.reshape(B, F, N, V, H, W, 3).to(self.device, f32).contiguous()
            self.input_depth = None
        else:
            d = torch.nan_to_num(input_depth.to(self.device, f32), nan=100)
            d = torch.clamp(d, min=0, max=100)
            self.input_points = unproject_points(d.reshape(-1, 1, H, W), input_w2c.reshape(-1, 4, 4).to(self.device),
                                                 input_intrinsics.reshape(-1, 3, 3).to(self.device)).reshape(B, F, N, V, H, W, 3)
            self.input_depth = d
            input_depth = d
        if self.filter_points_threshold < 1.0 and input_depth is not None:
            dm = reliable_depth_mask_range_batch(input_depth.reshape(-1, 1, H, W), ratio_thresh=self.filter_points_threshold)
            dm = dm.reshape(B, F, N, V, 1, H, W).to(f32)
            self.input_mask = dm if self.input_mask is None else self.input_mask * dm
        self.shard_group = None  # torch.distributed group over which render_cache splits its item pairs (None: render everything here)
        self.boundary_mask = None
        if foreground_masking:
            dm = reliable_depth_mask_range_batch(input_depth.reshape(-1, 1, H, W))
            self.boundary_mask = (~dm).reshape(B, F, N, V, 1, H, W)

    def _render_items(self, target_w2cs, target_intrinsics, render_depth: bool, start_frame_idx: int, Ft: int, lo: int, hi: int, step: int):
        """Items [lo, hi) of the flattened (B, F_t, N) axis through g3_render_items_f32 -> (frames [m,3,H,W] or None, masks [m,1,H,W],
        depths [m,H,W] or None). Source of item (b, f, n): cache entry (b, start_frame_idx + f if the cache holds one entry per frame else 0, n)."""
        B, F, N, V, C, H, W = self.input_image.shape
        dev = self.device
        m = hi - lo
        need_depth = bool(render_depth or self.foreground_masking)
        frames = torch.empty((m, 3, H, W), dtype=f32, device=dev)
        masks = torch.empty((m, 1, H, W), dtype=f32, device=dev)
        depths = torch.empty((m, H, W), dtype=f32, device=dev) if need_depth else None
        if m == 0:
            return (None if render_depth else frames), masks, (depths if render_depth else None)
        Fs = self.input_image[:, start_frame_idx:start_frame_idx + Ft].shape[1]
        assert Fs in (1, Ft), f"the cache holds {Fs} source frames for {Ft} target frames"
        # Per-call host work that only depends on the cache / the caller's tensors is kept between calls (keyed by tensor identity + in-place
        # version): the item -> source-view indices, the uint8 boundary mask and the intrinsics' inverses. The last one matters most: the
        # inverse is taken on the host (bit-equality with the reference's CPU inverse), i.e. a device -> host copy that waits
        # for every render still in flight on the stream and leaves the GPU idle until the next launch arrives.
        memo = self.__dict__.setdefault("_items_memo", {})
        if len(memo) > 16:
            memo.clear()

        def memoised(key, refs, make):
            if any(r.is_inference() for r in refs):  # no version counter: an in-place edit would go unnoticed -> recompute
                return make()
            hit = memo.get(key)
            if hit is None or any(a is not b for a, b in zip(hit[0], refs)):
                hit = memo[key] = (refs, make())  # `refs` keeps the keyed tensors alive, so an id() cannot be recycled under the entry
            return hit[1]

        def make_index():
            b_i = torch.arange(B, device=dev).view(B, 1, 1)
            f_i = (torch.arange(Ft, device=dev) if Fs == Ft else torch.zeros(Ft, dtype=torch.long, device=dev)).view(1, Ft, 1) + start_frame_idx
            n_i = torch.arange(N, device=dev).view(1, 1, N)
            return ((b_i * F + f_i) * N + n_i).reshape(-1).to(torch.int32)[lo:hi].contiguous()

        src_index = memoised(("idx", B, F, N, Ft, Fs, start_frame_idx, lo, hi), (), make_index)
        w2cs = target_w2cs.to(dev, f32).reshape(B, Ft, 1, 16).expand(B, Ft, N, 16).reshape(-1, 16)[lo:hi].contiguous()
        n_src = B * F * N
        img_src = self.input_image.reshape(n_src, C, H, W).contiguous()  # views when the cache tensors are contiguous (they are)
        pts_src = self.input_points.reshape(n_src, H, W, 3).contiguous()
        msk_src = None if self.input_mask is None else self.input_mask.reshape(n_src, H, W).contiguous()
        ti = target_intrinsics

        def make_K():
            Ks_ = ti.to(dev, f32).reshape(B, Ft, 1, 9).expand(B, Ft, N, 9).reshape(-1, 9)[lo:hi].contiguous()
            kinv_ = _host_inverse(Ks_.reshape(-1, 3, 3)).reshape(-1, 9).contiguous() if self.foreground_masking else None
            return Ks_, kinv_

        Ks, kinv = memoised(("K", id(ti), ti.data_ptr(), _tensor_version(ti), tuple(ti.shape), N, lo, hi, bool(self.foreground_masking)), (ti,), make_K)
        bnd_src = None
        if self.foreground_masking:
            bm = self.boundary_mask
            bnd_src = memoised(("bnd", id(bm), bm.data_ptr(), _tensor_version(bm), B, F, N, V, H, W), (bm,),
                               lambda: bm.expand(B, F, N, V, 1, H, W).reshape(n_src, H, W).to(torch.uint8).contiguous())
        lib = _lib.load()

        def launch(i, j, st):
            ws = _render_workspace(lib, j - i, H, W, 2, dev, st)
            rc = lib.g3_render_items_f32(_p(pts_src, "points_src"), _p(img_src, "image_src"), _p(msk_src, "mask_src"), _p(bnd_src, "boundary_src", torch.uint8),
                                               _p(src_index[i:j], "src_index", torch.int32), _p(w2cs[i:j], "w2c"), _p(Ks[i:j], "K"),
                                               _p(kinv[i:j], "Kinv") if kinv is not None else 0, ws.data_ptr(), _p(frames[i:j], "frame"),
                                               _p(masks[i:j], "mask"), _p(depths[i:j], "depth") if depths is not None else 0, 0, j - i, n_src, H, W, 2, st)
            if rc != 0:
                _drop_render_workspace(ws)
            _lib.check(rc, "g3_render_items_f32")

        chunks = [(i, min(i + step, m)) for i in range(0, m, step)]
        if _TWO_STREAM_CHUNKS and dev.type == "cuda":
            # every chunk in two halves (whole reference pairs) on two streams: one half's VALU-bound splat next to the other's memory-bound
            # resolve pass. Each stream has its own workspace (the cache is keyed by stream); the side stream is forked from / joined into the
            # caller's, so inputs and outputs need no further care.
            main = torch.cuda.current_stream(dev)
            side = _render_side_stream(dev)
            side.wait_stream(main)
            for (i, j) in chunks:
                half = ((j - i) // 2 + 1) // 2 * 2
                if half == 0 or half == j - i:
                    launch(i, j, main.cuda_stream)
                    continue
                launch(i, i + half, main.cuda_stream)
                with torch.cuda.stream(side):
                    launch(i + half, j, side.cuda_stream)
            main.wait_stream(side)
        else:
            for (i, j) in chunks:
                launch(i, j, _stream())
        return (None if render_depth else frames), masks, (depths if render_depth else None)

    def input_frame_count(self) -> int:
        return self.input_image.shape[1]

    def update_cache(self):
        raise NotImplementedError

    @torch.no_grad()
    def render_cache(self, target_w2cs, target_intrinsics, render_depth=False, start_frame_idx=0, items_per_launch: int = 32):
        """(B,F_t,4,4), (B,F_t,3,3) -> pixels (B,F_t,N,3,H,W) [or depth (B,F_t,N,H,W)], masks (B,F_t,N,1,H,W)
        (cache_3d.py:151-236). Items are flattened as (B F N) and warped in groups of warp_chunk_size = 2."""
        bs, Ft = target_w2cs.shape[:2]
        B, F, N, V, C, H, W = self.input_image.shape
        assert bs == B
        n = B * Ft * N
        step = max(2, items_per_launch // 2 * 2)  # never split a reference pair
        # Multi-GPU: the items are independent EXCEPT that bilinear_splatting normalises its depth weights by the
        # maximum over the 2 items of one reference call - so the unit that is sharded over the ranks of `shard_group` is the PAIR
        # (2j, 2j+1) of the flattened (B F N) axis; every rank renders a contiguous run of pairs and the results are all-gathered.
        group = getattr(self, "shard_group", None)
        lo, hi, counts = 0, n, None
        if group is not None and torch.distributed.get_world_size(group) > 1:
            from .parallel import shard_range
            world, rank = torch.distributed.get_world_size(group), torch.distributed.get_rank(group)
            n_pairs = (n + 1) // 2
            ranges = [shard_range(n_pairs, r, world) for r in range(world)]
            counts = [min(2 * b, n) - min(2 * a, n) for a, b in ranges]
            lo, hi = min(2 * ranges[rank][0], n), min(2 * ranges[rank][1], n)
        frames, masks, depths = [], [], []
        if _ITEMS_CALL:
            # one C call per chunk of items (g3_render_items_f32): every item NAMES its source view instead of getting a copy of it, the outputs
            # land in one preallocated tensor, the workspace (incl. its self-cleaning accumulator) is cached per chunk size
            fr, mk, dp = self._render_items(target_w2cs, target_intrinsics, render_depth, start_frame_idx, Ft, lo, hi, step)
            frames, masks, depths = [fr], [mk], [dp]
        else:
            sl = slice(start_frame_idx, start_frame_idx + Ft)
            ex = lambda t, tail: t[:, sl].expand(B, Ft, N, V, *tail).reshape(B * Ft * N, *tail)
            imgs = ex(self.input_image, (C, H, W))
            pts = ex(self.input_points, (H, W, 3))
            msk = None if self.input_mask is None else ex(self.input_mask, (1, H, W))
            bnd = None if self.boundary_mask is None else self.boundary_mask.expand(B, Ft, N, V, 1, H, W).reshape(B * Ft * N, H, W)
            w2cs = target_w2cs.to(self.device, f32).reshape(B, Ft, 1, 4, 4).expand(B, Ft, N, 4, 4).reshape(-1, 4, 4)
            Ks = target_intrinsics.to(self.device, f32).reshape(B, Ft, 1, 3, 3).expand(B, Ft, N, 3, 3).reshape(-1, 3, 3)
            for i in range(lo, hi, step):
                s = slice(i, min(i + step, hi))
                fr, mk, dp, _ = forward_warp(imgs[s], None if msk is None else msk[s], None, None, w2cs[s], Ks[s], Ks[s],
                                             render_depth=render_depth, world_points1=pts[s],
                                             foreground_masking=self.foreground_masking,
                                             boundary_mask=None if bnd is None else bnd[s], group_size=2)
                frames.append(fr)
                masks.append(mk)
                if render_depth:
                    depths.append(dp)

        def collect(parts, tail):
            parts = [p_ for p_ in parts if p_ is not None]
            local = (parts[0] if len(parts) == 1 else torch.cat(parts)) if parts else torch.empty((0, *tail), dtype=f32, device=self.device)
            if counts is None:
                return local
            from .parallel import gather_rows
            return gather_rows(local, counts, group)

        masks = collect(masks, (1, H, W)).reshape(bs, Ft, N, 1, H, W)
        if render_depth:
            return collect(depths, (H, W)).reshape(bs, Ft, N, H, W), masks
        return collect(frames, (3, H, W)).reshape(bs, Ft, N, 3, H, W), masks


class Cache3D_Buffer(Cache3D_Base):
    """cache_3d.py:239-343: newest-first frame buffer of at most `frame_buffer_max` entries + per-buffer noise."""

    def __init__(self, frame_buffer_max=0, noise_aug_strength=0, generator=None, **kwargs):
        super().__init__(**kwargs)
        self.frame_buffer_max, self.noise_aug_strength, self.generator = frame_buffer_max, noise_aug_strength, generator

    @torch.no_grad()
    def update_cache(self, new_image, new_depth, new_w2c, new_mask=None, new_intrinsics=None, depth_alignment=True,
                     alignment_method="non_rigid"):
        new_image = new_image.to(self.device, f32)
        new_depth = torch.clamp(torch.nan_to_num(new_depth.to(self.device, f32), nan=1e4), min=0, max=1e4)
        if depth_alignment:  # cache_3d.py:256-292: agree with what the cache already shows from the new camera
            from .camera_utils import align_depth
            if alignment_method not in ("rigid", "non_rigid"):
                raise NotImplementedError(alignment_method)
            w2c_d, k_d = new_w2c.to(self.device, f32), new_intrinsics.to(self.device, f32)
            target_depth, target_mask = self.render_cache(w2c_d.unsqueeze(1), k_d.unsqueeze(1), render_depth=True)
            target_depth, target_mask = target_depth[:, :, 0].squeeze(), target_mask[:, :, 0].squeeze()
            assert target_depth.dim() == 2, "depth alignment handles one image (B = 1), as the reference's squeeze() does"
            kw = {}
            if alignment_method == "non_rigid":
                kw = dict(k=k_d.squeeze(), c2w=_host_inverse(w2c_d.squeeze()), alignment_method="non_rigid", num_iters=100,
                          lambda_arap=0.1, smoothing_kernel_size=3)
            new_depth = align_depth(new_depth.squeeze(), target_depth, target_mask.bool(), **kw).reshape_as(new_depth)
        new_points = unproject_points(new_depth, new_w2c.to(self.device, f32), new_intrinsics.to(self.device, f32))
        B, F, N, V, C, H, W = self.input_image.shape
        if self.filter_points_threshold < 1.0:
            dm = reliable_depth_mask_range_batch(new_depth.reshape(-1, 1, H, W), ratio_thresh=self.filter_points_threshold)
            dm = dm.reshape(B, 1, H, W).to(f32)
            new_mask = dm if new_mask is None else new_mask.to(self.device, f32) * dm
        if self.frame_buffer_max > 1:
            if N < self.frame_buffer_max:
                self.input_image = torch.cat([new_image[:, None, None, None], self.input_image], 2)
                self.input_points = torch.cat([new_points[:, None, None, None], self.input_points], 2)
                if self.input_mask is not None:
                    self.input_mask = torch.cat([new_mask[:, None, None, None], self.input_mask], 2)
            else:
                self.input_image[:, :, 0] = new_image[:, None, None]
                self.input_points[:, :, 0] = new_points[:, None, None]
                if self.input_mask is not None:
                    self.input_mask[:, :, 0] = new_mask[:, None, None]
        else:
            self.input_image = new_image[:, None, None, None]
            self.input_points = new_points[:, None, None, None]

    @torch.no_grad()
    def render_cache(self, target_w2cs, target_intrinsics, render_depth: bool = False, start_frame_idx: int = 0):
        assert start_frame_idx == 0, "start_frame_idx must be 0 for Cache3D_Buffer"
        out_dev = target_w2cs.device
        pixels, masks = super().render_cache(target_w2cs, target_intrinsics, render_depth)
        pixels, masks = pixels.to(out_dev), masks.to(out_dev)
        if not render_depth and self.noise_aug_strength:
            noise = torch.randn(pixels.shape, generator=self.generator, device=pixels.device, dtype=pixels.dtype)
            per_buffer = torch.arange(pixels.shape[2] - 1, -1, -1, device=pixels.device) * self.noise_aug_strength
            pixels = pixels + noise * per_buffer.reshape(1, 1, -1, 1, 1, 1)
        return pixels, masks


class Cache3D_BufferSelector(Cache3D_Base):
    """cache_3d.py:346-421: many source frames on the N axis; per target keeps the `frame_buffer_max` buffers with the
    largest mask overlap and (mask_for_max_buffer_model) blanks every buffer but the first near-full one per frame.
    The selection itself is a few reductions over per-(frame, buffer) scalars and stays in torch."""

    def __init__(self, frame_buffer_max=1, mask_for_max_buffer_model: bool = True, mask_full_threshold: float = 0.9, **kwargs):
        super().__init__(**kwargs)
        self.frame_buffer_max = max(int(frame_buffer_max), 1)
        self.mask_for_max_buffer_model = bool(mask_for_max_buffer_model)
        self.mask_full_threshold = float(mask_full_threshold)

    def update_cache(self, *args, **kwargs):
        raise NotImplementedError("Cache3D_BufferSelector does not support update_cache")

    @torch.no_grad()
    def render_cache(self, target_w2cs, target_intrinsics, render_depth: bool = False, start_frame_idx: int = 0):
        out_dev = target_w2cs.device
        pixels_all, masks_all = super().render_cache(target_w2cs, target_intrinsics, render_depth, start_frame_idx)
        B, F, N = pixels_all.shape[:3]
        if N <= self.frame_buffer_max:
            pixels_sel, masks_sel = pixels_all, masks_all
        else:
            scores = masks_all.sum(dim=(1, 3, 4, 5))  # [B, N]
            # topk(largest, sorted); ties (equal overlap counts happen: the scores are integers) go to the LOWER buffer index, which is
            # what the reference's CPU run yields - torch.topk on the GPU breaks them the other way round
            idx = torch.sort(scores, dim=1, descending=True, stable=True).indices[:, :min(self.frame_buffer_max, N)]
            pixels_sel = torch.cat([pixels_all[b:b + 1, :, idx[b]] for b in range(B)], dim=0)
            masks_sel = torch.cat([masks_all[b:b + 1, :, idx[b]] for b in range(B)], dim=0)
        if self.mask_for_max_buffer_model and not render_depth:
            m = masks_sel.mean(dim=[3, 4, 5])  # [B, F, k]
            flat = m.reshape(-1, m.shape[-1])
            near_full = flat >= self.mask_full_threshold
            has = near_full.any(dim=1)
            first = near_full.float().argmax(dim=1)
            keep = torch.zeros_like(flat)
            rows = torch.arange(flat.shape[0], device=flat.device)
            keep[rows[has], first[has]] = 1
            keep[rows[~has]] = 1
            keep = keep.reshape(m.shape)[..., None, None, None]
            pixels_sel = (pixels_sel + 1) * keep - 1
            masks_sel = masks_sel * keep
        return pixels_sel.to(out_dev), masks_sel.to(out_dev)


class Cache4D(Cache3D_Base):
    """cache_3d.py:424-433: per-target-frame sources (dynamic video input); rendering is Cache3D_Base's with
    `start_frame_idx` selecting the source window."""

    def update_cache(self, **kwargs):
        raise NotImplementedError
