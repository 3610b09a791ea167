"""Context parallelism for the DiT over RCCL / xGMI (one process per GPU, torch.distributed backend "nccl" = RCCL).

Mirrors the reference surface
  * split_inputs_cp / cat_outputs_cp / broadcast   (cosmos_predict1/diffusion/module/parallel.py:25-163)
  * distributed.init()                             (cosmos_predict1/utils/distributed.py:49-79)
  * the slice of megatron-core `parallel_state` the GEN3C CLIs touch (gen3c_single_image.py:248-255, 480-484)
but NOT its communication pattern. The reference lets TransformerEngine run a P2P *ring* over the CP ranks
(general_dit.py:540-541), which on xGMI's point-to-point mesh is bound by a single ~153 GB/s link. Here every rank's
K / V shard is exchanged with a direct all-gather (all 7 links busy at once), split into head groups so that the
attention kernel of group g runs on the compute stream while RCCL is still gathering group g+1.

Token layout: rows are (s, b) pairs with b fastest and the latent frames sharded contiguously over ranks, so a
rank-major all-gather of row blocks IS the global (s, b) order - no re-sort after the collective.
"""
from __future__ import annotations

import os
from datetime import timedelta
from typing import Callable, List, Optional

import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------------------------------------
# reference-compatible helpers
# ------------------------------------------------------------------------------------------------------------------
def split_inputs_cp(x: torch.Tensor, seq_dim: int, cp_group) -> torch.Tensor:
    """This rank's contiguous slice of `x` along `seq_dim` (parallel.py:25-53)."""
    cp_size = dist.get_world_size(cp_group)
    rank = dist.get_rank(cp_group)
    n = x.shape[seq_dim]
    assert n % cp_size == 0, f"{n} cannot divide cp_size {cp_size}"
    step = n // cp_size
    return x.narrow(seq_dim, rank * step, step).contiguous()


def cat_outputs_cp(x: torch.Tensor, seq_dim: int, cp_group) -> torch.Tensor:
    """All-gather the per-rank slices and concatenate them along `seq_dim` (parallel.py:56-87)."""
    world = dist.get_world_size(cp_group)
    parts = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(parts, x.contiguous(), group=cp_group)
    return torch.cat(parts, dim=seq_dim)


def broadcast(item, to_tp: bool = True, to_cp: bool = True):
    """Broadcast a tensor / picklable object from the lowest rank of the CP group (parallel.py:90-163).
    There is no tensor parallelism on this path (Attention.tp_size = 1, attention.py:205): `to_tp` is accepted
    and ignored."""
    if not parallel_state.is_initialized():
        return item
    group = parallel_state.get_context_parallel_group()
    if not (to_cp and dist.get_world_size(group) > 1):
        return item
    src = min(dist.get_process_group_ranks(group))
    if isinstance(item, torch.Tensor):
        dev = item.device
        if dist.get_rank() == src:
            shape = torch.tensor(item.shape, dtype=torch.long, device=dev)
        else:
            shape = torch.empty(item.dim(), dtype=torch.long, device=dev)
        dist.broadcast(shape, src, group=group)
        if dist.get_rank() != src:
            item = item.new_empty(shape.tolist())
        item = item.contiguous()
        dist.broadcast(item, src, group=group)
        return item
    if item is not None:
        box = [item]
        dist.broadcast_object_list(box, src, group=group)
        return box[0]
    return item


class _ParallelState:
    """The part of megatron.core.parallel_state that gen3c_*.py use, for a pure context-parallel job."""

    def __init__(self):
        self._cp_group = None

    def initialize_model_parallel(self, context_parallel_size: int = 1, **_unused):
        world = dist.get_world_size()
        assert world % context_parallel_size == 0, f"world size {world} not divisible by cp {context_parallel_size}"
        my_group = None
        for start in range(0, world, context_parallel_size):
            ranks = list(range(start, start + context_parallel_size))
            g = dist.new_group(ranks)
            if dist.get_rank() in ranks:
                my_group = g
        self._cp_group = my_group

    def is_initialized(self) -> bool:
        return self._cp_group is not None

    def get_context_parallel_group(self):
        assert self._cp_group is not None, "context parallel group is not initialized"
        return self._cp_group

    def get_context_parallel_world_size(self) -> int:
        return dist.get_world_size(self._cp_group) if self._cp_group is not None else 1

    def get_context_parallel_rank(self) -> int:
        return dist.get_rank(self._cp_group) if self._cp_group is not None else 0

    def get_tensor_model_parallel_group(self):
        return None

    def get_tensor_model_parallel_world_size(self) -> int:
        return 1

    def destroy_model_parallel(self):
        self._cp_group = None


parallel_state = _ParallelState()


def init_distributed(backend: Optional[str] = None, timeout_s: Optional[int] = None) -> int:
    """Counterpart of cosmos_predict1.utils.distributed.init (utils/distributed.py:49-79) without the NVIDIA-only
    parts (pynvml affinity, libcudart cudaDeviceSetLimit). One process per GPU; rendezvous via env:// as set by
    torchrun. Returns the local device index."""
    local_rank = int(os.getenv("LOCAL_RANK", 0))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (required by RCCL on this driver)
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        # timeout_s: bench.py passes a short collective timeout - a hung exchange must end the run inside the driver's budget, not after 30 min
        timeout = timedelta(seconds=int(timeout_s if timeout_s is not None else os.getenv("TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC", 1800)))
        dist.init_process_group(backend=backend, init_method="env://", timeout=timeout)
    return local_rank


# ------------------------------------------------------------------------------------------------------------------
# sharding the replicated stages of a chunk: render items and tokenizer encodes are independent units
# ------------------------------------------------------------------------------------------------------------------
def shard_range(n_units: int, rank: int, world: int):
    """Contiguous, balanced split of `n_units` over `world` ranks -> [lo, hi) of `rank` (the first n % world ranks get one more)."""
    q, r = divmod(n_units, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_rows(local: torch.Tensor, counts: List[int], group) -> torch.Tensor:
    """All-gather of per-rank row blocks of different lengths (`counts[r]` rows on rank r, identical trailing shape): every rank pads
    its block to max(counts), one all_gather_into_tensor, the padding is cut out. Rank-major = the original unit order."""
    world = dist.get_world_size(group)
    assert len(counts) == world and local.shape[0] == counts[dist.get_rank(group)]
    m = max(counts)
    pad = local if local.shape[0] == m else torch.cat([local, local.new_zeros((m - local.shape[0],) + tuple(local.shape[1:]))], 0)
    full = torch.empty((world * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, pad.contiguous(), group=group)
    if all(c == m for c in counts):
        return full
    return torch.cat([full[r * m:r * m + c] for r, c in enumerate(counts)], 0)


def run_jobs_round_robin(jobs: List[Callable[[], torch.Tensor]], group, shape, dtype, device) -> List[torch.Tensor]:
    """Independent jobs whose results all have `shape` / `dtype` (the 2N+1 tokenizer encodes of a chunk): job j runs on rank
    j % world only, the results are all-gathered and returned in job order on every rank. group=None (or one rank): plain sequential
    execution. A rank without a job (more ranks than jobs) contributes zeros that nobody reads."""
    world = 1 if group is None else dist.get_world_size(group)
    if world == 1:
        return [j() for j in jobs]
    rank = dist.get_rank(group)
    slots = (len(jobs) + world - 1) // world
    local = torch.zeros((slots, *shape), dtype=dtype, device=device)
    for i, j in enumerate(range(rank, len(jobs), world)):
        out = jobs[j]()
        assert tuple(out.shape) == tuple(shape) and out.dtype == dtype, f"job {j}: {tuple(out.shape)} {out.dtype} vs declared {tuple(shape)} {dtype}"
        local[i] = out
    full = torch.empty((world * slots, *shape), dtype=dtype, device=device)
    dist.all_gather_into_tensor(full, local, group=group)
    return [full[(j % world) * slots + j // world] for j in range(len(jobs))]


# ------------------------------------------------------------------------------------------------------------------
# context-parallel self-attention
# ------------------------------------------------------------------------------------------------------------------
ATTN_KERNELS = {"auto": None, "w4b": 11, "wave8": 4}  # per-call kernel choice of g3_flash_attn_fwd_ex_bf16 ("auto": the even-fill rule below)
CP_SCHEDULES = ("gather_first", "local_first")


def _default_backend():
    from . import ops
    return dict(
        pack=lambda t: t.contiguous(),
        transpose_v=lambda v, S, B, H: ops.transpose_v(v, S, B, H),
        attention=lambda q, k, vt, Sq, Skv, B, H, out, variant=0: ops.flash_attn(q, k, vt, Sq, Skv, B, H, out=out, variant=variant),
        attention_partial=lambda q, k, vt, Sq, Skv, B, H, variant=0: ops.flash_attn(q, k, vt, Sq, Skv, B, H, variant=variant, partial=True),
        merge=lambda parts, Sq, B, H, out: ops.attn_merge(parts, Sq, B, H, out=out),
        timer=ops.HipTimer,
    )


class ContextParallelAttention:
    """softmax(Q_local K_all^T) V_all for a token-sharded sequence.

    q, k, v: [S_local*B, H*128] (v may be a strided column view). K and V shards are all-gathered head-group by
    head-group (async, RCCL stream) and consumed by the attention kernel in the same order.

    schedule "gather_first": one attention launch per head group over the gathered keys; only the first group's exchange is exposed (behind
    the Q projection), the rest hides under the attention of earlier groups.
    schedule "local_first" (what TransformerEngine's ring does behind attn_op.set_context_parallel_group, general_dit.py:540-541: start on the
    local shard at once): every head group first runs over THIS rank's K / V shard - no collective is waited for - and returns a normalised
    fp32 partial + log-sum-exp; the remote keys (ranks before / after this one in the gathered buffers) follow as the groups' exchanges land,
    and g3_attn_merge_partials_bf16 combines the parts. Costs one fp32 round trip of the group's output per part; hides the first exchange.

    kernel: "auto" (the one-wave-per-SIMD kernel when the groups' workgroups together fill the 256 CUs evenly, else the 8-wave kernel), "w4b",
    "wave8" - passed PER CALL through the C ABI; no process-wide option is touched (launches go out on two streams).

    `backend` (dict of callables) exists so the sharding + collective schedule can be exercised on CPU with gloo in tests; the product path
    always uses the HIP kernels. `stats` (set to a list to enable): one (kind, head group, HipTimer) per collective wait - the time the launch
    stream stalled for that exchange - read by bench.py.
    """

    def __init__(self, cp_group, head_groups: int = 4, backend: Optional[dict] = None, schedule: str = "gather_first", kernel: str = "auto"):
        assert schedule in CP_SCHEDULES and kernel in ATTN_KERNELS
        self.group = cp_group
        self.world = dist.get_world_size(cp_group)
        self.rank = dist.get_rank(cp_group)
        self.head_groups = head_groups
        self.backend = backend
        self.schedule = schedule
        self.kernel = kernel
        self.stats = None
        self.effective = None  # set by finish(): dict(schedule, kernel, head_groups) of the configuration that actually ran
        self.bytes_gathered = 0  # received from the other ranks since construction / the last reset (bench.py's "cp" object)

    def configure(self, head_groups: Optional[int] = None, schedule: Optional[str] = None, kernel: Optional[str] = None):
        if head_groups is not None:
            self.head_groups = int(head_groups)
        if schedule is not None:
            assert schedule in CP_SCHEDULES
            self.schedule = schedule
        if kernel is not None:
            assert kernel in ATTN_KERNELS
            self.kernel = kernel
        return self

    def start(self, k: torch.Tensor, v: torch.Tensor, S_local: int, B: int, H: int):
        """Issue the K / V all-gathers of every head group (async). Call as soon as K and V exist: whatever the caller
        launches before finish() - the Q projection and its norm/RoPE in the DiT - hides part of the first group's exchange.

        V is transposed LOCALLY (this rank's S_local keys) and the V^T shards are gathered rank-major; the attention kernel reads
        them as key segments, so no rank re-transposes the full-length V (needs S_local % 64 == 0, true for the 3 520-token
        latent frames; otherwise V is gathered row-major and transposed after the exchange)."""
        be = self.backend or _default_backend()
        G = self.head_groups
        while H % G != 0:
            G -= 1
        Hg = H // G
        W = Hg * 128
        rows = S_local * B
        segmented = S_local % 64 == 0
        works = []
        for g in range(G):
            ks = be["pack"](k[:, g * W:(g + 1) * W])
            kf = torch.empty((self.world * rows, W), dtype=k.dtype, device=k.device)
            wk = dist.all_gather_into_tensor(kf, ks, group=self.group, async_op=True)
            if segmented:
                vs = be["transpose_v"](v[:, g * W:(g + 1) * W], S_local, B, Hg)  # [B, Hg, 128, S_local]
                vf = torch.empty((self.world * vs.shape[0],) + tuple(vs.shape[1:]), dtype=v.dtype, device=v.device)  # dim-0 concat
            else:
                vs = be["pack"](v[:, g * W:(g + 1) * W])
                vf = torch.empty((self.world * rows, W), dtype=v.dtype, device=v.device)
            wv = dist.all_gather_into_tensor(vf, vs, group=self.group, async_op=True)
            self.bytes_gathered += (self.world - 1) * (ks.numel() * ks.element_size() + vs.numel() * vs.element_size())
            works.append((wk, wv, kf, vf, ks, vs))
        return dict(works=works, S_local=S_local, B=B, H=H, Hg=Hg, W=W, rows=rows, be=be, segmented=segmented)

    def _variant(self, S_local: int, S_all: int, B: int, H: int) -> int:
        """Kernel for every launch of this layer. One launch per head group covers only H / G heads - 0.9 to 3.4 rounds of the 256 CUs at
        cp = 8..2 - but the groups run back to back on two streams, so the chip sees their SUM: the one-wave-per-SIMD kernel's even-fill rule
        (attention.hip: attn_resolve_variant) is applied to all H heads."""
        forced = ATTN_KERNELS[self.kernel]
        if forced is not None:
            return forced if S_all % 64 == 0 else 4
        total_wg = ((S_local + 255) // 256) * H * B
        rounds = (total_wg + 255) // 256
        return 11 if (S_all > 2048 and S_all % 64 == 0 and total_wg * 100 >= rounds * 256 * 93) else 4

    def _wait(self, be, g: int, *works):
        """Work.wait() makes the CURRENT stream wait for the collective (torch.distributed semantics on the NCCL / RCCL backend). With stats
        enabled an event pair brackets it: elapsed = how long this stream had nothing to run because the exchange had not landed yet."""
        tm = None
        if self.stats is not None and "timer" in be:
            tm = be["timer"]()
            tm.start()
        for w in works:
            w.wait()
        if tm is not None:
            tm.stop()
            self.stats.append(("wait", g, tm))

    def finish(self, q: torch.Tensor, pending: dict) -> torch.Tensor:
        be, W, Hg, B, S_local = pending["be"], pending["W"], pending["Hg"], pending["B"], pending["S_local"]
        S_all = S_local * self.world
        out = torch.empty((pending["rows"], pending["H"] * 128), dtype=q.dtype, device=q.device)
        # Lifetime contract: the gathered buffers (kf, vf) and the packed send buffers (_ks, _vs) are consumed by RCCL on its own stream
        # and by HIP kernels launched through ctypes, which the caching allocator knows nothing about. They stay referenced by `pending`
        # (held by the caller's frame) until every attention launch below is enqueued on the compute stream; after that, stream order on
        # the compute stream protects them (a later allocation that reuses the memory is also enqueued there). Do not drop `pending`
        # entries inside this loop.
        assert all(len(w) == 6 and w[2] is not None and w[3] is not None for w in pending["works"]), "gathered K / V buffers must stay referenced"
        works = pending["works"]
        variant = self._variant(S_local, S_all, B, pending["H"]) if q.is_cuda else 0
        local_first = self.schedule == "local_first" and pending["segmented"] and self.world > 1 and "attention_partial" in be
        # what this layer actually runs (a requested "local_first" needs segmented V^T and a split-KV backend; a forced one-wave kernel needs
        # S_all % 64 == 0): read by bench.py so that its autotune table and `cp.chosen` never label a fallback with the requested name
        self.effective = dict(schedule="local_first" if local_first else "gather_first", kernel={11: "w4b", 4: "wave8"}.get(variant, str(variant)),
                              head_groups=len(pending["works"]))
        two_streams = q.is_cuda and len(works) >= 2
        main = side = q_ready = None
        if two_streams:
            # The groups are independent, so odd groups go to a second stream: the next group's workgroups fill the CUs the previous launch has drained.
            main = torch.cuda.current_stream(q.device)
            side = self._side_stream(q.device)
            q_ready = main.record_event()

        def on_stream(g):
            st = main if (not two_streams or g % 2 == 0) else side
            ctx = torch.cuda.stream(st) if two_streams else _NullCtx()
            return st, ctx

        try:
            local_parts = {}
            if local_first:
                # phase 1: every group over this rank's own shard - the packed K columns and the local V^T that were SENT (w[4], w[5]); no wait
                for g, (wk, wv, kf, vf, ks, vs) in enumerate(works):
                    st, ctx = on_stream(g)
                    with ctx:
                        if two_streams and st is side:
                            st.wait_event(q_ready)  # q (and `out`) were produced / allocated on the main stream
                        local_parts[g] = be["attention_partial"](q[:, g * W:(g + 1) * W], ks, vs.reshape(1, B, Hg, 128, -1), S_local, S_local, B, Hg, variant=variant)
            for g, (wk, wv, kf, vf, _ks, _vs) in enumerate(works):
                st, ctx = on_stream(g)
                with ctx:
                    if two_streams and st is side and not local_first:
                        st.wait_event(q_ready)
                    self._wait(be, g, wk, wv)
                    qg, og = q[:, g * W:(g + 1) * W], out[:, g * W:(g + 1) * W]
                    if local_first:
                        # phase 2: the other ranks' keys = the row blocks / V^T segments before and after this rank's in the gathered buffers
                        parts = [local_parts[g]]
                        vseg = vf.view(self.world, B, Hg, 128, -1)
                        rows = pending["rows"]
                        for (r0, r1) in ((0, self.rank), (self.rank + 1, self.world)):
                            if r1 > r0:
                                parts.append(be["attention_partial"](qg, kf[r0 * rows:r1 * rows], vseg[r0:r1], S_local, (r1 - r0) * S_local, B, Hg, variant=variant))
                        be["merge"](parts, S_local, B, Hg, og)
                    else:
                        vt = vf.view(self.world, B, Hg, 128, -1) if pending["segmented"] else be["transpose_v"](vf, S_all, B, Hg)
                        be["attention"](qg, kf, vt, S_local, S_all, B, Hg, og, variant=variant)
        finally:
            if two_streams:
                main.wait_stream(side)  # everything enqueued on the main stream from here on (out-projection, buffer reuse) follows both streams
        return out

    def _side_stream(self, device):
        st = getattr(self, "_side", None)
        if st is None or st.device != device:
            st = self._side = torch.cuda.Stream(device=device)
        return st

    def __call__(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, S_local: int, B: int, H: int) -> torch.Tensor:
        return self.finish(q, self.start(k, v, S_local, B, H))


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
