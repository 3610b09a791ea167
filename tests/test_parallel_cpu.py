"""CPU (gloo, world sizes 2 and 8): the context-parallel plumbing - split/cat/broadcast helpers and the all-gather-KV
head-group schedule of ContextParallelAttention - checked against the single-process oracle attention.
The HIP kernels themselves cannot run here, so the attention/transpose callables are the oracle's (this is the one
place a `backend` is injected; the product path never does)."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_backend():
    from oracle import dit_oracle

    def transpose_v(v, S, B, H):
        return v.reshape(S, B, H, 128).permute(1, 2, 3, 0).contiguous()  # [B,H,128,S]

    def attention(q, k, vt, Sq, Skv, B, H, out, variant=0):
        q4 = q.reshape(Sq, B, H, 128)
        k4 = k.reshape(Skv, B, H, 128)
        if vt.dim() == 5:  # rank-major V^T segments [n, B, H, 128, S_local] (context-parallel gather of local transposes)
            n, S_loc = vt.shape[0], vt.shape[-1]
            v4 = vt.permute(0, 4, 1, 2, 3).reshape(n * S_loc, B, H, 128)
        else:
            v4 = vt.permute(3, 0, 1, 2)  # [S,B,H,128]
        out.copy_(dit_oracle.attention_sbhd(q4, k4, v4).reshape(Sq * B, H * 128))
        return out

    def attention_partial(q, k, vt, Sq, Skv, B, H, variant=0):
        # normalised partial result + log2-domain log-sum-exp, like g3_flash_attn_fwd_ex_bf16 (fp64 reference arithmetic)
        q4 = q.reshape(Sq, B, H, 128).permute(1, 2, 0, 3).double()
        k4 = k.reshape(Skv, B, H, 128).permute(1, 2, 0, 3).double()
        n, S_loc = vt.shape[0], vt.shape[-1]
        v4 = vt.permute(0, 4, 1, 2, 3).reshape(n * S_loc, B, H, 128).permute(1, 2, 0, 3).double()
        sc = q4 @ k4.transpose(-1, -2) / math.sqrt(128.0)
        lse = torch.logsumexp(sc, dim=-1)  # [B,H,Sq] natural log
        o = torch.softmax(sc, dim=-1) @ v4  # [B,H,Sq,128]
        return o.permute(2, 0, 1, 3).reshape(Sq * B, H * 128).float(), (lse / math.log(2.0)).float().contiguous()

    def merge(parts, Sq, B, H, out):
        l = torch.stack([p[1].double() for p in parts])  # [n,B,H,Sq]
        w = torch.softmax(l * math.log(2.0), dim=0)
        acc = 0
        for i, (o, _) in enumerate(parts):
            acc = acc + o.double().reshape(Sq, B, H, 128) * w[i].permute(2, 0, 1)[..., None]
        out.copy_(acc.reshape(Sq * B, H * 128).to(out.dtype))
        return out

    return dict(pack=lambda t: t.contiguous(), transpose_v=transpose_v, attention=attention, attention_partial=attention_partial, merge=merge)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gen3c_amd import parallel
    from oracle import dit_oracle
    parallel.init_distributed("gloo")
    parallel.parallel_state.initialize_model_parallel(context_parallel_size=world)
    group = parallel.parallel_state.get_context_parallel_group()
    try:
        # --- split / cat round trip along T
        g = torch.Generator().manual_seed(0)
        x = torch.randn(1, 4, 3 * world, 3, 5, generator=g)
        xs = parallel.split_inputs_cp(x, 2, group)
        assert xs.shape == (1, 4, 3, 3, 5) and torch.equal(xs, x[:, :, rank * 3:(rank + 1) * 3])
        assert torch.equal(parallel.cat_outputs_cp(xs, 2, group), x)
        # --- broadcast: tensors (shape known only on src) and python objects
        t = torch.arange(10.0).reshape(2, 5) if rank == 0 else torch.zeros(1, 1)
        assert torch.equal(parallel.broadcast(t), torch.arange(10.0).reshape(2, 5))
        assert parallel.broadcast("hello" if rank == 0 else "x") == "hello"
        assert parallel.broadcast(None) is None
        # --- context-parallel attention == full attention on the gathered sequence
        # S_local = 24: V gathered row-major, transposed after the exchange; S_local = 64: V^T shards gathered as key segments
        for S, B, H in ((24 * world, 2, 4), (64 * world, 1, 4)):
            Sl = S // world
            q = torch.randn(S * B, H * 128, generator=g)
            k = torch.randn(S * B, H * 128, generator=g)
            v = torch.randn(S * B, H * 128, generator=g)
            ref = dit_oracle.attention_sbhd(q.reshape(S, B, H, 128), k.reshape(S, B, H, 128), v.reshape(S, B, H, 128))
            ref = ref.reshape(S * B, H * 128)
            rows = slice(rank * Sl * B, (rank + 1) * Sl * B)
            qkv_local = torch.cat([q[rows], k[rows], v[rows]], dim=1)  # v passed as a strided column view, like the DiT does
            D = H * 128
            cpa = parallel.ContextParallelAttention(group, head_groups=3, backend=_oracle_backend())  # 3 !| 4 -> falls back to 2
            pending = cpa.start(qkv_local[:, D:2 * D], qkv_local[:, 2 * D:], Sl, B, H)
            assert pending["segmented"] == (Sl % 64 == 0)
            out = cpa.finish(qkv_local[:, :D], pending)
            torch.testing.assert_close(out, ref[rows], rtol=1e-5, atol=1e-5)
            # local-KV-first schedule: own shard first (no wait), remote segments after the exchange, merged (falls back to gather-first when
            # V cannot be gathered as key segments, S_local % 64 != 0)
            cpl = parallel.ContextParallelAttention(group, head_groups=2, backend=_oracle_backend(), schedule="local_first")
            b0 = cpl.bytes_gathered
            out2 = cpl(qkv_local[:, :D], qkv_local[:, D:2 * D], qkv_local[:, 2 * D:], Sl, B, H)
            torch.testing.assert_close(out2, ref[rows], rtol=1e-5, atol=1e-5)
            assert cpl.bytes_gathered - b0 == (world - 1) * 2 * Sl * B * D * 4  # fp32 K + V shards of the other ranks
        with open(os.path.join(tmp, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])  # 8 = the driver's multi-GPU run: interior ranks have key segments before AND after their own
def test_context_parallel_world(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_parallel_state_defaults():
    from gen3c_amd.parallel import broadcast, parallel_state
    assert not parallel_state.is_initialized()
    assert parallel_state.get_context_parallel_world_size() == 1
    x = torch.ones(3)
    assert broadcast(x) is x  # no-op when model parallel is not initialised (module/parallel.py:102-103)
