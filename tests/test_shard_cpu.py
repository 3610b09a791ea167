"""CPU (gloo, world sizes 2, 3 and 8): the sharding of a chunk's replicated stages over the context-parallel ranks -
render item PAIRS (cache_3d.py:163-183: the depth weights of bilinear_splatting are normalised per 2-item call, so a pair must stay
on one rank) and the 2N+1 independent tokenizer encodes - must give bit-identical results to the unsharded path on every rank.
The HIP kernels cannot run here: the renderer's forward_warp is replaced by the numpy oracle's (the one place a stand-in is injected;
the product path never does), the encodes by a deterministic stand-in."""
import os
import socket

import numpy as np
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


def _oracle_forward_warp(frame1, mask1, depth1, transformation1, transformation2, intrinsic1, intrinsic2, render_depth=False, world_points1=None,
                         foreground_masking=False, boundary_mask=None, group_size=2):
    from oracle import warp_oracle
    n = frame1.shape[0]
    outs = []
    for i in range(0, n, group_size):  # the reference's 2-item calls
        s = slice(i, i + group_size)
        outs.append(warp_oracle.forward_warp(frame1[s].numpy(), None if mask1 is None else mask1[s].numpy(), world_points1[s].numpy(),
                                             transformation2[s].numpy(), intrinsic2[s].numpy(), render_depth=render_depth))
    cat = lambda j: torch.from_numpy(np.concatenate([o[j] for o in outs]))
    return cat(0), cat(1), (cat(2) if render_depth else None), cat(3)


def _scene(n_src, h=24, w=32):
    from oracle import warp_oracle
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    imgs, pts = [], []
    K = np.array([[30.0, 0, w / 2], [0, 30.0, h / 2], [0, 0, 1]], np.float32)
    for i in range(n_src):
        depth = (2.0 + 0.5 * i + 0.02 * xs + 0.01 * ys).astype(np.float32)  # different depth ranges: the per-pair max matters
        depth = np.where((ys - 10) ** 2 + (xs - 12 - 3 * i) ** 2 < 25, 1.0 + 0.3 * i, depth).astype(np.float32)
        imgs.append(np.stack([np.sin(xs * 0.3 + c + i) * np.cos(ys * 0.2 - c) for c in range(3)], 0).astype(np.float32))
        pts.append(warp_oracle.unproject_points(depth[None, None], np.eye(4, dtype=np.float32)[None], K[None])[0])
    return np.stack(imgs), np.stack(pts), K


def _render(n_src, n_frames, group, render_depth=False):
    from gen3c_amd import renderer
    imgs, pts, K = _scene(n_src)
    c = renderer.Cache3D_Base(input_image=torch.from_numpy(imgs)[None], input_depth=None, input_w2c=None, input_intrinsics=None,
                              input_points=torch.from_numpy(pts)[None], input_format=["B", "N", "C", "H", "W"], device="cpu")
    c.shard_group = group
    w2cs = torch.eye(4).repeat(1, n_frames, 1, 1)
    w2cs[0, :, 0, 3] = torch.linspace(0.0, 0.4, n_frames)
    Ks = torch.from_numpy(K).repeat(1, n_frames, 1, 1)
    return c.render_cache(w2cs, Ks, render_depth=render_depth)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gen3c_amd import parallel, renderer
    parallel.init_distributed("gloo")
    parallel.parallel_state.initialize_model_parallel(context_parallel_size=world)
    group = parallel.parallel_state.get_context_parallel_group()
    renderer.forward_warp = _oracle_forward_warp
    renderer._ITEMS_CALL = False  # the expand-and-loop form calls forward_warp (the stand-in); the item ranges / pair sharding / gather are shared with the
    # one-call form (g3_render_items_f32), whose sharded == replicated check runs with the real kernels in tools/cp_check.py (tests/test_cp_gpu.py)
    try:
        # --- unit ranges cover everything exactly once, contiguous, balanced
        for n in (0, 1, 5, 121, 242):
            rs = [parallel.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1
        # --- renders: N=1 (pairs = consecutive frames, odd item count), N=2 (pairs = both buffers of a frame), depth rendering,
        #     and a single item (the depth-alignment render of update_cache): sharded == unsharded bit for bit on every rank
        for n_src, n_frames, depth in ((1, 7, False), (2, 5, False), (2, 3, True), (1, 1, True)):
            full = _render(n_src, n_frames, None, depth)
            shard = _render(n_src, n_frames, group, depth)
            for a, b in zip(full, shard):
                assert a.shape == b.shape and torch.equal(a, b), (n_src, n_frames, depth)
        # wrong grouping WOULD be visible: rendering buffer 0 (the nearer one) of the N=2 scene on its own changes its colours (per-call log-depth max)
        both = _render(2, 1, None)[0][0, 0, 0]
        imgs, pts, K = _scene(2)
        alone = renderer.Cache3D_Base(input_image=torch.from_numpy(imgs[:1])[None], input_depth=None, input_w2c=None, input_intrinsics=None,
                                      input_points=torch.from_numpy(pts[:1])[None], input_format=["B", "N", "C", "H", "W"], device="cpu")
        px = alone.render_cache(torch.eye(4).repeat(1, 1, 1, 1), torch.from_numpy(K).repeat(1, 1, 1, 1))[0][0, 0, 0]
        assert not torch.equal(px, both)
        # --- the 2N+1 encodes: job j on rank j % world, gathered in job order; also with fewer jobs than ranks
        ran = []
        for n_jobs in (5, 3, 1):
            jobs = [(lambda j=j: (ran.append(j), torch.full((1, 2, 3), float(j + 1)).to(torch.bfloat16))[1]) for j in range(n_jobs)]
            ran.clear()
            outs = parallel.run_jobs_round_robin(jobs, group, (1, 2, 3), torch.bfloat16, torch.device("cpu"))
            assert ran == list(range(rank, n_jobs, world)), (ran, rank)
            assert len(outs) == n_jobs and all(float(o.float().mean()) == j + 1 for j, o in enumerate(outs))
        assert [float(t.mean()) for t in parallel.run_jobs_round_robin([lambda: torch.ones(2), lambda: torch.zeros(2)], None, (2,), torch.float32, "cpu")] == [1.0, 0.0]
        with open(os.path.join(tmp, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])  # 8: more ranks than tokenizer encodes / render pairs of a small chunk (ranks without a unit)
def test_sharded_render_and_encodes(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
