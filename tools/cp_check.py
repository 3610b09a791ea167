"""Context-parallel end-to-end check with the REAL HIP kernels on ONE GPU: N ranks (torchrun) share cuda:0 and talk over
gloo (RCCL refuses several ranks on one device); each rank compares the CP denoise step on its frame shard against
the same step computed without CP on the full latent. Exercises dit.enable_context_parallel, table slicing,
ContextParallelAttention (all-gather-KV in head groups) and the sampler's CP splits exactly as bench.py --gpus N does, then the
sharded render pairs / tokenizer encodes of a chunk (bit-identical to the replicated computation).

  torchrun --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/cp_check.py
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd.dit import VideoExtendGeneralDIT  # noqa: E402
from gen3c_amd.parallel import init_distributed, parallel_state, split_inputs_cp  # noqa: E402
from gen3c_amd.sampler import Gen3CDenoiser, VideoExtendCondition, add_condition_video_indicator_and_video_input_mask  # noqa: E402


def main():
    torch.cuda.set_device(0)
    backend = os.environ.get("G3_CP_CHECK_BACKEND", "gloo")  # "nccl" works with ONE rank only on a 1-GPU box (API-path smoke)
    init_distributed(backend)
    world, rank = dist.get_world_size(), dist.get_rank()
    parallel_state.initialize_model_parallel(context_parallel_size=world)
    dev = torch.device("cuda:0")
    net = VideoExtendGeneralDIT(max_img_h=64, max_img_w=64, max_frames=32, in_channels=81, model_channels=512, num_blocks=2, num_heads=4,
                                adaln_lora_dim=64, crossattn_emb_channels=256, rope_t_extrapolation_ratio=2.0, device=dev, init_weights=False)
    net.initialize_weights(randomize_adaln=True, seed=11)
    B, T, H, W, M = 1, 4 * world, 16, 24, 64
    rs = np.random.RandomState(3)
    nrm = lambda shape, std: torch.from_numpy((rs.standard_normal(shape) * std).astype(np.float32)).to(torch.bfloat16).to(dev)
    den = Gen3CDenoiser(net, state_shape=(16, T, H, W))
    den.scheduler.set_timesteps(35)
    xt = nrm((B, 16, T, H, W), den.scheduler.init_noise_sigma)
    gt, pose, ctx = nrm((B, 16, T, H, W), 0.5), nrm((B, 64, T, H, W), 0.5), nrm((B, M, 256), 0.2)
    pad = torch.zeros(B, 1, 8 * H, 8 * W, device=dev, dtype=torch.bfloat16)

    def cond(p):
        c = VideoExtendCondition(crossattn_emb=ctx, padding_mask=pad, fps=torch.tensor([24.0], device=dev), video_cond_bool=True,
                                 condition_video_pose=p)
        return add_condition_video_indicator_and_video_input_mask(gt, c, 1)

    c, u = cond(pose), cond(torch.zeros_like(pose))
    full = den.denoise_step(xt, 5, c, u, 1.0, 0.001, 1)  # no CP: every rank computes the whole thing
    net.enable_context_parallel(parallel_state.get_context_parallel_group())
    ref = split_inputs_cp(full, 2, net.cp_group).float()
    good = True
    # both collective schedules of ContextParallelAttention, 2 head groups (alternating streams): "gather_first" = one launch per head group over
    # the gathered keys (same arithmetic per row as the non-CP call); "local_first" = own shard first, remote segments after the exchange,
    # fp32 partials merged (one extra rounding pattern: not bitwise, same tolerance)
    for sched in ("gather_first", "local_first"):
        net._cp_attn.configure(head_groups=2, schedule=sched)
        part = den.denoise_step(split_inputs_cp(xt, 2, net.cp_group), 5, c, u, 1.0, 0.001, 1)
        torch.cuda.synchronize()
        rel = float((part.float() - ref).norm() / ref.norm())
        mx = float((part.float() - ref).abs().max())
        print(f"[cp_check] rank {rank}/{world}: CP ({sched}) vs non-CP denoise step rel_l2={rel:.3e} max_abs={mx:.3e}", flush=True)
        good = good and rel < 5e-3 and np.isfinite(rel)
    net._cp_attn.configure(head_groups=4, schedule="gather_first")

    # ---- the chunk's other stages, sharded over the same group: render item pairs and tokenizer encodes with the real
    # kernels must be bit-identical to the replicated computation on every rank
    from gen3c_amd import renderer
    from gen3c_amd.pipeline import DiffusionGen3CModel
    from gen3c_amd.tokenizer import VideoTokenizer
    group = net.cp_group
    hh, ww, n_frames = 64, 96, 9
    ys, xs = np.mgrid[0:hh, 0:ww].astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    K = t(np.array([[80.0, 0, ww / 2], [0, 80.0, hh / 2], [0, 0, 1]], np.float32))
    for n_buf, fg in ((1, False), (2, True)):
        depth = (3.0 + 0.01 * xs + 0.004 * ys).astype(np.float32)
        depth = np.where((ys - 30) ** 2 + (xs - 40) ** 2 < 15 ** 2, 1.5 + 0.002 * xs, depth).astype(np.float32)
        img = np.stack([np.sin(xs * 0.2 + c) * np.cos(ys * 0.15 - c) for c in range(3)], 0).astype(np.float32)
        cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=t(img)[None], input_depth=t(depth)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                        input_intrinsics=K[None], filter_points_threshold=0.05, foreground_masking=fg, input_format=["B", "C", "H", "W"])
        if n_buf == 2:
            w2 = torch.eye(4, device=dev)
            w2[0, 3] = -0.2
            cache.update_cache(t(img[::-1].copy())[None], t(depth * 1.1)[None, None], w2[None], new_intrinsics=K[None], depth_alignment=False)
        w2cs = torch.eye(4, device=dev).repeat(1, n_frames, 1, 1)
        w2cs[0, :, 0, 3] = torch.linspace(0.0, 0.3, n_frames, device=dev)
        Ks = K.repeat(1, n_frames, 1, 1)
        full_r = cache.render_cache(w2cs, Ks)
        cache.shard_group = group
        shard_r = cache.render_cache(w2cs, Ks)
        # masks (hence every splat index / occlusion decision) bit-identical; colours are sums of fp32 atomics whose order is not
        # reproducible between two launches even on one GPU (the reference's index_put_(accumulate=True) has the same property)
        same = torch.equal(full_r[1], shard_r[1]) and torch.allclose(full_r[0], shard_r[0], rtol=1e-4, atol=1e-5)
        print(f"[cp_check] rank {rank}: sharded render (N={n_buf}, foreground_masking={fg}) == replicated: {same} "
              f"(max colour diff {float((full_r[0] - shard_r[0]).abs().max()):.2e})", flush=True)
        good = good and same
    tk = VideoTokenizer(pixel_chunk_duration=n_frames, channels=16, device=dev)
    tk.net.init_random(seed=2)
    tk.register_mean_std(torch.zeros(16, 32), torch.ones(16, 32))
    model = DiffusionGen3CModel(net, tk, latent_shape=(16, 2, hh // 8, ww // 8))
    renders, masks = shard_r
    net.disable_context_parallel()
    lat_full = model.encode_warped_frames(renders, masks, torch.bfloat16)
    net.enable_context_parallel(group)
    lat_shard = model.encode_warped_frames(renders, masks, torch.bfloat16)
    same = torch.equal(lat_full, lat_shard) and lat_full.shape == (1, 64, 2, hh // 8, ww // 8)
    print(f"[cp_check] rank {rank}: sharded tokenizer encodes (4 clips over {world} ranks) == replicated: {same}", flush=True)
    good = good and same
    torch.cuda.synchronize()
    ok = torch.tensor([1.0 if good else 0.0], device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    if ok.item() != 1.0:
        sys.exit(1)
    if rank == 0:
        print("[cp_check] OK", flush=True)


if __name__ == "__main__":
    main()
