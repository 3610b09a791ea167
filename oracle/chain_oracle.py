"""Oracle of ONE generated chunk (TEST INFRASTRUCTURE - never imported by gen3c_amd/): the composition
Gen3cPipeline.generate -> DiffusionGen3CModel.encode_warped_frames / _get_conditions -> generate_samples_from_batch -> decode
(gen3c_pipeline.py:108-184, model/model_gen3c.py:32-139, model/model_v2w.py:84-155, world_generation_pipeline.py:244-245) assembled from the
pinned component oracles (tokenizer_oracle, dit_oracle, sampler_oracle). Plain torch on whatever device the tensors live on.

`net_dtype`: the precision the NETWORKS run in. torch.float32 = the parity oracle. torch.bfloat16 = the reference's own precision
(`precision="bfloat16"`, config/base/model.py:29: bf16 parameters and activations in the DiT and the tokenizer, fp32 sampler algebra with the
network input / output cast at the call, model_v2w.py:137-146) - the yardstick tools/psnr_vs_oracle.py holds the HIP path against.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import dit_oracle, sampler_oracle
from . import tokenizer_oracle as tok


def generate_chunk(dit_sd: Dict[str, torch.Tensor], tok_sd: Dict[str, torch.Tensor], latent_mean: torch.Tensor, latent_std: torch.Tensor,
                   image: torch.Tensor, renders: torch.Tensor, masks: torch.Tensor, prompt: torch.Tensor, negative_prompt: Optional[torch.Tensor],
                   xt: torch.Tensor, *, num_steps: int, guidance: float, num_blocks: int, num_heads: int, frame_buffer_max: int = 2,
                   net_dtype: torch.dtype = torch.float32, seed: int = 1, fps: float = 24.0, return_latent: bool = False):
    """image [1,3,1,H,W] in [-1,1]; renders [1,T,N,3,H,W], masks [1,T,N,1,H,W] (Cache3D.render_cache); prompt / negative_prompt [1,M,C];
    xt [1,16,t,h,w] initial noise times init_noise_sigma. Weights: reference-named state dicts (DiT without the `net.` prefix).
    -> video [T,H,W,3] float in [0,1]."""
    dev = xt.device
    f32 = torch.float32
    T, N = renders.shape[1], renders.shape[2]
    H, W = renders.shape[-2:]
    cast = lambda sd: {k: (v.to(dev) if k == "pos_embedder.seq" else v.to(dev, net_dtype)) for k, v in sd.items()}
    dsd, tsd = cast(dit_sd), cast(tok_sd)
    mean, std = latent_mean.to(dev, net_dtype), latent_std.to(dev, net_dtype)

    def encode(v):  # tokenizer.encode(x) * sigma_data (model_t2w.py:133)
        return (tok.encode(tsd, v.to(dev, net_dtype), mean, std) * 0.5).to(f32)

    clip = torch.cat([image.to(dev, f32), torch.zeros(1, 3, T - 1, H, W, device=dev)], dim=2)  # condition frame + zero padding (inference_utils.py:650-700)
    gt = encode(clip)
    lat = []
    for n in range(N):  # encode_warped_frames: each buffer's render, and its mask as 3 channels in [-1,1] (model_gen3c.py:32-57)
        rv = renders[0, :, n].permute(1, 0, 2, 3)[None]
        mv = (masks[0, :, n] * 2 - 1).repeat(1, 3, 1, 1).permute(1, 0, 2, 3)[None]
        lat += [encode(rv), encode(mv)]
    t_lat, h_lat, w_lat = gt.shape[2:]
    for _ in range(frame_buffer_max - N):
        lat += [torch.zeros(1, 16, t_lat, h_lat, w_lat, device=dev)] * 2
    pose = torch.cat(lat, dim=1)
    ind = torch.zeros(1, 1, t_lat, 1, 1, device=dev)
    ind[:, :, :1] = 1  # num_condition_t = 1
    mask_in = ind.expand(1, 1, t_lat, h_lat, w_lat).contiguous()
    pad = torch.zeros(1, 1, H, W, device=dev)
    fps_t = torch.tensor([fps], device=dev)

    def net(ctx):
        c = ctx.to(dev, net_dtype)
        return lambda x, tt, pose_: dit_oracle.dit_forward(dsd, x.to(net_dtype), tt.to(dev, net_dtype), c, mask_in.to(net_dtype), pose_.to(net_dtype),
                                                           pad.to(net_dtype), fps_t, num_blocks=num_blocks, num_heads=num_heads).to(f32)

    f_cond = net(prompt)
    f_unc = net(negative_prompt if negative_prompt is not None else prompt)  # no negative embedding: the positive text (conditioner.py:267-323)
    x = xt.to(dev, f32)
    gt_in = gt.to(torch.bfloat16).to(f32)  # the latent condition is handed over in bf16 (tensor_kwargs, model_v2w.py:121-128)
    for i in range(num_steps):
        # denoise_step calls net_fn exactly twice per step, in this order: with the pose (conditional branch, the prompt's context), then with
        # zeros (unconditional branch, the negative prompt's context; model_v2w.py:137-141, model_gen3c.py:126-127)
        calls = []

        def net_fn(xx, tt, pp):
            calls.append(1)
            return (f_cond if len(calls) == 1 else f_unc)(xx, tt, pp)

        x = sampler_oracle.denoise_step(net_fn, x, i, gt_in, ind, pose, num_steps, guidance, 0.001, seed)
        assert len(calls) == 2
    if return_latent:
        return x
    y = tok.decode(tsd, (x / 0.5).to(net_dtype), mean, std).to(f32)
    return ((1.0 + y).clamp(0, 2) / 2)[0].permute(1, 2, 3, 0)
