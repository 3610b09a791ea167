"""EDM-Euler sampling loop of GEN3C on the HIP path.

Mirrors (same names, argument meaning, CP behaviour):
  * diffusers 0.32.2 `EDMEulerScheduler(sigma_max=80, sigma_min=0.0002, sigma_data=0.5)` as used at
    cosmos_predict1/diffusion/model/model_t2w.py:65 (restated - diffusers is a third-party dependency that is not
    vendored in the reference; anchored on diffusers' own full-loop known answer, tests/test_scheduler_kat_cpu.py);
  * `DiffusionV2WModel.generate_samples_from_batch / _augment_noise_with_latent / _reverse_precondition_*`
    (model_v2w.py:84-155, 201-259) and `add_condition_video_indicator_and_video_input_mask` (model_v2w.py:32-82);
  * `VideoExtendCondition` (conditioner.py:107-134).

Per step the only device work outside the two network calls is two fused HIP kernels (csrc/sampler.hip). Scalar
coefficients are evaluated here on the host with 0-dim torch tensors in the SAME dtypes the reference's expressions
produce (several of them are bf16 because the loop's `sigma` is cast with `.to(**tensor_kwargs)`).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, fields
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from .dit import DataType, cacheable, tensor_version
from .parallel import cat_outputs_cp, split_inputs_cp


class EDMEulerScheduler:
    """Karras-rho EDM schedule + Euler step (diffusers 0.32.2 semantics, prediction_type='epsilon', final sigma 0)."""

    def __init__(self, sigma_max: float = 80.0, sigma_min: float = 0.0002, sigma_data: float = 0.5, rho: float = 7.0):
        self.sigma_max, self.sigma_min, self.sigma_data, self.rho = sigma_max, sigma_min, sigma_data, rho
        self.sigmas: Optional[torch.Tensor] = None
        self.timesteps: Optional[torch.Tensor] = None
        self.num_inference_steps: Optional[int] = None

    @property
    def init_noise_sigma(self) -> float:
        return (self.sigma_max ** 2 + 1) ** 0.5

    def set_timesteps(self, num_inference_steps: int):
        self.num_inference_steps = num_inference_steps
        ramp = torch.linspace(0, 1, num_inference_steps)
        min_inv_rho = self.sigma_min ** (1 / self.rho)
        max_inv_rho = self.sigma_max ** (1 / self.rho)
        sigmas = ((max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** self.rho).to(torch.float32)
        self.timesteps = 0.25 * torch.log(sigmas)  # precondition_noise
        self.sigmas = torch.cat([sigmas, torch.zeros(1, dtype=torch.float32)])

    def index_for_timestep(self, timestep: torch.Tensor) -> int:
        idx = (self.timesteps == timestep).nonzero()
        pos = 1 if len(idx) > 1 else 0
        return int(idx[pos].item())


@dataclass
class VideoExtendCondition:
    """Field-for-field mirror of conditioner.py:107-134 (BaseVideoCondition + VideoExtendCondition)."""
    crossattn_emb: torch.Tensor
    crossattn_mask: Optional[torch.Tensor] = None
    data_type: DataType = DataType.VIDEO
    padding_mask: Optional[torch.Tensor] = None
    fps: Optional[torch.Tensor] = None
    num_frames: Optional[torch.Tensor] = None
    image_size: Optional[torch.Tensor] = None
    scalar_feature: Optional[torch.Tensor] = None
    frame_repeat: Optional[torch.Tensor] = None
    video_cond_bool: Optional[bool] = None
    gt_latent: Optional[torch.Tensor] = None
    condition_video_indicator: Optional[torch.Tensor] = None
    condition_video_input_mask: Optional[torch.Tensor] = None
    condition_video_augment_sigma: Optional[torch.Tensor] = None
    condition_video_pose: Optional[torch.Tensor] = None

    def to_dict(self) -> Dict[str, Optional[torch.Tensor]]:
        return {f.name: getattr(self, f.name) for f in fields(self)}


def arch_invariant_rand(shape, dtype, device, seed=None) -> torch.Tensor:
    """utils/misc.py:133-154: numpy RandomState normals (GPU-architecture independent)."""
    arr = np.random.RandomState(seed).standard_normal(shape).astype(np.float32)
    return torch.from_numpy(arr).to(dtype=dtype, device=device)


def add_condition_video_indicator_and_video_input_mask(latent_state: torch.Tensor, condition: VideoExtendCondition,
                                                       num_condition_t: int) -> VideoExtendCondition:
    """model_v2w.py:32-82 (inference branch: first `num_condition_t` latent frames are the condition region)."""
    B, C, T, H, W = latent_state.shape
    assert num_condition_t is not None and num_condition_t <= T
    ind = torch.zeros(1, 1, T, 1, 1, device=latent_state.device, dtype=latent_state.dtype)
    ind[:, :, :num_condition_t] += 1.0
    condition.gt_latent = latent_state
    condition.condition_video_indicator = ind
    assert condition.video_cond_bool is not None, "video_cond_bool should be set"
    if condition.video_cond_bool:
        condition.condition_video_input_mask = ind.expand(B, 1, T, H, W).contiguous()
    else:
        condition.condition_video_input_mask = torch.zeros((B, 1, T, H, W), dtype=latent_state.dtype, device=latent_state.device)
    return condition


class Gen3CDenoiser:
    """The sampling half of `DiffusionGen3CModel` (model_gen3c.py:26-139 on top of model_v2w.py / model_t2w.py).

    `net` is a gen3c_amd.dit.VideoExtendGeneralDIT. Conditions are VideoExtendCondition objects that already carry
    `condition_video_pose` (the tokenizer-encoded warp buffers; zeros for the unconditional branch)."""

    def __init__(self, net, sigma_data: float = 0.5, state_shape=(16, 16, 88, 160)):
        self.net = net
        self.sigma_data = sigma_data
        self.state_shape = list(state_shape)
        self.scheduler = EDMEulerScheduler(sigma_max=80, sigma_min=0.0002, sigma_data=sigma_data)
        self._noise_cache: Dict[tuple, torch.Tensor] = {}
        self._fused_cache = None
        self.fuse_cond_uncond = os.environ.get("G3_FUSE_COND_UNCOND", "1") != "0"  # one batched forward for the conditional and the unconditional branch of a step (False: two calls)

    # ---- host-side scalar algebra, in the reference's dtypes ------------------------------------------------------
    def _coefficients(self, sigma32: torch.Tensor, sigma_next32: torch.Tensor, augment_sigma: float) -> dict:
        sd = self.sigma_data
        s_bf = sigma32.to(torch.bfloat16)  # `sigma = ....to(**self.tensor_kwargs)` (model_v2w.py:132)
        c_in_bf16 = 1 / ((s_bf ** 2 + sd ** 2) ** 0.5)                    # model_v2w.py:250 (bf16 0-dim tensor)
        c_skip_bf16 = sd ** 2 / (s_bf ** 2 + sd ** 2)                      # model_v2w.py:256
        c_out_bf16 = s_bf * sd / (s_bf ** 2 + sd ** 2) ** 0.5              # model_v2w.py:257
        c_in_step = 1 / ((sigma32 ** 2 + sd ** 2) ** 0.5)                  # scheduler.precondition_inputs (fp32)
        c_skip = sd ** 2 / (sigma32 ** 2 + sd ** 2)                        # scheduler.precondition_outputs
        c_out = sigma32 * sd / (sigma32 ** 2 + sd ** 2) ** 0.5
        c_in_aug = 1 / ((augment_sigma ** 2 + sd ** 2) ** 0.5)             # python floats
        return dict(c_in_bf16=float(c_in_bf16), c_skip_bf16=float(c_skip_bf16), c_out_bf16=float(c_out_bf16),
                    c_in_step=float(c_in_step), c_skip=float(c_skip), c_out=float(c_out), c_in_aug=float(c_in_aug),
                    sigma=float(sigma32), sigma_next=float(sigma_next32), indicator_off=bool(augment_sigma >= float(s_bf)))

    @staticmethod
    def _fused_cond_uncond_kwargs(condition: "VideoExtendCondition", uncondition: "VideoExtendCondition", B: int) -> Optional[dict]:
        """Keyword arguments of one net call over [cond batch | uncond batch], or None when the two conditions cannot share a call
        (different shapes / flags). Per-sample tensors (leading dim B) are concatenated; broadcast tensors (leading dim 1) and non-tensors
        must agree and are passed once."""
        dc, du = condition.to_dict(), uncondition.to_dict()
        out = {}
        for k, vc in dc.items():
            vu = du.get(k)
            if isinstance(vc, torch.Tensor) and isinstance(vu, torch.Tensor):
                if vc.shape != vu.shape or vc.dtype != vu.dtype:
                    return None
                if k == "gt_latent":  # not an input of the net
                    out[k] = vc
                elif vc.dim() >= 1 and vc.shape[0] == B:
                    out[k] = torch.cat([vc, vu], dim=0)
                elif vc.dim() >= 1 and vc.shape[0] == 1 and B != 1:
                    if not torch.equal(vc, vu):
                        return None
                    out[k] = vc
                else:
                    out[k] = torch.cat([vc, vu], dim=0) if vc.dim() >= 1 else vc
            elif vc is None and vu is None:
                out[k] = None
            elif isinstance(vc, torch.Tensor) or isinstance(vu, torch.Tensor):
                return None
            else:
                if vc != vu:
                    return None
                out[k] = vc
        return out

    def _augment_noise(self, shape, device, seed: int) -> torch.Tensor:
        key = (tuple(shape), str(device), seed)
        if key not in self._noise_cache:  # the reference regenerates the SAME numpy normals every step (seeded)
            self._noise_cache = {key: arch_invariant_rand(shape, torch.float32, device, seed)}
        return self._noise_cache[key]

    # ---- one denoise step (the unit bench.py times) ----------------------------------------------------------------
    @torch.no_grad()
    def denoise_step(self, xt: torch.Tensor, step_index: int, condition: VideoExtendCondition,
                     uncondition: VideoExtendCondition, guidance: float, condition_augment_sigma: float,
                     seed: int) -> torch.Tensor:
        """model_v2w.py:130-149 for scheduler step `step_index`; xt is this rank's [B,C,T_local,H,W] bf16 shard."""
        sch = self.scheduler
        sigma32, sigma_next32 = sch.sigmas[step_index], sch.sigmas[step_index + 1]
        co = self._coefficients(sigma32, sigma_next32, condition_augment_sigma)
        to_cp = self.net.is_context_parallel_enabled
        gt = condition.gt_latent
        ind = condition.condition_video_indicator.float()
        if co["indicator_off"]:
            ind = torch.zeros_like(ind)  # `if augment_sigma >= sigma: indicator = zeros` (model_v2w.py:229-230)
        noise = self._augment_noise(gt.shape, gt.device, seed)
        if to_cp:
            gt = split_inputs_cp(gt, 2, self.net.cp_group)
            ind = split_inputs_cp(ind, 2, self.net.cp_group)
            noise = split_inputs_cp(noise, 2, self.net.cp_group)
        B, C, T, H, W = xt.shape
        ind_t = ind.reshape(-1).contiguous()
        xt = xt.to(torch.bfloat16).contiguous()
        gt = gt.to(torch.bfloat16).contiguous()
        new_xt, new_xt_scaled = ops.edm_prepare_input(xt, gt, noise.contiguous(), ind_t, T, H * W, condition_augment_sigma,
                                                      co["c_in_aug"], co["c_in_bf16"], co["c_in_step"])
        t = sch.timesteps[step_index].to(device=xt.device, dtype=torch.bfloat16)
        fused = None
        if self.fuse_cond_uncond:
            # the conditions are the same objects for all steps of a chunk: build the batched arguments once (this also keeps the DiT's
            # per-context cross-attention K / V cache valid, which is keyed on the context tensor)
            # identity AND content version of every value: an in-place edit of a condition tensor between steps / chunks (crossattn_emb.copy_,
            # a refilled pose buffer) must rebuild the concatenated copies, exactly as the DiT's own K / V cache does
            def _ident(v):
                return (id(v), v.data_ptr(), tensor_version(v)) if isinstance(v, torch.Tensor) else (id(v),)
            key = (B,) + tuple(_ident(v) for v in condition.to_dict().values()) + tuple(_ident(v) for v in uncondition.to_dict().values())
            fc = self._fused_cache
            if not cacheable(*condition.to_dict().values(), *uncondition.to_dict().values()):
                # inference tensors carry no version counter: an in-place edit would go unnoticed -> rebuild the batched arguments every step
                fc = (key, self._fused_cond_uncond_kwargs(condition, uncondition, B))
            elif fc is None or fc[0] != key:
                # (the cache entry keeps the condition objects alive, so the ids in the key cannot be recycled while it is valid)
                fc = self._fused_cache = (key, self._fused_cond_uncond_kwargs(condition, uncondition, B), condition.to_dict(), uncondition.to_dict())
            fused = fc[1]
        if fused is not None:
            # the conditional and the unconditional forward (model_v2w.py:137-141: two net calls) as ONE forward over a batch of 2 B: every
            # row is computed exactly as in its own call (GEMM / attention / norm kernels work row-, head- and batch-wise), but the chip sees
            # launches twice as long - 55 instead of 27.5 rounds of attention workgroups, no half-empty last round - and half as many of them
            out = self.net(x=torch.cat([new_xt_scaled, new_xt_scaled], dim=0), timesteps=t, **fused)
            out_c, out_u = out[:B], out[B:]
        else:
            out_c = self.net(x=new_xt_scaled, timesteps=t, **condition.to_dict())
            out_u = self.net(x=new_xt_scaled, timesteps=t, **uncondition.to_dict())
        return ops.edm_cfg_euler_step(out_c, out_u, new_xt, gt, ind_t, T, H * W, guidance, co["c_skip_bf16"], co["c_out_bf16"],
                                      co["c_skip"], co["c_out"], co["sigma"], co["sigma_next"])

    @torch.no_grad()
    def generate_samples_from_batch(self, condition: VideoExtendCondition, uncondition: VideoExtendCondition,
                                    guidance: float = 1.0, seed: int = 1, state_shape=None, n_sample: int = 1,
                                    num_steps: int = 35, condition_augment_sigma: float = 0.001,
                                    xt: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Sampling loop of model_v2w.py:84-155 given ready conditions. `xt` (optional) injects the initial noise
        (already multiplied by init_noise_sigma) - torch.randn on the device is not reproducible across vendors, parity
        runs inject it (SURVEY.md 7 'RNG parity')."""
        state_shape = list(state_shape or self.state_shape)
        self.scheduler.set_timesteps(num_steps)
        dev = condition.gt_latent.device
        if xt is None:
            g = torch.Generator(device=dev).manual_seed(seed)
            xt = torch.randn((n_sample, *state_shape), device=dev, dtype=torch.bfloat16, generator=g) * self.scheduler.init_noise_sigma
        to_cp = self.net.is_context_parallel_enabled
        if to_cp:
            xt = split_inputs_cp(xt, 2, self.net.cp_group)
        for i in range(num_steps):
            xt = self.denoise_step(xt, i, condition, uncondition, guidance, condition_augment_sigma, seed)
        if to_cp:
            xt = cat_outputs_cp(xt, 2, self.net.cp_group)
        return xt
