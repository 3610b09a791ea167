"""CPU oracle for the EDM-Euler sampling step of GEN3C (TEST INFRASTRUCTURE - never imported by gen3c_amd/).

PARITY: the scheduler arithmetic lives in diffusers==0.32.2 (requirements.txt:5 of the reference), which is neither
vendored under /root/reference nor installed here, so it cannot be pinned by running it ("parity unpinned" in that
sense). It is anchored on diffusers' OWN known answer instead - denoise_step run as the loop of diffusers'
tests/schedulers/test_scheduler_edm_euler.py::test_full_loop_no_noise reproduces that test's constant (sum|x| = 34.1855
+- 1e-3; tests/test_scheduler_kat_cpu.py) - and on float64 evaluations of Karras et al. 2022 eq. 5 / Table 1. Restated
from the published algorithm
(EDMEulerScheduler: Karras rho=7 sigmas, timesteps = 0.25*ln(sigma), init_noise_sigma = sqrt(sigma_max^2+1),
scale_model_input = x / sqrt(sigma^2 + sigma_data^2), step: x0 = c_skip*x + c_out*eps_out, Euler update with dt =
sigma_next - sigma). The loop body follows the reference's own call sites: model_v2w.py:130-149 and 201-259.
All math here is fp32 (no bf16 rounding points) - the HIP path is compared within a stated bf16 tolerance.
"""
from __future__ import annotations

import numpy as np
import torch

SIGMA_MAX, SIGMA_MIN, SIGMA_DATA, RHO = 80.0, 0.0002, 0.5, 7.0


def karras_sigmas(num_steps: int) -> torch.Tensor:
    ramp = torch.linspace(0, 1, num_steps)
    lo, hi = SIGMA_MIN ** (1 / RHO), SIGMA_MAX ** (1 / RHO)
    sig = ((hi + ramp * (lo - hi)) ** RHO).float()
    return torch.cat([sig, torch.zeros(1)])


def denoise_step(net_fn, xt, step_index, gt_latent, indicator, pose, num_steps, guidance, augment_sigma, seed):
    """One iteration of the loop at model_v2w.py:130-149. net_fn(x, timesteps, pose) -> network output;
    the unconditional branch gets pose = zeros (model_gen3c.py:126-127)."""
    sig = karras_sigmas(num_steps).to(gt_latent.device)  # (device-agnostic: the full-size chains of tools/psnr_vs_oracle.py run on the GPU's torch)
    sigma, sigma_next = sig[step_index], sig[step_index + 1]
    ind = indicator.clone()
    if augment_sigma >= float(sigma):
        ind = torch.zeros_like(ind)
    noise = torch.from_numpy(np.random.RandomState(seed).standard_normal(tuple(gt_latent.shape)).astype(np.float32)).to(gt_latent.device)
    aug = gt_latent + noise * augment_sigma
    aug = aug * (1 / (augment_sigma ** 2 + SIGMA_DATA ** 2) ** 0.5)        # scheduler.precondition_inputs
    aug = aug / (1 / (sigma ** 2 + SIGMA_DATA ** 2) ** 0.5)               # _reverse_precondition_input
    new_xt = ind * aug + (1 - ind) * xt
    c_in = 1 / (sigma ** 2 + SIGMA_DATA ** 2) ** 0.5
    t = (0.25 * torch.log(sigma)).reshape(1).to(torch.bfloat16).float()  # `t.to(**tensor_kwargs)` (model_v2w.py:141)
    out_c = net_fn(new_xt * c_in, t, pose)
    out_u = net_fn(new_xt * c_in, t, torch.zeros_like(pose))
    net_out = out_c + guidance * (out_c - out_u)
    c_skip = SIGMA_DATA ** 2 / (sigma ** 2 + SIGMA_DATA ** 2)
    c_out = sigma * SIGMA_DATA / (sigma ** 2 + SIGMA_DATA ** 2) ** 0.5
    latent_unscaled = (gt_latent - c_skip * new_xt) / c_out               # _reverse_precondition_output
    new_out = ind * latent_unscaled + (1 - ind) * net_out
    x0 = c_skip * new_xt + c_out * new_out                                 # scheduler.step
    return new_xt + (new_xt - x0) / sigma * (sigma_next - sigma)
