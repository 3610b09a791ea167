"""CPU: known-answer test of the EDM-Euler schedule. diffusers 0.32.2 - where the reference's EDMEulerScheduler lives
(requirements.txt:5; model_t2w.py:65 constructs it with sigma_max 80, sigma_min 0.0002, sigma_data 0.5) - is neither vendored in the reference
nor installable here, so its arithmetic cannot be pinned by running it. What CAN be anchored is the published algorithm it implements:
Karras et al. 2022, "Elucidating the Design Space of Diffusion-Based Generative Models", eq. 5 (sigma_i = (sigma_max^(1/rho) + i/(N-1)
(sigma_min^(1/rho) - sigma_max^(1/rho)))^rho) and Table 1 (c_skip, c_out, c_in, c_noise = ln(sigma)/4). The constants below were evaluated
from those formulas in float64, independently of this repository's code; the product scheduler and the oracle must both reproduce them."""
import math

import torch

# (index, sigma_i, c_noise_i) for N = 35, rho = 7, sigma_max = 80, sigma_min = 0.0002 - float64 evaluation of eq. 5 / Table 1
KAT = [(0, 80.0, 1.0955066586684703), (1, 67.12601686534994, 1.0516429253865789), (2, 56.070267422077706, 1.0066514198891654),
       (10, 10.926856940976407, 0.5978059245630466), (17, 1.7492122807575898, 0.1397913903293126), (25, 0.09351188025110804, -0.5924166973124871),
       (33, 0.0005527091292231598, -1.8751696704399559), (34, 0.00019999999999999987, -2.1292982978540596)]


def test_karras_schedule_and_preconditioning_known_answers():
    from gen3c_amd.sampler import EDMEulerScheduler
    from oracle import sampler_oracle
    sch = EDMEulerScheduler(sigma_max=80.0, sigma_min=0.0002, sigma_data=0.5)
    sch.set_timesteps(35)
    assert sch.sigmas.shape == (36,) and float(sch.sigmas[-1]) == 0.0 and sch.timesteps.shape == (35,)
    assert abs(sch.init_noise_sigma - 80.00624975587844) < 1e-12  # sqrt(sigma_max^2 + 1)
    osig = sampler_oracle.karras_sigmas(35)
    for i, s, cn in KAT:
        for got in (float(sch.sigmas[i]), float(osig[i])):
            assert abs(got - s) <= 2e-6 * s, (i, got, s)  # fp32 evaluation of a 7th power: a few ulp
        assert abs(float(sch.timesteps[i]) - cn) <= 2e-6, (i, float(sch.timesteps[i]), cn)
    assert all(float(sch.sigmas[i]) > float(sch.sigmas[i + 1]) for i in range(35))
    # Table 1 at sigma_10 with sigma_data = 0.5, as the sampler evaluates them (fp32 forms of gen3c_amd/sampler.py:_coefficients)
    s = torch.tensor(10.926856940976407, dtype=torch.float32)
    sd = 0.5
    c_skip, c_out, c_in = sd ** 2 / (s ** 2 + sd ** 2), s * sd / (s ** 2 + sd ** 2) ** 0.5, 1 / (s ** 2 + sd ** 2) ** 0.5
    for got, want in ((c_skip, 0.002089493812168051), (c_out, 0.4994773533874764), (c_in, 0.09142196261660654)):
        assert abs(float(got) - want) <= 1e-6 * max(want, 1e-3)
    # the Euler update is x + (x - x0) / sigma * (sigma_next - sigma): with x0 = 0 it contracts x by sigma_next / sigma
    x = torch.tensor([3.0])
    nxt = x + (x - 0.0) / sch.sigmas[10] * (sch.sigmas[11] - sch.sigmas[10])
    assert math.isclose(float(nxt), 3.0 * float(sch.sigmas[11]) / float(sch.sigmas[10]), rel_tol=1e-6)


# diffusers' OWN known answer for this scheduler: tests/schedulers/test_scheduler_edm_euler.py::EDMEulerSchedulerTest::test_full_loop_no_noise
# (public repository, v0.32.x) runs EDMEulerScheduler(num_train_timesteps=256, sigma_min=0.002, sigma_max=80.0) for 10 steps on
# SchedulerCommonTest's deterministic sample (arange(4*3*8*8) / 768, reshaped (3, 8, 8, 4), permuted to (4, 3, 8, 8), times
# init_noise_sigma) with the dummy model `sample * t / (t + 1)` (t = the scheduler's timestep 0.25 ln sigma), through scale_model_input
# and step, and asserts |sum(|x|) - 34.1855| < 1e-3 and |mean(|x|) - 0.044| < 1e-3. diffusers cannot be imported here (not vendored, no
# network), so the two constants are quoted from that file; they are self-consistent (34.1855 / 768 = 0.0445) and an independent
# implementation reproducing the sum to 3e-4 is the check that both the quotation and the restatement are right.
DIFFUSERS_FULL_LOOP_SUM, DIFFUSERS_FULL_LOOP_MEAN, DIFFUSERS_TOL = 34.1855, 0.044, 1e-3


def _diffusers_dummy_sample():
    n = 4 * 3 * 8 * 8
    return (torch.arange(n).reshape(3, 8, 8, 4) / n).permute(3, 0, 1, 2)


def test_product_scheduler_reproduces_diffusers_full_loop_known_answer():
    """sigmas / timesteps / init_noise_sigma of gen3c_amd.sampler.EDMEulerScheduler under diffusers' test configuration, driven through
    the published scale_model_input / step arithmetic (c_in, c_skip, c_out, Euler with dt = sigma_next - sigma) in fp32."""
    from gen3c_amd.sampler import EDMEulerScheduler
    sch = EDMEulerScheduler(sigma_max=80.0, sigma_min=0.002, sigma_data=0.5)
    sch.set_timesteps(10)
    sd = 0.5
    x = _diffusers_dummy_sample() * sch.init_noise_sigma
    for i, t in enumerate(sch.timesteps):
        s = sch.sigmas[i]
        scaled = x * (1 / (s ** 2 + sd ** 2) ** 0.5)                                  # scale_model_input
        out = scaled * t / (t + 1)                                                    # dummy model
        x0 = sd ** 2 / (s ** 2 + sd ** 2) * x + s * sd / (s ** 2 + sd ** 2) ** 0.5 * out  # precondition_outputs (epsilon prediction)
        x = x + (x - x0) / s * (sch.sigmas[i + 1] - s)                                # step
    assert abs(float(x.abs().sum()) - DIFFUSERS_FULL_LOOP_SUM) < DIFFUSERS_TOL, float(x.abs().sum())
    assert abs(float(x.abs().mean()) - DIFFUSERS_FULL_LOOP_MEAN) < DIFFUSERS_TOL


def test_oracle_denoise_step_reproduces_diffusers_full_loop_known_answer(monkeypatch):
    """The same loop through oracle.sampler_oracle.denoise_step (the checker of the HIP sampling step): no condition region (indicator 0),
    guidance 0, the dummy model fed the exact fp32 timestep (the oracle hands the network the bf16-cast timestep the reference's loop
    produces, model_v2w.py:141 - a property of GEN3C's loop, not of the scheduler)."""
    from oracle import sampler_oracle as so
    monkeypatch.setattr(so, "SIGMA_MIN", 0.002)
    n_steps = 10
    sig = so.karras_sigmas(n_steps)
    x = _diffusers_dummy_sample()[None].reshape(1, 4, 3, 8, 8) * (so.SIGMA_MAX ** 2 + 1) ** 0.5
    zeros = torch.zeros_like(x)
    for i in range(n_steps):
        t32 = 0.25 * torch.log(sig[i])
        x = so.denoise_step(lambda inp, t, pose: inp * t32 / (t32 + 1), x, i, gt_latent=zeros, indicator=torch.zeros(1, 1, 3, 1, 1), pose=zeros,
                            num_steps=n_steps, guidance=0.0, augment_sigma=0.001, seed=0)
    assert abs(float(x.abs().sum()) - DIFFUSERS_FULL_LOOP_SUM) < DIFFUSERS_TOL, float(x.abs().sum())
    assert abs(float(x.abs().mean()) - DIFFUSERS_FULL_LOOP_MEAN) < DIFFUSERS_TOL
