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
