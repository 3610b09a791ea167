"""CPU: pins oracle/dit_oracle.py against fixtures produced by the reference's own Python (tools/gen_golden.py)."""
import pytest
import torch

from oracle import dit_oracle
from tests.golden_io import load_dit_case


@pytest.mark.parametrize("name", ["dit_tiny", "dit_small"])
def test_dit_oracle_matches_reference_golden(name):
    cfg, sd, inp, y_ref = load_dit_case(name)
    sd32 = {k: v.float() for k, v in sd.items()}
    y = dit_oracle.dit_forward(
        sd32, inp["x"].float(), inp["timesteps"].float(), inp["ctx"].float(), inp["mask"].float(), inp["pose"].float(),
        inp["padding_mask"], inp["fps"], num_blocks=cfg["blocks"], num_heads=cfg["heads"])
    assert y.shape == y_ref.shape
    # both are fp32 evaluations of the same graph; only summation order differs
    torch.testing.assert_close(y, y_ref, rtol=2e-4, atol=2e-4)
