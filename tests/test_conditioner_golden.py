"""CPU: cond / uncond text conditions of DiffusionGen3CModel._text_conditions against the reference's `video_cond` conditioner
(tests/golden/conditioner.npz, written by tools/gen_golden_conditioner.py from cosmos_predict1/diffusion/conditioner.py:234-292)."""
import numpy as np
import torch

from gen3c_amd.pipeline import DiffusionGen3CModel
from tests.golden_io import GOLD


def _batch(z, with_neg):
    b = {"t5_text_embeddings": torch.from_numpy(z["pos"]), "t5_text_mask": torch.from_numpy(z["pos_mask"]), "fps": torch.tensor([24.0]),
         "num_frames": torch.tensor([121.0]), "image_size": torch.tensor([[64.0, 96.0, 64.0, 96.0]]), "padding_mask": torch.zeros(1, 1, 64, 96)}
    if with_neg:
        b["neg_t5_text_embeddings"] = torch.from_numpy(z["neg"])
        b["neg_t5_text_mask"] = torch.from_numpy(z["neg_mask"])
    return b


def test_text_conditions_match_reference_builders():
    z = np.load(GOLD / "conditioner.npz")
    n = 0
    for tag, with_neg in (("noneg", False), ("neg", True)):
        for builder, is_neg in (("get_condition_with_negative_prompt", True), ("get_condition_uncondition", False)):
            c, u = DiffusionGen3CModel._text_conditions(None, _batch(z, with_neg), is_neg)
            for side, v in (("cond", c), ("uncond", u)):
                np.testing.assert_array_equal(v.crossattn_emb.numpy(), z[f"{tag}:{builder}:{side}:crossattn_emb"])
                np.testing.assert_array_equal(v.crossattn_mask.numpy(), z[f"{tag}:{builder}:{side}:crossattn_mask"])
                np.testing.assert_array_equal(v.fps.numpy(), z[f"{tag}:{builder}:{side}:fps"])
                n += 1
    assert n == 8


def test_pipeline_default_keeps_positive_text_for_the_unconditional_branch():
    """Gen3cPipeline always passes is_negative_prompt=True (gen3c_pipeline.py:250): without a negative embedding the
    unconditional branch sees the POSITIVE text (only the pose condition is zeroed), never zero embeddings."""
    z = np.load(GOLD / "conditioner.npz")
    c, u = DiffusionGen3CModel._text_conditions(None, _batch(z, False), True)
    assert torch.equal(u.crossattn_emb, c.crossattn_emb) and float(u.crossattn_emb.abs().sum()) > 0
    assert np.array_equal(z["noneg:get_condition_with_negative_prompt:uncond:crossattn_emb"], z["pos"])
    assert not np.any(z["noneg:get_condition_uncondition:uncond:crossattn_emb"])
