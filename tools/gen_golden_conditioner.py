"""tests/golden/conditioner.npz: what the reference's `video_cond` conditioner (VideoExtendConditioner with the embedders and
dropout rates of config/base/conditioner.py:27-30,195-222) returns from its two builders, on a small batch, with and without
`neg_t5_text_embeddings`. Pins gen3c_amd.pipeline.DiffusionGen3CModel._text_conditions (cond / uncond text for CFG)."""
import sys
from pathlib import Path
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import ref_shims  # noqa: E402

ref_shims.install()  # its lazy_config.instantiate is the identity: `obj` below is the already-built embedder
sys.modules["cosmos_predict1.utils.lazy_config"].LazyCall = lambda cls: (lambda **kw: (cls, kw))  # config module builds L(...)() at import
from cosmos_predict1.diffusion.conditioner import TextAttr, VideoExtendConditioner  # noqa: E402
from cosmos_predict1.diffusion.config.base.conditioner import BooleanFlag, ReMapkey  # noqa: E402

cond = VideoExtendConditioner(
    text=NS(obj=TextAttr(), dropout_rate=0.2, input_keys=["t5_text_embeddings", "t5_text_mask"]),
    fps=NS(obj=ReMapkey(output_key="fps", dtype=None), dropout_rate=0.0, input_key="fps"),
    num_frames=NS(obj=ReMapkey(output_key="num_frames", dtype=None), dropout_rate=0.0, input_key="num_frames"),
    image_size=NS(obj=ReMapkey(output_key="image_size", dtype=None), dropout_rate=0.0, input_key="image_size"),
    padding_mask=NS(obj=ReMapkey(output_key="padding_mask", dtype=None), dropout_rate=0.0, input_key="padding_mask"),
    video_cond_bool=NS(obj=BooleanFlag(output_key="video_cond_bool"), dropout_rate=0.2, input_key="fps"),
)
g = torch.Generator().manual_seed(3)
M, C = 16, 8
batch = {
    "t5_text_embeddings": torch.randn(1, M, C, generator=g),
    "t5_text_mask": torch.ones(1, M),
    "fps": torch.tensor([24.0]), "num_frames": torch.tensor([121.0]), "image_size": torch.tensor([[64.0, 96.0, 64.0, 96.0]]),
    "padding_mask": torch.zeros(1, 1, 64, 96),
}
neg = torch.randn(1, M, C, generator=g)
neg_mask = torch.ones(1, M)
neg_mask[:, M // 2:] = 0
out = {"pos": batch["t5_text_embeddings"].numpy(), "pos_mask": batch["t5_text_mask"].numpy(), "neg": neg.numpy(), "neg_mask": neg_mask.numpy()}
for tag, b in (("noneg", dict(batch)), ("neg", {**batch, "neg_t5_text_embeddings": neg, "neg_t5_text_mask": neg_mask})):
    for builder in ("get_condition_with_negative_prompt", "get_condition_uncondition"):
        c, u = getattr(cond, builder)(b)
        for side, v in (("cond", c), ("uncond", u)):
            out[f"{tag}:{builder}:{side}:crossattn_emb"] = v.crossattn_emb.numpy()
            out[f"{tag}:{builder}:{side}:crossattn_mask"] = v.crossattn_mask.numpy()
            out[f"{tag}:{builder}:{side}:fps"] = v.fps.numpy()
np.savez_compressed(ROOT / "tests" / "golden" / "conditioner.npz", **out)
print(len(out), "entries")
