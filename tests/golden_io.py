"""Loader for tests/golden/*.npz (written by tools/gen_golden.py from the reference's own Python)."""
from pathlib import Path

import numpy as np
import torch

GOLD = Path(__file__).resolve().parent / "golden"


def from_bf16_bits(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.uint16).view(np.int16).copy()).view(torch.bfloat16)


def load_dit_case(name: str):
    z = np.load(GOLD / f"{name}.npz")
    cfg = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    sd = {}
    for k in z.files:
        if k.startswith("w:"):
            n = k[2:]
            sd[n] = torch.from_numpy(z[k]) if n == "pos_embedder.seq" else from_bf16_bits(z[k])
    inputs = dict(
        x=from_bf16_bits(z["x"]), pose=from_bf16_bits(z["pose"]), mask=torch.from_numpy(z["mask"]),
        ctx=from_bf16_bits(z["ctx"]), timesteps=from_bf16_bits(z["timesteps"]), fps=torch.from_numpy(z["fps"]),
        padding_mask=torch.from_numpy(z["padding_mask"]),
    )
    return cfg, sd, inputs, torch.from_numpy(z["y_ref"])


def load_tokenizer_case(name: str = "tokenizer_small"):
    z = np.load(GOLD / f"{name}.npz")
    sd = {k[2:]: from_bf16_bits(z[k]) for k in z.files if k.startswith("w:")}
    return sd, from_bf16_bits(z["x"]), torch.from_numpy(z["z_ref"]), from_bf16_bits(z["zin"]), torch.from_numpy(z["y_ref"])
