"""LayerNorm + AdaLN modulate at the step's shape (M = 2 x 56 320 rows, D = 4096): one workgroup per row (two LDS reductions + barriers) vs one wave per row
(round 6, option ln_wave_rows), plain and with the position embedding added in the same pass; outputs compared. usage (GPU box): python tools/ln_wave_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import _lib, ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
S, B, D = 56320, 2, 4096
T, Hp, Wp = 16, 44, 80
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.randn(S * B, D, device=dev, generator=g) * 2 + 0.3).to(torch.bfloat16)
mod = torch.randn(B, 2 * D, device=dev, generator=g).to(torch.bfloat16)
pos = torch.randn(S, D, device=dev, generator=g).to(torch.bfloat16)
lib = _lib.load()
outs = {}
for rnd in range(3):
    line = []
    for wave in (0, 1):
        lib.g3_set_option(b"ln_wave_rows", wave)
        ms = timeit(lambda: ops.layernorm_modulate(x, mod[:, :D], mod[:, D:]), 5)
        outs[("plain", wave)] = ops.layernorm_modulate(x, mod[:, :D], mod[:, D:])
        line.append(f"[plain wave={wave}: {ms:.3f} ms = {2 * x.numel() * 2 / ms / 1e9:.2f} TB/s]")
        xs = x.clone()
        ms = timeit(lambda: ops.posemb_layernorm_modulate(xs, pos, None, None, None, T, Hp, Wp, B, mod[:, :D], mod[:, D:]), 5)
        xs = x.clone()
        outs[("pos", wave)] = ops.posemb_layernorm_modulate(xs, pos, None, None, None, T, Hp, Wp, B, mod[:, :D], mod[:, D:])
        line.append(f"[+pos wave={wave}: {ms:.3f} ms]")
    print("  ".join(line), flush=True)
lib.g3_set_option(b"ln_wave_rows", 0)
for k in ("plain", "pos"):
    a, b = outs[(k, 0)].float(), outs[(k, 1)].float()
    print(f"{k}: wave-per-row vs workgroup-per-row: elements differing {int((a != b).sum())} of {a.numel()}, max abs {float((a - b).abs().max()):.3e}, rel-L2 {float((a - b).norm() / a.norm()):.2e}")
