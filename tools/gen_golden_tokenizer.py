"""tests/golden/tokenizer_small.npz: the reference's own CausalContinuousVideoTokenizer (CV8x8x8_720p config with
`channels` reduced 128 -> 16, seeded weights rounded to bf16) run in fp32 on a small clip; encoder_jit()/decoder_jit()
are exactly what the reference traces into encoder.jit / decoder.jit."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"


def bf16_bits(t):
    return t.detach().to(torch.bfloat16).view(torch.int16).numpy().astype(np.uint16)


def main():
    import ref_shims
    ref_shims.install()
    from cosmos_predict1.tokenizer.networks import TokenizerConfigs, TokenizerModels
    torch.manual_seed(7)
    cfg = dict(TokenizerConfigs.CV8x8x8_720p.value)
    cfg["channels"] = 16
    model = TokenizerModels.CV.value(**cfg).float().eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("norm.weight"):
                p.uniform_(0.5, 1.5)
            elif n.endswith(".bias"):
                p.normal_(0, 0.05)
            else:
                p.mul_(1.5)  # keep activations O(1) through ~40 layers
            p.copy_(p.to(torch.bfloat16).float())
    x = torch.rand(1, 3, 9, 32, 48) * 2 - 1
    x = x.to(torch.bfloat16).float()
    with torch.no_grad():
        z = model.encoder_jit()(x)[0] if isinstance(model.encoder_jit()(x), tuple) else model.encoder_jit()(x)
        zin = z.to(torch.bfloat16).float()
        y = model.decoder_jit()(zin)
    out = {"x": bf16_bits(x), "z_ref": z.numpy().astype(np.float32), "zin": bf16_bits(zin), "y_ref": y.numpy().astype(np.float32),
           "channels": np.array(cfg["channels"])}
    for n, p in model.state_dict().items():
        if any(s in n for s in ("wavelets", "_arange", "patch_size_buffer")):
            continue
        out["w:" + n] = bf16_bits(p)
    np.savez_compressed(GOLD / "tokenizer_small.npz", **out)
    print("z", tuple(z.shape), float(z.abs().mean()), "y", tuple(y.shape), float(y.abs().mean()), "MB",
          (GOLD / "tokenizer_small.npz").stat().st_size / 1e6, "n tensors", len(out))


if __name__ == "__main__":
    main()
