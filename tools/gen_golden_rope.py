"""Pins oracle/dit_oracle.te_rope_fused to the reference's own copy of TransformerEngine's RoPE (VERDICT r2 next #8a):
cosmos_predict1/autoregressive/modules/embedding.py:46-85 (`_rotate_half_te`, `_apply_rotary_pos_emb_te`, "Adopted from TransformerEngine").
The two function bodies are cut out of the reference file with ast (the module's import chain needs megatron / einops extras these functions
do not use) and executed on seeded inputs in the DiT's layout: t [s, b, h, d], cos / sin [s, 1, 1, d] fp32, d = 128 = rot_dim.

  python tools/gen_golden_rope.py     (build container only) -> tests/golden/te_rope.npz
"""
import ast
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/cosmos_predict1/autoregressive/modules/embedding.py")


def main():
    src = REF.read_text()
    ns = {"torch": torch}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in ("_rotate_half_te", "_apply_rotary_pos_emb_te"):
            exec(compile(ast.get_source_segment(src, node), str(REF), "exec"), ns)
    g = torch.Generator().manual_seed(2024)
    s, b, h, d = 37, 2, 3, 128
    t32 = torch.randn(s, b, h, d, generator=g)
    tbf = torch.randn(s, b, h, d, generator=g).to(torch.bfloat16)
    freqs = (torch.rand(s, 1, 1, d // 2, generator=g) * 50.0)
    freqs = torch.cat([freqs, freqs], dim=-1).float()  # halves layout, as VideoRopePosition3DEmb builds it (position_embedding.py:176-186)
    cos, sin = torch.cos(freqs), torch.sin(freqs)
    out32 = ns["_apply_rotary_pos_emb_te"](t32, cos, sin)
    outbf = ns["_apply_rotary_pos_emb_te"](tbf, cos, sin)  # bf16 * fp32 promotes: computed in fp32
    assert out32.dtype == torch.float32 and outbf.dtype == torch.float32
    np.savez_compressed(ROOT / "tests" / "golden" / "te_rope.npz", t32=t32.numpy(), tbf=tbf.float().numpy(), freqs=freqs.numpy(),
                        out32=out32.numpy(), outbf=outbf.numpy())
    print("wrote tests/golden/te_rope.npz")


if __name__ == "__main__":
    main()
