"""CPU: the two oracle functions that restate TransformerEngine arithmetic which cannot be pinned here (TE is neither vendored in the reference
nor installable: `RMSNorm`, `DotProductAttention` - attention.py:130-131, 228-238) held against PyTorch's own, independently written
implementations of the same published operators. This does not make them "pinned to TE" (DESIGN.md §4 keeps saying so); it rules out a
restatement error that the HIP kernels - checked against these very functions - would inherit."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import dit_oracle as o


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_te_rmsnorm_restatement_equals_torch_rms_norm(dtype):
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(5, 7, 32, 128, generator=g) * 3).to(dtype)
    w = (1 + 0.2 * torch.randn(128, generator=g)).to(dtype)
    want = F.rms_norm(x.float(), (128,), w.float(), eps=1e-6).to(dtype)  # fp32 math, one rounding: what TE's kernel does for bf16 tensors
    got = o.te_rmsnorm(x, w, eps=1e-6)
    assert got.dtype == dtype
    if dtype == torch.float32:
        assert torch.allclose(got, want, rtol=2e-6, atol=1e-6)
    else:
        assert float((got.float() - want.float()).abs().max()) <= 2 ** -7 * float(want.float().abs().max())  # at most one bf16 ulp
        assert float((got != want).float().mean()) < 2e-3


@pytest.mark.parametrize("sq,skv", [(48, 48), (33, 70)])
def test_attention_restatement_equals_torch_sdpa(sq, skv):
    g = torch.Generator().manual_seed(5)
    b, h, d = 2, 4, 128
    q = torch.randn(sq, b, h, d, generator=g)
    k = torch.randn(skv, b, h, d, generator=g)
    v = torch.randn(skv, b, h, d, generator=g)
    want = F.scaled_dot_product_attention(q.permute(1, 2, 0, 3), k.permute(1, 2, 0, 3), v.permute(1, 2, 0, 3), scale=1 / math.sqrt(d))
    want = want.permute(2, 0, 1, 3).reshape(sq, b, h * d)
    got = o.attention_sbhd(q, k, v)
    assert got.shape == want.shape and torch.allclose(got, want, rtol=1e-5, atol=2e-6), float((got - want).abs().max())
