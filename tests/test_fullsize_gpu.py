"""GPU: the two MFMA kernels at BASELINE.json's full size (56 320 tokens, 32 x 128 heads, D = 4096), checked through properties
that do not need the (far too slow) CPU oracle at this size:
  * sampled rows against an fp32 torch reference of the same op on the same device (the op is floating point; tolerances below);
  * permutation invariance of attention over the key/value tokens;
  * linearity of the GEMM in its token rows (row subsets give the same rows)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
S, D, HD = 56320, 4096, 128


def _rel_l2(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def test_attention_full_sequence_sampled_rows_and_kv_permutation():
    from gen3c_amd import ops
    dev = torch.device("cuda:0")
    H = 4  # four of the 32 heads keep the test at ~2 s; every head runs the same code path
    g = torch.Generator(device=dev).manual_seed(7)
    q = torch.randn(S, H * HD, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(S, H * HD, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(S, H * HD, device=dev, generator=g).to(torch.bfloat16)
    out = ops.flash_attn(q, k, ops.transpose_v(v, S, 1, H), S, S, 1, H)
    rows = torch.cat([torch.arange(0, 64, device=dev), torch.randint(0, S, (128,), device=dev, generator=g), torch.arange(S - 64, S, device=dev)])
    for h in range(H):
        sl = slice(h * HD, (h + 1) * HD)
        sc = (q[rows, sl].float() @ k[:, sl].float().t()) / math.sqrt(HD)
        ref = torch.softmax(sc, dim=-1) @ v[:, sl].float()
        r = _rel_l2(out[rows, sl], ref)
        # bf16 Q*scale, P and output roundings give 2.9e-3 here; a staging race in the first tiles (K read before its LDS-DMA landed,
        # or overwritten while a late wave still reads it) showed up as 6.6e-3 on this very check
        assert r < 4e-3, f"head {h}: rel-L2 {r:.3e} vs fp32 reference on sampled rows"
    # softmax(QK^T)V does not depend on the order of the key/value tokens
    perm = torch.randperm(S, device=dev, generator=g)
    out_p = ops.flash_attn(q, k[perm].contiguous(), ops.transpose_v(v[perm].contiguous(), S, 1, H), S, S, 1, H)
    r = _rel_l2(out_p, out)
    print(f"[attn 56320] kv-permutation rel-L2 {r:.3e}")
    # P is rounded to bf16 relative to the running maximum of ITS tile order, so the two runs carry independent 2^-9 roundings; with
    # zero-mean random V the output is itself a random-walk sum and that rounding noise does not average out: expect ~3e-3
    assert r < 6e-3


@pytest.mark.parametrize("N,K,epi", [(12288, 4096, 0), (16384, 4096, 1), (4096, 16384, 2)])
def test_gemm_full_size_sampled_rows_and_row_subsets(N, K, epi):
    from gen3c_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(N + K)
    a = torch.randn(S, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    gate = torch.randn(1, N, device=dev, generator=g).to(torch.bfloat16)
    res = torch.randn(S, N, device=dev, generator=g).to(torch.bfloat16)
    kw = dict(gate=gate, residual=res) if epi == 2 else {}
    out = ops.gemm_nt(a, w, epilogue=epi, **kw)
    rows = torch.cat([torch.arange(0, 32, device=dev), torch.randint(0, S, (192,), device=dev, generator=g), torch.arange(S - 32, S, device=dev)])
    ref = a[rows].float() @ w.float().t()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    elif epi == 2:
        ref = res[rows].float() + gate.float() * ref
    r = _rel_l2(out[rows], ref)
    assert r < 4e-3, f"rel-L2 {r:.3e} vs fp32 reference on sampled rows"
    # the same rows computed as their own (ragged, 1 000-row) problem are bit-identical: every output element accumulates over K in
    # the same order whatever tile it lands in
    sub = slice(12345, 13345)
    kw2 = dict(gate=gate, residual=res[sub].contiguous()) if epi == 2 else {}
    out_sub = ops.gemm_nt(a[sub].contiguous(), w, epilogue=epi, **kw2)
    assert torch.equal(out_sub, out[sub])
