"""A few launches of the two MFMA kernels at the benchmark shapes, for rocprofv3 --pmc passes.
(full-size self-attention as bench.py launches it: S=56320, 32 heads, B=2, strided q / k views; the block GEMMs)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402

import os  # noqa: E402
ONLY = os.environ.get("G3_PMC_ONLY", "")  # "attn": just the self-attention launch (bench.py's in-run traffic passes)
dev = torch.device("cuda:0")
S, H, B = 56320, 32, 2  # the benchmark's launch: conditional + unconditional branch as one B = 2 problem, q / k column views of the fused QKV buffer
qkv = torch.randn(S * B, 3 * H * 128, device=dev).to(torch.bfloat16)
q, k = qkv[:, :H * 128], qkv[:, H * 128:2 * H * 128]
vt = ops.transpose_v(qkv[:, 2 * H * 128:], S, B, H)
out = torch.empty(S * B, H * 128, device=dev, dtype=torch.bfloat16)
for _ in range(2):
    ops.flash_attn(q, k, vt, S, S, B, H, out=out)
torch.cuda.synchronize()
del q, k, qkv, vt, out
if ONLY == "attn":
    print("done")
    sys.exit(0)
for (M, N, K, epi) in [(56320, 12288, 4096, 0), (56320, 16384, 4096, 1), (56320, 4096, 16384, 2)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    gate = torch.randn(1, N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(gate=gate, residual=res) if epi == 2 else {}
    for _ in range(2):
        ops.gemm_nt(a, w, out=o, epilogue=epi, **kw)
    torch.cuda.synchronize()
    del a, w, gate, res, o
# round 4: the tokenizer's d = 512 flash attention at the benchmark shape (16 frames x 14 080 pixels)
from gen3c_amd import _lib  # noqa: E402
T, HW, C = 16, 14080, 512
q5, k5, v5 = (torch.randn(T, HW, C, device=dev).to(torch.bfloat16) for _ in range(3))
vT5 = v5.reshape(T * HW, C).t().contiguous()
o5 = torch.empty_like(q5)
for _ in range(2):
    _lib.check(_lib.load().g3_spatial_attn_d512_bf16(q5.data_ptr(), k5.data_ptr(), vT5.data_ptr(), T * HW, HW, o5.data_ptr(), T, HW, C ** -0.5,
                                                     torch.cuda.current_stream().cuda_stream), "g3_spatial_attn_d512_bf16")
torch.cuda.synchronize()
print("done")
