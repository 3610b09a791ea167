"""Two launches of the self-attention kernel variant named by G3_ATTN_VARIANT at the benchmark shape, for rocprofv3 --pmc passes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
S, H = 56320, 32
q, k, v = (torch.randn(S, H * 128, device=dev).to(torch.bfloat16) for _ in range(3))
vt = ops.transpose_v(v, S, 1, H)
out = torch.empty_like(q)
for _ in range(2):
    ops.flash_attn(q, k, vt, S, S, 1, H, out=out)
torch.cuda.synchronize()
print("done")
