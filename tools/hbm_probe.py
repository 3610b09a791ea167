"""Streaming write / copy rate of the box (torch fill_ / copy_), the yardstick for the GEMM epilogue's 2.5 TB/s."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402
dev = torch.device("cuda:0")
M, N = 56320, 12288
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
src = torch.randn(M, N, device=dev).to(torch.bfloat16)
gb = M * N * 2 / 1e9
ms = timeit(lambda: out.zero_(), 10); print(f"zero_  {gb:.2f} GB: {ms:.3f} ms  {gb / ms:.2f} TB/s write")
ms = timeit(lambda: out.copy_(src), 10); print(f"copy_  {gb:.2f} GB: {ms:.3f} ms  {2 * gb / ms:.2f} TB/s read+write")
ms = timeit(lambda: ops.add_inplace(out, src), 10); print(f"g3 add_inplace: {ms:.3f} ms  {3 * gb / ms:.2f} TB/s (2 reads + 1 write)")
