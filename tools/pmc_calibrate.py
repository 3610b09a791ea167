"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE (KB) on THIS box against kernels whose memory-side traffic is known, by load width - the renderer's kernels
read 4 bytes per lane (points, image planes) and, since round 5, 16 bytes per lane (the gather pass's window loads); the attention / GEMM operands arrive by 16-byte
LDS-DMA. (on which load widths was the calibration done?) Three product kernels, each over a buffer far larger than the caches:
  g3_add_inplace_bf16      16-byte loads + stores: reads 2 n x 2 B, writes n x 2 B
  g3_reliable_depth_mask   4-byte loads, 25 per pixel (5 x 5 window, neighbours from the caches): reads >= 4 B per pixel, writes 1 B per pixel
  g3_unproject_points      4-byte loads + 4-byte stores: reads 4 B, writes 12 B per pixel
run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -o p -- python tools/pmc_calibrate.py   (and again with WRITE_SIZE);
tools/pmc_calibrate.py --expected prints the expected bytes per launch."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
N_ADD = 1 << 29            # bf16 elements: 1 GiB per operand
H, W, NIMG = 704, 1280, 64  # 64 depth maps of 704 x 1280 = 57.7 M pixels
EXPECTED = {
    "add_inplace_kernel": dict(fetch=2 * N_ADD * 2, write=N_ADD * 2, loads="16 B / lane"),
    "reliable_mask_kernel": dict(fetch=NIMG * H * W * 4, write=NIMG * H * W * 1, loads="4 B / lane, 25 per pixel (24 from the caches)"),
    "unproject_kernel": dict(fetch=NIMG * H * W * 4, write=NIMG * H * W * 12, loads="4 B / lane"),
}
if "--expected" in sys.argv:
    for k, v in EXPECTED.items():
        print(f"{k}: expected FETCH {v['fetch'] / 1e6:.1f} MB, WRITE {v['write'] / 1e6:.1f} MB per launch ({v['loads']})")
    sys.exit(0)

from gen3c_amd import ops, renderer  # noqa: E402

dev = torch.device("cuda:0")
x = torch.zeros(N_ADD, dtype=torch.bfloat16, device=dev)
y = torch.ones(N_ADD, dtype=torch.bfloat16, device=dev)
depth = (torch.rand(NIMG, 1, H, W, device=dev) * 3 + 1)
K = torch.tensor([[1000.0, 0, W / 2], [0, 1000.0, H / 2], [0, 0, 1]], device=dev).expand(NIMG, 3, 3).contiguous()
w2c = torch.eye(4, device=dev).expand(NIMG, 4, 4).contiguous()
for _ in range(3):
    ops.add_inplace(x, y)
    renderer.reliable_depth_mask_range_batch(depth, ratio_thresh=0.05)
    renderer.unproject_points(depth, w2c, K)
torch.cuda.synchronize()
print("done")
