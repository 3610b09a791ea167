"""Cache renderer at 704x1280 in the PRODUCT configuration (window splat, foreground masking on / off), 32 items x 3 repetitions: the command
behind profiles/r3_render_kernel_stats.csv (single-configuration rocprofv3 profile)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from gen3c_amd import ops, renderer  # noqa: E402
from bench_render import scene  # noqa: E402

dev = torch.device("cuda:0")
h, w, F = 704, 1280, 32
depth, img, K = scene(h, w)
t = lambda a: torch.from_numpy(a).to(dev)
w2cs = torch.eye(4, device=dev).repeat(1, F, 1, 1)
w2cs[0, :, 0, 3] = torch.linspace(0, 0.3, F, device=dev)
Ks = t(K)[None, None].expand(1, F, 3, 3).contiguous()
import os
CONFIGS = ((False, 1), (True, 1)) + (((True, 0), (True, 1)) if os.environ.get("G3_RENDER_AB") else ())
if os.environ.get("G3_RENDER_ONLY_FG"):  # bench.py's in-run traffic passes: the benchmarked configuration only (4 renders x 32 items)
    CONFIGS = ((True, 1),)
FUSED = int(os.environ.get("G3_RENDER_FUSED_ARM", "1"))  # 1: projection inside the splat (round 5, default); 0: z / flow / validity planes
for fg, overlap in CONFIGS:
    ops.set_option("render_overlap", overlap)
    ops.set_option("render_fused", FUSED)
    cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=t(img)[None], input_depth=t(depth)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                    input_intrinsics=t(K)[None], filter_points_threshold=0.05, foreground_masking=fg, input_format=["B", "C", "H", "W"])
    cache.render_cache(w2cs, Ks)
    torch.cuda.synchronize()
    tm = ops.HipTimer()
    tm.start()
    for _ in range(3):
        pix, msk = cache.render_cache(w2cs, Ks)
    tm.stop()
    per_item = tm.elapsed_ms() / 3 / F
    print(f"render 704x1280 foreground_masking={fg} occlusion_on_side_stream={overlap} fused_projection={FUSED}: {per_item:.4f} ms/item = {43.2e6 / (per_item * 1e-3) / 1e9:.0f} GB/s algorithmic; coverage {float(msk.mean()):.3f}", flush=True)
