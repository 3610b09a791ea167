"""A/B of g3_render_items_f32's side-stream occlusion pass (option render_overlap) on ONE cache: alternating settings, 10 renders each."""
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
cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=t(img)[None], input_depth=t(depth)[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                input_intrinsics=t(K)[None], filter_points_threshold=0.05, foreground_masking=True, input_format=["B", "C", "H", "W"])
for _ in range(3):
    cache.render_cache(w2cs, Ks)
import os
TWO = os.environ.get("AB") == "two_streams"  # A/B the two-stream chunk halves instead of the side-stream occlusion pass
res = {0: [], 1: []}
for rnd in range(4):
    for ov in (0, 1):
        renderer._TWO_STREAM_CHUNKS = bool(ov) if TWO else renderer._TWO_STREAM_CHUNKS
        if not TWO:
            ops.set_option("render_overlap", ov)
        cache.render_cache(w2cs, Ks)
        torch.cuda.synchronize()
        tm = ops.HipTimer()
        tm.start()
        for _ in range(10):
            cache.render_cache(w2cs, Ks)
        tm.stop()
        res[ov].append(tm.elapsed_ms() / 10 / F)
for ov in (0, 1):
    v = res[ov]
    print(f"{'chunk halves on two streams' if TWO else 'occlusion pass on a side stream'} = {ov}: ms/item per round {['%.4f' % x for x in v]}  mean {sum(v) / len(v):.4f} = {43.2e6 / (sum(v) / len(v) * 1e-3) / 1e9:.0f} GB/s")
