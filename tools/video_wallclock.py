"""Wall-clock of one full-size 121-frame GEN3C video on one MI355X (cache build + render + tokenizer encodes +
35 denoise steps + decode), random-init weights, synthetic image/depth.  usage (GPU box): python tools/video_wallclock.py [--steps 35]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import renderer  # noqa: E402
from gen3c_amd.camera_utils import generate_camera_trajectory  # noqa: E402
from gen3c_amd.dit import VideoExtendGeneralDIT  # noqa: E402
from gen3c_amd.pipeline import DiffusionGen3CModel, Gen3cPipeline  # noqa: E402
from gen3c_amd.tokenizer import VideoTokenizer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=35)
ap.add_argument("--out", type=str, default="gpurun_out/video_wallclock.json")
args = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W, T = 704, 1280, 121
times = {}


def timed(name, fn, *a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn(*a, **k)
    torch.cuda.synchronize()
    times[name] = times.get(name, 0.0) + time.perf_counter() - t0
    return r


def build():
    net = VideoExtendGeneralDIT(in_channels=16 + 16 * 4 + 1, rope_t_extrapolation_ratio=2.0, device=dev, init_weights=False)
    net.initialize_weights(randomize_adaln=True, seed=1)
    tk = VideoTokenizer(pixel_chunk_duration=T, device=dev)
    tk.net.init_random(seed=1)
    tk.register_mean_std(torch.zeros(16, 32), torch.ones(16, 32))
    return net, tk


net, tk = timed("model_build_random_init", build)
model = DiffusionGen3CModel(net, tk, latent_shape=(16, tk.get_latent_num_frames(T), H // 8, W // 8))
pipe = Gen3cPipeline(model, guidance=1.0, num_steps=args.steps, height=H, width=W, fps=24, num_video_frames=T, seed=1)

ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev), indexing="ij")
depth = 3.0 + 0.001 * xs + 0.0005 * ys
depth[((xs - 400) ** 2 + (ys - 300) ** 2) < 120 ** 2] = 1.5
depth[((xs - 900) ** 2 + (ys - 420) ** 2) < 90 ** 2] = 2.2
img = torch.stack([torch.sin(xs / 37.0), torch.cos(ys / 23.0), torch.sin((xs + ys) / 51.0)])[None]
K = torch.tensor([[1000.0, 0, 640], [0, 1000.0, 352], [0, 0, 1]], device=dev)
w2c0 = torch.eye(4, device=dev)

cache = timed("cache_build", renderer.Cache3D_Buffer, frame_buffer_max=2, noise_aug_strength=0.0, input_image=img, input_depth=depth[None, None],
              input_w2c=w2c0[None], input_intrinsics=K[None], filter_points_threshold=0.05, foreground_masking=True, input_format=["B", "C", "H", "W"])
w2cs, Ks = generate_camera_trajectory("left", w2c0, K, T, 0.3, "center_facing", center_depth=3.0, device=dev)
renders, masks = timed("render_121_frames", cache.render_cache, w2cs, Ks)

for name in ("encode", "decode"):
    orig = getattr(model, name)
    setattr(model, name, (lambda o, n: (lambda *a, **k: timed("tokenizer_" + n, o, *a, **k)))(orig, name))
cond_image = (img[:, :, None] * 0.99).to(torch.bfloat16)
emb = torch.zeros(1, 512, net.crossattn_emb_channels, dtype=torch.bfloat16)
video = timed("pipeline_generate_total", pipe.generate_from_embeddings, emb, cond_image, renders, masks)
assert video.shape == (T, H, W, 3)
times["denoise_loop_and_glue"] = times["pipeline_generate_total"] - times.get("tokenizer_encode", 0) - times.get("tokenizer_decode", 0)
times["video_total_excl_model_build"] = times["cache_build"] + times["render_121_frames"] + times["pipeline_generate_total"]
res = {"workload": f"121x704x1280 video, Cosmos-7B random-init, {args.steps} steps, guidance 1, 1 cache buffer, foreground masking", "seconds": {k: round(v, 3) for k, v in times.items()},
       "steps_per_sec": round(args.steps / times["denoise_loop_and_glue"], 4), "video_finite": bool(np.isfinite(video.astype(np.float32)).all()),
       "video_mean": float(video.mean())}
print(json.dumps(res))
Path(args.out).parent.mkdir(parents=True, exist_ok=True)
Path(args.out).write_text(json.dumps(res, indent=1))
