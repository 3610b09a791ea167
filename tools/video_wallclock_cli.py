"""Wall-clock of BASELINE.json configs 4 and 5 on ONE MI355X through the CLI entry points themselves (random-init weights, synthetic
inputs at 704x1280): config 4 = single image, 361 frames = 3 autoregressive chunks with cache update + depth alignment between them;
config 5 = 3 input views fused by Cache3D_BufferSelector with --foreground_masking, 121 frames.
usage (GPU box): python tools/video_wallclock_cli.py [--steps 35] [--configs 4,5]   -> gpurun_out/r2_video_wallclock.json"""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=35)
ap.add_argument("--configs", type=str, default="4,5")
ap.add_argument("--out", type=str, default="gpurun_out/r2_video_wallclock.json")
a = ap.parse_args()
H, W = 704, 1280
tmp = Path(tempfile.mkdtemp(prefix="g3wall_"))
ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
K = np.array([[1000.0, 0, W / 2], [0, 1000.0, H / 2], [0, 0, 1]], np.float32)
res = {"steps": a.steps, "resolution": [H, W], "weights": "random-init Cosmos-7B / CV8x8x8 tokenizer", "runs": {}}


def scene_depth(shift=0.0):
    d = 3.0 + 0.001 * xs + 0.0005 * ys + shift
    d[((xs - 400) ** 2 + (ys - 300) ** 2) < 120 ** 2] = 1.5 + shift
    d[((xs - 900) ** 2 + (ys - 420) ** 2) < 90 ** 2] = 2.2 + shift
    return d.astype(np.float32)


def run(name, demo, args, frames):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    video = demo(args)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert video.shape[0] == frames, video.shape
    res["runs"][name] = {"seconds_total_incl_model_build": round(dt, 2), "frames": frames, "chunks": (frames - 1) // 120, "video_shape": list(video.shape),
                         "finite": bool(np.isfinite(video.astype(np.float32)).all()), "seconds_per_chunk": round(dt / ((frames - 1) // 120), 2)}
    print(name, json.dumps(res["runs"][name]), flush=True)


if "4" in a.configs.split(","):
    from PIL import Image
    from gen3c_amd import gen3c_single_image as cli
    Image.fromarray((np.stack([np.sin(xs / 37.0), np.cos(ys / 23.0), np.sin((xs + ys) / 51.0)], -1) * 127 + 128).astype(np.uint8)).save(tmp / "in.png")
    np.savez(tmp / "depth.npz", depth=scene_depth(), intrinsics=K)
    args = cli.create_parser().parse_args(["--input_image_path", str(tmp / "in.png"), "--depth_path", str(tmp / "depth.npz"), "--num_steps", str(a.steps), "--guidance", "1",
                                           "--random_init", "--video_save_folder", str(tmp / "out4"), "--video_save_name", "cfg4", "--num_video_frames", "361",
                                           "--trajectory", "left", "--foreground_masking"])
    run("config4_single_image_361_frames_3_chunks", cli.demo, args, 361)

if "5" in a.configs.split(","):
    from gen3c_amd import gen3c_multiview as cli
    N, T = 3, 121
    imgs = np.stack([np.stack([np.sin(xs / (37 + 5 * f)), np.cos(ys / (23 + 3 * f)), np.sin((xs + ys) / 51)]) for f in range(N)]).astype(np.float32)
    depth = np.stack([scene_depth(0.05 * f)[None] for f in range(N)])
    mask = np.ones((N, 1, H, W), np.float32)
    Ks = np.repeat(K[None], N, 0)
    w2c = np.repeat(np.eye(4, dtype=np.float32)[None], N, 0)
    w2c[:, 0, 3] = -0.15 * np.arange(N)
    w2cs_all = np.repeat(np.eye(4, dtype=np.float32)[None], T, 0)
    w2cs_all[:, 0, 3] = -0.3 * np.arange(T) / (T - 1)
    np.savez(tmp / "mv.npz", images_key_frames=imgs, depth_key_frames=depth, mask_key_frames=mask, K_key_frames=Ks, w2cs_key_frames=w2c, w2cs_all=w2cs_all)
    args = cli.create_parser().parse_args(["--npz_path", str(tmp / "mv.npz"), "--num_steps", str(a.steps), "--guidance", "1", "--random_init", "--num_video_frames", str(T),
                                           "--foreground_masking", "--video_save_folder", str(tmp / "out5"), "--video_save_name", "cfg5"])
    run("config5_multiview_3_views_foreground_masking_121_frames", cli.demo, args, T)

Path(a.out).parent.mkdir(parents=True, exist_ok=True)
Path(a.out).write_text(json.dumps(res, indent=1))
print(json.dumps(res))
