"""Dynamic video (per-frame RGB-D + cameras) -> novel-trajectory GEN3C video on the MI355X path: counterpart of
cosmos_predict1/diffusion/inference/gen3c_dynamic.py (:190-320).

--input_image_path is a packaged .pt or a directory in the distributed format (gen3c_amd/data_loader_utils.py); every
source frame f becomes the 3D cache of target frame f (Cache4D, cache_3d.py:424-433; chunks render their own window through
start_frame_idx). The camera trajectory starts from the first frame's pose. T5 embeddings are inputs (see cli_common)."""
from __future__ import annotations

import argparse

import numpy as np
import torch

from gen3c_amd.cli_common import Session, add_common_args


def create_parser() -> argparse.ArgumentParser:
    p = add_common_args(argparse.ArgumentParser(description="GEN3C dynamic video -> video on MI355X"))
    p.add_argument("--input_image_path", type=str, required=True, help="packaged .pt or directory (rgb.npz|rgb.mp4, depth.npz, mask.npz, camera.npz)")
    p.add_argument("--trajectory", type=str, default="left",
                   choices=["left", "right", "up", "down", "zoom_in", "zoom_out", "clockwise", "counterclockwise"])
    p.add_argument("--camera_rotation", type=str, default="center_facing", choices=["center_facing", "no_rotation", "trajectory_aligned"])
    p.add_argument("--movement_distance", type=float, default=0.3)
    return p


def demo(args) -> np.ndarray:
    from gen3c_amd import renderer
    from gen3c_amd.camera_utils import generate_camera_trajectory
    from gen3c_amd.data_loader_utils import load_data_auto_detect
    ses = Session(args)
    dev = ses.dev
    image, depth, mask, w2c, K = (x.to(dev, torch.float32) for x in load_data_auto_detect(args.input_image_path))
    assert image.shape[0] >= args.num_video_frames, f"{image.shape[0]} source frames < --num_video_frames {args.num_video_frames}"
    cache = renderer.Cache4D(input_image=image.clone(), input_depth=depth, input_mask=mask, input_w2c=w2c, input_intrinsics=K,
                             filter_points_threshold=args.filter_points_threshold, input_format=["F", "C", "H", "W"],
                             foreground_masking=args.foreground_masking)
    w2cs, Ks = generate_camera_trajectory(args.trajectory, w2c[0], K[0], args.num_video_frames, args.movement_distance, args.camera_rotation,
                                          center_depth=1.0, device=dev)

    def render(start: int, _last01):
        return cache.render_cache(w2cs[:, start:start + ses.chunk], Ks[:, start:start + ses.chunk], start_frame_idx=start)

    video = ses.finalize(ses.run_chunks(image[0][None, :, None], render))
    ses.save(video)
    return video


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    demo(create_parser().parse_args())
