"""Dynamic video (per-frame RGB-D + cameras) -> novel-trajectory GEN3C video on the MI355X path: counterpart of
cosmos_predict1/diffusion/inference/gen3c_dynamic.py (:190-320).

--input_image_path is a packaged .pt or a directory in the distributed format (gen3c_amd/data_loader_utils.py), or --vipe_path a ViPE
result folder / clip (gen3c_amd/vipe_utils.py; --vipe_starting_frame_idx, resized to 720x1280 and centre-cropped to 704x1280 like
gen3c_dynamic.py:200-212); every source frame f becomes the 3D cache of target frame f (Cache4D, cache_3d.py:424-433; chunks render
their own window through start_frame_idx). The camera trajectory starts from the first frame's pose. All reference flags are accepted
(cli_common.add_common_args); text conditions: cli_common.TextEmbedder."""
from __future__ import annotations

import argparse

import numpy as np
import torch

from gen3c_amd.cli_common import Session, add_common_args


def create_parser() -> argparse.ArgumentParser:
    p = add_common_args(argparse.ArgumentParser(description="GEN3C dynamic video -> video on MI355X"))
    p.add_argument("--input_image_path", type=str, default=None, help="packaged .pt or directory (rgb.npz|rgb.mp4, depth.npz, mask.npz, camera.npz)")
    p.add_argument("--trajectory", type=str, default="left",
                   choices=["left", "right", "up", "down", "zoom_in", "zoom_out", "clockwise", "counterclockwise"])
    p.add_argument("--camera_rotation", type=str, default="center_facing", choices=["center_facing", "no_rotation", "trajectory_aligned"])
    p.add_argument("--movement_distance", type=float, default=0.3)
    p.add_argument("--vipe_path", type=str, default=None, help="ViPE result folder (rgb/ depth/ pose/ intrinsics/) or one of its rgb clips")
    p.add_argument("--vipe_starting_frame_idx", type=int, default=0)
    return p


def generate_one(ses: Session, args, source) -> np.ndarray:
    from gen3c_amd import renderer
    from gen3c_amd.camera_utils import generate_camera_trajectory
    dev = ses.dev
    image, depth, mask, w2c, K = (x.to(dev, torch.float32) for x in source)
    assert image.shape[0] >= args.num_video_frames, f"{image.shape[0]} source frames < --num_video_frames {args.num_video_frames}"
    cache = renderer.Cache4D(input_image=image.clone(), input_depth=depth, input_mask=mask, input_w2c=w2c, input_intrinsics=K,
                             filter_points_threshold=args.filter_points_threshold, input_format=["F", "C", "H", "W"],
                             foreground_masking=args.foreground_masking)
    cache.shard_group = ses.cp_group  # multi-GPU: every rank renders its share of the item pairs
    w2cs, Ks = generate_camera_trajectory(args.trajectory, w2c[0], K[0], args.num_video_frames, args.movement_distance, args.camera_rotation,
                                          center_depth=1.0, device=dev)
    ses.rendered_warps.clear()

    def render(start: int, _last01):
        return cache.render_cache(w2cs[:, start:start + ses.chunk], Ks[:, start:start + ses.chunk], start_frame_idx=start)

    return ses.finalize(ses.run_chunks(image[0][None, :, None], render))


def demo(args) -> np.ndarray:
    from gen3c_amd.cli_common import read_prompts_from_file
    from gen3c_amd.data_loader_utils import load_data_auto_detect
    from gen3c_amd.vipe_utils import load_vipe_data
    ses = Session(args)
    if args.batch_input_path:
        records = read_prompts_from_file(args.batch_input_path)
    else:
        records = [{"prompt": args.prompt, "visual_input": args.input_image_path or args.vipe_path}]  # gen3c_dynamic.py:181-186
    video = None
    for i, rec in enumerate(records):
        path = rec.get("visual_input")
        if path is None:
            print(f"[gen3c_amd] record {i}: visual input is missing, skipping world generation")
            continue
        try:
            if args.vipe_path is not None:
                source = load_vipe_data(args.vipe_path, args.vipe_starting_frame_idx, resize_hw=(720, 1280), crop_hw=(704, 1280),
                                        num_frames=args.num_video_frames)
            else:
                source = load_data_auto_detect(path)
        except Exception as e:  # gen3c_dynamic.py:223-225: a bad record is reported and skipped
            print(f"[gen3c_amd] record {i}: failed to load visual input from {path}: {e}")
            continue
        ses.set_prompt(rec.get("prompt"))
        video = generate_one(ses, args, source)
        ses.save(video, name=str(i) if args.batch_input_path else None)
    ses.close()
    return video


def main(argv=None) -> None:
    torch.set_grad_enabled(False)
    args = create_parser().parse_args(argv)
    if args.prompt is None:
        args.prompt = ""
    demo(args)


if __name__ == "__main__":
    main()
