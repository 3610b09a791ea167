"""Multi-view key frames -> GEN3C video on the MI355X path: counterpart of
cosmos_predict1/diffusion/inference/gen3c_multiview.py (:180-268).

--npz_path holds (same keys as the reference): images_key_frames [N,3,H,W] in [-1,1], depth_key_frames [N,1,H,W],
mask_key_frames [N,1,H,W], K_key_frames [N,3,3], w2cs_key_frames [N,4,4], w2cs_all [T,4,4] (the trajectory to render) and
optionally Ks_all [T,3,3] (else the last key frame's K). The N key frames form a Cache3D_BufferSelector: per target frame
the 2 buffers with the largest mask overlap are rendered (cache_3d.py:346-421). Chunks after the first reuse the cache
unchanged and are conditioned on the last generated frame. T5 embeddings are inputs (see cli_common)."""
from __future__ import annotations

import argparse

import numpy as np
import torch

from gen3c_amd.cli_common import Session, add_common_args


def create_parser() -> argparse.ArgumentParser:
    p = add_common_args(argparse.ArgumentParser(description="GEN3C multi-view key frames -> video on MI355X"))
    p.add_argument("--npz_path", type=str, required=True)
    # declared by the reference's parser (gen3c_multiview.py:50-85) although the trajectory comes from the npz; accepted, unused there too
    p.add_argument("--trajectory", type=str, default="left",
                   choices=["left", "right", "up", "down", "zoom_in", "zoom_out", "clockwise", "counterclockwise", "none"])
    p.add_argument("--camera_rotation", type=str, default="center_facing", choices=["center_facing", "no_rotation", "trajectory_aligned"])
    p.add_argument("--movement_distance", type=float, default=0.3)
    p.add_argument("--noise_aug_strength", type=float, default=0.0)
    return p


def demo(args) -> np.ndarray:
    from gen3c_amd import renderer
    ses = Session(args)
    dev = ses.dev
    npz = np.load(args.npz_path)
    t = lambda k: torch.tensor(npz[k], dtype=torch.float32, device=dev)
    images_key, depth_key, mask_key, K_key, w2c_key = (t(k) for k in ("images_key_frames", "depth_key_frames", "mask_key_frames",
                                                                      "K_key_frames", "w2cs_key_frames"))
    cache = renderer.Cache3D_BufferSelector(frame_buffer_max=2, input_image=images_key[None], input_depth=depth_key[None], input_mask=mask_key[None],
                                            input_w2c=w2c_key[None], input_intrinsics=K_key[None], filter_points_threshold=args.filter_points_threshold,
                                            input_format=["B", "N", "C", "H", "W"], foreground_masking=args.foreground_masking)
    cache.shard_group = ses.cp_group  # multi-GPU: every rank renders its share of the item pairs
    w2cs = t("w2cs_all")[: args.num_video_frames][None]
    assert w2cs.shape[1] == args.num_video_frames, f"w2cs_all holds {w2cs.shape[1]} poses, --num_video_frames {args.num_video_frames}"
    Ks = t("Ks_all")[: args.num_video_frames][None] if "Ks_all" in npz.files else K_key[-1][None, None].repeat(1, w2cs.shape[1], 1, 1)

    def render(start: int, _last01):  # no cache update between chunks (gen3c_multiview.py:253-259)
        return cache.render_cache(w2cs[:, start:start + ses.chunk], Ks[:, start:start + ses.chunk])

    video = ses.finalize(ses.run_chunks(images_key[None, 0][:, :, None], render))
    ses.save(video)
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
