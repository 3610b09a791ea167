"""Resident GEN3C model for repeated requests: counterpart of
cosmos_predict1/diffusion/inference/gen3c_persistent.py:55-569 (`Gen3cPersistentModel`), the object the reference's API
server keeps per GPU worker. Models are built once; `seed_model_from_values` builds the 3D cache from user images,
`inference_on_cameras` renders + generates along user cameras (autoregressive when more than 121 frames are requested).

Differences, all because the models around the path are inputs here: there is no MoGe, so seeding from a single
image REQUIRES `depths_np` (the reference instead refuses it and predicts depth), and the per-chunk depth of autoregressive
single-image requests comes from `depth_estimator` (callable image[3,H,W] in [0,1] -> (depth[1,1,H,W], mask)) or, by default,
from the cache's own rendering; the text prompt follows cli_common.TextEmbedder
(embedding file, T5 checkpoint, or dummy zeros); the video is written as .mp4 when an encoder is importable, else .npz."""
from __future__ import annotations

import argparse
import os
import time
from typing import Callable, Optional

import numpy as np
import torch

from gen3c_amd import renderer
from gen3c_amd.cli_common import Session, add_common_args


def create_parser() -> argparse.ArgumentParser:
    p = add_common_args(argparse.ArgumentParser(description="GEN3C persistent model on MI355X"))
    p.add_argument("--noise_aug_strength", type=float, default=0.0)
    return p


def resize_intrinsics(intrinsics, old_size, new_size, crop_size=None):
    """gen3c_persistent.py:35-52: scale fx, cx by the width ratio and fy, cy by the height ratio; sizes are (height, width)."""
    out = intrinsics.clone() if isinstance(intrinsics, torch.Tensor) else np.array(intrinsics, copy=True)
    out[..., 0, :] *= new_size[1] / old_size[1]
    out[..., 1, :] *= new_size[0] / old_size[0]
    if crop_size is not None:
        out[..., 0, 2] -= (new_size[1] - crop_size[1]) / 2
        out[..., 1, 2] -= (new_size[0] - crop_size[0]) / 2
    return out


class Gen3cPersistentModel:
    @torch.no_grad()
    def __init__(self, args: argparse.Namespace, depth_estimator: Optional[Callable] = None):
        self.session = Session(args)
        self.args = args
        self.frames_per_batch = self.sample_n_frames = self.session.chunk
        self.inference_overlap_frames = 1
        self.frame_buffer_max = self.session.model.frame_buffer_max
        self.device = self.session.dev
        self.generator = torch.Generator(device=self.device).manual_seed(args.seed)
        self.depth_estimator = depth_estimator
        self.pipeline = self.session.pipe
        self.cache = None
        self.model_was_seeded = False
        self.seeding_image: Optional[torch.Tensor] = None  # [B, C, T, H, W] in [-1, 1]

    # ---- seeding (gen3c_persistent.py:138-268)
    @torch.no_grad()
    def seed_model_from_values(self, images_np, depths_np, world_to_cameras_np, focal_lengths_np, principal_point_rel_np, resolutions,
                               masks_np=None):
        n = images_np.shape[0]
        assert images_np.shape[-1] == 3
        assert world_to_cameras_np.shape == (n, 4, 4) and focal_lengths_np.shape == (n, 2)
        assert principal_point_rel_np.shape == (n, 2) and resolutions.shape == (n, 2)
        assert (depths_np is None) or (depths_np.shape == images_np.shape[:-1])
        assert (masks_np is None) or (masks_np.shape == images_np.shape[:-1])
        if depths_np is None:
            raise NotImplementedError("no monocular depth model in this build: seeding needs depths_np (the reference predicts it with MoGe)")
        dev = self.device
        K = np.zeros((n, 3, 3), dtype=np.float32)
        K[:, 0, 0], K[:, 1, 1] = focal_lengths_np[:, 0], focal_lengths_np[:, 1]
        K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = principal_point_rel_np[:, 0] * self.W, principal_point_rel_np[:, 1] * self.H, 1.0
        image = torch.from_numpy(images_np.transpose(0, 3, 1, 2).astype(np.float32)).to(dev) * 2.0 - 1.0   # 0..1 -> -1..1
        depth = torch.from_numpy(depths_np[:, None].astype(np.float32)).to(dev)
        w2c = torch.from_numpy(world_to_cameras_np.astype(np.float32)).to(dev)
        Kt = torch.from_numpy(K).to(dev)
        common = dict(filter_points_threshold=self.args.filter_points_threshold, foreground_masking=self.args.foreground_masking)
        if n == 1:
            self.cache = renderer.Cache3D_Buffer(frame_buffer_max=self.frame_buffer_max, generator=self.generator,
                                                 noise_aug_strength=self.args.noise_aug_strength, input_image=image, input_depth=depth,
                                                 input_w2c=w2c, input_intrinsics=Kt, input_format=["B", "C", "H", "W"], **common)
            seeding = torch.from_numpy(images_np[0].transpose(2, 0, 1)[None].astype(np.float32) * 255.0 / 128.0 - 1.0).to(dev)  # x/128-1 (:202)
        else:
            if masks_np is None:
                raise NotImplementedError("Seeding from multiple frames requires providing mask values.")
            mask = torch.from_numpy(masks_np[:, None].astype(np.float32)).to(dev)
            self.cache = renderer.Cache4D(input_image=image.clone(), input_depth=depth, input_mask=mask, input_w2c=w2c, input_intrinsics=Kt,
                                          input_format=["F", "C", "H", "W"], **common)
            seeding = image
        self.cache.shard_group = self.session.cp_group  # multi-GPU: render item pairs are split over the ranks
        if seeding.shape[2] != self.H or seeding.shape[3] != self.W:
            seeding = torch.nn.functional.interpolate(seeding, size=(self.H, self.W), mode="bicubic", antialias=True, align_corners=False)
        self.seeding_image = seeding[:, :, None]
        self.model_was_seeded = True
        return world_to_cameras_np, focal_lengths_np, principal_point_rel_np, np.tile([[self.W, self.H]], (n, 1))

    # ---- inference (gen3c_persistent.py:272-516)
    def _depth_for_frame(self, frame_hwc_uint8, w2c, K):
        pred01 = torch.from_numpy(np.ascontiguousarray(frame_hwc_uint8)).to(self.device).permute(2, 0, 1).to(torch.float32) / 255.0
        if self.depth_estimator is not None:
            d, m = self.depth_estimator(pred01)
        else:  # the cache's own geometry at that camera, holes filled with the median
            d, m = self.cache.render_cache(w2c[:, None], K[:, None], render_depth=True)
            d, m = d[:, 0, 0], m[:, 0, 0, 0] > 0
            d = torch.where(m, d, d[m].median() if bool(m.any()) else d.new_tensor(1.0))[:, None]
        return d, m, pred01

    @torch.no_grad()
    def inference_on_cameras(self, view_cameras_w2cs, view_camera_intrinsics, fps, overlap_frames: int = 1, return_estimated_depths: bool = False,
                             video_save_quality: int = 5, save_buffer: Optional[bool] = None):
        assert self.model_was_seeded, "seed_model_from_values first"
        ses, n_chunk = self.session, self.sample_n_frames
        self.pipeline.fps = int(fps)
        save_buffer = self.args.save_buffer if save_buffer is None else save_buffer
        name = self.args.video_save_name or f"video_{time.strftime('%Y-%m-%d_%H-%M-%S')}"
        multiframe = isinstance(self.cache, renderer.Cache4D)
        w2cs, Ks = self.prepare_camera_for_inference(view_cameras_w2cs, view_camera_intrinsics, (self.H, self.W), (self.H, self.W))
        n_total = w2cs.shape[1]
        num_ar = (n_total - overlap_frames) // (n_chunk - overlap_frames)
        renders, masks = self.cache.render_cache(w2cs[:, :n_chunk], Ks[:, :n_chunk], start_frame_idx=0)
        warps = [renders.clone().cpu()] if save_buffer else []
        depths = []
        start_img = self.seeding_image[0:1] if multiframe else self.seeding_image
        video = self.pipeline.generate_from_embeddings(ses._emb, start_img.to(torch.bfloat16), renders, masks, negative_prompt_embedding=ses._neg)
        pred_depth = pred01 = None
        if return_estimated_depths or (num_ar > 1 and not multiframe):
            idx = min(n_chunk - overlap_frames, n_total - 1)
            pred_depth, _, pred01 = self._depth_for_frame(video[-1], w2cs[:, idx], Ks[:, idx])
            if return_estimated_depths:
                d0 = np.full((video.shape[0], 1, self.H, self.W), np.nan, dtype=np.float32)
                d0[-1] = pred_depth.cpu().numpy()[0]
                depths.append(d0)
        for it in range(1, num_ar):
            start = it * (n_chunk - overlap_frames)
            end = start + n_chunk
            cache_start = 0
            if multiframe:
                pred01 = torch.from_numpy(video[-1]).to(self.device).permute(2, 0, 1).to(torch.float32) / 255.0
                cache_start = min(start, self.cache.input_frame_count() - (end - start))  # hold on the last window (:407-412)
            else:
                self.cache.update_cache(new_image=pred01[None] * 2 - 1, new_depth=pred_depth, new_w2c=w2cs[:, start], new_intrinsics=Ks[:, start])
            renders, masks = self.cache.render_cache(w2cs[:, start:end], Ks[:, start:end], start_frame_idx=cache_start)
            if save_buffer:
                warps.append(renders[:, overlap_frames:].clone().cpu())
            video_new = self.pipeline.generate_from_embeddings(ses._emb, (pred01[None, :, None] * 2 - 1).to(torch.bfloat16), renders, masks,
                                               negative_prompt_embedding=ses._neg)
            video = np.concatenate([video, video_new[overlap_frames:]], axis=0)
            if return_estimated_depths or (it < num_ar - 1 and not multiframe):
                idx = min(end - overlap_frames, n_total - 1)
                pred_depth, _, pred01 = self._depth_for_frame(video_new[-1], w2cs[:, idx], Ks[:, idx])
            if return_estimated_depths:
                di = np.full((video_new.shape[0] - overlap_frames, 1, self.H, self.W), np.nan, dtype=np.float32)
                di[-1] = pred_depth.cpu().numpy()[0]
                depths.append(di)
        ses.rendered_warps = warps
        final = ses.finalize(video, save_buffer)
        ses.rendered_warps = []
        ses.save(final, name)
        video_b = video.transpose(0, 3, 1, 2)[None]  # [1, n_frames, C, H, W] (:497)
        return {"rendered_warp_images": renders, "video": video_b, "rendered_warp_images_no_overlap": renders, "video_no_overlap": video_b,
                "predicted_depth": np.concatenate(depths, axis=0) if return_estimated_depths else None,
                "video_save_path": getattr(ses, "saved_path", os.path.join(self.args.video_save_folder, name + ".npz"))}

    # ---- helpers (gen3c_persistent.py:518-569)
    def prepare_camera_for_inference(self, view_cameras, view_camera_intrinsics, old_size, new_size):
        if isinstance(view_cameras, np.ndarray):
            view_cameras = torch.from_numpy(view_cameras).float().contiguous()
        if view_cameras.ndim == 3:
            view_cameras = view_cameras.unsqueeze(0)
        if isinstance(view_camera_intrinsics, np.ndarray):
            view_camera_intrinsics = torch.from_numpy(view_camera_intrinsics).float().contiguous()
        view_camera_intrinsics = resize_intrinsics(view_camera_intrinsics, old_size, new_size).unsqueeze(0)
        assert view_camera_intrinsics.ndim == 4
        return view_cameras.to(self.device), view_camera_intrinsics.to(self.device)

    def get_cache_input_depths(self):
        return None if self.cache is None else self.cache.input_depth

    @property
    def W(self) -> int:
        return self.args.width

    @property
    def H(self) -> int:
        return self.args.height

    def clear_cache(self) -> None:
        self.cache = None
        self.model_was_seeded = False

    def cleanup(self) -> None:
        if self.args.num_gpus > 1:
            import torch.distributed as dist
            from gen3c_amd.parallel import parallel_state
            parallel_state.destroy_model_parallel()
            dist.destroy_process_group()
