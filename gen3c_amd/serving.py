"""Model side of the GEN3C serving boundary: what the reference's API server calls on the object it keeps resident.

  reference                                                      here
  gui/api/server_base.py:30-204      InferenceModel              InferenceModel   request bookkeeping: request_inference -> asyncio task,
                                                                                  inference_result_or_none, result cache + eviction, validity
  gui/api/server_cosmos_base.py:31-267  CosmosBaseModel          Gen3cInferenceModel.seed_model / run_inference: api_types records <->
  gui/api/server_cosmos.py:49-138       CosmosModel              Gen3cPersistentModel.seed_model_from_values / inference_on_cameras, the pose
                                                                                  history that lets consecutive requests overlap by one frame
  gui/api/server_debug.py:22-114     DebugInferenceModel         DebugInferenceModel (deterministic stand-in; the contract test drives both)

The HTTP layer (FastAPI routes /seed-model, /request-inference, /inference-result, ...; gui/api/server.py:123-236) and the per-GPU worker
processes with their queues (gui/api/multi_gpu.py:40-354) are control plane and are NOT built: a multi-GPU resident model here is N ranks of
`Gen3cPersistentModel` under torchrun (context parallel inside the model), and whoever fronts it calls the methods below.
"""
from __future__ import annotations

import asyncio
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

from .api_types import (CompressedInferenceResult, CompressionFormat, InferenceRequest, InferenceResult, SeedingRequest, SeedingResult,
                        compress_images)


class InferenceModel:
    """Request bookkeeping shared by every servable model (server_base.py:30-204). Subclasses implement seed_model / run_inference / metadata
    and the frame-count limits."""

    def __init__(self, data_path: Optional[str] = None, checkpoint_path: Optional[str] = None, fake_delay_ms: float = 0,
                 inference_cache_size: int = 15, compress_inference_results: bool = True) -> None:
        self.data_path, self.checkpoint_path = data_path, checkpoint_path
        self.fake_delay_ms, self.inference_cache_size = fake_delay_ms, inference_cache_size
        self.inference_tasks: Dict[str, asyncio.Task] = {}
        self.inference_results: Dict[str, InferenceResult] = {}
        self.request_history: set = set()
        self.compress_inference_results = compress_inference_results
        self.inference_lock = asyncio.Lock()  # one request at a time on the model
        self.model_seeded = False

    # -- to implement
    async def make_test_image(self):
        raise NotImplementedError("make_test_image")

    async def seed_model(self, req: SeedingRequest):
        self.model_seeded = True

    async def run_inference(self, req: InferenceRequest) -> InferenceResult:
        raise NotImplementedError("run_inference")

    def metadata(self) -> dict:
        raise NotImplementedError("metadata")

    def min_frames_per_request(self) -> int:
        raise NotImplementedError("min_frames_per_request")

    def max_frames_per_request(self) -> int:
        raise NotImplementedError("max_frames_per_request")

    def inference_time_per_frame(self) -> float:
        raise NotImplementedError("inference_time_per_frame")

    def inference_resolution(self) -> Optional[List[Tuple[int, int]]]:
        return None  # any (width, height)

    def default_framerate(self) -> Optional[float]:
        return None

    def requires_seeding(self) -> bool:
        return False

    def cleanup(self):
        pass

    # -- requests
    def check_valid_request(self, req: InferenceRequest) -> bool:
        lo, hi = self.min_frames_per_request(), self.max_frames_per_request()
        if not lo <= len(req) <= hi:
            raise ValueError(f"This model can produce between {lo} and {hi} frames per request, but the request specified {len(req)} camera poses.")
        return True

    def request_inference(self, req: InferenceRequest) -> asyncio.Task:
        if not self.model_seeded:
            raise ValueError(f"Received request id '{req.request_id}', but the model was not seeded.")
        if req.request_id in self.inference_tasks or req.request_id in self.inference_results:
            raise ValueError(f"Invalid request id '{req.request_id}': request already exists.")
        self.check_valid_request(req)
        task = asyncio.create_task(self.run_inference(req))
        self.inference_tasks[req.request_id] = task
        self.request_history.add(req.request_id)
        return task

    async def request_inference_sync(self, req: InferenceRequest) -> InferenceResult:
        await self.request_inference(req)
        result = self.inference_result_or_none(req.request_id)
        assert isinstance(result, InferenceResult)
        return result

    def inference_result_or_none(self, request_id: str) -> Optional[InferenceResult]:
        """Finished -> the result (moved into the bounded cache); still running -> None; failed -> its exception; unknown / evicted -> KeyError."""
        task = self.inference_tasks.get(request_id)
        if task is not None:
            if not task.done():
                return None
            result = task.result()  # raises what run_inference raised
            self.inference_results[request_id] = result
            del self.inference_tasks[request_id]
            self.evict_results()
            return result
        if request_id in self.inference_results:
            return self.inference_results[request_id]
        if request_id in self.request_history:
            raise KeyError(f"Request with id '{request_id}' was known, but does not have any result. Perhaps it was evicted from the cache or failed.")
        raise KeyError(f"Invalid request id '{request_id}': request not known.")

    def evict_results(self, keep_max: Optional[int] = None):
        keep = self.inference_cache_size if keep_max is None else keep_max
        for k in list(self.inference_results)[:max(0, len(self.inference_results) - keep)]:
            del self.inference_results[k]

    def get_latest_rgb(self) -> Optional[np.ndarray]:
        if not self.inference_results:
            return None
        return next(reversed(self.inference_results.values())).images[-1]


class DebugInferenceModel(InferenceModel):
    """Deterministic model without weights (server_debug.py:22-114): colour ramps whose blue channel / depth encode the frame index. The contract
    test runs the same call sequence against this and against Gen3cInferenceModel."""

    def __init__(self, *args, gpu_count: int = 0, **kwargs) -> None:
        super().__init__(*args, compress_inference_results=False, **kwargs)
        self.model_seeded = True
        self.aabb_min, self.aabb_max = np.full(3, -1.0, np.float32), np.full(3, 1.0, np.float32)

    async def make_test_image(self):
        req = InferenceRequest(request_id="debug-startup", timestamps=np.zeros(1, np.float32), cameras_to_world=np.zeros((1, 3, 4), np.float32),
                               focal_lengths=np.ones((1, 2), np.float32), principal_points=np.full((1, 2), 0.5, np.float32),
                               resolutions=np.array([[16, 8]], np.int32), return_depths=True)
        result = await self.run_inference(req)
        self.inference_results[req.request_id] = result
        self.request_history.add(req.request_id)
        return result

    async def seed_model(self, req: SeedingRequest) -> SeedingResult:
        self.model_seeded = True
        fallback = None
        if req.depths is None:
            w, h = req.resolution()
            fallback = np.ones((len(req), h, w), np.float32)
        return SeedingResult.from_request(req, fallback_depths=fallback)

    async def run_inference(self, req: InferenceRequest) -> InferenceResult:
        w, h = req.resolution()
        xx, yy = np.meshgrid(np.linspace(0.0, 1.0, w, dtype=np.float32), np.linspace(0.0, 1.0, h, dtype=np.float32))
        n = len(req)
        level = [np.float32((i + 1) / max(n, 1)) for i in range(n)]
        return InferenceResult(request_id=req.request_id, result_ids=[f"{req.request_id}__debug_{i}" for i in range(n)], timestamps=req.timestamps.copy(),
                               cameras_to_world=req.cameras_to_world.copy(), focal_lengths=req.focal_lengths.copy(), principal_points=req.principal_points.copy(),
                               resolutions=req.resolutions.copy(), frame_count_without_padding=req.frame_count_without_padding,
                               images=np.stack([np.stack([xx, yy, np.full_like(xx, v)], axis=-1) for v in level]),
                               depths=np.stack([np.full((h, w), v, np.float32) for v in level]), runtime_ms=0.0)

    def metadata(self) -> dict:
        return _metadata(self, "DebugInferenceModel")

    def min_frames_per_request(self) -> int:
        return 1

    def max_frames_per_request(self) -> int:
        return 16

    def inference_time_per_frame(self) -> float:
        return 0.0

    def inference_resolution(self) -> List[Tuple[int, int]]:
        return [(16, 8), (64, 32)]

    def default_framerate(self) -> float:
        return 24.0

    def requires_seeding(self) -> bool:
        return False


def _metadata(m: InferenceModel, name: str) -> dict:
    return {"model_name": name, "model_version": (1, 0, 0), "aabb_min": m.aabb_min.tolist(), "aabb_max": m.aabb_max.tolist(),
            "min_frames_per_request": m.min_frames_per_request(), "max_frames_per_request": m.max_frames_per_request(),
            "inference_resolution": m.inference_resolution(), "inference_time_per_frame": m.inference_time_per_frame(),
            "default_framerate": m.default_framerate(), "requires_seeding": m.requires_seeding()}


class Gen3cInferenceModel(InferenceModel):
    """The resident GEN3C model behind the request records (CosmosBaseModel + CosmosModel, server_cosmos_base.py:31-267, server_cosmos.py:49-138).

    `model`: a Gen3cPersistentModel (gen3c_amd/gen3c_persistent.py) - or anything with its surface: seed_model_from_values, inference_on_cameras,
    clear_cache, get_cache_input_depths, W, H, frames_per_batch, inference_overlap_frames. From the second request on, the last
    `inference_overlap_frames` cameras of the previous request are put in front of the new ones (and as many dropped at the end), the model
    regenerates those frames as its autoregressive overlap, and the result carries only the new frames."""

    def __init__(self, model, **kwargs):
        super().__init__(**kwargs)
        self.model = model
        self.pose_history_w2c: List[np.ndarray] = []
        self.intrinsics_history: List[np.ndarray] = []
        self.default_focal_length, self.default_principal_point = (338.29, 338.29), (0.5, 0.5)
        self.aabb_min, self.aabb_max = np.full(3, -16), np.full(3, 16)

    async def make_test_image(self) -> InferenceResult:
        raise NotImplementedError("Not implemented: make_test_image()")

    async def seed_model(self, req: SeedingRequest) -> SeedingResult:
        self.model.clear_cache()
        self.pose_history_w2c.clear()
        self.intrinsics_history.clear()
        got = self.model.seed_model_from_values(images_np=req.images, depths_np=req.depths, masks_np=req.masks, world_to_cameras_np=req.world_to_cameras(),
                                                focal_lengths_np=req.focal_lengths, principal_point_rel_np=req.principal_points, resolutions=req.resolutions)
        self.model_seeded = True
        out_depths = None
        if req.depths is None:
            out_depths = _to_numpy(self.model.get_cache_input_depths())
        if got is None:
            return SeedingResult.from_request(req, fallback_depths=out_depths)
        w2c, focal, pp_abs, res = (_to_numpy(g) for g in got)
        # as in the reference (server_cosmos_base.py:80-90): the estimated world-to-camera rows go out under `cameras_to_world`, and the third
        # value is divided by the working resolution (the model's contract calls it absolute)
        return SeedingResult(request_id=req.request_id, cameras_to_world=w2c[:, :3, :], focal_lengths=focal, principal_points=pp_abs / res,
                             resolutions=res, depths=out_depths)

    async def run_inference(self, req: InferenceRequest) -> InferenceResult:
        async with self.inference_lock:
            t0 = time.time()
            w2c = req.world_to_cameras()
            # intrinsics arrive in pixels of the REQUESTED resolution; the model wants pixels of its working resolution
            working = req.resolutions.copy()
            working[:, 0], working[:, 1] = self.model.W, self.model.H
            K = req.intrinsics_matrix(for_resolutions=working)
            overlap = 0
            if self.pose_history_w2c:
                overlap = self.model.inference_overlap_frames
                assert overlap < self.min_frames_per_request()
                w2c = np.concatenate([self.pose_history_w2c[-1][-overlap:], w2c[:-overlap]], axis=0)
                K = np.concatenate([self.intrinsics_history[-1][-overlap:], K[:-overlap]], axis=0)
            self.pose_history_w2c.append(w2c)
            self.intrinsics_history.append(K)
            out = self.model.inference_on_cameras(w2c, K, fps=req.framerate, overlap_frames=overlap, return_estimated_depths=req.return_depths,
                                                  video_save_quality=req.video_encoding_quality, save_buffer=req.show_cache_renderings)
            if isinstance(out, dict):
                frames, depth, video_path = out["video_no_overlap"], out["predicted_depth"], out.get("video_save_path")
            else:
                (_, _, _, frames, depth), video_path = out, None
            await self._device_idle()
            if self.fake_delay_ms > 0:
                await asyncio.sleep(self.fake_delay_ms / 1000.0)
        frames = _to_numpy(frames)
        if frames.ndim == 5:
            assert frames.shape[0] == 1, frames.shape
            frames = frames[0]
        depths = None
        if req.return_depths:
            depths = _to_numpy(depth)
            if depths.ndim == 4:
                assert depths.shape[1] == 1, depths.shape
                depths = depths[:, 0]
        # The resident model returns the overlap frame(s) it regenerated at the head of the batch ("video_no_overlap" is the whole video in the
        # reference too - gen3c_persistent.py:504-506 "TODO: handle overlap" - so its uncompressed record would carry n frames for n - overlap
        # cameras and trip its own shape check); here they are cut so that frames, depths and cameras agree.
        trimmed = overlap > 0 and frames.shape[0] == w2c.shape[0]
        if trimmed:
            frames = frames[overlap:]
            depths = depths[overlap:] if depths is not None else None
        images = frames.transpose(0, 2, 3, 1)  # [n, C, H, W] -> [n, H, W, C]
        n = images.shape[0]
        upper = -overlap if overlap > 0 else None
        common = dict(request_id=req.request_id, result_ids=[f"{req.request_id}__frame_{k}" for k in range(n)], timestamps=np.zeros((n,)),
                      cameras_to_world=req.cameras_to_world[:upper], focal_lengths=req.focal_lengths[:upper], principal_points=req.principal_points[:upper],
                      frame_count_without_padding=req.frame_count_without_padding, runtime_ms=1000 * (time.time() - t0))
        # The model's MP4 holds every frame it produced, the regenerated overlap frame(s) included: after a trim it would decompress to `overlap`
        # more images than there are cameras / depths (frame i paired with camera i - overlap). Such a request goes out uncompressed instead.
        if self.compress_inference_results and not trimmed and video_path is not None and str(video_path).endswith(".mp4"):
            with open(video_path, "rb") as f:
                video_bytes = f.read()
            return CompressedInferenceResult(images=None, depths=None, resolutions=np.tile([[images.shape[2], images.shape[1]]], (n, 1)),
                                             images_compressed=[video_bytes], images_format=CompressionFormat.MP4,
                                             depths_compressed=compress_images(depths, CompressionFormat.NPZ, is_depth=True), depths_format=CompressionFormat.NPZ, **common)
        return InferenceResult(images=images, depths=depths, **common)

    @staticmethod
    async def _device_idle():
        """Yield to the event loop until the GPU has finished the request (an event + polling instead of a blocking synchronize)."""
        try:
            import torch
            if not torch.cuda.is_available():
                return
            ev = torch.cuda.Event()
            ev.record()
            while not ev.query():
                await asyncio.sleep(0.0005)
        except ImportError:
            return

    def min_frames_per_request(self) -> int:
        return self.model.frames_per_batch

    def max_frames_per_request(self) -> int:
        return self.model.frames_per_batch * 100  # autoregressive chunks (server_cosmos.py:117-121)

    def inference_resolution(self) -> List[Tuple[int, int]]:
        return [(self.model.W, self.model.H)]

    def inference_time_per_frame(self) -> float:
        return 4.0

    def default_framerate(self) -> float:
        return 24.0

    def requires_seeding(self) -> bool:
        return True

    def metadata(self) -> dict:
        return _metadata(self, "CosmosModel")

    def cleanup(self):
        if hasattr(self.model, "cleanup"):
            self.model.cleanup()


def _to_numpy(x):
    if x is None or isinstance(x, np.ndarray):
        return x
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
