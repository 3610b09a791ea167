"""Fixture generator (build container only): runs the reference's OWN serving records and debug model - gui/api/api_types.py, encoding.py,
server_base.py, server_debug.py under /root/reference - on seeded inputs and commits what they produce to tests/golden/api_types.npz.
tests/test_api_types_cpu.py replays the same calls on gen3c_amd.api_types / gen3c_amd.serving. Shims: `cv2` (absent here; only the NPZ
paths are exercised, which never touch it) and `loguru`."""
import asyncio
import dataclasses
import json
import sys
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, "/root/reference/gui/api")
sys.modules["cv2"] = types.ModuleType("cv2")


class _Log:
    def __getattr__(self, _k):
        return lambda *a, **k: None


sys.modules["loguru"] = types.ModuleType("loguru")
sys.modules["loguru"].logger = _Log()

import api_types as ref  # noqa: E402
from encoding import CompressionFormat  # noqa: E402
from server_debug import DebugInferenceModel  # noqa: E402


def cameras(rs, n):
    c2w = np.tile(np.eye(4, dtype=np.float32)[None, :3], (n, 1, 1))
    c2w[:, :, 3] = rs.standard_normal((n, 3)).astype(np.float32)
    a = rs.uniform(-0.3, 0.3, n).astype(np.float32)
    c2w[:, 0, 0], c2w[:, 0, 2], c2w[:, 2, 0], c2w[:, 2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
    return dict(cameras_to_world=c2w, focal_lengths=rs.uniform(300, 900, (n, 2)).astype(np.float32),
                principal_points=rs.uniform(0.4, 0.6, (n, 2)).astype(np.float32))


def main():
    rs = np.random.RandomState(7)
    out = {}
    fields = {c.__name__: [f.name for f in dataclasses.fields(c)] for c in
              (ref.RequestBase, ref.SeedingRequest, ref.CompressedSeedingRequest, ref.SeedingResult, ref.InferenceRequest, ref.InferenceResult,
               ref.CompressedInferenceResult, ref.PendingRequest)}
    out["fields_json"] = np.array(json.dumps(fields))
    out["request_states_json"] = np.array(json.dumps({s.name: s.value for s in ref.RequestState}))
    out["formats_json"] = np.array(json.dumps({s.name: s.value for s in CompressionFormat}))

    # ---- inference request: geometry helpers, padding and trimming
    n = 5
    cam = cameras(rs, n)
    for k, v in cam.items():
        out[f"inf_{k}"] = v
    req = ref.InferenceRequest(request_id="r0", timestamps=np.arange(n, dtype=np.float32), resolutions=np.tile([[640, 352]], (n, 1)), **cam)
    out["inf_w2c"] = req.world_to_cameras()
    out["inf_K"] = req.intrinsics_matrix(None)
    out["inf_K_resized"] = req.intrinsics_matrix(np.tile([[1280, 704]], (n, 1)))
    out["inf_defaults_json"] = np.array(json.dumps(dict(framerate=req.framerate, return_depths=req.return_depths, video_encoding_quality=req.video_encoding_quality,
                                                        show_cache_renderings=req.show_cache_renderings)))
    req.pad_to_frame_count(8)
    out["inf_padded_c2w"], out["inf_padded_ts"], out["inf_padded_res"] = req.cameras_to_world, req.timestamps, req.resolutions
    out["inf_padded_count"] = np.array(req.frame_count_without_padding)
    req.trim_to_original_frame_count()
    out["inf_trimmed_ts"] = req.timestamps

    # ---- seeding request: resolution from the images, NPZ compression round trip, SeedingResult.from_request
    m = 2
    cam2 = cameras(rs, m)
    for k, v in cam2.items():
        out[f"seed_{k}"] = v
    images = rs.uniform(0, 1, (m, 6, 10, 3)).astype(np.float32)
    depths = rs.uniform(1, 5, (m, 6, 10)).astype(np.float32)
    masks = rs.uniform(0, 1, (m, 6, 10)) > 0.5
    out["seed_images"], out["seed_depths"], out["seed_masks"] = images, depths, masks
    sreq = ref.SeedingRequest(request_id="s0", images=images, depths=depths, masks=masks, **cam2)
    out["seed_resolutions"] = sreq.resolutions
    comp = sreq.compress(format_rgb=CompressionFormat.NPZ, format_depth=CompressionFormat.NPZ, format_mask=CompressionFormat.NPZ)
    out["comp_placeholder_shapes"] = np.array([comp.images.shape, comp.depths.shape + (0,), comp.masks.shape + (0,)], dtype=np.int64)
    out["comp_buffer_counts"] = np.array([len(comp.images_compressed), len(comp.depths_compressed), len(comp.masks_compressed)])
    comp.decompress()
    out["comp_images"], out["comp_depths"], out["comp_masks"] = comp.images, comp.depths, comp.masks
    sres = ref.SeedingResult.from_request(ref.SeedingRequest(request_id="s1", images=images, depths=None, **cam2), fallback_depths=np.ones((m, 12, 20), np.float32))
    out["sres_resolutions"], out["sres_depths"] = sres.resolutions, sres.depths

    # ---- the debug model through the request bookkeeping: seed -> request_inference -> result
    async def drive():
        model = DebugInferenceModel()
        seeded = await model.seed_model(ref.SeedingRequest(request_id="s2", images=images, depths=None, **cam2))
        r = ref.InferenceRequest(request_id="r1", timestamps=np.arange(n, dtype=np.float32), resolutions=np.tile([[16, 8]], (n, 1)), return_depths=True, **cam)
        task = model.request_inference(r)
        pending = model.inference_result_or_none("r1")
        await task
        done = model.inference_result_or_none("r1")
        test = await model.make_test_image()
        return model, seeded, pending, done, test

    model, seeded, pending, done, test = asyncio.run(drive())
    assert pending is None
    out["dbg_seed_depths"], out["dbg_seed_resolutions"] = seeded.depths, seeded.resolutions
    out["dbg_images"], out["dbg_depths"], out["dbg_timestamps"] = done.images, done.depths, done.timestamps
    out["dbg_result_ids_json"] = np.array(json.dumps(done.result_ids))
    out["dbg_test_images"] = test.images
    out["dbg_metadata_json"] = np.array(json.dumps(model.metadata()))
    np.savez_compressed(ROOT / "tests" / "golden" / "api_types.npz", **out)
    print("wrote tests/golden/api_types.npz:", sorted(out))


if __name__ == "__main__":
    main()
