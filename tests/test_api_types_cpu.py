"""CPU: the serving boundary's records and model-side contract against goldens produced by the reference's OWN classes
(tools/gen_golden_api.py ran gui/api/api_types.py, encoding.py, server_base.py, server_debug.py from /root/reference -> tests/golden/api_types.npz)."""
import asyncio
import dataclasses
import json

import numpy as np
import pytest

from gen3c_amd import api_types as api
from gen3c_amd import serving


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(golden_dir / "api_types.npz")


def _cam(gold, pre):
    return {k: gold[f"{pre}_{k}"] for k in ("cameras_to_world", "focal_lengths", "principal_points")}


def test_records_have_the_reference_fields_in_the_reference_order(gold):
    ref_fields = json.loads(str(gold["fields_json"]))
    for name, names in ref_fields.items():
        assert [f.name for f in dataclasses.fields(getattr(api, name))] == names, name
    assert {s.name: s.value for s in api.RequestState} == json.loads(str(gold["request_states_json"]))
    assert {s.name: s.value for s in api.CompressionFormat} == json.loads(str(gold["formats_json"]))


def test_inference_request_geometry_padding_and_defaults(gold):
    n = 5
    req = api.InferenceRequest(request_id="r0", timestamps=np.arange(n, dtype=np.float32), resolutions=np.tile([[640, 352]], (n, 1)), **_cam(gold, "inf"))
    assert np.array_equal(req.world_to_cameras(), gold["inf_w2c"])
    assert np.array_equal(req.intrinsics_matrix(None), gold["inf_K"])
    assert np.array_equal(req.intrinsics_matrix(np.tile([[1280, 704]], (n, 1))), gold["inf_K_resized"])
    d = json.loads(str(gold["inf_defaults_json"]))
    assert (req.framerate, req.return_depths, req.video_encoding_quality, req.show_cache_renderings) == \
           (d["framerate"], d["return_depths"], d["video_encoding_quality"], d["show_cache_renderings"])
    req.pad_to_frame_count(8)
    assert req.frame_count_without_padding == int(gold["inf_padded_count"]) == 5
    assert np.array_equal(req.cameras_to_world, gold["inf_padded_c2w"]) and np.array_equal(req.timestamps, gold["inf_padded_ts"])
    assert np.array_equal(req.resolutions, gold["inf_padded_res"])
    req.trim_to_original_frame_count()
    assert np.array_equal(req.timestamps, gold["inf_trimmed_ts"]) and len(req) == 5
    with pytest.raises(ValueError):
        api.InferenceRequest(request_id="x", timestamps=np.zeros(n, np.float32), **_cam(gold, "inf"))  # no images, no resolutions
    with pytest.raises(AssertionError):
        api.InferenceRequest(request_id="x", timestamps=np.zeros(n + 1, np.float32), resolutions=np.tile([[4, 4]], (n, 1)), **_cam(gold, "inf"))


def test_seeding_request_compression_round_trip_and_result(gold):
    images, depths, masks = gold["seed_images"], gold["seed_depths"], gold["seed_masks"]
    sreq = api.SeedingRequest(request_id="s0", images=images, depths=depths, masks=masks, **_cam(gold, "seed"))
    assert np.array_equal(sreq.resolutions, gold["seed_resolutions"])  # (width, height) taken from the images
    F = api.CompressionFormat
    comp = sreq.compress(format_rgb=F.NPZ, format_depth=F.NPZ, format_mask=F.NPZ)
    assert isinstance(comp, api.CompressedSeedingRequest) and comp.images.shape[0] == 0 and comp.depths.shape[0] == 0 and comp.masks.shape[0] == 0
    assert [len(comp.images_compressed), len(comp.depths_compressed), len(comp.masks_compressed)] == gold["comp_buffer_counts"].tolist()
    comp.decompress()
    assert comp.images.dtype == gold["comp_images"].dtype and np.array_equal(comp.images, gold["comp_images"])  # uint8 quantised colours, as the reference's NPZ path
    assert np.array_equal(comp.depths, gold["comp_depths"]) and comp.masks.dtype == bool and np.array_equal(comp.masks, gold["comp_masks"])
    with pytest.raises(RuntimeError):
        sreq.pad_to_frame_count(4)
    res = api.SeedingResult.from_request(api.SeedingRequest(request_id="s1", images=images, depths=None, **_cam(gold, "seed")),
                                         fallback_depths=np.ones((2, 12, 20), np.float32))
    assert np.array_equal(res.resolutions, gold["sres_resolutions"]) and np.array_equal(res.depths, gold["sres_depths"])
    # codecs that need OpenCV say so instead of substituting another format
    try:
        import cv2  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="OpenCV"):
            sreq.compress()  # defaults: JPG / EXR / NPZ


def test_pad_or_trim_helpers():
    a = np.arange(6).reshape(3, 2)
    assert np.array_equal(api.pad_or_trim_array(a, 5), np.concatenate([a, a[-1:], a[-1:]])) and api.pad_or_trim_array(a, 2).shape == (2, 2)
    assert api.pad_or_trim_array(None, 3) is None and api.pad_or_trim_array(a, 3) is a
    buf = api.compress_images(np.arange(24, dtype=np.float32).reshape(2, 3, 4), api.CompressionFormat.NPZ, is_depth=True)
    out = api.decompress_buffer(api.pad_or_trim_encoded_buffers(buf, api.CompressionFormat.NPZ, 4), api.CompressionFormat.NPZ, is_depth=True)
    assert out.shape == (4, 3, 4) and np.array_equal(out[3], out[1])
    assert api.pad_or_trim_encoded_buffers([b"a", b"b"], api.CompressionFormat.PNG, 4) == [b"a", b"b", b"b", b"b"]


def test_debug_model_answers_like_the_reference_debug_model(gold):
    """seed_model -> request_inference -> inference_result_or_none, the sequence the reference's server runs (server.py:123-236), on the stand-in
    model: same depths / images / ids / metadata as the reference's DebugInferenceModel."""
    n = 5

    async def drive():
        model = serving.DebugInferenceModel()
        seeded = await model.seed_model(api.SeedingRequest(request_id="s2", images=gold["seed_images"], depths=None, **_cam(gold, "seed")))
        r = api.InferenceRequest(request_id="r1", timestamps=np.arange(n, dtype=np.float32), resolutions=np.tile([[16, 8]], (n, 1)), return_depths=True, **_cam(gold, "inf"))
        task = model.request_inference(r)
        assert model.inference_result_or_none("r1") is None  # still running
        with pytest.raises(ValueError):
            model.request_inference(r)  # duplicate id
        await task
        done = model.inference_result_or_none("r1")
        assert model.inference_result_or_none("r1") is done  # cached
        with pytest.raises(KeyError):
            model.inference_result_or_none("nope")
        with pytest.raises(ValueError):
            model.check_valid_request(api.InferenceRequest(request_id="big", timestamps=np.zeros(17, np.float32), cameras_to_world=np.zeros((17, 3, 4), np.float32),
                                                           focal_lengths=np.ones((17, 2), np.float32), principal_points=np.ones((17, 2), np.float32),
                                                           resolutions=np.tile([[16, 8]], (17, 1))))
        return model, seeded, done, await model.make_test_image()

    model, seeded, done, test = asyncio.run(drive())
    assert np.array_equal(seeded.depths, gold["dbg_seed_depths"]) and np.array_equal(seeded.resolutions, gold["dbg_seed_resolutions"])
    assert np.array_equal(done.images, gold["dbg_images"]) and np.array_equal(done.depths, gold["dbg_depths"]) and np.array_equal(done.timestamps, gold["dbg_timestamps"])
    assert done.result_ids == json.loads(str(gold["dbg_result_ids_json"])) and np.array_equal(test.images, gold["dbg_test_images"])
    assert json.loads(json.dumps(model.metadata())) == json.loads(str(gold["dbg_metadata_json"]))
    assert np.array_equal(model.get_latest_rgb(), test.images[-1])


class _StandInPersistentModel:
    """The surface of gen3c_amd.gen3c_persistent.Gen3cPersistentModel that the serving adaptor touches, in numpy (the real one needs the GPU:
    tests/test_cli_gpu.py drives the adaptor over it)."""
    W, H, frames_per_batch, inference_overlap_frames = 32, 16, 5, 1

    def __init__(self):
        self.calls, self.depths = [], None

    def clear_cache(self):
        self.calls.append(("clear",))

    def seed_model_from_values(self, images_np, depths_np, world_to_cameras_np, focal_lengths_np, principal_point_rel_np, resolutions, masks_np=None):
        self.calls.append(("seed", images_np.shape, None if depths_np is None else depths_np.shape, world_to_cameras_np.shape))
        self.depths = np.full(images_np.shape[:3], 2.0, np.float32) if depths_np is None else depths_np
        n = images_np.shape[0]
        return world_to_cameras_np, focal_lengths_np, principal_point_rel_np * np.array([[self.W, self.H]]), np.tile([[self.W, self.H]], (n, 1))

    def get_cache_input_depths(self):
        return self.depths

    def inference_on_cameras(self, w2cs, Ks, fps, overlap_frames=1, return_estimated_depths=False, video_save_quality=5, save_buffer=None):
        self.calls.append(("infer", w2cs.copy(), Ks.copy(), fps, overlap_frames, return_estimated_depths))
        n = w2cs.shape[0] - overlap_frames
        video = np.zeros((1, n, 3, self.H, self.W), np.uint8) + np.arange(n, dtype=np.uint8)[None, :, None, None, None]
        return {"video_no_overlap": video, "predicted_depth": np.ones((n, 1, self.H, self.W), np.float32) if return_estimated_depths else None,
                "video_save_path": "/nonexistent/video.npz"}


def test_gen3c_inference_model_maps_requests_onto_the_persistent_model(gold):
    n = 5
    cam = _cam(gold, "inf")

    async def drive():
        pm = _StandInPersistentModel()
        model = serving.Gen3cInferenceModel(pm)
        assert model.requires_seeding() and not model.model_seeded
        r = api.InferenceRequest(request_id="a", timestamps=np.zeros(n, np.float32), resolutions=np.tile([[640, 352]], (n, 1)), return_depths=True, **cam)
        with pytest.raises(ValueError, match="not seeded"):
            model.request_inference(r)
        seeded = await model.seed_model(api.SeedingRequest(request_id="s", images=gold["seed_images"][:1], depths=None,
                                                           **{k: v[:1] for k, v in _cam(gold, "seed").items()}))
        first = await model.request_inference_sync(r)
        r2 = api.InferenceRequest(request_id="b", timestamps=np.zeros(n, np.float32), resolutions=np.tile([[640, 352]], (n, 1)), **cam)
        second = await model.request_inference_sync(r2)
        return pm, model, seeded, first, second, r

    pm, model, seeded, first, second, r = asyncio.run(drive())
    assert [c[0] for c in pm.calls] == ["clear", "seed", "infer", "infer"]
    # seeding: world-to-camera matrices go in; estimated depths come back because the request had none; principal point relative again
    assert pm.calls[1][3] == (1, 4, 4) and seeded.depths.shape == (1, 6, 10) and np.allclose(seeded.principal_points, gold["seed_principal_points"][:1])
    assert np.array_equal(seeded.resolutions, [[32, 16]])
    # first request: cameras inverted, intrinsics rescaled from the requested 640 x 352 to the model's 32 x 16, no overlap
    _, w2c0, K0, fps0, ov0, rd0 = pm.calls[2]
    assert np.array_equal(w2c0, r.world_to_cameras()) and np.array_equal(K0, r.intrinsics_matrix(np.tile([[32, 16]], (n, 1)))) and (fps0, ov0, rd0) == (30.0, 0, True)
    assert first.images.shape == (n, 16, 32, 3) and first.depths.shape == (n, 16, 32) and len(first.result_ids) == n and first.result_ids[2] == "a__frame_2"
    assert np.array_equal(first.cameras_to_world, r.cameras_to_world) and np.array_equal(first.resolutions, np.tile([[32, 16]], (n, 1)))
    # second request: the previous request's last camera is put in front, the last requested camera is dropped, one frame fewer comes back
    _, w2c1, K1, _, ov1, _ = pm.calls[3]
    assert ov1 == 1 and np.array_equal(w2c1[0], w2c0[-1]) and np.array_equal(w2c1[1:], r.world_to_cameras()[:-1]) and np.array_equal(K1[0], K0[-1])
    assert second.images.shape[0] == n - 1 and second.cameras_to_world.shape[0] == n - 1 and second.depths is None
    md = model.metadata()
    assert md["min_frames_per_request"] == 5 and md["max_frames_per_request"] == 500 and md["inference_resolution"] == [(32, 16)] and md["requires_seeding"]


def test_pad_trim_and_npz_round_trips_are_identities_property():
    """Size-independent properties of the records (hypothesis): pad_to_frame_count(m) then trim_to_original_frame_count() restores every per-frame field;
    the padding repeats the LAST entry; NPZ compression of depths / masks / 8-bit colours round-trips exactly."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(n=st.integers(1, 9), extra=st.integers(0, 7), seed=st.integers(0, 10_000))
    def run(n, extra, seed):
        rs = np.random.RandomState(seed)
        req = api.InferenceRequest(request_id="p", timestamps=rs.rand(n).astype(np.float32), cameras_to_world=rs.randn(n, 3, 4).astype(np.float32),
                                   focal_lengths=rs.rand(n, 2).astype(np.float32) + 1, principal_points=rs.rand(n, 2).astype(np.float32),
                                   resolutions=np.tile([[64, 32]], (n, 1)))
        before = {k: getattr(req, k).copy() for k in ("timestamps", "cameras_to_world", "focal_lengths", "principal_points", "resolutions")}
        req.pad_to_frame_count(n + extra)
        assert len(req) == n + extra and req.frame_count_without_padding == n
        for k, v in before.items():
            got = getattr(req, k)
            assert np.array_equal(got[:n], v) and all(np.array_equal(got[i], v[-1]) for i in range(n, n + extra))
        req.trim_to_original_frame_count()
        assert all(np.array_equal(getattr(req, k), v) for k, v in before.items())
        h, w = int(rs.randint(1, 6)), int(rs.randint(1, 7))
        depth = rs.rand(n, h, w).astype(np.float32) * 10
        mask = rs.rand(n, h, w) > 0.5
        col = rs.randint(0, 256, (n, h, w, 3)).astype(np.float32) / 255.0
        F = api.CompressionFormat.NPZ
        assert np.array_equal(api.decompress_buffer(api.compress_images(depth, F, is_depth=True), F, is_depth=True), depth)
        assert np.array_equal(api.decompress_buffer(api.compress_images(mask, F, is_bool=True), F, is_bool=True), mask)
        assert np.array_equal(api.decompress_buffer(api.compress_images(col, F), F), (col * 255.0).astype(np.uint8))

    run()
