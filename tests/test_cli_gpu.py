"""GPU: the single-image CLI end to end (tiny random-weight models, 9-frame chunk) and the multi-buffer cache selector."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gen3c_single_image_cli_tiny(tmp_path):
    from PIL import Image
    from gen3c_amd import gen3c_single_image as cli
    H, W = 64, 96
    ys, xs = np.mgrid[0:H, 0:W]
    img = np.stack([(xs * 2) % 256, (ys * 3) % 256, ((xs + ys) * 2) % 256], -1).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "in.png")
    depth = (2.0 + 0.01 * xs).astype(np.float32)
    depth[10:20, 10:30] = 1.2
    np.savez(tmp_path / "depth.npz", depth=depth, intrinsics=np.array([[80, 0, W / 2], [0, 80, H / 2], [0, 0, 1]], np.float32))
    args = cli.create_parser().parse_args([
        "--input_image_path", str(tmp_path / "in.png"), "--depth_path", str(tmp_path / "depth.npz"), "--height", str(H), "--width", str(W),
        "--num_steps", "2", "--random_init", "--tiny", "--video_save_folder", str(tmp_path / "out"), "--video_save_name", "v",
        "--trajectory", "clockwise", "--foreground_masking"])
    video = cli.demo(args)
    assert video.shape == (9, H, W, 3) and video.dtype == np.uint8
    saved = np.load(tmp_path / "out" / "v.npz")["video"]
    assert np.array_equal(saved, video) and (tmp_path / "out" / "v_first.png").exists()


def test_buffer_selector_topk_and_exclusive_mask():
    from gen3c_amd import renderer
    dev = torch.device("cuda:0")
    H, W, N = 32, 48, 3
    g = torch.Generator().manual_seed(0)
    imgs = (torch.rand(1, N, 3, H, W, generator=g) * 2 - 1).to(dev)
    depth = torch.full((1, N, 1, H, W), 2.0, device=dev)
    depth[:, 1, :, :, : W // 2] = 0.0   # buffer 1 covers only half of the view
    depth[:, 2] = 0.0                    # buffer 2 covers nothing
    K = torch.tensor([[40.0, 0, W / 2], [0, 40.0, H / 2], [0, 0, 1]], device=dev)
    cache = renderer.Cache3D_BufferSelector(frame_buffer_max=2, input_image=imgs, input_depth=depth, input_w2c=torch.eye(4, device=dev).expand(1, N, 4, 4).contiguous(),
                                            input_intrinsics=K.expand(1, N, 3, 3).contiguous(), input_format=["B", "N", "C", "H", "W"])
    w2cs = torch.eye(4, device=dev)[None, None].repeat(1, 2, 1, 1)
    pix, msk = cache.render_cache(w2cs, K[None, None].repeat(1, 2, 1, 1))
    assert pix.shape == (1, 2, 2, 3, H, W) and msk.shape == (1, 2, 2, 1, H, W)
    cover = msk.mean(dim=(3, 4, 5))[0]  # [F, k]
    # top-2 by overlap are buffers 0 (full) and 1 (half); buffer 0 is near-full, so buffer 1 is blanked in every frame
    assert torch.all(cover[:, 0] > 0.95) and torch.all(cover[:, 1] == 0)
    assert torch.all(pix[0, :, 1] == -1)


def test_update_cache_aligns_new_depth_to_the_cache():
    """Cache3D_Buffer.update_cache(depth_alignment=True) (cache_3d.py:246-316): a mis-scaled monocular depth for the new frame is
    pulled onto the geometry the cache already holds; the buffer grows newest-first up to frame_buffer_max."""
    from gen3c_amd import renderer
    dev = torch.device("cuda:0")
    H, W = 48, 64
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    depth = (2.0 + 0.004 * xs + 0.002 * ys).to(dev)
    img = (torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    K = torch.tensor([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]], device=dev)
    eye = torch.eye(4, device=dev)
    for method in ("rigid", "non_rigid"):
        cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=img, input_depth=depth[None, None], input_w2c=eye[None],
                                        input_intrinsics=K[None], filter_points_threshold=0.05, input_format=["B", "C", "H", "W"])
        w2c_new = eye.clone()
        w2c_new[0, 3] = -0.05  # camera moved 5 cm to the right
        wrong = 1.0 / (0.6 / depth + 0.05)  # affine-in-inverse-depth distortion, ~1.5x too far
        cache.update_cache(new_image=img, new_depth=wrong[None, None], new_w2c=w2c_new[None], new_intrinsics=K[None], alignment_method=method)
        assert cache.input_image.shape[2] == 2 and cache.input_points.shape[2] == 2
        z_new = cache.input_points[0, 0, 0, 0, ..., 2]      # newest first; world z == camera z for this pure x-translation
        m = cache.input_mask[0, 0, 0, 0, 0] > 0
        rel = (z_new[m] / depth[m] - 1).abs()
        print(f"[update_cache {method}] aligned depth vs scene: max rel {float(rel.max()):.3e} (unaligned {float((wrong / depth - 1).abs().max()):.2f})")
        assert float(rel.max()) < 0.02
        cache.update_cache(new_image=img, new_depth=wrong[None, None], new_w2c=w2c_new[None], new_intrinsics=K[None], alignment_method=method)
        assert cache.input_image.shape[2] == 2  # full buffer: slot 0 is overwritten


def test_gen3c_single_image_cli_autoregressive_tiny(tmp_path):
    """Two autoregressive chunks (8*2+1 frames with the tiny models): chunks overlap by one frame, the cache is updated with the
    last generated frame + aligned depth, the second chunk is conditioned on that frame (gen3c_single_image.py:378-419)."""
    from PIL import Image
    from gen3c_amd import gen3c_single_image as cli
    H, W = 64, 96
    ys, xs = np.mgrid[0:H, 0:W]
    Image.fromarray(np.stack([(xs * 2) % 256, (ys * 3) % 256, ((xs + ys) * 2) % 256], -1).astype(np.uint8)).save(tmp_path / "in.png")
    np.savez(tmp_path / "depth.npz", depth=(2.0 + 0.01 * xs).astype(np.float32), intrinsics=np.array([[80, 0, W / 2], [0, 80, H / 2], [0, 0, 1]], np.float32))
    args = cli.create_parser().parse_args([
        "--input_image_path", str(tmp_path / "in.png"), "--depth_path", str(tmp_path / "depth.npz"), "--height", str(H), "--width", str(W),
        "--num_steps", "2", "--random_init", "--tiny", "--video_save_folder", str(tmp_path / "out"), "--video_save_name", "ar",
        "--num_video_frames", "17", "--trajectory", "left"])
    video = cli.demo(args)
    assert video.shape == (17, H, W, 3) and video.dtype == np.uint8


def _scene(H, W, F, seed=0):
    rng = np.random.RandomState(seed)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    imgs = np.stack([np.stack([np.sin(xs / (7 + f)), np.cos(ys / (5 + f)), np.sin((xs + ys) / 11)]) for f in range(F)]).astype(np.float32)
    depth = np.stack([(2.0 + 0.01 * xs + 0.02 * f)[None] for f in range(F)]).astype(np.float32)
    mask = (rng.rand(F, 1, H, W) < 0.9).astype(np.float32)
    K = np.repeat(np.array([[80, 0, W / 2], [0, 80, H / 2], [0, 0, 1]], np.float32)[None], F, 0)
    w2c = np.repeat(np.eye(4, dtype=np.float32)[None], F, 0)
    w2c[:, 0, 3] = -0.02 * np.arange(F)
    return imgs, depth, mask, K, w2c


def test_gen3c_multiview_cli_tiny_with_save_buffer(tmp_path):
    """gen3c_multiview.py:180-268: NPZ key frames -> Cache3D_BufferSelector (top-2 buffers per target frame) -> 2 chunks."""
    from gen3c_amd import gen3c_multiview as cli
    H, W, N, T = 64, 96, 3, 17
    imgs, depth, mask, K, w2c = _scene(H, W, N)
    w2cs_all = np.repeat(np.eye(4, dtype=np.float32)[None], T, 0)
    w2cs_all[:, 0, 3] = -0.004 * np.arange(T)
    np.savez(tmp_path / "mv.npz", images_key_frames=imgs, depth_key_frames=depth, mask_key_frames=mask, K_key_frames=K, w2cs_key_frames=w2c,
             w2cs_all=w2cs_all)
    args = cli.create_parser().parse_args(["--npz_path", str(tmp_path / "mv.npz"), "--height", str(H), "--width", str(W), "--num_steps", "2",
                                           "--random_init", "--tiny", "--num_video_frames", str(T), "--save_buffer",
                                           "--video_save_folder", str(tmp_path / "out"), "--video_save_name", "mv"])
    video = cli.demo(args)
    assert video.shape == (T, H, 3 * W, 3) and video.dtype == np.uint8  # 2 selected buffers + the generated frame, side by side
    assert np.array_equal(np.load(tmp_path / "out" / "mv.npz")["video"], video)


@pytest.mark.parametrize("fmt", ["pt", "dir"])
def test_gen3c_dynamic_cli_tiny(tmp_path, fmt):
    """gen3c_dynamic.py:190-320 + data_loader_utils.py:137-193: per-frame RGB-D sources (Cache4D), both input formats."""
    from gen3c_amd import gen3c_dynamic as cli
    H, W, F = 64, 96, 17
    imgs, depth, mask, K, w2c = _scene(H, W, F, seed=1)
    if fmt == "pt":
        src = tmp_path / "dyn.pt"
        torch.save(tuple(torch.from_numpy(a) for a in (imgs, depth, mask, w2c, K)), src)
    else:
        src = tmp_path / "dyn"
        src.mkdir()
        np.savez(src / "rgb.npz", rgb=np.clip((imgs.transpose(0, 2, 3, 1) + 1) * 127.5, 0, 255).astype(np.uint8))
        np.savez(src / "depth.npz", depth=depth[:, 0])
        np.savez(src / "mask.npz", mask=mask[:, 0])
        np.savez(src / "camera.npz", w2c=w2c, intrinsics=K)
    args = cli.create_parser().parse_args(["--input_image_path", str(src), "--height", str(H), "--width", str(W), "--num_steps", "2", "--random_init",
                                           "--tiny", "--num_video_frames", str(F), "--trajectory", "right",
                                           "--video_save_folder", str(tmp_path / "out"), "--video_save_name", "dyn"])
    video = cli.demo(args)
    assert video.shape == (F, H, W, 3) and video.dtype == np.uint8


def test_persistent_model_seed_and_two_requests(tmp_path):
    """gen3c_persistent.py:55-569: models built once; single-image seeding -> 2-chunk autoregressive request with estimated depths
    returned; then multi-frame (Cache4D) seeding on the same object."""
    from gen3c_amd import gen3c_persistent as gp
    H, W = 64, 96
    args = gp.create_parser().parse_args(["--height", str(H), "--width", str(W), "--num_steps", "2", "--random_init", "--tiny",
                                          "--video_save_folder", str(tmp_path / "out"), "--video_save_name", "req"])
    model = gp.Gen3cPersistentModel(args)
    imgs, depth, mask, K, w2c = _scene(H, W, 9, seed=2)
    img01 = ((imgs + 1) / 2).transpose(0, 2, 3, 1)
    fl = np.stack([K[:, 0, 0], K[:, 1, 1]], 1)
    pp = np.stack([K[:, 0, 2] / W, K[:, 1, 2] / H], 1)
    res = np.tile([[W, H]], (9, 1))
    out = model.seed_model_from_values(img01[:1], depth[:1, 0], w2c[:1], fl[:1], pp[:1], res[:1])
    assert out[3].tolist() == [[W, H]] and model.seeding_image.shape == (1, 3, 1, H, W)
    T = 17
    cams = np.repeat(np.eye(4, dtype=np.float32)[None], T, 0)
    cams[:, 0, 3] = -0.003 * np.arange(T)
    r = model.inference_on_cameras(cams, np.repeat(K[:1], T, 0), fps=24, return_estimated_depths=True)
    assert r["video"].shape == (1, T, 3, H, W) and r["predicted_depth"].shape == (T, 1, H, W)
    assert np.isfinite(r["predicted_depth"][8]).all() and np.isnan(r["predicted_depth"][3]).all()
    assert model.cache.input_image.shape[2] == 2  # the AR step pushed the last frame into the buffer
    model.clear_cache()
    model.seed_model_from_values(img01, depth[:, 0], w2c, fl, pp, res, masks_np=mask[:, 0])
    r = model.inference_on_cameras(cams[:9], np.repeat(K[:1], 9, 0), fps=12, save_buffer=True)
    assert r["video"].shape == (1, 9, 3, H, W) and model.pipeline.fps == 12
    assert np.load(r["video_save_path"])["video"].shape == (9, H, 2 * W, 3)


def test_serving_boundary_over_the_resident_model(tmp_path):
    """SURVEY.md 8-f4 / VERDICT r3 #8: the call sequence the reference's server makes (server.py:123-236 -> server_cosmos_base.py:46-214) -
    seed_model(SeedingRequest) -> request_inference(InferenceRequest) -> inference_result_or_none - through gen3c_amd.serving.Gen3cInferenceModel
    on the REAL resident model (HIP kernels, tiny widths): record fields, shapes and the one-frame overlap of consecutive requests."""
    import asyncio
    from gen3c_amd import api_types as api
    from gen3c_amd import gen3c_persistent as gp
    from gen3c_amd.serving import Gen3cInferenceModel
    H, W = 64, 96
    args = gp.create_parser().parse_args(["--height", str(H), "--width", str(W), "--num_steps", "2", "--random_init", "--tiny",
                                          "--video_save_folder", str(tmp_path / "out"), "--video_save_name", "srv"])
    resident = gp.Gen3cPersistentModel(args)
    n = resident.frames_per_batch
    imgs, depth, mask, K, w2c = _scene(H, W, 1, seed=4)
    c2w = np.linalg.inv(w2c)[:, :3].astype(np.float32)
    fl = np.stack([K[:, 0, 0], K[:, 1, 1]], 1).astype(np.float32)
    pp = np.stack([K[:, 0, 2] / W, K[:, 1, 2] / H], 1).astype(np.float32)

    def cams(x0):
        c = np.repeat(np.eye(4, dtype=np.float32)[None, :3], n, 0)
        c[:, 0, 3] = x0 + 0.003 * np.arange(n)
        return c

    async def drive():
        model = Gen3cInferenceModel(resident, compress_inference_results=False)
        with pytest.raises(ValueError, match="not seeded"):
            model.request_inference(api.InferenceRequest(request_id="early", timestamps=np.zeros(n, np.float32), cameras_to_world=cams(0), focal_lengths=np.repeat(fl, n, 0),
                                                         principal_points=np.repeat(pp, n, 0), resolutions=np.tile([[W, H]], (n, 1))))
        seeded = await model.seed_model(api.SeedingRequest(request_id="seed", images=((imgs + 1) / 2).transpose(0, 2, 3, 1).astype(np.float32), depths=depth[:, 0],
                                                           cameras_to_world=c2w, focal_lengths=fl, principal_points=pp))
        results = []
        for i, x0 in enumerate((0.0, 0.003 * (n - 1))):
            req = api.InferenceRequest(request_id=f"req{i}", timestamps=np.zeros(n, np.float32), cameras_to_world=cams(x0), focal_lengths=np.repeat(fl, n, 0),
                                       principal_points=np.repeat(pp, n, 0), resolutions=np.tile([[2 * W, 2 * H]], (n, 1)), framerate=24.0, return_depths=(i == 0))
            task = model.request_inference(req)
            await task
            results.append(model.inference_result_or_none(req.request_id))
        return model, seeded, results

    model, seeded, (r0, r1) = asyncio.run(drive())
    assert isinstance(seeded, api.SeedingResult) and seeded.depths is None and seeded.resolutions.tolist() == [[W, H]]  # the request brought its depths
    assert isinstance(r0, api.InferenceResult) and r0.images.shape == (n, H, W, 3) and r0.depths.shape == (n, H, W) and len(r0.result_ids) == n
    assert r0.request_id == "req0" and r0.runtime_ms > 0 and np.isfinite(r0.depths[-1]).all()
    assert r1.images.shape == (n - 1, H, W, 3) and r1.depths is None and r1.cameras_to_world.shape == (n - 1, 3, 4)  # one overlap frame regenerated, not returned
    assert len(model.pose_history_w2c) == 2 and np.array_equal(model.pose_history_w2c[1][0], model.pose_history_w2c[0][-1])
    assert model.metadata()["inference_resolution"] == [(W, H)] and model.min_frames_per_request() == n


def test_cli_loads_checkpoint_layout(tmp_path):
    """The checkpoint path of the entry points (world_generation_pipeline.py:182-186, inference_utils.py:240-242,327-347):
    checkpoints/Gen3C-Cosmos-7B/model.pt with `net.*` keys (+ TE `_extra_state` blobs and `logvar.*` that must be ignored) and
    checkpoints/Cosmos-Tokenize1-CV8x8x8-720p/{encoder.jit,decoder.jit,mean_std.pt}; tiny configuration, strict loading."""
    from PIL import Image
    from gen3c_amd import gen3c_single_image as cli
    from gen3c_amd.dit import VideoExtendGeneralDIT
    from gen3c_amd.tokenizer import CausalVideoTokenizerNet
    from tests.test_tokenizer_gpu import _script_archive
    dev = torch.device("cuda:0")
    H, W = 64, 96
    net = VideoExtendGeneralDIT(max_img_h=240, max_img_w=240, max_frames=16, in_channels=81, model_channels=256, num_blocks=2, num_heads=2,
                                adaln_lora_dim=32, rope_t_extrapolation_ratio=2.0, device=dev, init_weights=False)
    net.initialize_weights(randomize_adaln=True, seed=42)
    sd = {"net." + k: v.detach().cpu() for k, v in net.state_dict().items()}
    sd["net.blocks.block0.blocks.0.block.attn.attn_op._extra_state"] = torch.zeros(4, dtype=torch.uint8)
    sd["logvar.0.freqs"] = torch.zeros(3)
    (tmp_path / "ckpt" / "Gen3C-Cosmos-7B").mkdir(parents=True)
    torch.save({"model": sd}, tmp_path / "ckpt" / "Gen3C-Cosmos-7B" / "model.pt")
    tdir = tmp_path / "ckpt" / "Cosmos-Tokenize1-CV8x8x8-720p"
    tdir.mkdir()
    tsd = CausalVideoTokenizerNet(channels=16, device=dev).init_random(seed=9)
    _script_archive({k: v.float().cpu() for k, v in tsd.items() if k.startswith(("encoder.", "quant_conv."))}, tdir / "encoder.jit")
    _script_archive({k: v.float().cpu() for k, v in tsd.items() if k.startswith(("post_quant_conv.", "decoder."))}, tdir / "decoder.jit")
    torch.save((torch.zeros(16 * 32), torch.ones(16 * 32)), tdir / "mean_std.pt")
    # the checkpoint PSNR harness on this very layout - HIP vs the oracle chain in the reference's precision (bf16) vs fp32, per frame,
    # non-zero exit if a HIP frame is more than 0.1 dB worse than the reference-precision frame
    import json
    from tools import psnr_vs_oracle
    rc = psnr_vs_oracle.main(["--checkpoint_dir", str(tmp_path / "ckpt"), "--tiny", "--height", str(H), "--width", str(W), "--num_steps", "3",
                              "--video_save_folder", str(tmp_path / "out"), "--json", str(tmp_path / "psnr.json")])
    rep = json.loads((tmp_path / "psnr.json").read_text())
    print(f"[psnr harness, tiny checkpoint] hip vs fp32 {min(rep['psnr_hip_vs_fp32']):.2f}..{max(rep['psnr_hip_vs_fp32']):.2f} dB, reference precision vs fp32 "
          f"{min(rep['psnr_ref_vs_fp32']):.2f}..{max(rep['psnr_ref_vs_fp32']):.2f} dB, worst delta {rep['worst_delta_db']:+.2f} dB")
    assert rc == 0 and rep["passed"] and rep["frames"] == 9 and len(rep["psnr_hip_vs_fp32"]) == 9
    ys, xs = np.mgrid[0:H, 0:W]
    Image.fromarray(np.stack([(xs * 2) % 256, (ys * 3) % 256, ((xs + ys) * 2) % 256], -1).astype(np.uint8)).save(tmp_path / "in.png")
    np.savez(tmp_path / "depth.npz", depth=(2.0 + 0.01 * xs).astype(np.float32))
    argv = ["--input_image_path", str(tmp_path / "in.png"), "--depth_path", str(tmp_path / "depth.npz"), "--height", str(H), "--width", str(W),
            "--num_steps", "2", "--tiny", "--checkpoint_dir", str(tmp_path / "ckpt"), "--video_save_folder", str(tmp_path / "out")]
    v1 = cli.demo(cli.create_parser().parse_args(argv + ["--video_save_name", "a"]))
    v2 = cli.demo(cli.create_parser().parse_args(argv + ["--video_save_name", "b"]))
    v3 = cli.demo(cli.create_parser().parse_args(argv + ["--video_save_name", "c", "--random_init"]))
    assert v1.shape == (9, H, W, 3)
    d12 = float(np.abs(v1.astype(np.float32) - v2.astype(np.float32)).mean())
    d13 = float(np.abs(v1.astype(np.float32) - v3.astype(np.float32)).mean())
    print(f"[checkpoint cli] same checkpoint twice: mean |diff| {d12:.3f}; checkpoint vs random init: {d13:.3f}")
    # same checkpoint and seed -> the same video up to the splat's atomic summation order; other weights -> another video
    assert d12 < 0.5 and d13 > 10 * max(d12, 0.05)


# ---------------------------------------------------------------------------------------------------------------------------------
# rows f2 / f3 / f4 against the REFERENCE's own Python (tests/golden/cache_rows_f.npz, tools/gen_golden_cache.py)
# ---------------------------------------------------------------------------------------------------------------------------------
def _gold():
    from tests.golden_io import GOLD
    return np.load(GOLD / "cache_rows_f.npz")


def _close_pixels(got, ref, what):
    err = np.abs(got - ref)
    bad = err > (1e-4 + 1e-3 * np.abs(ref))
    assert bad.mean() < 1e-4 and err.max() < 5e-2, f"{what}: {int(bad.sum())} px off, max {err.max():.3e}"


@pytest.mark.parametrize("tag,kw", [("top2", dict(frame_buffer_max=2)), ("top2_nomax", dict(frame_buffer_max=2, mask_for_max_buffer_model=False)),
                                    ("thr70", dict(frame_buffer_max=2, mask_full_threshold=0.7)), ("all", dict(frame_buffer_max=4))])
def test_buffer_selector_matches_reference(tag, kw):
    """Cache3D_BufferSelector.render_cache (cache_3d.py:346-421): top-K buffers by mask overlap and the near-full exclusivity mask
    (thr70: frame 0 keeps buffer 0, middle frames keep buffer 1, late frames keep both). Masks - hence the selection - bit-exact."""
    from gen3c_amd import renderer
    dev = torch.device("cuda:0")
    z = _gold()
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    c = renderer.Cache3D_BufferSelector(input_image=t("sel_images")[None], input_depth=t("sel_depth")[None], input_mask=t("sel_mask")[None],
                                        input_w2c=t("sel_w2c")[None], input_intrinsics=t("sel_K")[None], filter_points_threshold=0.05,
                                        input_format=["B", "N", "C", "H", "W"], foreground_masking=False, **kw)
    px, mk = c.render_cache(t("sel_tw2c")[None], t("sel_tK")[None])
    ref_m = z[f"sel:{tag}:masks"]
    assert tuple(mk.shape) == ref_m.shape
    nd = int((mk.cpu().numpy() != ref_m).sum())
    assert nd == 0, f"{tag}: {nd} mask px differ from the reference"
    if f"sel:{tag}:pixels" in z.files:
        _close_pixels(px.cpu().numpy(), z[f"sel:{tag}:pixels"], tag)
    if tag == "thr70":  # the case really exercises all three branches of the exclusivity logic
        per = ref_m.mean(axis=(3, 4, 5))[0]
        assert (per[:, 1] == 0).any() and (per[:, 0] == 0).any() and ((per[:, 0] > 0) & (per[:, 1] > 0)).any()
    if tag == "all":
        d, dm = c.render_cache(t("sel_tw2c")[None], t("sel_tK")[None], render_depth=True)
        assert np.array_equal(dm.cpu().numpy(), z["sel:all:depth_masks"])
        _close_pixels(d.cpu().numpy(), z["sel:all:depth"], "depth")


@pytest.mark.parametrize("start", [0, 4])
def test_cache4d_windows_match_reference(start):
    """Cache4D.render_cache(start_frame_idx) (cache_3d.py:151-236, 424-433): target frame f is rendered from source frame start + f."""
    from gen3c_amd import renderer
    dev = torch.device("cuda:0")
    z = _gold()
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    c = renderer.Cache4D(input_image=t("c4_images").clone(), input_depth=t("c4_depth"), input_mask=t("c4_mask"), input_w2c=t("c4_w2c"),
                         input_intrinsics=t("c4_K"), filter_points_threshold=0.05, input_format=["F", "C", "H", "W"], foreground_masking=False)
    px, mk = c.render_cache(t("c4_tw2c")[start:start + 5][None], t("c4_K")[start:start + 5][None], start_frame_idx=start)
    assert np.array_equal(mk.cpu().numpy(), z[f"c4:{start}:masks"])
    _close_pixels(px.cpu().numpy(), z[f"c4:{start}:pixels"], f"window {start}")


def test_persistent_model_seeding_and_cameras_match_reference(tmp_path):
    """Gen3cPersistentModel.seed_model_from_values (multi-frame branch, gen3c_persistent.py:203-268), prepare_camera_for_inference
    (:518-536) and resize_intrinsics (:35-52) against the reference's own functions."""
    from gen3c_amd import gen3c_persistent as gp
    z = _gold()
    PH, PW = (int(v) for v in z["seed_hw"])
    args = gp.create_parser().parse_args(["--height", str(PH), "--width", str(PW), "--num_steps", "2", "--random_init", "--tiny",
                                          "--video_save_folder", str(tmp_path), "--video_save_name", "p"])
    m = gp.Gen3cPersistentModel(args)
    ret = m.seed_model_from_values(z["seed_images01"], z["seed_depths"], z["seed_w2c"], z["seed_focal"], z["seed_pp_rel"], z["seed_res"],
                                   masks_np=z["seed_masks"])
    for got, key in zip(ret, ("seed_ret_w2c", "seed_ret_focal", "seed_ret_pp", "seed_ret_res")):
        np.testing.assert_array_equal(np.asarray(got), z[key])
    np.testing.assert_allclose(m.seeding_image.cpu().numpy(), z["seed_seeding_image"], rtol=0, atol=2e-6)  # bicubic-antialias resize
    np.testing.assert_allclose(m.cache.input_points.cpu().numpy(), z["seed_cache_points"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(m.cache.input_mask.cpu().numpy(), z["seed_cache_mask"])
    np.testing.assert_array_equal(m.cache.input_image.cpu().numpy(), z["seed_cache_image"])
    H0, W0 = z["c4_images"].shape[-2:]
    cams, intr = m.prepare_camera_for_inference(z["c4_tw2c"][:5], z["c4_K"][:5], (H0, W0), (PH, PW))
    np.testing.assert_array_equal(cams.cpu().numpy(), z["pc_w2c"])
    np.testing.assert_array_equal(intr.cpu().numpy(), z["pc_K"])
    np.testing.assert_array_equal(gp.resize_intrinsics(z["ri_in"], (H0, W0), (96, 160)), z["ri_plain"])
    np.testing.assert_array_equal(gp.resize_intrinsics(torch.from_numpy(z["ri_in"]), (H0, W0), (90, 160), crop_size=(88, 152)).numpy(), z["ri_crop"])
