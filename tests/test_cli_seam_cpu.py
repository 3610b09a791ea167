"""CLI seam: the reference's documented command lines (/root/reference/README.md:84-230) must parse unchanged against
the entry points at the reference's own locations (cosmos_predict1/diffusion/inference/gen3c_*.py in this repository), every flag of
inference_utils.add_common_arguments (:53-170) must exist with the reference's default, and flags without a counterpart are logged."""
import importlib
import runpy
import shlex
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

# verbatim from the reference README (line continuations joined, ${NUM_GPUS} -> 8, the launcher part cut at the script path)
SINGLE_8GPU = ("--checkpoint_dir checkpoints --input_image_path assets/diffusion/000000.png --video_save_name test_single_image_multigpu "
               "--num_gpus 8 --guidance 1 --foreground_masking")
SINGLE_OFFLOAD = ("--checkpoint_dir checkpoints --input_image_path assets/diffusion/000000.png --video_save_name test_single_image --guidance 1 "
                  "--foreground_masking --offload_diffusion_transformer --offload_tokenizer --offload_text_encoder_model --offload_prompt_upsampler "
                  "--offload_guardrail_models --disable_guardrail --disable_prompt_encoder")
DYNAMIC_8GPU = ("--checkpoint_dir checkpoints --input_image_path assets/diffusion/dynamic_video_samples/batch_0000 "
                "--video_save_name test_dynamic_video_multigpu --num_gpus 8 --guidance 1")
DYNAMIC_VIPE = ("--checkpoint_dir checkpoints --vipe_path /data/vipe_results --vipe_starting_frame_idx 0 --video_save_name gen3c_test_dynamic_vipe "
                "--disable_prompt_upsampler --num_gpus 8 --guidance 1 --num_video_frames 121")
MULTIVIEW = ("--checkpoint_dir checkpoints --video_save_name gen3c_test_multiview --num_video_frames 361 --height 704 --width 1280 "
             "--npz_path assets/diffusion/dynamic_video_samples/mv_0.npz --num_gpus 8 --filter_points_threshold 0.05 --foreground_masking --guidance 1")


@pytest.mark.parametrize("module,cmd", [("gen3c_single_image", SINGLE_8GPU), ("gen3c_single_image", SINGLE_OFFLOAD), ("gen3c_dynamic", DYNAMIC_8GPU),
                                        ("gen3c_dynamic", DYNAMIC_VIPE), ("gen3c_multiview", MULTIVIEW)])
def test_reference_command_lines_parse_verbatim(module, cmd):
    mod = importlib.import_module(f"cosmos_predict1.diffusion.inference.{module}")  # the reference's import path
    args = mod.create_parser().parse_args(shlex.split(cmd))
    assert args.checkpoint_dir == "checkpoints" and args.guidance == 1
    if "--num_gpus 8" in cmd:
        assert args.num_gpus == 8
    if module == "gen3c_single_image":
        assert args.depth_path is None and args.foreground_masking  # no --depth_path in the reference's command: MoGe when importable


def test_common_flags_and_defaults_are_the_references():
    from gen3c_amd.gen3c_single_image import create_parser
    d = vars(create_parser().parse_args([]))
    expect = dict(checkpoint_dir="checkpoints", tokenizer_dir="Cosmos-Tokenize1-CV8x8x8-720p", video_save_name="output", video_save_folder="outputs/",
                  prompt=None, batch_input_path=None, num_steps=35, guidance=1, height=704, width=1280, fps=24, seed=1, num_gpus=1,
                  disable_prompt_upsampler=False, offload_diffusion_transformer=False, offload_tokenizer=False, offload_text_encoder_model=False,
                  offload_prompt_upsampler=False, offload_guardrail_models=False, disable_guardrail=False, disable_prompt_encoder=False,
                  prompt_upsampler_dir="Pixtral-12B", trajectory="left", camera_rotation="center_facing", movement_distance=0.3, noise_aug_strength=0.0,
                  save_buffer=False, filter_points_threshold=0.05, foreground_masking=False, input_image_path=None)
    for k, v in expect.items():
        assert k in d, f"missing flag --{k}"
        assert d[k] == v, (k, d[k], v)
    assert d["negative_prompt"].startswith("The video captures a series of frames showing ugly scenes")


def test_flags_without_counterpart_are_logged_not_rejected():
    from gen3c_amd.cli_common import log_ignored_flags
    from gen3c_amd.gen3c_single_image import create_parser
    args = create_parser().parse_args(shlex.split(SINGLE_OFFLOAD))
    lines = []
    given = log_ignored_flags(args, log=lines.append)
    assert set(given) == {"offload_diffusion_transformer", "offload_tokenizer", "offload_text_encoder_model", "offload_prompt_upsampler",
                          "offload_guardrail_models", "disable_guardrail"}
    assert len(lines) == 6 and all(l.startswith("[gen3c_amd] --") for l in lines)
    assert args.disable_prompt_encoder  # honoured (dummy zero embeddings), not merely ignored


def test_entry_points_run_as_script_paths():
    """`python cosmos_predict1/diffusion/inference/gen3c_single_image.py --help` (no PYTHONPATH): the script finds the package itself."""
    import os
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    for name in ("gen3c_single_image", "gen3c_dynamic", "gen3c_multiview"):
        r = subprocess.run([sys.executable, str(ROOT / "cosmos_predict1" / "diffusion" / "inference" / f"{name}.py"), "--help"], capture_output=True,
                           text=True, cwd="/tmp", env=env, timeout=300)
        assert r.returncode == 0 and "--checkpoint_dir" in r.stdout and "--offload_tokenizer" in r.stdout, r.stderr[-500:]


def test_text_embedder_falls_back_to_dummy_zeros(tmp_path):
    import argparse
    import torch
    from gen3c_amd.cli_common import TextEmbedder
    lines = []
    te = TextEmbedder(argparse.Namespace(checkpoint_dir=str(tmp_path), disable_prompt_encoder=False), 1024, "cpu", log=lines.append)
    assert te.source == "dummy"
    e = te("a prompt")
    assert e.shape == (1, 512, 1024) and e.dtype == torch.bfloat16 and not e.any()  # DummyT5TextEncoder (t5_text_encoder.py:111-132)
    assert any("all-zero text" in l for l in lines)
    p = tmp_path / "emb.pt"
    torch.save(torch.ones(1, 512, 1024), p)
    assert float(te(None, str(p)).float().mean()) == 1.0


def test_video_writer_chain_falls_back_to_npz(tmp_path):
    from gen3c_amd.cli_common import write_video
    video = (np.arange(3 * 8 * 16 * 3) % 255).astype(np.uint8).reshape(3, 8, 16, 3)
    out = write_video(video, 24, str(tmp_path / "clip"), log=lambda *_: None)
    have_encoder = any(importlib.util.find_spec(m) for m in ("imageio", "cv2"))
    assert out.endswith(".mp4" if have_encoder else ".npz") and Path(out).exists()
    if not have_encoder:
        z = np.load(out)
        assert np.array_equal(z["video"], video) and int(z["fps"]) == 24


def test_single_image_without_moge_asks_for_depth():
    from gen3c_amd import gen3c_single_image as g
    if g.load_moge() is not None:
        pytest.skip("moge is importable here")
    with pytest.raises(SystemExit, match="depth_path"):
        g._depth_inputs(None, "x.png", None, 704, 1280, "cpu", None)


def test_tokenizer_plugin_seam_video_vae_alias():
    """SURVEY 8b "Tokenizer plugin": helpers outside the class reach the video tokenizer through `model.tokenizer.video_vae.*`
    (inference_utils.py:677-691 compute_num_latent_frames, :768-782 compute_num_frames_condition). Replay those accesses on
    gen3c_amd.tokenizer.VideoTokenizer, and - when the reference tree is present - call the reference's own functions on it."""
    import types
    from gen3c_amd.tokenizer import VideoTokenizer
    tk = VideoTokenizer(pixel_chunk_duration=121, channels=16, device="cpu")
    model = types.SimpleNamespace(tokenizer=tk)
    v = model.tokenizer.video_vae
    assert v.pixel_chunk_duration == 121 and v.latent_chunk_duration == 16 and getattr(v, "is_casual", None) is True
    for name in ("load_weights", "encode", "decode", "reset_dtype", "get_latent_num_frames", "get_pixel_num_frames"):
        assert callable(getattr(tk, name)), name
    assert (tk.channel, tk.spatial_compression_factor, tk.temporal_compression_factor) == (16, 8, 8)

    def compute_num_latent_frames(model, num_input_frames, downsample_factor=8):  # the reference's attribute accesses, inference_utils.py:677-691
        n = num_input_frames // model.tokenizer.video_vae.pixel_chunk_duration * model.tokenizer.video_vae.latent_chunk_duration
        if num_input_frames % model.tokenizer.video_vae.latent_chunk_duration == 1:
            n += 1
        elif num_input_frames % model.tokenizer.video_vae.latent_chunk_duration > 1:
            assert (num_input_frames % model.tokenizer.video_vae.pixel_chunk_duration - 1) % downsample_factor == 0
            n += 1 + (num_input_frames % model.tokenizer.video_vae.pixel_chunk_duration - 1) // downsample_factor
        return n

    from gen3c_amd import pipeline
    for n_in in (1, 9, 17, 122):
        assert compute_num_latent_frames(model, n_in) == pipeline.compute_num_latent_frames(model, n_in)
    ref_file = Path("/root/reference/cosmos_predict1/diffusion/inference/inference_utils.py")
    if ref_file.is_file():  # build container only: the reference's own function BODIES (cut out with ast - the module's import chain needs
        import ast          # omegaconf / imageio / megatron, none of which these two functions use), executed on our tokenizer
        src = ref_file.read_text()
        ns = {}
        for node in ast.parse(src).body:
            if isinstance(node, ast.FunctionDef) and node.name in ("compute_num_latent_frames", "compute_num_frames_condition"):
                exec(compile("from __future__ import annotations\n" + ast.get_source_segment(src, node), str(ref_file), "exec"), ns)
        assert len([k for k in ns if k.startswith("compute_")]) == 2
        for n_in in (1, 9, 17, 122):
            assert ns["compute_num_latent_frames"](model, n_in) == pipeline.compute_num_latent_frames(model, n_in)
        assert [ns["compute_num_frames_condition"](model, k) for k in (1, 2, 16)] == [1, 9, 121]
