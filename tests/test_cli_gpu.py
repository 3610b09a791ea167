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
