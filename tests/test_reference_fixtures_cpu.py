"""CPU half of the reference-fixture parity tests: the data files under tests/golden/ref_inputs/ are the reference's own (byte copies, checksums
pinned here), and the inputs derived from them (tests/ref_fixture_inputs.py) have the geometry the GPU tests rely on."""
import hashlib

import numpy as np

from tests import ref_fixture_inputs as rf


def test_reference_input_files_are_byte_copies():
    for name, sha in rf.SHA256.items():
        assert hashlib.sha256((rf.REF_IN / name).read_bytes()).hexdigest() == sha, name
    assert rf.load_rgb("diffusion_000000.png").shape == (720, 1280, 3) and rf.load_rgb("tokenizer_image.png").shape == (764, 1024, 3)


def test_derived_inputs():
    from oracle import warp_oracle as wo
    rgb = rf.load_rgb("diffusion_000000.png", size=(1280, 704))
    d = rf.pseudo_depth(rgb)
    assert d.shape == (704, 1280) and d.dtype == np.float32 and 1.5 < d.min() < d.max() < 4.6
    rel = wo.reliable_depth_mask(d[None, None], ratio_thresh=0.05)
    bnd = ~wo.reliable_depth_mask(d[None, None])[0, 0]
    assert 0.5 < rel.mean() < 0.97 and bnd.sum() > 20000  # ragged depth edges along the image's own outlines
    clip = rf.pan_clip(rf.load_rgb("diffusion_000000.png"), 17, 512, 512, step=4)
    assert clip.shape == (3, 17, 512, 512) and -1.0 <= clip.min() and clip.max() <= 1.0
    assert np.array_equal(clip[:, 1, :, :-4], clip[:, 0, :, 4:])  # frame t + 1 is frame t moved by 4 pixels
    assert 0.05 < np.abs(clip[:, 1] - clip[:, 0]).mean()  # ... which is real motion on natural content
