"""CPU: pins oracle/tokenizer_oracle.py against the reference's own tokenizer modules (tests/golden/tokenizer_small.npz)."""
import torch

from oracle import tokenizer_oracle as tok
from tests.golden_io import load_tokenizer_case


def test_encoder_decoder_match_reference_golden():
    sd, x, z_ref, zin, y_ref = load_tokenizer_case()
    sd32 = {k: v.float() for k, v in sd.items()}
    z = tok.encoder(sd32, x.float())
    assert z.shape == z_ref.shape
    torch.testing.assert_close(z, z_ref, rtol=1e-4, atol=1e-4)
    y = tok.decoder(sd32, zin.float())
    assert y.shape == y_ref.shape
    torch.testing.assert_close(y, y_ref, rtol=1e-4, atol=2e-4)


def test_haar_roundtrip_is_identity():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, 9, 16, 24, generator=g)
    h = tok.haar_patch3d(x)
    assert h.shape == (1, 192, 3, 4, 6)
    torch.testing.assert_close(tok.haar_unpatch3d(h), x, rtol=1e-5, atol=1e-5)


def test_taps_convolution_backend_equals_conv3d():
    """oracle.tokenizer_oracle.CONV_IMPL = "taps" (one matmul per kernel tap; used by the full-size on-device evaluation in
    tests/test_fullsize_gpu.py) is the same convolution as the pinned F.conv3d form: whole encoder / decoder on the golden case."""
    sd, x, z_ref, zin, y_ref = load_tokenizer_case()
    sd32 = {k: v.float() for k, v in sd.items()}
    tok.CONV_IMPL = "taps"
    try:
        z = tok.encoder(sd32, x.float())
        y = tok.decoder(sd32, zin.float())
    finally:
        tok.CONV_IMPL = "torch"
    torch.testing.assert_close(z, z_ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y, y_ref, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(z, tok.encoder(sd32, x.float()), rtol=1e-5, atol=1e-5)
