"""GPU parity of the HIP causal video tokenizer against (a) the golden outputs of the reference's own tokenizer modules
(tests/golden/tokenizer_small.npz, channels=16 -> register-staged conv path) and (b) the CPU oracle with wider channels
(channels=64 -> LDS-DMA conv path).

Stated tolerance: bf16 activations through ~45 layers (GroupNorm renormalises every block): relative L2 <= 3e-2 on the
latent and <= 4e-2 on the reconstruction against the fp32 evaluation of the same bf16 weights."""
import pytest
import torch

from tests.golden_io import load_tokenizer_case

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def test_tokenizer_matches_reference_golden():
    from gen3c_amd.tokenizer import CausalVideoTokenizerNet
    dev = torch.device("cuda:0")
    sd, x, z_ref, zin, y_ref = load_tokenizer_case()
    net = CausalVideoTokenizerNet(channels=16, device=dev)
    net.load_state_dict(sd, strict=True)
    z = net.encoder(x.to(dev))
    torch.cuda.synchronize()
    assert z.shape == z_ref.shape
    rz = _rel(z, z_ref)
    y = net.decoder(zin.to(dev))
    torch.cuda.synchronize()
    assert y.shape == y_ref.shape
    ry = _rel(y, y_ref)
    print(f"[tokenizer golden] encoder rel_l2={rz:.3e}  decoder rel_l2={ry:.3e}")
    assert torch.isfinite(z.float()).all() and torch.isfinite(y.float()).all()
    assert rz <= 2e-2 and ry <= 1.6e-2  # measured 1.3e-2 / 0.9e-2; bound = measured + margin


def test_tokenizer_wide_channels_vs_oracle():
    from gen3c_amd.tokenizer import CausalVideoTokenizerNet
    from oracle import tokenizer_oracle as tok
    dev = torch.device("cuda:0")
    net = CausalVideoTokenizerNet(channels=64, device=dev)
    sd = net.init_random(seed=3)
    sd32 = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(1, 3, 17, 32, 64, generator=g) * 2 - 1).to(torch.bfloat16)
    z = net.encoder(x.to(dev))
    z_ref = tok.encoder(sd32, x.float())
    rz = _rel(z, z_ref)
    zin = z_ref.to(torch.bfloat16)
    y = net.decoder(zin.to(dev))
    y_ref = tok.decoder(sd32, zin.float())
    ry = _rel(y, y_ref)
    torch.cuda.synchronize()
    print(f"[tokenizer ch64] encoder rel_l2={rz:.3e}  decoder rel_l2={ry:.3e}  shapes {tuple(z.shape)} {tuple(y.shape)}")
    assert z.shape == z_ref.shape and y.shape == y_ref.shape
    assert rz <= 2e-2 and ry <= 1.6e-2  # measured 1.3e-2 / 0.9e-2; bound = measured + margin


@pytest.mark.parametrize("T,H,W", [(17, 352, 640), (9, 704, 1280)])
def test_tokenizer_production_width_vs_oracle(T, H, W):
    """The shipped configuration: channels=128 (256 / 512-wide levels, layers3d.py:669-949). (17,352,640) covers 3 latent frames
    (two causal temporal strides); (9,704,1280) is the production resolution: its mid-block spatial attention runs over
    88*160 = 14 080 pixels (layers3d.py:345-383) - the 14 080^2 score matrix that the HIP path rounds to bf16 before the row
    softmax (as the bf16 reference does; the fp32 oracle does not). Oracle: fp32 evaluation of the same bf16 weights on the CPU
    (about 20 s / 70 s). Tolerances = measured (profiles/r2_parity_measured.txt) + margin."""
    from gen3c_amd.tokenizer import CausalVideoTokenizerNet
    from oracle import tokenizer_oracle as tok
    dev = torch.device("cuda:0")
    net = CausalVideoTokenizerNet(channels=128, device=dev)
    sd = net.init_random(seed=3)
    sd32 = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    g = torch.Generator().manual_seed(5)
    # smooth content + noise: closer to video statistics than white noise (GroupNorm / attention see structure)
    base = torch.nn.functional.interpolate(torch.rand(1, 3, max(T // 4, 2), H // 16, W // 16, generator=g), size=(T, H, W), mode="trilinear")
    x = ((base * 2 - 1) * 0.8 + 0.2 * (torch.rand(1, 3, T, H, W, generator=g) * 2 - 1)).clamp(-1, 1).to(torch.bfloat16)
    z = net.encoder(x.to(dev))
    torch.cuda.synchronize()
    z_ref = tok.encoder(sd32, x.float())
    rz = _rel(z, z_ref)
    zin = z_ref.to(torch.bfloat16)
    y = net.decoder(zin.to(dev))
    torch.cuda.synchronize()
    y_ref = tok.decoder(sd32, zin.float())
    ry = _rel(y, y_ref)
    print(f"[tokenizer ch128 {T}x{H}x{W}] encoder rel_l2={rz:.3e}  decoder rel_l2={ry:.3e}  shapes {tuple(z.shape)} {tuple(y.shape)}")
    assert z.shape == z_ref.shape and y.shape == y_ref.shape
    assert torch.isfinite(z.float()).all() and torch.isfinite(y.float()).all()
    assert rz <= 1.5e-2 and ry <= 1.8e-2  # measured 7.0e-3 / 9.2e-3 at both sizes


def test_spatial_attention_14080_pixels_vs_fp32_softmax():
    """CausalAttnBlock at the production size in isolation: one frame, 14 080 pixels, 512 channels. Sampled query rows against
    softmax(q k^T / sqrt(C)) v evaluated in fp64 from the q/k/v the HIP path itself produced (so only the attention arithmetic -
    score rounding to bf16, row softmax, P.V - is measured). Scores are given a realistic spread (std ~2.5)."""
    from gen3c_amd.tokenizer import CausalVideoTokenizerNet
    dev = torch.device("cuda:0")
    C, Hh, Ww = 512, 88, 160
    net = CausalVideoTokenizerNet(channels=128, device=dev)
    net.init_random(seed=9)
    name = "encoder.mid.attn_1.0"
    g = torch.Generator(device=dev).manual_seed(2)
    # sharpen q/k a little: score std ~2.7 (x36 would make the softmax one-hot - a regime where ANY bf16 score is off by half an e-fold)
    for n, sc in (("q", 1.6), ("k", 1.6)):
        net._w[f"{name}.{n}.conv3d.weight"] = (net._w[f"{name}.{n}.conv3d.weight"].float() * sc).to(torch.bfloat16)
    x = torch.randn(1, Hh, Ww, C, device=dev, generator=g).to(torch.bfloat16)
    y = net._spatial_attn(x, name)
    torch.cuda.synchronize()
    hn = net._gn(x, f"{name}.norm", False)
    q = net._conv(hn, f"{name}.q", "p1").view(-1, C).double()
    k = net._conv(hn, f"{name}.k", "p1").view(-1, C).double()
    v = net._conv(hn, f"{name}.v", "p1").view(-1, C).double()
    rows = torch.arange(0, Hh * Ww, 137, device=dev)
    s = (q[rows] @ k.T) * C ** -0.5
    spread = float(s.std())
    o = (torch.softmax(s, dim=-1) @ v).float().to(torch.bfloat16)
    ref = net._conv(o.view(1, -1, 1, C).contiguous(), f"{name}.proj_out", "p1").view(-1, C).float() + x.view(-1, C)[rows].float()
    got = y.view(-1, C)[rows].float()
    r_out = _rel(got, ref)
    # the attention branch alone (residual removed) - the part that carries the score rounding
    r_branch = _rel(got - x.view(-1, C)[rows].float(), ref - x.view(-1, C)[rows].float())
    print(f"[spatial attn 14080] score std {spread:.2f}  rel_l2 output {r_out:.3e}  attention branch {r_branch:.3e}")
    assert 1.5 < spread < 6.0
    assert r_out <= 1e-2 and r_branch <= 3e-2


@pytest.mark.parametrize("T,HW", [(16, 14080), (3, 3520), (8, 448)])
def test_flash_spatial_attention_d512_kernel_vs_fp32(T, HW):
    """csrc/attention_d512.hip through the C ABI (g3_spatial_attn_d512_bf16), round 4: single head, d = 512, per frame softmax(q k^T / sqrt(512)) v
    (layers3d.py:362-377). (16, 14 080): the benchmarked shape - 16 frames = 2 per XCD (frame-per-XCD workgroup mapping), 110 query blocks, 220 key tiles;
    (3, 3 520): frames not a multiple of 8 (plain mapping) and HW % 128 != 0 (ragged last query block); (8, 448): few tiles. Sampled rows of every frame
    against an fp32 softmax; operands with a realistic score spread (std ~2.5: the running maximum moves a few times per row -> the deferred rescale
    path). Tolerance as for the DiT kernels (rel-L2 <= 4e-3 measured ~3e-3 there; here the sum runs over 4x the dims)."""
    from gen3c_amd import _lib
    dev = torch.device("cuda:0")
    C = 512
    g = torch.Generator(device=dev).manual_seed(T * 1000 + HW)
    q = (torch.randn(T, HW, C, device=dev, generator=g) * 0.75).to(torch.bfloat16)
    k = (torch.randn(T, HW, C, device=dev, generator=g) * 0.75).to(torch.bfloat16)
    k[:, ::97] *= 3.0  # a few dominant keys: rows whose maximum arrives late
    v = torch.randn(T, HW, C, device=dev, generator=g).to(torch.bfloat16)
    vT = v.reshape(T * HW, C).t().contiguous()  # [C, T*HW]: frame f = columns [f HW, (f+1) HW)
    o = torch.full((T, HW, C), float("nan"), device=dev, dtype=torch.bfloat16)
    lib = _lib.load()
    _lib.check(lib.g3_spatial_attn_d512_bf16(q.data_ptr(), k.data_ptr(), vT.data_ptr(), T * HW, HW, o.data_ptr(), T, HW, C ** -0.5,
                                             torch.cuda.current_stream().cuda_stream), "g3_spatial_attn_d512_bf16")
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    rows = torch.cat([torch.arange(0, min(64, HW), device=dev), torch.randint(0, HW, (128,), device=dev, generator=g), torch.arange(HW - 64, HW, device=dev)])
    worst, spread = 0.0, 0.0
    for f in range(T):
        sc = (q[f, rows].float() @ k[f].float().t()) * C ** -0.5
        spread = max(spread, float(sc.std()))
        ref = torch.softmax(sc, dim=-1) @ v[f].float()
        worst = max(worst, _rel(o[f, rows], ref))
    print(f"[flash spatial attention d512, T={T} HW={HW}] worst frame rel-L2 vs fp32 softmax on {rows.numel()} rows: {worst:.3e} (score std {spread:.2f})")
    assert worst <= 4e-3


def test_flash_spatial_attention_matches_the_three_kernel_path(monkeypatch):
    """The same CausalAttnBlock through both forms (A/B switch G3_TOK_FLASH_ATTN): flash kernel vs scores GEMM -> row softmax -> P.V GEMM. The three-kernel
    form rounds the SCORES to bf16 before the softmax (its score matrix lives in HBM as bf16), the flash kernel keeps them in fp32 and rounds only P: it
    is the more accurate of the two (2.4e-3 vs fp32 softmax in the test above; the score rounding alone is ~7e-3) - agreement at the level of the coarser one.
    Measured 8.4e-3."""
    from gen3c_amd import tokenizer as tkmod
    dev = torch.device("cuda:0")
    C, Hh, Ww, T = 512, 44, 80, 3
    net = tkmod.CausalVideoTokenizerNet(channels=128, device=dev)
    net.init_random(seed=9)
    name = "encoder.mid.attn_1.0"
    x = torch.randn(T, Hh, Ww, C, device=dev, generator=torch.Generator(device=dev).manual_seed(4)).to(torch.bfloat16)
    monkeypatch.setattr(tkmod, "_FLASH_SPATIAL_ATTN", True)
    y_flash = net._spatial_attn(x, name)
    net._pending_stats = None
    monkeypatch.setattr(tkmod, "_FLASH_SPATIAL_ATTN", False)
    y_three = net._spatial_attn(x, name)
    torch.cuda.synchronize()
    r = _rel(y_flash.float() - x.float(), y_three.float() - x.float())
    print(f"[spatial attention, flash vs three-kernel path, attention branch] rel-L2 {r:.3e}")
    assert r <= 1.3e-2


def test_video_tokenizer_interface_roundtrip_shapes():
    from gen3c_amd.tokenizer import VideoTokenizer
    dev = torch.device("cuda:0")
    tk = VideoTokenizer(pixel_chunk_duration=9, channels=16, device=dev)
    tk.net.init_random(seed=1)
    tk.register_mean_std(torch.zeros(16, 32), torch.ones(16, 32))
    assert tk.latent_chunk_duration == 2 and tk.get_latent_num_frames(18) == 4 and tk.get_pixel_num_frames(4) == 18
    x = torch.rand(1, 3, 18, 32, 32, device=dev).to(torch.bfloat16) * 2 - 1
    z = tk.encode(x)
    assert z.shape == (1, 16, 4, 4, 4)
    y = tk.decode(z)
    assert y.shape == x.shape and torch.isfinite(y.float()).all()


def _script_archive(weights: dict, path):
    """A TorchScript archive whose state_dict() has exactly `weights`' dotted names (what encoder.jit / decoder.jit provide to the
    reference's loader, tokenizer/inference/utils.py:50-92)."""
    class Node(torch.nn.Module):
        pass

    class Root(torch.nn.Module):
        def forward(self, x: torch.Tensor) -> torch.Tensor:
            return x

    root = Root()
    for name, w in weights.items():
        mod = root
        parts = name.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                setattr(mod, p, Node())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], torch.nn.Parameter(w.clone(), requires_grad=False))
    torch.jit.script(root).save(str(path))


def test_video_tokenizer_load_weights_from_jit_archives(tmp_path):
    """VideoTokenizer.load_weights (pretrained_vae.py:194-214, 342-359): encoder.jit / decoder.jit TorchScript archives +
    mean_std.pt, then encode = (net(x) - mean) / std and decode = net^-1(z * std + mean) against the reference goldens."""
    from gen3c_amd.tokenizer import VideoTokenizer
    dev = torch.device("cuda:0")
    sd, x, z_ref, zin, y_ref = load_tokenizer_case()
    enc = {k: v.float() for k, v in sd.items() if k.startswith(("encoder.", "quant_conv."))}
    dec = {k: v.float() for k, v in sd.items() if k.startswith(("post_quant_conv.", "decoder."))}
    assert len(enc) + len(dec) == len(sd)
    _script_archive(enc, tmp_path / "encoder.jit")
    _script_archive(dec, tmp_path / "decoder.jit")
    g = torch.Generator().manual_seed(11)
    mean, std = torch.randn(16 * 32, generator=g) * 0.1, 0.5 + torch.rand(16 * 32, generator=g)
    torch.save((mean, std), tmp_path / "mean_std.pt")
    tk = VideoTokenizer(pixel_chunk_duration=9, channels=16, device=dev)
    tk.load_weights(str(tmp_path))
    m = mean.view(16, 32)[:, :2].reshape(1, 16, 2, 1, 1)
    s = std.view(16, 32)[:, :2].reshape(1, 16, 2, 1, 1)
    z = tk.encode(x.to(dev))
    rz = _rel(z, (z_ref - m) / s)
    y = tk.decode(((zin.float() - m) / s).to(torch.bfloat16).to(dev))
    ry = _rel(y, y_ref)
    print(f"[tokenizer jit archives] encode rel_l2={rz:.3e} decode rel_l2={ry:.3e}")
    assert rz <= 2e-2 and ry <= 1.6e-2  # measured 1.3e-2 / 0.9e-2; bound = measured + margin


@pytest.mark.parametrize("rows,n,ld", [(37, 14080, 14080), (16, 1024, 1100 // 8 * 8), (5, 16392, 16392), (9, 100, 104), (3, 16384, 16384)])
def test_softmax_rows_both_paths(rows, n, ld):
    """g3_softmax_rows_bf16: register-resident path (n <= 16384, n % 8 == 0) and the streaming path, against torch fp32."""
    from gen3c_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(rows + n)
    x = (torch.randn(rows, ld, device=dev, generator=g) * 3).to(torch.bfloat16)
    ref = torch.softmax(x[:, :n].float() * 0.37, dim=-1)
    y = x.clone()
    _lib.check(_lib.load().g3_softmax_rows_bf16(y.data_ptr(), ld, rows, n, 0.37, torch.cuda.current_stream().cuda_stream), "softmax")
    torch.cuda.synchronize()
    assert torch.equal(y[:, n:], x[:, n:])  # padding columns untouched
    assert _rel(y[:, :n], ref) < 4e-3 and float((y[:, :n].float().sum(-1) - 1).abs().max()) < 2e-2


_CONV_GEOMS = {  # kt, kh, kw, st, sh, sw, ot, oh, ow (gen3c_amd/tokenizer.py: _GEOM)
    "s3": (1, 3, 3, 1, 1, 1, 0, -1, -1), "t3": (3, 1, 1, 1, 1, 1, -2, 0, 0), "p1": (1, 1, 1, 1, 1, 1, 0, 0, 0),
    "s3s2": (1, 3, 3, 1, 2, 2, 0, 0, 0), "t3s2": (3, 1, 1, 2, 1, 1, -2, 0, 0),
}


@pytest.mark.parametrize("kind,K,N,T,H,W", [("s3", 256, 256, 3, 24, 40), ("t3", 256, 256, 5, 16, 24), ("p1", 128, 512, 2, 17, 23), ("s3", 64, 128, 2, 9, 31),
                                             ("s3s2", 128, 256, 2, 23, 37), ("t3s2", 192, 192, 7, 8, 16), ("t3", 512, 512, 4, 8, 20), ("s3", 192, 128, 1, 40, 64)])
@pytest.mark.parametrize("res", [False, True])
def test_conv_one_wave_kernel_bitwise_equals_pingpong_and_delivers_groupnorm_statistics(kind, K, N, T, H, W, res):
    """gemm_w4_conv.hpp (one wave per SIMD, gathered token rows by per-lane address) against the 8-wave ping-pong implicit GEMM it replaces:
    same accumulation order over (tap, channel) => BITWISE equal outputs, on every tokenizer geometry incl. borders (zero page), the causal
    front replication, strides, ragged M / N tiles; and the GroupNorm statistics delivered by its epilogue against fp64 sums of the output.
    conv_w4 = 1 (default) computes a tap change's token addresses in the MFMA gaps of the barrier K step (32-bit form), conv_w4 = 2 between two
    statements (the form tensors of 4 GiB and more keep): both are checked."""
    from gen3c_amd import _lib, ops
    dev = torch.device("cuda:0")
    lib = _lib.load()
    kt, kh, kw, st, sh, sw, ot, oh, ow = _CONV_GEOMS[kind]
    To, Ho, Wo = T, H, W
    if kind == "s3s2":
        Ho, Wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
    elif kind == "t3s2":
        To = (T + 2 - 3) // 2 + 1
    g = torch.Generator(device=dev).manual_seed(K + N + T)
    x = torch.randn(T, H, W, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(kt * kh * kw, N, K, device=dev, generator=g) / (K * kt * kh * kw) ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
    r = torch.randn(To, Ho, Wo, N, device=dev, generator=g).to(torch.bfloat16) if res else None
    outs = {}
    for w4 in (1, 2, 0):
        ops.set_option("conv_w4", w4)
        o = torch.full((To, Ho, Wo, N), float("nan"), device=dev, dtype=torch.bfloat16)
        stats = torch.zeros(To, 2, device=dev, dtype=torch.float64)
        rc = lib.g3_conv3d_cl_gnstats_bf16(x.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), r.data_ptr() if res else None, N, o.data_ptr(), N, K, N, T, H, W,
                                           To, Ho, Wo, kt, kh, kw, st, sh, sw, ot, oh, ow, stats.data_ptr(), Ho * Wo, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "g3_conv3d_cl_gnstats_bf16")
        torch.cuda.synchronize()
        outs[w4] = (o, stats)
    ops.set_option("conv_w4", 1)
    assert torch.isfinite(outs[1][0].float()).all()
    assert torch.equal(outs[1][0], outs[0][0]), f"one-wave conv != ping-pong conv on {(outs[1][0] != outs[0][0]).sum().item()} elements"
    assert torch.equal(outs[2][0], outs[0][0]), f"one-wave conv (tap change between statements) != ping-pong conv on {(outs[2][0] != outs[0][0]).sum().item()} elements"
    of = outs[1][0].double().reshape(To, -1)
    ref = torch.stack([of.sum(1), (of * of).sum(1)], dim=1)
    for w4 in (1, 2, 0):
        st_ = outs[w4][1]
        err = ((st_ - ref).abs() / (ref.abs() + 1.0)).max().item()
        assert err < 2e-6, f"conv_w4={w4}: GroupNorm statistics off by {err:.2e}"


@pytest.mark.parametrize("T,HW,C", [(16, 1000, 512), (3, 77, 64), (5, 256, 256), (1, 40, 128)])
def test_temporal_attention_per_pixel_kernel_bitwise_equals_first_form_and_matches_fp32(T, HW, C):
    """CausalTemporalAttnBlock core (layers3d.py:386-427): the one-wave-per-pixel kernel (K / V rows of a pixel read once for all query frames)
    against the one-wave-per-(pixel, query frame) kernel - same operation order => bitwise - and against an fp32 causal softmax."""
    from gen3c_amd import _lib, ops
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(T * 7 + C)
    q, k, v = (torch.randn(T, HW, C, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    outs = []
    for px in (1, 0):
        ops.set_option("tok_tattn_px", px)
        o = torch.empty_like(q)
        _lib.check(lib.g3_temporal_attn_cl_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), T, HW, C, float(C) ** -0.5,
                                                torch.cuda.current_stream().cuda_stream), "g3_temporal_attn_cl_bf16")
        outs.append(o)
    ops.set_option("tok_tattn_px", 1)
    assert torch.equal(outs[0], outs[1])
    qf, kf, vf = (t.float().permute(1, 0, 2) for t in (q, k, v))  # [HW, T, C]
    sc = (qf @ kf.transpose(1, 2)) * float(C) ** -0.5
    sc = sc.masked_fill(torch.triu(torch.ones(T, T, device=dev, dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(sc, dim=-1) @ vf).permute(1, 0, 2)
    r = _rel(outs[0], ref)
    assert r < 8e-3, f"rel-L2 {r:.3e} vs fp32 causal softmax"


def test_image_branch_single_frame_vs_oracle():
    """The image half of JointImageVideoSharedJITTokenizer (pretrained_vae.py:520-545, 588-611): a T == 1 input goes through the SAME encoder /
    decoder with the per-channel statistics of image_mean_std.pt. GEN3C's entry points never take it; the plug-in offers it."""
    from gen3c_amd.tokenizer import VideoTokenizer
    from oracle import tokenizer_oracle as tok
    dev = torch.device("cuda:0")
    tk = VideoTokenizer(pixel_chunk_duration=9, channels=16, device=dev)
    sd = tk.net.init_random(seed=4)
    g = torch.Generator().manual_seed(2)
    mean, std = torch.randn(16, generator=g) * 0.1, torch.rand(16, generator=g) * 0.5 + 0.75
    tk.register_mean_std(torch.zeros(16, 32), torch.ones(16, 32))
    tk.register_image_mean_std(mean, std)
    assert tk.get_latent_num_frames(1) == 1 and tk.get_pixel_num_frames(1) == 1
    x = (torch.rand(2, 3, 1, 32, 48, generator=g) * 2 - 1).to(torch.bfloat16)
    z = tk.encode(x.to(dev))
    assert z.shape == (2, 16, 1, 4, 6)
    y = tk.decode(z)
    assert y.shape == x.shape
    tsd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    m5, s5 = mean.to(torch.bfloat16).float().reshape(1, 16, 1, 1, 1), std.to(torch.bfloat16).float().reshape(1, 16, 1, 1, 1)
    z_ref = torch.cat([tok.encode(tsd, x[b:b + 1].float(), m5, s5) for b in range(2)])
    y_ref = torch.cat([tok.decode(tsd, z[b:b + 1].float().cpu(), m5, s5) for b in range(2)])
    rz, ry = _rel(z.cpu(), z_ref), _rel(y.cpu(), y_ref)
    print(f"[tokenizer image branch] encode rel_l2={rz:.3e} decode rel_l2={ry:.3e}")
    assert rz <= 2e-2 and ry <= 2e-2  # measured 1.06e-2 / 1.18e-2 (channels = 16: the reduced-width goldens measure 1.3e-2 / 0.9e-2)


@pytest.mark.parametrize("T,H,W", [(17, 512, 512), (9, 360, 640)])
def test_plugin_encode_decode_reference_test_shape(T, H, W):
    """The reference's only hot-path test (tokenizer/modules/layers3d_test.py:32-114) encodes a 17 x 512 x 512 centre crop to a latent
    (1, 16, (17 - 1) // 8 + 1, 512 // 8, 512 // 8) = (1, 16, 3, 64, 64) and decodes it back to the input's shape: replayed here through the plug-in
    surface (`VideoTokenizer.encode` / `.decode`, with latent mean / std as pretrained_vae.py:365-405 applies them) - and, since random weights are
    all there is, with VALUES against the fp32 oracle, which the reference's shape-only test does not check. H = W = 512 is the only non-16:9 case in
    the suite (mid-level attention over 64 x 64 = 4 096 pixels -> the flash d = 512 kernel). (9, 360, 640): a size whose mid level is 45 x 80 =
    3 600 pixels - not a multiple of 64, so the spatial attention takes the score-matrix path, and 3 x 45 x 80 rows leave ragged 256-row
    convolution tiles at every level."""
    from gen3c_amd.tokenizer import VideoTokenizer
    from oracle import tokenizer_oracle as tok
    dev = torch.device("cuda:0")
    tk = VideoTokenizer(pixel_chunk_duration=T, channels=128, device=dev)
    sd = tk.net.init_random(seed=11)
    sd32 = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    g = torch.Generator().manual_seed(7)
    tl = (T - 1) // 8 + 1
    mean = torch.randn(16, tl, generator=g) * 0.1
    std = torch.rand(16, tl, generator=g) * 0.5 + 0.75
    tk.register_mean_std(mean, std)
    base = torch.nn.functional.interpolate(torch.rand(1, 3, max(T // 4, 2), H // 16, W // 16, generator=g), size=(T, H, W), mode="trilinear")
    x = ((base * 2 - 1) * 0.8 + 0.2 * (torch.rand(1, 3, T, H, W, generator=g) * 2 - 1)).clamp(-1, 1).to(torch.bfloat16)
    z = tk.encode(x.to(dev))
    torch.cuda.synchronize()
    assert tuple(z.shape) == (1, 16, tl, H // 8, W // 8)  # layers3d_test.py:96-99
    m32 = mean.to(torch.bfloat16).float().view(1, 16, tl, 1, 1)
    s32 = std.to(torch.bfloat16).float().view(1, 16, tl, 1, 1)
    z_ref = tok.encode(sd32, x.float(), m32, s32)
    rz = _rel(z, z_ref)
    zin = z_ref.to(torch.bfloat16)
    y = tk.decode(zin.to(dev))
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(x.shape)  # layers3d_test.py:113
    y_ref = tok.decode(sd32, zin.float(), m32, s32)
    ry = _rel(y, y_ref)
    print(f"[tokenizer plug-in {T}x{H}x{W}] encode rel_l2={rz:.3e}  decode rel_l2={ry:.3e}  latent {tuple(z.shape)}")
    assert torch.isfinite(z.float()).all() and torch.isfinite(y.float()).all()
    assert rz <= 1.5e-2 and ry <= 1.8e-2  # the tolerance of test_tokenizer_production_width_vs_oracle (same depth, same arithmetic)
