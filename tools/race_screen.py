"""Race screen: every LDS hand-off of the MFMA kernels must be fenced, so results may not depend on wave timing. Runs the
product library against the same sources built with -DG3_AB_JITTER=<n> (pseudo-random waves sleep n*64 cycles at tile / phase
boundaries) and demands BITWISE equal outputs.
  build (here, before gpurun):  python tools/race_screen.py --build [n]
  run (GPU box):                python tools/race_screen.py"""
import ctypes as C
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

if "--build" in sys.argv:
    from gen3c_amd import build
    n = [a for a in sys.argv[1:] if a.isdigit()]
    extra = tuple(a for a in sys.argv[1:] if a.startswith("-D"))  # e.g. -DG3_AB_OMIT_PROLOGUE_BARRIER: self-test of the screen
    # default perturbations: sleeping waves at tile / phase boundaries AND no static wave priorities (the variant that exposed the
    # attention prologue races of round 1; -DG3_AB_OMIT_PROLOGUE_BARRIER re-introduces one of them as a self-test of this screen)
    jit = () if "--no-jitter" in sys.argv else (f"-DG3_AB_JITTER={n[0] if n else 30}",)
    extra = extra + ("-DG3_AB_NO_ATTN_SETPRIO", "-DG3_AB_NO_GEMM_SETPRIO")
    print(build.build(extra_flags=jit + extra, suffix="_ab", force=True))
    sys.exit(0)

import torch  # noqa: E402
from gen3c_amd import _lib, ops  # noqa: E402

base = _lib.load()
alt = C.CDLL(str(ROOT / "gen3c_amd" / "lib" / "libgen3c_hip_ab.so"))
for name, argtypes in _lib.SIGNATURES.items():
    getattr(alt, name).argtypes = argtypes
    getattr(alt, name).restype = _lib._RESTYPES.get(name, C.c_int)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(3)
bad = 0


def both(fn):
    outs = []
    for lib in (base, alt):
        outs.append(fn(lib))
    torch.cuda.synchronize()
    return outs


def report(what, a, b):
    global bad
    eq = bool(torch.equal(a, b))
    bad += not eq
    print(f"{'ok  ' if eq else 'DIFF'} {what}" + ("" if eq else f"  rel-l2 {float((a.float() - b.float()).norm() / a.float().norm()):.3e}"), flush=True)


# ---- attention (long and short contexts, ragged tails, segmented V^T)
for (Sq, Skv, H, segs) in [(56320, 56320, 4, 1), (7040, 56320, 4, 8), (1000, 449, 3, 1), (56320, 512, 8, 1), (300, 128, 2, 2)]:
    q = torch.randn(Sq, H * 128, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(Skv, H * 128, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(Skv, H * 128, device=dev, generator=g).to(torch.bfloat16)
    if segs == 1:
        vt = ops.transpose_v(v, Skv, 1, H)
    else:
        sl = Skv // segs
        vt = torch.stack([ops.transpose_v(v[i * sl:(i + 1) * sl], sl, 1, H) for i in range(segs)]).contiguous()
    ld = vt.shape[-1]
    for variant in (3, 4, 6, 8, 9, 10, 11):
        def run(lib):
            lib.g3_set_option(b"attn_variant", variant)
            o = torch.empty_like(q)
            args = (q.data_ptr(), H * 128, H * 128, 128, k.data_ptr(), H * 128, H * 128, 128, vt.data_ptr(), ld, H * 128 * ld, 128 * ld)
            tail = (o.data_ptr(), H * 128, H * 128, 128, Sq, Skv, 1, H, 128, 1.0 / math.sqrt(128), st)
            rc = lib.g3_flash_attn_fwd_kvseg_bf16(*args, Skv // segs, H * 128 * ld, *tail) if segs > 1 else lib.g3_flash_attn_fwd_bf16(*args, *tail)
            assert rc == 0, lib.g3_last_error()
            return o
        a, b = both(run)
        report(f"attention Sq={Sq} Skv={Skv} H={H} segs={segs} variant={variant}", a, b)
    del q, k, v, vt

# ---- cross-attention form (round 6): Q norm inside the Q load + zero key tail in closed form
for (Sq, Skv, H, live) in [(56320, 512, 8, 64), (1000, 512, 3, 0), (3000, 256, 2, 100)]:
    q = torch.randn(Sq, H * 128, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(Skv, H * 128, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(Skv, H * 128, device=dev, generator=g).to(torch.bfloat16)
    if live:
        k[live:] = 0
        v[live:] = 0
    wq = (torch.rand(128, device=dev, generator=g) + 0.5).to(torch.bfloat16)
    vt = ops.transpose_v(v, Skv, 1, H)
    ld = vt.shape[-1]

    def run(lib):
        lib.g3_set_option(b"attn_variant", 0)
        o = torch.empty_like(q)
        rc = lib.g3_cross_attn_fwd_bf16(q.data_ptr(), H * 128, H * 128, 128, wq.data_ptr(), 1e-6, k.data_ptr(), H * 128, H * 128, 128, vt.data_ptr(), ld, H * 128 * ld, 128 * ld,
                                        o.data_ptr(), H * 128, H * 128, 128, Sq, Skv, live, 1, H, 128, 1.0 / math.sqrt(128), st)
        assert rc == 0, lib.g3_last_error()
        return o
    a, b = both(run)
    report(f"cross-attention form Sq={Sq} Skv={Skv} H={H} live={live} (q norm in the kernel)", a, b)
    del q, k, v, vt

# ---- GEMM: every K-loop structure x epilogues x ragged shapes
for (M, N, K, epi) in [(56320, 4096, 4096, 2), (7040, 12288, 4096, 0), (4096, 16384, 4096, 1), (3000, 4096, 16384, 2), (513, 264, 192, 3), (300, 520, 128, 0),
                       (256, 256, 64, 0), (8192, 12288, 4096, 0), (16384, 16384, 4096, 1), (8448, 4096, 2432, 2)]:
    a_ = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w_ = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    gate = torch.randn(1, N, device=dev, generator=g).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    # 13 = one wave per SIMD as a persistent tile loop (round 3; runs where a workgroup gets >= 2 tiles); 23 = persistent with the DEFERRED epilogue
    # (round 5, gemm_w4e.hpp: where M, N are multiples of 256, K of 128 with >= 38 K tiles - the jitter hits its tile boundaries and period K tiles)
    for pp in (0, 1, 2, 3, 13, 23):
        def run(lib):
            lib.g3_set_option(b"gemm_pingpong", pp % 10)
            lib.g3_set_option(b"gemm_persistent", 1 if pp == 13 else 0)
            lib.g3_set_option(b"gemm_deferred", 1 if pp == 23 else 0)
            o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            rc = lib.g3_gemm_bf16_nt(a_.data_ptr(), K, w_.data_ptr(), K, o.data_ptr(), N, M, N, K, epi, gate.data_ptr() if epi >= 2 else None, 1, N,
                                     res.data_ptr() if epi in (2, 4) else None, N, st)
            assert rc == 0, lib.g3_last_error()
            return o
        x, y = both(run)
        report(f"gemm {M}x{N}x{K} epi{epi} pingpong={pp}", x, y)
    del a_, w_, gate, res

# ---- implicit-GEMM convolutions (tokenizer geometries: 1x3x3, causal 3x1x1, strided)
for (C_in, C_out, T, Hh, Ww, kt, kh, kw, st_, sh, sw) in [(128, 256, 5, 48, 64, 1, 3, 3, 1, 1, 1), (256, 256, 6, 40, 48, 3, 1, 1, 1, 1, 1), (128, 128, 5, 48, 64, 1, 3, 3, 1, 2, 2),
                                                           (256, 256, 7, 24, 32, 3, 1, 1, 2, 1, 1)]:
    x_ = torch.randn(T * Hh * Ww, C_in, device=dev, generator=g).to(torch.bfloat16)
    w_ = (torch.randn(kt * kh * kw, C_out, C_in, device=dev, generator=g) / math.sqrt(C_in * kt * kh * kw)).to(torch.bfloat16)
    bias = torch.randn(C_out, device=dev, generator=g).to(torch.bfloat16)
    To = (T + st_ - 1) // st_ if kt == 3 else T
    Ho, Wo = (Hh // sh, Ww // sw)
    ot, oh, ow = (-(kt - 1), 0 if sh == 2 else -(kh // 2), 0 if sw == 2 else -(kw // 2))
    if kt == 3 and st_ == 2:
        To = (T - 1) // 2 + 1
    resid = torch.randn(To * Ho * Wo, C_out, device=dev, generator=g).to(torch.bfloat16)
    for (pp, w4, with_res) in ((0, 0, False), (2, 0, False), (2, 1, False), (2, 1, True), (2, 0, True)):  # w4 = 1: gemm_w4_conv.hpp (round 3)
        def run(lib):
            lib.g3_set_option(b"gemm_pingpong", pp)
            lib.g3_set_option(b"conv_w4", w4)
            o = torch.empty(To * Ho * Wo, C_out, device=dev, dtype=torch.bfloat16)
            rc = lib.g3_conv3d_cl_bf16(x_.data_ptr(), C_in, w_.data_ptr(), C_in, bias.data_ptr(), resid.data_ptr() if with_res else None, C_out, o.data_ptr(), C_out,
                                       C_in, C_out, T, Hh, Ww, To, Ho, Wo, kt, kh, kw, st_, sh, sw, ot, oh, ow, st)
            assert rc == 0, lib.g3_last_error()
            return o
        x, y = both(run)
        report(f"conv {C_in}->{C_out} k=({kt},{kh},{kw}) s=({st_},{sh},{sw}) on {T}x{Hh}x{Ww} pingpong={pp} conv_w4={w4} residual={with_res}", x, y)
# ---- split-KV attention: partial outputs + merge (round 3)
for (Sq, Skv, H, variant) in [(7040, 14080, 4, 11), (7040, 14080, 4, 4), (1000, 448, 2, 4)]:
    q = torch.randn(Sq, H * 128, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(Skv, H * 128, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(Skv, H * 128, device=dev, generator=g).to(torch.bfloat16)
    vt = ops.transpose_v(v, Skv, 1, H)
    ld = vt.shape[-1]

    def run(lib):
        op = torch.empty(Sq, H * 128, device=dev, dtype=torch.float32)
        lse = torch.empty(1, H, Sq, device=dev, dtype=torch.float32)
        rc = lib.g3_flash_attn_fwd_ex_bf16(q.data_ptr(), H * 128, H * 128, 128, k.data_ptr(), H * 128, H * 128, 128, vt.data_ptr(), ld, H * 128 * ld, 128 * ld, 0, 0,
                                           None, op.data_ptr(), lse.data_ptr(), H * 128, H * 128, 128, Sq, Skv, 1, H, 128, 1.0 / math.sqrt(128), variant, st)
        assert rc == 0, lib.g3_last_error()
        return torch.cat([op.reshape(-1), lse.reshape(-1)])
    a, b = both(run)
    report(f"split-KV partial attention Sq={Sq} Skv={Skv} H={H} variant={variant}", a, b)
# ---- the whole tokenizer at the size bench.py times it (round 4): 121 x 704 x 1280, channels = 128 - the 1.79 GB activations, the spatial attention's frames
# on two streams with per-stream score buffers, GroupNorm statistics from conv epilogues, all 16 latent frames in the temporal attention
if "--no-tokenizer" not in sys.argv:
    import bench  # noqa: E402
    from gen3c_amd.tokenizer import CausalVideoTokenizerNet  # noqa: E402
    for lib in (base, alt):
        lib.g3_set_option(b"attn_variant", 0)
        lib.g3_set_option(b"gemm_pingpong", 3)
        lib.g3_set_option(b"gemm_persistent", 0)
        lib.g3_set_option(b"gemm_deferred", 1)
        lib.g3_set_option(b"conv_w4", 1)
    tnet = CausalVideoTokenizerNet(channels=128, device=dev)
    tnet.init_random(seed=3)
    clip = bench.tokenizer_bench_clip(dev)
    outs = []
    for lib in (base, alt, base):
        _lib._lib = lib  # the product's loader hands out this handle
        z = tnet.encoder(clip)
        y = tnet.decoder(z)
        torch.cuda.synchronize()
        outs.append((z, y))
    _lib._lib = base
    # the fp64 atomics that collect GroupNorm statistics may add in any order: bitwise equality is expected in practice (measured), a last-bit
    # difference of a statistic would show as ~1e-7, a race as >= 1e-3
    for what, i in (("encode", 0), ("decode", 1)):
        report(f"tokenizer {what} 121x704x1280 channels=128, product vs jitter build", outs[0][i], outs[1][i])
        report(f"tokenizer {what} 121x704x1280 channels=128, product run twice", outs[0][i], outs[2][i])
    del outs, clip, tnet
for lib in (base, alt):
    lib.g3_set_option(b"attn_variant", 0)
    lib.g3_set_option(b"gemm_pingpong", 3)
    lib.g3_set_option(b"gemm_persistent", 0)
    lib.g3_set_option(b"conv_w4", 1)
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FOUND {bad} DIFFERENCES")
sys.exit(1 if bad else 0)
