#!/usr/bin/env python
"""bench.py - denoise-steps/sec of the GEN3C-Cosmos-7B DiT on MI355X (BASELINE.json metric).

One "step" = one EDM-Euler denoise step of `DiffusionV2WModel.generate_samples_from_batch` (model_v2w.py:130-149):
build the network input, net(cond), net(uncond), CFG, conditioning-frame replacement, Euler update - on the
121x704x1280 video latent [1,16,16,88,160] (56 320 tokens), random-init Cosmos-7B weights (28 blocks x 4096, 32 heads),
synthetic conditions of SURVEY.md 8d. Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N>1 = context parallel over the 16 latent frames, one process per GPU over RCCL. Both launch forms work:
  * under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`): RANK / WORLD_SIZE come
    from the environment;
  * bare (`python bench.py --gpus N`): no WORLD_SIZE in the environment -> this process re-executes itself under
    `torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (self_launch_argv), the counterpart of the
    reference's `torchrun --nproc_per_node=N gen3c_single_image.py --num_gpus N` (gen3c_single_image.py:248-255).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak per MI355X (/opt/skills/guides/MI355X_MICROARCH.md:42)


def dit_forward_flops(N, D=4096, M=512, Dctx=1024, L=28, patch_dim=328, out_dim=64):
    """L*(28 N D^2 + 4 N^2 D + 4 N M D + 4 M Dctx D) + 2 N patch D + 2 N D out."""
    return L * (28 * N * D * D + 4 * N * N * D + 4 * N * M * D + 4 * M * Dctx * D) + 2 * N * patch_dim * D + 2 * N * D * out_dim


def cpu_baseline(threads: int, blocks: int = 1):
    """Oracle ('port') timed on the host cores on a bounded sample of BASELINE.json's configs[0] shape: ONE Cosmos-7B-width block
    (D=4096, 32 heads, MLP 16384, context 512x1024) of a DiT forward on the 16x64x64 latent = 16 384 tokens, fp32, once. Measured: that
    block (incl. patch embedding / final layer). Extrapolated by FLOPs: the other 27 blocks, the second forward of the step and the
    56 320-token size of configs[1] (the full-size CPU step would take ~1.7 h)."""
    from oracle import dit_oracle
    torch.set_num_threads(threads)
    D, H, T, Hh, Ww, M = 4096, 32, 16, 64, 64, 512
    g = torch.Generator().manual_seed(0)
    sd = {}
    def W(*s, scale=0.02):
        return torch.randn(*s, generator=g) * scale
    sd["x_embedder.proj.1.weight"] = W(D, 328)
    sd["pos_embedder.seq"] = torch.arange(128, dtype=torch.float)
    for a, n in (("t", 128), ("h", 120), ("w", 120)):
        sd[f"extra_pos_embedder.pos_emb_{a}"] = W(n, D)
    sd["t_embedder.1.linear_1.weight"] = W(D, D)
    sd["t_embedder.1.linear_2.weight"] = W(3 * D, D)
    for bi in range(blocks):
        pre = f"blocks.block{bi}.blocks"
        for j, cd in ((0, D), (1, 1024)):
            a = f"{pre}.{j}.block.attn"
            sd[f"{a}.to_q.0.weight"] = W(D, D); sd[f"{a}.to_q.1.weight"] = torch.ones(128)
            sd[f"{a}.to_k.0.weight"] = W(D, cd); sd[f"{a}.to_k.1.weight"] = torch.ones(128)
            sd[f"{a}.to_v.0.weight"] = W(D, cd); sd[f"{a}.to_out.0.weight"] = W(D, D)
        sd[f"{pre}.2.block.layer1.weight"] = W(4 * D, D); sd[f"{pre}.2.block.layer2.weight"] = W(D, 4 * D)
        for j in range(3):
            sd[f"{pre}.{j}.adaLN_modulation.1.weight"] = W(256, D); sd[f"{pre}.{j}.adaLN_modulation.2.weight"] = W(3 * D, 256)
    sd["final_layer.linear.weight"] = W(64, D)
    sd["final_layer.adaLN_modulation.1.weight"] = W(256, D); sd["final_layer.adaLN_modulation.2.weight"] = W(2 * D, 256)
    sd["affline_norm.weight"] = torch.ones(D)
    x = torch.randn(1, 16, T, Hh, Ww, generator=g)
    pose = torch.randn(1, 64, T, Hh, Ww, generator=g)
    mask = torch.zeros(1, 1, T, Hh, Ww)
    ctx = torch.randn(1, M, 1024, generator=g) * 0.2
    N = T * (Hh // 2) * (Ww // 2)
    flops = dit_forward_flops(N, L=blocks)
    step_flops = 2 * dit_forward_flops(56320)

    def run(sd_, cast):
        t0 = time.perf_counter()
        with torch.no_grad():
            dit_oracle.dit_forward(sd_, cast(x), cast(torch.tensor([0.3])), cast(ctx), cast(mask), cast(pose), cast(torch.zeros(1, 1, 8 * Hh, 8 * Ww)),
                                   torch.tensor([24.0]), num_blocks=blocks, num_heads=H)
        return time.perf_counter() - t0

    dt32 = run(sd, lambda t: t)
    legs = {"fp32": dict(value=flops / dt32 / step_flops, tflops=round(flops / dt32 / 1e12, 3), seconds=round(dt32, 2))}
    # The reference runs this network with bf16 parameters and activations (config/base/model.py:29 `precision="bfloat16"`): the same sample with the
    # oracle's bf16 evaluation (= the reference's rounding points: bf16 Linear outputs, fp32 norm statistics, P rounded to bf16) is the CPU figure that
    # corresponds to what the reference would do on these cores. A host without native bf16 GEMM kernels can be far slower in bf16 than in fp32 - a
    # one-matmul probe decides whether the leg is worth its seconds.
    def probe(dtype):
        a, b = torch.randn(2048, D, generator=g).to(dtype), torch.randn(D, D, generator=g).to(dtype)
        a @ b.t()
        t0 = time.perf_counter()
        a @ b.t()
        return 2 * 2048 * D * D / (time.perf_counter() - t0) / 1e12
    p32, p16 = probe(torch.float32), probe(torch.bfloat16)
    note16 = f"one-GEMM probe on this host: bf16 {p16:.2f} vs fp32 {p32:.2f} TFLOP/s"
    if p16 >= 0.5 * p32:
        sd16 = {k_: (v_ if k_ == "pos_embedder.seq" else v_.to(torch.bfloat16)) for k_, v_ in sd.items()}
        # attention of the bf16 leg through torch's fused CPU kernel (F.scaled_dot_product_attention in bf16) - what the reference's attention falls back
        # to off-GPU and how profiles/r4_cpu_reference.json timed it; the oracle's own attention_sbhd materialises fp32 score matrices (a checker's
        # form, 10x slower than any CPU path the reference would take). Timing leg only: the parity oracle is untouched.
        def sdpa_sbhd(q, k, v):
            qs, ks, vs = (t.permute(1, 2, 0, 3) for t in (q, k, v))  # b h s d
            return torch.nn.functional.scaled_dot_product_attention(qs, ks, vs).permute(2, 0, 1, 3).reshape(q.shape[0], q.shape[1], -1)
        keep = dit_oracle.attention_sbhd
        dit_oracle.attention_sbhd = sdpa_sbhd
        try:
            dt16 = run(sd16, lambda t: t.to(torch.bfloat16))
        finally:
            dit_oracle.attention_sbhd = keep
        legs["bf16"] = dict(value=flops / dt16 / step_flops, tflops=round(flops / dt16 / 1e12, 3), seconds=round(dt16, 2))
    else:
        note16 += " - bf16 leg skipped (no native bf16 GEMM on this host), value = the fp32 leg"
    # the baseline is the best the host does: the faster of the measured legs (a bf16 leg that ran slower than fp32 must not become the denominator)
    prec = max(legs, key=lambda k_: legs[k_]["value"])
    head = legs[prec]
    return dict(value=head["value"], unit="denoise-steps/sec", cores=threads, kind="port", precision=prec,
                seconds=head["seconds"], blocks=blocks, legs=legs,
                sample=f"oracle/dit_oracle.py: {blocks} of 28 blocks (D=4096,H=32) of one DiT forward on the configs[0] latent 16x64x64 = {N} tokens, timed in fp32 "
                       f"({dt32:.1f}s = {legs['fp32']['tflops']} TFLOP/s) and in bf16 = the reference's precision, attention through torch's fused CPU SDPA (" +
                       (f"{legs['bf16']['seconds']}s = {legs['bf16']['tflops']} TFLOP/s" if "bf16" in legs else "skipped") + f"; {note16}); `value` = the faster measured leg ({prec}), "
                       f"extrapolated by FLOPs to 28 blocks x 2 forwards at 56320 tokens (the 4.419 PFLOP step); "
                       f"per-block linearity of the extrapolation checked once: profiles/r3_cpu_baseline_linearity.txt")


def cpu_baseline_render(threads: int):
    """Renderer leg of the CPU baseline: oracle/warp_oracle.py ("port": numpy restatement of the reference's forward_warp, pinned bit-exact to it on
    flow / masks) on the host, ONE reference pair (2 items) of bench.py's 704 x 1280 scene, without and with foreground masking (the mesh
    occlusion through oracle/c/ray_tri.c, OpenMP over the rays). The reference's own forward_warp timed on CPU tensors: profiles/r4_cpu_reference.json."""
    from oracle import warp_oracle
    h, w = 704, 1280
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    depth = 4.0 + 0.0004 * xs + 0.0002 * ys
    for (cy, cx, r, zz) in ((h * 0.4, w * 0.3, h * 0.22, 1.6), (h * 0.65, w * 0.7, h * 0.18, 2.4)):
        depth = np.where((ys - cy) ** 2 + (xs - cx) ** 2 < r * r, zz + 0.0001 * xs, depth)
    depth = depth.astype(np.float32)
    img = np.stack([np.sin(xs * 0.021 + c) * np.cos(ys * 0.017 - c) for c in range(3)], 0).astype(np.float32)
    K = np.array([[1000.0, 0, w / 2], [0, 1000.0, h / 2], [0, 0, 1]], np.float32)
    pts = warp_oracle.unproject_points(depth[None, None], np.eye(4, dtype=np.float32)[None], K[None])
    rel = warp_oracle.reliable_depth_mask(depth[None, None], ratio_thresh=0.05).astype(np.float32)
    bnd = ~warp_oracle.reliable_depth_mask(depth[None, None])
    w2c = np.stack([np.eye(4, dtype=np.float32)] * 2)
    w2c[:, 0, 3] = (0.1, 0.11)
    out = {}
    pairs = 4  # ~1.5 s + ~5 s on the GPU box's host
    for fg in (False, True):
        t0 = time.perf_counter()
        for j in range(pairs):
            w2c[:, 0, 3] = (0.3 * (2 * j) / 31, 0.3 * (2 * j + 1) / 31)
            warp_oracle.forward_warp(np.stack([img] * 2), np.concatenate([rel] * 2), np.concatenate([pts] * 2), w2c, np.stack([K] * 2),
                                     foreground_masking=fg, boundary_mask=np.concatenate([bnd[:, 0]] * 2) if fg else None,
                                     ray_triangle_fn=warp_oracle.ray_triangle_depth_c if fg else None)
        dt = time.perf_counter() - t0
        out["foreground_masking" if fg else "plain"] = dict(value=round(43.2e6 * 2 * pairs / dt / 1e9, 4), unit="GB/s", ms_per_item=round(dt / (2 * pairs) * 1e3, 1),
                                                              seconds=round(dt, 2))
    out.update(cores=threads, kind="port", sample=f"oracle/warp_oracle.py (numpy; ray x triangle in C, OpenMP), {pairs} reference pairs = {2 * pairs} items of 704x1280 of the bench scene "
                                                  f"(camera sliding left), 43.2 MB algorithmic per item")
    return out


def cpu_baseline_tokenizer(threads: int):
    """Tokenizer leg: oracle/tokenizer_oracle.py ("port", fp32 torch on the host) encode + decode of a 9 x 704 x 1280 clip at channels = 128 = 2/16 of
    the benchmark clip's latent volume at its full resolution. The reference's own modules on CPU: profiles/r4_cpu_reference.json."""
    from oracle import tokenizer_oracle as tok
    from gen3c_amd.tokenizer import CausalVideoTokenizerNet
    torch.set_num_threads(threads)
    keys = CausalVideoTokenizerNet(channels=128, device="cpu").expected_keys()
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shape in keys.items():
        sd[k] = (torch.rand(shape, generator=g) + 0.5) if k.endswith("norm.weight") else (torch.randn(shape, generator=g) * (0.05 if k.endswith(".bias") else (1.0 / max(1, int(np.prod(shape[1:])))) ** 0.5))
    T, H, W = 9, 704, 1280  # 2 of the 16 latent frames at the full resolution (the 14 080-pixel spatial attention included): ~7 + ~11 s
    x = torch.rand(1, 3, T, H, W, generator=g) * 2 - 1
    frac = (1 + (T - 1) / 8) * H * W / (16 * 704 * 1280)
    out = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        z = tok.encoder(sd, x)
        te = time.perf_counter() - t0
        t0 = time.perf_counter()
        tok.decoder(sd, z)
        td = time.perf_counter() - t0
    for name, dt, tflop in (("encode", te, 35.7), ("decode", td, 61.3)):
        out[name] = dict(value=round(tflop * frac / dt, 4), unit="TFLOP/s", seconds=round(dt, 2), full_clip_seconds_extrapolated=round(dt / frac, 1))
    out.update(cores=threads, kind="port", sample=f"oracle/tokenizer_oracle.py fp32, channels=128, one {T}x{H}x{W} clip = {frac:.4f} of the 121x704x1280 clip's work")
    return out


def parse_pmc_csv(text: str, kernel_substr: str, counter: str):
    """Mean Counter_Value of `counter` over the dispatches of kernels whose name contains `kernel_substr` in a rocprofv3 counter_collection.csv."""
    import csv
    import io
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(io.StringIO(text)) if r.get("Counter_Name") == counter and kernel_substr in r.get("Kernel_Name", "")]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def sum_pmc_csv(text: str, kernel_substrs, counter: str) -> float:
    """Sum of `counter` over all dispatches of kernels whose name contains any of `kernel_substrs`."""
    import csv
    import io
    return sum(float(r["Counter_Value"]) for r in csv.DictReader(io.StringIO(text))
               if r.get("Counter_Name") == counter and any(k in r.get("Kernel_Name", "") for k in kernel_substrs))


def _pmc_pass(counter: str, probe: str, extra_env: dict, timeout_s: int):
    """One rocprofv3 pass (--kernel-trace --pmc <counter> only) over tools/<probe> in a child process, from /tmp -> text of its counter_collection.csv files."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(exe).exists():
        raise RuntimeError("rocprofv3 not found")
    d = tempfile.mkdtemp(prefix="g3pmc_", dir="/tmp")
    try:
        r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, str(ROOT / "tools" / probe)],
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", **extra_env), capture_output=True, text=True, timeout=timeout_s)
        files = list(Path(d).rglob("*counter_collection.csv"))
        if r.returncode != 0 or not files:
            raise RuntimeError(f"rocprofv3 --pmc {counter} over {probe} failed (rc {r.returncode})")
        return "\n".join(f.read_text() for f in files)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_render_traffic(timeout_s: int = 150):
    """`roofline_render.traffic` measured in this run: FETCH_SIZE and WRITE_SIZE passes over tools/bench_render_single.py restricted to the benchmarked configuration
    (foreground masking; 4 renders x 32 items in the child process), summed over every renderer kernel (warp_* / mesh_*) and divided by the 128 items.
    bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: calibrated in round 6 on this pool against product kernels of known traffic (tools/pmc_calibrate.py,
    profiles/r6_pmc_calibration.txt): the counters are in KiB, WRITE_SIZE is exact, and FETCH_SIZE tallies HALF the bytes not only of 16 B/lane streams (the guide's
    gfx950 note) but also of coalesced 4 B/lane loads - the renderer's two load forms. Rounds 3-5 quoted the raw sum x 1000 (`traffic_raw_r5_convention`)."""
    try:
        parts = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            kb = sum_pmc_csv(_pmc_pass(counter, "bench_render_single.py", {"G3_RENDER_ONLY_FG": "1"}, timeout_s), ("warp_", "mesh_"), counter)
            if kb <= 0:
                return None, f"no renderer dispatch in the {counter} pass"
            parts[counter] = kb
        tot = (2.0 * parts["FETCH_SIZE"] + parts["WRITE_SIZE"]) * 1024.0
        return int(tot / 128), (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/bench_render_single.py (foreground masking, 4 x 32 items "
                                f"in a child process), all warp_* / mesh_* kernels, per item; (2 x FETCH_SIZE {parts['FETCH_SIZE']:.4g} KiB + WRITE_SIZE {parts['WRITE_SIZE']:.4g} KiB) x 1024 / 128 "
                                f"[profiles/r6_pmc_calibration.txt: FETCH_SIZE counts half the bytes of coalesced 4 B and 16 B per-lane loads on gfx950; the raw sum x 1000 of rounds 3-5 would read "
                                f"{int((parts['FETCH_SIZE'] + parts['WRITE_SIZE']) * 1000.0 / 128)}]")
    except Exception as e:  # noqa: BLE001
        return None, repr(e)


def measure_attention_traffic(kernel_substr: str = "flash_attn_fwd_w4b", timeout_s: int = 150):
    """`roofline.traffic` measured IN THIS RUN: two rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE: one counter per
    pass, nothing else, as MI355X_MICROARCH.md's HBM section prescribes) over tools/pmc_probe.py, which issues the benchmark's own self-attention launch
    (S = 56 320, H = 32, B = 2, strided q / k views) twice in a child process on this box, from /tmp. bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the guide's gfx950
    correction: 16 B/lane coalesced reads are tallied at half size; counters in KiB - both confirmed on known byte counts, profiles/r6_pmc_calibration.txt).
    Returns (bytes per launch, description) or (None, why)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(exe).exists():
        return None, "rocprofv3 not found"
    got = {}
    env = dict(os.environ, TMPDIR="/tmp", G3_PMC_ONLY="attn")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="g3pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, str(ROOT / "tools" / "pmc_probe.py")],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = list(Path(d).rglob("*counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            v, n = parse_pmc_csv("\n".join(f.read_text() for f in files), kernel_substr, counter)
            if v is None:
                return None, f"no {kernel_substr} dispatch in the {counter} pass"
            got[counter] = v
        except Exception as e:  # noqa: BLE001 - the traffic entry must never hide the measurement
            return None, f"rocprofv3 --pmc {counter}: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    total = int((got["FETCH_SIZE"] * 2 + got["WRITE_SIZE"]) * 1024)
    return total, (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_probe.py = this launch in a child process; "
                   f"(FETCH_SIZE {got['FETCH_SIZE']:.4g} KiB x 2 (gfx950 correction of the guide) + WRITE_SIZE {got['WRITE_SIZE']:.4g} KiB) x 1024")


TOK_FIXTURE = ROOT / "tests" / "golden" / "tokenizer_fullsize_samples.npz"


def tokenizer_bench_clip(dev, T=121, H=704, W=1280, seed=5):
    """The fixed-seed 121x704x1280 clip the tokenizer entry encodes (smooth content + noise, generated on the device). The same function
    feeds tests/test_fullsize_gpu.py::test_tokenizer_full_clip_vs_fp32_oracle, which holds encode / decode of THIS clip against the fp32
    oracle and wrote the committed samples of the oracle's outputs (TOK_FIXTURE) that `stage_rooflines` compares with."""
    g = torch.Generator(device=dev).manual_seed(seed)
    base = torch.nn.functional.interpolate(torch.rand(1, 3, max(T // 4, 2), H // 16, W // 16, device=dev, generator=g), size=(T, H, W), mode="trilinear")
    x = ((base * 2 - 1) * 0.8 + 0.2 * (torch.rand(1, 3, T, H, W, device=dev, generator=g) * 2 - 1)).clamp(-1, 1)
    return x.to(torch.bfloat16)


def tokenizer_sample_index(numel: int, n: int = 8192, seed: int = 11):
    """Fixed pseudo-random flat positions (host RNG) at which oracle outputs are committed / compared."""
    return torch.from_numpy(np.random.RandomState(seed).randint(0, numel, size=n).astype(np.int64))


def stage_rooflines(dev):
    """The chunk's other two stages at the benchmark size, timed once each with hipEvents (rank 0, N = 1, outside the timed region):
    tokenizer encode / decode of one 121x704x1280 clip (algorithmic work SURVEY.md 8a-a15: 35.7 / 61.3 TFLOP, MFMA-bound) and the cache
    renderer (algorithmic 43.2 MB per 704x1280 item, SURVEY.md 8d, HBM-bound). Random weights / synthetic scene."""
    from gen3c_amd import ops, renderer
    from gen3c_amd.tokenizer import CausalVideoTokenizerNet
    out = {}
    net = CausalVideoTokenizerNet(channels=128, device=dev)
    net.init_random(seed=3)  # the weights of tests/test_fullsize_gpu.py::test_tokenizer_full_clip_vs_fp32_oracle
    x = tokenizer_bench_clip(dev)
    z = None
    tok = {}
    fix = None
    try:
        fix = np.load(TOK_FIXTURE)
    except Exception:
        pass
    for name, fn, tflop in (("encode", net.encoder, 35.7), ("decode", net.decoder, 61.3)):
        arg = x if name == "encode" else z
        res = fn(arg)  # warm-up (also the decode input)
        torch.cuda.synchronize()
        runs = []  # median of three single passes (one pass after one warm-up varied by +-4 % from run to run: allocator state, clocks after the DiT loop)
        for _ in range(3):
            tm = ops.HipTimer()
            tm.start()
            res = fn(arg)
            tm.stop()
            runs.append(tm.elapsed_ms())
        ms = sorted(runs)[1]
        finite = bool(torch.isfinite(res.float()).all())
        tok[name] = dict(ms=round(ms, 2), runs_ms=[round(r, 2) for r in runs], achieved=round(tflop / ms * 1e3, 1), frac=round(tflop / ms * 1e3 / PEAK_BF16_TFLOPS, 4), output_finite=finite)
        if fix is not None:  # the timed result itself against the committed samples of the fp32 oracle's output on this clip / these weights
            idx = tokenizer_sample_index(res.numel()).to(dev)
            got = res.reshape(-1)[idx].float().cpu()
            ref = torch.from_numpy(fix["z_ref" if name == "encode" else "y_ref"])
            tok[name]["rel_l2_vs_oracle_samples"] = round(float((got - ref).norm() / ref.norm()), 5)
        if not finite:
            raise RuntimeError(f"tokenizer {name} of the benchmark clip produced non-finite values")
        if name == "encode":
            z = res
    out["roofline_tokenizer"] = dict(bound="mfma", unit="TFLOP/s", peak=PEAK_BF16_TFLOPS, workload="CV8x8x8 tokenizer, one 121x704x1280 clip, bf16",
                                     parity="rel_l2_vs_oracle_samples: this run's outputs at 8192 fixed positions vs oracle/tokenizer_oracle.py in fp32 on the same clip and weights "
                                            "(tests/golden/tokenizer_fullsize_samples.npz, written by tests/test_fullsize_gpu.py::test_tokenizer_full_clip_vs_fp32_oracle; "
                                            "decode here takes this run's own latent, the test the oracle's)" if fix is not None else None, **tok)
    del net, x, z, res
    h, w, F = 704, 1280, 32
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=dev), torch.arange(w, dtype=torch.float32, device=dev), indexing="ij")
    depth = 4.0 + 0.0004 * xs + 0.0002 * ys
    for (cy, cx, r, zz) in ((h * 0.4, w * 0.3, h * 0.22, 1.6), (h * 0.65, w * 0.7, h * 0.18, 2.4)):
        depth = torch.where((ys - cy) ** 2 + (xs - cx) ** 2 < r * r, zz + 0.0001 * xs, depth)
    img = torch.stack([torch.sin(xs * 0.021 + c) * torch.cos(ys * 0.017 - c) for c in range(3)], 0)
    K = torch.tensor([[1000.0, 0, w / 2], [0, 1000.0, h / 2], [0, 0, 1]], device=dev)
    w2cs = torch.eye(4, device=dev).repeat(1, F, 1, 1)
    w2cs[0, :, 0, 3] = torch.linspace(0, 0.3, F, device=dev)
    Ks = K[None, None].expand(1, F, 3, 3).contiguous()
    cache = renderer.Cache3D_Buffer(frame_buffer_max=2, input_image=img[None], input_depth=depth[None, None], input_w2c=torch.eye(4, device=dev)[None],
                                    input_intrinsics=K[None], filter_points_threshold=0.05, foreground_masking=True, input_format=["B", "C", "H", "W"])
    for _ in range(2):  # workspace / host memo set up, clocks back up after the host-side cache construction
        cache.render_cache(w2cs, Ks)
    torch.cuda.synchronize()
    tm = ops.HipTimer()
    reps = 10  # ~13 ms of GPU work (3 repetitions = 4 ms sat inside the clock ramp after an idle gap)
    tm.start()
    for _ in range(reps):
        cache.render_cache(w2cs, Ks)
    tm.stop()
    per_item = tm.elapsed_ms() / reps / F
    gbs = 43.2e6 / (per_item * 1e-3) / 1e9
    traffic = traffic_source = None  # memory-side bytes per item: QUOTED from the committed rocprofv3 PMC passes of this configuration
    try:
        tf = next(f for f in ("r6_render_traffic.json", "r5_render_traffic.json", "r4_render_traffic.json", "r3_render_traffic.json") if (ROOT / "profiles" / f).exists())  # the latest committed PMC passes
        tj = json.loads((ROOT / "profiles" / tf).read_text())
        traffic, traffic_source = tj["traffic_bytes_per_item"], f"profiles/{tf} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration, per item); not re-measured in this run"
    except Exception:
        pass
    traffic_quoted = traffic
    del cache
    torch.cuda.empty_cache()
    measured, how = measure_render_traffic()
    if measured is not None:
        traffic, traffic_source = measured, how
    elif traffic_source:
        traffic_source += f" [in-run measurement unavailable: {how}]"
    out["roofline_render"] = dict(bound="hbm", unit="GB/s", peak=8000.0, achieved=round(gbs, 1), frac=round(gbs / 8000.0, 4), ms_per_item=round(per_item, 4),
                                  traffic=traffic, traffic_source=traffic_source, traffic_quoted=traffic_quoted,
                                  workload="cache render (z pre-pass + projecting splat + mesh occlusion + resolve), 704x1280 items, foreground masking, 43.2 MB algorithmic per item")
    return out


def video_wallclock(dev, net, ms_per_step: float, steps: int = 35):
    """The second half of BASELINE.json's metric - wall-clock per video - for configs[1] (one 121 x 704 x 1280 chunk, 35 steps, guidance 1, one cache
    buffer, foreground masking), rank 0, N = 1, after the timed region (gen3c_single_image.py:366-419, gen3c_pipeline.py:108-184). Everything except
    the denoise loop is MEASURED here end to end through the product's own entry points, each once and cold as one video pays it: cache build,
    render_cache of the 121 items, then ONE real Gen3cPipeline.generate_from_embeddings at num_steps = 1 on the bench's own 28-block network
    (the 1 + 2 N tokenizer encodes, condition assembly, noise, the scheduler, decode, uint8 conversion and the D2H copy of the video). The denoise
    loop is `steps` x the ms_per_step this run measured (the one step inside the 1-step generate is subtracted as measured there, so the
    first-call allocation of that step is not charged 35 times). Model construction / checkpoint load is not part of it (the reference loads
    once per process)."""
    from gen3c_amd import renderer
    from gen3c_amd.camera_utils import generate_camera_trajectory
    from gen3c_amd.pipeline import DiffusionGen3CModel, Gen3cPipeline
    from gen3c_amd.tokenizer import VideoTokenizer
    H, W, T = 704, 1280, 121
    sec = {}

    def timed(name, fn_, *a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn_(*a, **k)
        torch.cuda.synchronize()
        sec[name] = sec.get(name, 0.0) + time.perf_counter() - t0
        return r

    tk = VideoTokenizer(pixel_chunk_duration=T, device=dev)
    tk.net.init_random(seed=1)
    tk.register_mean_std(torch.zeros(16, 32), torch.ones(16, 32))
    model = DiffusionGen3CModel(net, tk, latent_shape=(16, tk.get_latent_num_frames(T), H // 8, W // 8))
    pipe = Gen3cPipeline(model, guidance=1.0, num_steps=1, height=H, width=W, fps=24, num_video_frames=T, seed=1)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev), indexing="ij")
    depth = 3.0 + 0.001 * xs + 0.0005 * ys
    depth[((xs - 400) ** 2 + (ys - 300) ** 2) < 120 ** 2] = 1.5
    depth[((xs - 900) ** 2 + (ys - 420) ** 2) < 90 ** 2] = 2.2
    img = torch.stack([torch.sin(xs / 37.0), torch.cos(ys / 23.0), torch.sin((xs + ys) / 51.0)])[None]
    K = torch.tensor([[1000.0, 0, 640], [0, 1000.0, 352], [0, 0, 1]], device=dev)
    w2c0 = torch.eye(4, device=dev)
    cache = timed("cache_build", renderer.Cache3D_Buffer, frame_buffer_max=2, noise_aug_strength=0.0, input_image=img, input_depth=depth[None, None],
                  input_w2c=w2c0[None], input_intrinsics=K[None], filter_points_threshold=0.05, foreground_masking=True, input_format=["B", "C", "H", "W"])
    w2cs, Ks = generate_camera_trajectory("left", w2c0, K, T, 0.3, "center_facing", center_depth=3.0, device=dev)
    renders, masks = timed("render_121_items", cache.render_cache, w2cs, Ks)
    for name in ("encode", "decode"):
        setattr(model, name, (lambda o, n: (lambda *a, **k: timed("tokenizer_" + n, o, *a, **k)))(getattr(model, name), name))
    den_step = model.denoiser.denoise_step
    model.denoiser.denoise_step = lambda *a, **k: timed("one_denoise_step_in_generate", den_step, *a, **k)
    emb = torch.zeros(1, 512, net.crossattn_emb_channels, dtype=torch.bfloat16)
    video = timed("generate_1_step_total", pipe.generate_from_embeddings, emb, (img[:, :, None] * 0.99).to(torch.bfloat16), renders, masks)
    assert video.shape == (T, H, W, 3)
    glue = sec["generate_1_step_total"] - sec.get("tokenizer_encode", 0.0) - sec.get("tokenizer_decode", 0.0) - sec.get("one_denoise_step_in_generate", 0.0)
    non_dit = sec["cache_build"] + sec["render_121_items"] + sec.get("tokenizer_encode", 0.0) + sec.get("tokenizer_decode", 0.0) + glue
    dit = steps * ms_per_step * 1e-3
    total = non_dit + dit
    sec["host_glue_in_generate"] = glue
    del cache, renders, masks, model, pipe, tk
    torch.cuda.empty_cache()
    return dict(value=round(total, 2), unit="s/video", higher_is_better=False,
                workload=f"configs[1]: one 121x704x1280 chunk, {steps} steps, guidance 1, 1 cache buffer, foreground masking, random-init weights, synthetic image + depth",
                denoise_loop_s=round(dit, 2), non_dit_s=round(non_dit, 3), non_dit_share=round(non_dit / total, 4),
                measured_seconds={k: round(v, 3) for k, v in sec.items()},
                method=f"measured: cache build, render of 121 items, one Gen3cPipeline.generate_from_embeddings at num_steps=1 (tokenizer encodes + decode + host glue; its "
                       f"own denoise step subtracted); denoise loop = {steps} x this run's ms_per_step; model construction excluded",
                video_finite=bool(np.isfinite(video.astype(np.float32)).all()))


def self_launch_argv(n_gpus: int, argv: list, port: int | None = None) -> list:
    """Command line that runs this script as `n_gpus` ranks of one node (used when bench.py is started bare with --gpus N>1)."""
    if port is None:
        import socket
        with socket.socket() as s:  # a free rendezvous port (hard-coded ports collide on shared boxes)
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), str(Path(__file__).resolve()), *argv]


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=28, help="(debug only) number of DiT blocks; anything but 28 is not the benchmark")
    ap.add_argument("--latent", type=str, default="16,88,160", help="(debug only) latent T,H,W")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true",
                    help="A/B: no hipEvent pairs around the launches of the timed region (the roofline / roofline_gemm entries are then null)")
    ap.add_argument("--no-extras", action="store_true", help="skip the tokenizer / renderer roofline entries (a few seconds after the timed region)")
    ap.add_argument("--cp-config", type=str, default="auto",
                    help="N > 1: 'auto' = time the context-parallel configurations (head groups x attention kernel x collective schedule) on a few blocks "
                         "before the warm-up and run on the fastest; or 'G,kernel,schedule' e.g. '4,auto,gather_first' to pin one")
    return ap


def cp_report(cpa, self_attn, gemms, steps: int, rccl_ranks: int) -> dict:
    """Diagnosis of the one multi-GPU run the driver makes: where a step's time goes on THIS rank (rank 0), per step. Inputs: the
    ContextParallelAttention object (its `stats` = (kind, head group, HipTimer) per Work.wait(), `bytes_gathered`), the (meta, ms) lists of the
    self-attention and GEMM launches of the timed region."""
    waits = [tm.elapsed_ms() for (_k, _g, tm) in (cpa.stats or [])]
    steps = max(1, steps)
    return dict(
        rccl_ranks=rccl_ranks,
        attention_ms_per_step=round(sum(ms for _, ms in self_attn) / steps, 2),
        attention_launches_per_step=len(self_attn) // steps,
        gemm_ms_per_step=round(sum(ms for _, ms in gemms) / steps, 2),
        exposed_collective_wait_ms_per_step=round(sum(waits) / steps, 3),
        collective_waits_per_step=len(waits) // steps,
        worst_single_wait_ms=round(max(waits), 3) if waits else 0.0,
        gathered_bytes_per_step=int(cpa.bytes_gathered // steps),
        note="hipEvent pairs on the launch streams: attention / gemm = sum over launches (head groups alternate between two streams, so "
             "attention sums can exceed wall time); exposed wait = time a launch stream sat idle in Work.wait() for an all-gather",
    )


CP_TUNE_BLOCKS = 4  # DiT blocks per autotune forward (every block has the same shapes and collectives)
CP_FALLBACK = (4, "auto", "local_first")  # the DiT's own default (dit.py: enable_context_parallel): what runs when the autotune has nothing to offer
DIST_TIMEOUT_S = int(os.environ.get("G3_BENCH_DIST_TIMEOUT_S", "420"))  # process-group timeout of the N > 1 run (parallel.init_distributed)
PHASE_DEADLINE_S = {"init": 400, "autotune": 240, "timed": 360}  # watchdog: a phase that overruns ends the run WITH a JSON line


def _injected(phase: str, rank: int, cand=None) -> bool:
    """Test hook (tests/test_cp_gpu.py): G3_BENCH_INJECT="autotune:4,w4b,local_first:1" raises inside that candidate on rank 1,
    "timed:1" raises in the timed region on rank 1. Never set by the driver."""
    spec = os.environ.get("G3_BENCH_INJECT", "")
    if not spec:
        return False
    parts = spec.split(":")
    if parts[0] != phase or int(parts[-1]) != rank:
        return False
    return phase != "autotune" or ",".join(str(c) for c in cand) == parts[1]


def autotune_cp(net, den, xt, cond, uncond, dev, dist, rank: int = 0, progress: dict | None = None):
    """Pick (head_groups, attention kernel, collective schedule) of ContextParallelAttention by measurement: the driver's multi-GPU run is the
    only one this code ever gets on real xGMI links, so it tunes itself. Every candidate runs the same denoise step on the first CP_TUNE_BLOCKS
    blocks (1 untimed + 1 timed, barrier + synchronize on both sides); ranks agree on each time via all_reduce(MAX), so every rank picks the
    same winner. Untimed by the benchmark (before the warm-up steps); the state `xt` is not advanced.

    A candidate that RAISES on any rank is dropped on every rank (the same all_reduce carries a failure flag) and listed in the returned
    `failed`; if nothing survives, CP_FALLBACK is configured. Returns (best or None, table, failed)."""
    cands = [(G, kern, sched) for sched in ("gather_first", "local_first") for kern in ("w4b", "wave8") for G in (1, 2, 4, 8)]
    cpa = net._cp_attn
    net._tune_blocks = CP_TUNE_BLOCKS
    table, failed = [], []
    try:
        for cand in cands:
            (G, kern, sched) = cand
            if progress is not None:
                progress.update(candidate=cand, table=table, failed=failed)
            ms, err = [0.0, 0.0], None

            def all_ok() -> bool:
                """Barrier + agreement in one collective: every rank reports whether it can still run this candidate. Keeps the collective
                sequence identical on all ranks when ONE of them raised outside a step (a rank that skipped a barrier the others enter would
                hang them); a raise in the middle of a step's exchanges cannot be repaired - RunGuard ends the run with a line then."""
                f = torch.tensor([0.0 if err is None else 1.0], device=dev, dtype=torch.float64)
                dist.all_reduce(f, op=dist.ReduceOp.MAX)
                return float(f.item()) == 0.0

            try:
                if _injected("autotune", rank, cand):
                    raise RuntimeError(f"injected failure in candidate {cand} on rank {rank}")
                cpa.configure(head_groups=G, kernel=kern, schedule=sched)
            except Exception as e:  # this rank cannot even set the candidate up
                err = repr(e)
            times = []
            for rep in range(2):
                if not all_ok():
                    err = err or "failed on another rank"
                    break
                try:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    den.denoise_step(xt, 0, cond, uncond, 1.0, 0.001, 1)
                    torch.cuda.synchronize()
                    times.append((time.perf_counter() - t0) * 1e3)
                except Exception as e:  # (if every rank raises at the same point the sequence stays aligned; otherwise see all_ok)
                    err = repr(e)
            if err is None:
                ms = times
            t = torch.tensor([ms[1] if err is None else 0.0, 0.0 if err is None else 1.0], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # [slowest rank's time, 1 if any rank failed]
            eff = getattr(cpa, "effective", None) or {}
            if float(t[1].item()) > 0:
                failed.append(dict(head_groups=G, kernel=kern, schedule=sched, error=err or "failed on another rank"))
                continue
            table.append(dict(head_groups=G, kernel=kern, schedule=sched, ms=round(float(t[0].item()), 3),
                              ran=dict(kernel=eff.get("kernel", kern), schedule=eff.get("schedule", sched), head_groups=eff.get("head_groups", G))))
    finally:
        net._tune_blocks = None
    if not table:
        cpa.configure(head_groups=CP_FALLBACK[0], kernel=CP_FALLBACK[1], schedule=CP_FALLBACK[2])
        return None, table, failed
    best = min(table, key=lambda r: r["ms"])
    cpa.configure(head_groups=best["head_groups"], kernel=best["kernel"], schedule=best["schedule"])
    return best, table, failed


class _FileErrorStore:
    """set / check / get of keys through files in the node's temp directory (RunGuard's fallback side channel). One file per key, created 0600.
    A leftover of an earlier run (same MASTER_PORT / run id / recycled parent pid) must not fail this one: rank 0 unlinks the files when it
    constructs the store and at exit, and every rank ignores files older than its own start (the per-run nonce every rank has without a
    launcher's help: no rank of THIS run can have written before the youngest rank's interpreter started, minus a clock-granularity margin)."""

    def __init__(self, tag: str, rank: int = 0, keys=("g3_bench_error",)):
        import atexit
        import tempfile
        self.base = os.path.join(tempfile.gettempdir(), f"g3_bench_error_{tag}_{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}_{os.getppid()}")
        self.path = self._file(keys[0])
        self.t0 = time.time() - 2.0
        if rank == 0:
            for k in keys:
                self._unlink(self._file(k))
            atexit.register(lambda: [self._unlink(self._file(k)) for k in keys])

    def _file(self, key) -> str:
        return f"{self.base}_{key}"

    @staticmethod
    def _unlink(path):
        try:
            os.unlink(path)
        except OSError:
            pass

    def set(self, key, msg):
        tmp = f"{self._file(key)}.{os.getpid()}"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, "w") as f:
            f.write(msg)
        os.replace(tmp, self._file(key))

    def _fresh(self, key) -> bool:
        try:
            return os.stat(self._file(key)).st_mtime >= self.t0
        except OSError:
            return False

    def check(self, keys):
        return all(self._fresh(k) for k in keys)

    def get(self, key):
        with open(self._file(key), "rb") as f:
            return f.read()


class RunGuard:
    """The one multi-GPU run the driver makes must end with a JSON line whatever happens. One daemon thread per rank (N > 1 only):
      * a rank whose phase raised publishes the exception in the process group's key-value store (`fail`); every rank's thread polls the
        store, so rank 0 learns of a failure on rank 5 without a collective;
      * a phase that overruns its deadline (a hung collective: no Python exception ever surfaces) trips the same path;
    then rank 0 prints a line with `value: null`, the phase, what was known (`progress`: autotune candidate / table so far) and the error,
    and every rank leaves with os._exit(1) - before the process-group timeout (DIST_TIMEOUT_S) lets the NCCL watchdog abort silently."""
    KEY = "g3_bench_error"

    def __init__(self, rank: int, world: int, base_line: dict):
        import threading
        self.rank, self.world, self.base = rank, world, dict(base_line)
        self.phase, self.deadline, self.progress, self.done = "init", time.monotonic() + PHASE_DEADLINE_S["init"], {}, False
        self.store, self._lock, self._emitted = None, threading.Lock(), False
        try:
            from torch.distributed.distributed_c10d import _get_default_store
            self.store = _get_default_store()
        except Exception:
            pass
        if self.store is None and world > 1:
            # (private torch API gone / no default group): without a side channel a failure on rank != 0 would never reach rank 0 and the job
            # would end without a JSON line. One node, one /tmp: a file keyed by the rendezvous port does the same job.
            self.store = _FileErrorStore(os.environ.get("MASTER_PORT", "0"), rank=rank, keys=(self.KEY,))
            print(f"bench.py: rank {rank}: no process-group store, RunGuard falls back to {self.store.path}", file=sys.stderr, flush=True)
        self._t = threading.Thread(target=self._watch, daemon=True)
        self._t.start()

    def enter(self, phase: str, seconds: float | None = None):
        self.phase, self.deadline = phase, time.monotonic() + (seconds if seconds is not None else PHASE_DEADLINE_S.get(phase, 300))

    def finish(self):
        self.done = True

    def fail(self, where: str, exc: BaseException):
        """Called by the rank whose phase raised: publish, then let the watcher threads end the run (rank 0 prints)."""
        msg = f"rank {self.rank} in {where}: {exc!r}"
        print(f"bench.py: {msg}", file=sys.stderr, flush=True)
        try:
            if self.store is not None:
                self.store.set(self.KEY, msg)
        except Exception:
            pass
        if self.rank == 0:
            self._emit(msg)
        time.sleep(15)  # rank 0's watcher prints within its poll interval; then leave (torchrun ends the other ranks)
        os._exit(1)

    def _emit(self, error: str):
        with self._lock:
            first, self._emitted = not self._emitted, True
        if not first:
            time.sleep(30)  # another thread of this rank is printing the line and leaving
            return
        line = dict(self.base)
        prog = {k: v for k, v in self.progress.items()}
        line.update(value=None, ms_per_step=None, error=error, failed_phase=self.phase, progress=prog)
        print(json.dumps(line, default=str), flush=True)
        os._exit(1)

    def _watch(self):
        while not self.done:
            time.sleep(1.0)
            err = None
            try:
                if self.store is not None and self.store.check([self.KEY]):
                    err = self.store.get(self.KEY).decode()
            except Exception:
                pass
            if err is None and time.monotonic() > self.deadline:
                err = f"rank {self.rank}: phase '{self.phase}' exceeded its {PHASE_DEADLINE_S.get(self.phase, 300)} s deadline (hung collective?)"
            if err is not None and not self.done:
                if self.rank == 0:
                    self._emit(err)
                time.sleep(5)
                os._exit(1)


def main():
    args = build_parser().parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher of N ranks (one per GPU, RCCL) and relay rank 0's JSON line
        share = os.environ.get("G3_BENCH_SHARE_GPU") == "1"  # plumbing runs on a 1-GPU box (all ranks on cuda:0, gloo)
        if not share and torch.cuda.device_count() < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible on this node")
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required by RCCL on this driver
        env.setdefault("OMP_NUM_THREADS", "8")
        sys.exit(subprocess.call(self_launch_argv(args.gpus, sys.argv[1:]), env=env))

    from gen3c_amd import _lib, ops
    from gen3c_amd.dit import VideoExtendGeneralDIT
    from gen3c_amd.parallel import init_distributed, parallel_state
    from gen3c_amd.sampler import Gen3CDenoiser, VideoExtendCondition, add_condition_video_indicator_and_video_input_mask
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        # G3_BENCH_BACKEND=gloo + G3_BENCH_SHARE_GPU=1: plumbing check of this script's N>1 path on a 1-GPU box (all ranks on
        # cuda:0, collectives over gloo); not a measurement. The driver's runs use the defaults: RCCL, one GPU per rank.
        local = init_distributed(os.environ.get("G3_BENCH_BACKEND", "nccl"), timeout_s=DIST_TIMEOUT_S)
        if os.environ.get("G3_BENCH_SHARE_GPU") == "1":
            local = 0
        elif torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
            sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible (one GPU per rank is required)")
        torch.cuda.set_device(local)
        parallel_state.initialize_model_parallel(context_parallel_size=world)
    else:
        local = 0
        torch.cuda.set_device(0)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device(f"cuda:{local}")

    T, Hl, Wl = (int(v) for v in args.latent.split(","))
    N_tok = T * (Hl // 2) * (Wl // 2)
    step_flops = 2 * dit_forward_flops(N_tok, L=args.blocks)
    base_line = {
        "metric": "denoise-steps/sec (121x1280x704 latent, Cosmos-7B)", "value": None, "unit": "denoise-steps/sec",
        "n_gpus": world, "rccl_ranks": (dist.get_world_size() if world > 1 and dist.get_backend() == "nccl" else (1 if world == 1 else 0)),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"GEN3C-Cosmos-7B denoise step: 2 DiT forwards (cond+uncond) + CFG + latent replace + Euler on latent "
                               f"[1,16,{T},{Hl},{Wl}] = {N_tok} tokens, {args.blocks} blocks x 4096, 32 heads, guidance=1, 35-step Karras schedule, "
                               f"random-init weights", "parallelism": f"cp{world}" if world > 1 else "single-gpu",
                   "step_pflop": round(step_flops / 1e15, 4)},
    }
    # N > 1: from here on the run ends with a JSON line whatever happens (exception on any rank, hung collective) - see RunGuard
    guard = RunGuard(rank, world, base_line) if world > 1 else None

    def guarded(phase, fn):
        if guard is not None:
            guard.enter(phase)
        try:
            return fn()
        except Exception as e:  # noqa: BLE001 - whatever it is, the line must still be printed
            if guard is not None:
                guard.fail(phase, e)
            line = dict(base_line)
            line.update(error=repr(e), failed_phase=phase)
            print(json.dumps(line), flush=True)
            sys.exit(1)

    def setup():
        net = VideoExtendGeneralDIT(in_channels=16 + 16 * 4 + 1, rope_t_extrapolation_ratio=2.0, num_blocks=args.blocks,
                                    device=dev, init_weights=False)
        net.initialize_weights(randomize_adaln=True, seed=1234)  # same weights on every rank
        # the timed region is the DENSE workload: every one of the 512 context tokens goes through the cross-attention's tile loop, although the synthetic
        # context (like a real T5 embedding) is zero-padded beyond its first 64 tokens and the product would by default take that tail in closed form
        # (dit.py: cross_attention_skip_zero_context); the product default is timed beside it, outside the timed region (`cross_attention_zero_tail`)
        net.cross_attention_skip_zero_context = False
        if world > 1:
            net.enable_context_parallel(parallel_state.get_context_parallel_group())
        return net

    net = guarded("init", setup)

    # ---- synthetic inputs, identical on every rank (host RNG), resident in HBM before timing
    rs = np.random.RandomState(1)
    def normal(shape, std):
        return torch.from_numpy((rs.standard_normal(shape) * std).astype(np.float32)).to(torch.bfloat16).to(dev)
    B = 1
    den = Gen3CDenoiser(net, state_shape=(16, T, Hl, Wl))
    den.scheduler.set_timesteps(35)
    xt_full = normal((B, 16, T, Hl, Wl), den.scheduler.init_noise_sigma)
    gt = normal((B, 16, T, Hl, Wl), 0.5)
    pose = normal((B, 64, T, Hl, Wl), 0.5)
    ctx = normal((B, 512, 1024), 0.2)
    ctx[:, 64:] = 0
    pad = torch.zeros(B, 1, 8 * Hl, 8 * Wl, device=dev, dtype=torch.bfloat16)
    fps = torch.tensor([24.0], device=dev)

    def make_cond(p):
        c = VideoExtendCondition(crossattn_emb=ctx, crossattn_mask=None, padding_mask=pad, fps=fps, video_cond_bool=True,
                                 condition_video_pose=p)
        return add_condition_video_indicator_and_video_input_mask(gt, c, 1)

    cond, uncond = make_cond(pose), make_cond(torch.zeros_like(pose))
    from gen3c_amd.parallel import split_inputs_cp
    xt = split_inputs_cp(xt_full, 2, net.cp_group) if world > 1 else xt_full

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cp_info = None
    if world > 1:
        def tune():
            if args.cp_config == "auto":
                best, table, failed = autotune_cp(net, den, xt, cond, uncond, dev, dist, rank=rank, progress=guard.progress)
                info = dict(chosen=best if best is not None else dict(head_groups=CP_FALLBACK[0], kernel=CP_FALLBACK[1], schedule=CP_FALLBACK[2], ms=None),
                            autotune_blocks=CP_TUNE_BLOCKS, autotune_ms=table)
                if failed:
                    info["autotune_failed"] = failed
                if best is None:
                    info["autotune_error"] = "no candidate survived on every rank: running the fixed fallback configuration"
                return info
            G, kern, sched = args.cp_config.split(",")
            net._cp_attn.configure(head_groups=int(G), kernel=kern, schedule=sched)
            return dict(chosen=dict(head_groups=int(G), kernel=kern, schedule=sched), autotune_ms=None)
        cp_info = guarded("autotune", tune)
        guard.progress.clear()
        guard.progress.update(cp=cp_info)

    state = dict(xt=xt, step_id=0)

    def timed_region():
        xt_, step_id = state["xt"], 0
        for _ in range(args.warmup):
            xt_ = den.denoise_step(xt_, step_id, cond, uncond, 1.0, 0.001, 1)
            step_id += 1
        barrier()
        ops.enable_kernel_timers(not args.no_kernel_timers)
        if world > 1:
            net._cp_attn.stats = []
            net._cp_attn.bytes_gathered = 0
        if _injected("timed", rank):
            raise RuntimeError(f"injected failure in the timed region on rank {rank}")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            xt_ = den.denoise_step(xt_, step_id, cond, uncond, 1.0, 0.001, 1)
            step_id += 1
        torch.cuda.synchronize()
        own = time.perf_counter() - t0  # this rank's own time to finish its K steps (before the closing barrier): load imbalance shows here
        barrier()
        elapsed_ = time.perf_counter() - t0
        state["xt"] = xt_
        return elapsed_, own

    elapsed, own_elapsed = guarded("timed", timed_region)
    xt = state["xt"]
    timers = ops.collected_kernel_timers()
    ops.enable_kernel_timers(False)
    finite = bool(torch.isfinite(xt.float()).all())

    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    rank_ms = None
    if world > 1:
        def reduce_times():
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            mine = torch.tensor([own_elapsed / args.steps * 1e3], device=dev, dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            return [round(float(t.item()), 2) for t in allr]
        rank_ms = guarded("timed", reduce_times)
    elapsed = float(tmax.item())

    # ---- dominant kernel: self-attention flash-attention forward, timed live with hipEvents on its launch stream
    if guard is not None:
        guard.enter("report", 120)
    self_attn = [(meta, tm.elapsed_ms()) for (name, meta, tm) in timers if name == "flash_attn_fwd" and meta["Skv"] > 2048]
    roof = None
    if self_attn:
        avg_ms = sum(ms for _, ms in self_attn) / len(self_attn)
        flops_launch = sum(4.0 * m["Sq"] * m["Skv"] * m["H"] * 128 * m["B"] for m, _ in self_attn) / len(self_attn)
        ach = flops_launch / (avg_ms * 1e-3) / 1e12
        m0 = self_attn[0][0]
        kname = m0.get("kernel") or _lib.load().g3_flash_attn_kernel_name(m0["Sq"], m0["Skv"], m0["B"], m0["H"]).decode()
        traffic = traffic_source = None  # HBM bytes per launch: QUOTED from the committed rocprofv3 PMC passes of this kernel at this shape
        for tf in ("r4_attn_traffic.json", "r3_attn_traffic.json", "r2_attn_traffic.json", "r1_attn_traffic.json"):
            try:
                tj = json.loads((ROOT / "profiles" / tf).read_text())
                same_kernel = tj["kernel"].replace(" ", "").split("<")[0] == kname.replace(" ", "").split("<")[0]
                js = tj["shape"]
                if world == 1 and same_kernel and (js["Sq"], js["Skv"], js["H"]) == (m0["Sq"], m0["Skv"], m0["H"]):
                    # measured at batch js["B"]; a launch over B samples is B independent problems in one grid (per-sample K / V^T panels)
                    traffic = tj["traffic_bytes_per_launch"] * m0["B"] // js["B"]
                    traffic_source = f"profiles/{tf} (rocprofv3 --pmc passes of this kernel, measured at B={js['B']}" + \
                                     (")" if js["B"] == m0["B"] else f", x{m0['B'] // js['B']} for this launch's batch)") + "; not re-measured in this run"
                    break
            except Exception:
                continue
        roof = dict(bound="mfma", kernel=kname, achieved=round(ach, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / PEAK_BF16_TFLOPS, 4), traffic=traffic, traffic_source=traffic_source, launches=len(self_attn), avg_launch_ms=round(avg_ms, 3),
                    flops_per_launch=flops_launch)

    # second MFMA kernel: the block GEMMs (ping-pong kernel), same live hipEvent timing; reported next to `roofline`
    gemms = [(meta, tm.elapsed_ms()) for (name, meta, tm) in timers if name == "gemm_nt"]
    roof_gemm = None
    if gemms:
        g_ms = sum(ms for _, ms in gemms)
        g_fl = sum(2.0 * m["M"] * m["N"] * m["K"] for m, _ in gemms)
        gm = max(gemms, key=lambda r: r[0]["M"] * r[0]["N"] * r[0]["K"])[0]
        roof_gemm = dict(bound="mfma", kernel=_lib.load().g3_gemm_kernel_name(gm["M"], gm["N"], gm["K"], gm["epilogue"]).decode(), achieved=round(g_fl / (g_ms * 1e-3) / 1e12, 1), peak=PEAK_BF16_TFLOPS,
                         unit="TFLOP/s", frac=round(g_fl / (g_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), launches=len(gemms),
                         total_ms_per_step=round(g_ms / args.steps, 2))
        # per launch class (epilogue + shape), so that the driver's run shows which class loses: the block's projections / MLP halves
        lib = _lib.load()
        epi_name = {0: "none", 1: "gelu", 2: "gated_residual", 3: "bias", 4: "bias_residual"}
        classes = {}
        for m, ms in gemms:
            c = classes.setdefault((m["epilogue"], m["M"], m["N"], m["K"]), [0, 0.0])
            c[0] += 1
            c[1] += ms
        roof_gemm["classes"] = [dict(epilogue=epi_name.get(e, str(e)), M=M, N=N, K=K, launches=n, avg_ms=round(ms / n, 3),
                                     achieved=round(2.0 * M * N * K * n / (ms * 1e-3) / 1e12, 1), frac=round(2.0 * M * N * K * n / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                     kernel=lib.g3_gemm_kernel_name(M, N, K, e).decode())
                                for (e, M, N, K), (n, ms) in sorted(classes.items(), key=lambda kv: -kv[1][1]) if ms > 0]

    if world > 1:
        cp_info.update(cp_report(net._cp_attn, self_attn, gemms, args.steps, dist.get_world_size() if dist.get_backend() == "nccl" else 0))
        net._cp_attn.stats = None

    if world > 1:
        cp_info["rank_own_ms_per_step"] = dict(min=min(rank_ms), max=max(rank_ms), per_rank=rank_ms,
                                               note="each rank's own wall time per step up to ITS synchronize, before the closing barrier")
        guard.finish()
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = args.steps / elapsed
        out = dict(base_line)
        out.update({
            "value": round(value, 5), "ms_per_step": round(ms_per_step, 2),
            "step_tflops_per_gpu": round(step_flops / (elapsed / args.steps) / 1e12 / world, 1),
            "step_mfma_frac": round(step_flops / (elapsed / args.steps) / 1e12 / world / PEAK_BF16_TFLOPS, 4),
            "output_finite": finite,
            "kernel_timers": not args.no_kernel_timers,
            "roofline": roof,
            "roofline_gemm": roof_gemm,
        })
        if cp_info is not None:
            out["cp"] = cp_info
        out["outside_timed_region"] = ("in-run rocprofv3 PMC passes, tokenizer / renderer entries, the wall-clock-per-video entry and the CPU baseline legs: rank 0, after the timed loop, "
                                       + ("run in this invocation (n_gpus = 1)" if world == 1 else "SKIPPED in this invocation (they run at n_gpus = 1 only)"))
        if not args.no_extras and world == 1 and roof is not None and "w4b" in roof["kernel"] and (N_tok, args.blocks) == (56320, 28):
            # the dominant kernel's memory-side traffic, measured now on this box instead of quoted from a committed file (the quoted figure stays as fallback)
            measured, how = measure_attention_traffic()
            if measured is not None:
                out["roofline"]["traffic_quoted"] = out["roofline"]["traffic"]
                out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured, how
            else:
                out["roofline"]["traffic_source"] = (out["roofline"].get("traffic_source") or "") + f" [in-run measurement unavailable: {how}]"
        if not args.no_extras and world == 1:
            try:
                out.update(stage_rooflines(dev))
            except Exception as e:  # the extras must never hide the measurement
                out["roofline_extras_error"] = repr(e)
        if not args.no_extras and world == 1:
            try:  # the product's default cross-attention (zero-padded context tail in closed form), two steps, beside the dense timed region
                net.cross_attention_skip_zero_context = True
                xz = den.denoise_step(xt, 0, cond, uncond, 1.0, 0.001, 1)  # (rebuilds nothing: the K / V cache entry carries the live-key count)
                torch.cuda.synchronize()
                ops.enable_kernel_timers(True)
                tz = time.perf_counter()
                for i_ in range(2):
                    xz = den.denoise_step(xz, 1 + i_, cond, uncond, 1.0, 0.001, 1)
                torch.cuda.synchronize()
                dz = (time.perf_counter() - tz) / 2 * 1e3
                tmz = [(m_, t_.elapsed_ms()) for (n_, m_, t_) in ops.collected_kernel_timers() if n_ == "flash_attn_fwd" and m_["Skv"] <= 2048]
                ops.enable_kernel_timers(False)
                dense_ca = [t__.elapsed_ms() for (n__, m__, t__) in timers if n__ == "flash_attn_fwd" and m__["Skv"] <= 2048]
                out["cross_attention_zero_tail"] = dict(
                    context_tokens=int(ctx.shape[1]), live_tokens=int((ctx != 0).any(-1).any(0).sum()), keys_through_the_loop=tmz[0][0].get("kv_dense") if tmz else None,
                    ms_per_step=round(dz, 2), steps_per_sec=round(1e3 / dz, 5), dense_ms_per_step=round(ms_per_step, 2),
                    cross_attention_ms_per_step=round(sum(ms for _, ms in tmz) / 2, 2) if tmz else None,
                    dense_cross_attention_ms_per_step=round(sum(dense_ca) / args.steps, 2) if dense_ca else None, output_finite=bool(torch.isfinite(xz.float()).all()),
                    note="NOT the headline: `value` times the dense workload (all 512 context tokens through the attention loop); this is the product default on the same "
                         "zero-padded context (g3_cross_attn_fwd_bf16: same softmax, the all-zero K / V tail in closed form)")
            except Exception as e:
                out["cross_attention_zero_tail"] = {"error": repr(e)}
            finally:
                net.cross_attention_skip_zero_context = False
        if not args.no_extras and world == 1 and (N_tok, args.blocks) == (56320, 28):
            try:
                out["video_wallclock"] = video_wallclock(dev, net, ms_per_step)
            except Exception as e:  # the extras must never hide the measurement
                out["video_wallclock"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                nthr = min(32, os.cpu_count() or 1)  # 32 threads measured fastest on the 2x64-core EPYC host
                out["cpu_baseline"] = cpu_baseline(threads=nthr)
                for leg, fn in (("render", cpu_baseline_render), ("tokenizer", cpu_baseline_tokenizer)):  # the chunk's other two stages
                    try:
                        out["cpu_baseline"][leg] = fn(nthr)
                    except Exception as e:
                        out["cpu_baseline"][leg] = {"error": repr(e)}
                out["cpu_baseline"]["reference_on_cpu"] = "the reference's own DiT / forward_warp / tokenizer timed on CPU in the build container: profiles/r4_cpu_reference.json"
            except Exception as e:  # the baseline must never hide the measurement
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
