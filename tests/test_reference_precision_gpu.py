"""GPU: where the HIP path sits relative to the REFERENCE'S OWN precision.

The reference runs its network with bf16 parameters and activations (`precision="bfloat16"`, config/base/model.py:29); the parity oracle is an fp32
evaluation. "rel-L2 5e-3 vs fp32" means little without the distance the reference's own arithmetic has to fp32. Three comparisons, every number printed
(copied to profiles/r4_parity_measured.txt):
  1. golden cases: HIP vs fp32 next to the reference's own class run in bf16 vs fp32 (tests/golden/dit_bf16_ref.npz, tools/gen_golden_bf16.py);
  2. attention alone: fp32 softmax vs {torch bf16 SDPA, the oracle's TransformerEngine restatement in bf16, the one-wave kernel w4b, the 8-wave kernel};
  3. the full-size block (56 320 tokens): oracle/dit_oracle.py in bf16 on the device (= the reference's rounding points: bf16 Linear outputs, fp32 RMSNorm
     statistics, P rounded to bf16 before P.V) vs the same in fp32, next to HIP vs fp32.
The assertion in each: the HIP path is no further from fp32 than 1.25 x the reference-precision evaluation is."""
import math

import numpy as np
import pytest
import torch

from tests.golden_io import GOLD, load_dit_case
from tests.test_dit_gpu import build_net

pytestmark = pytest.mark.gpu
SLACK = 1.25


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


@pytest.mark.parametrize("name", ["dit_tiny", "dit_small"])
def test_hip_is_as_close_to_fp32_as_the_reference_class_in_bf16(name):
    dev = torch.device("cuda:0")
    cfg, sd, inp, y_ref = load_dit_case(name)
    y_ref_bf16 = torch.from_numpy(np.load(GOLD / "dit_bf16_ref.npz")[f"{name}_y_bf16"])
    net = build_net(cfg, sd, dev)
    bf = lambda t: t.to(dev).to(torch.bfloat16)
    y = net(x=bf(inp["x"]), timesteps=bf(inp["timesteps"]), crossattn_emb=bf(inp["ctx"]), crossattn_mask=None, fps=inp["fps"].to(dev),
            padding_mask=bf(inp["padding_mask"]), condition_video_indicator=bf(inp["mask"][:, :, :, :1, :1]), condition_video_input_mask=bf(inp["mask"]),
            condition_video_pose=bf(inp["pose"])).float().cpu()
    r_hip, r_ref = _rel(y, y_ref), _rel(y_ref_bf16, y_ref)
    print(f"[{name}] vs the reference class in fp32: HIP {r_hip:.3e} | the reference class in bf16 (CPU) {r_ref:.3e} | HIP vs reference-bf16 {_rel(y, y_ref_bf16):.3e}")
    assert r_hip <= SLACK * r_ref


def test_attention_kernels_next_to_bf16_sdpa():
    from gen3c_amd import ops
    from oracle import dit_oracle
    dev = torch.device("cuda:0")
    S, H, HD = 8192, 4, 128
    g = torch.Generator(device=dev).manual_seed(31)
    q, k, v = (torch.randn(S, H * HD, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
    q4, k4, v4 = (t.view(S, 1, H, HD) for t in (q, k, v))  # sbhd
    ref = dit_oracle.attention_sbhd(q4.float(), k4.float(), v4.float()).view(S, H * HD)  # fp32 softmax(QK^T/sqrt(d))V
    rows = {}
    rows["oracle TE restatement in bf16 (P -> bf16, fp32 accumulate)"] = _rel(dit_oracle.attention_sbhd(q4, k4, v4).view(S, H * HD), ref)
    try:
        qs, ks, vs = (t.view(S, H, HD).permute(1, 0, 2)[None] for t in (q, k, v))
        sd = torch.nn.functional.scaled_dot_product_attention(qs, ks, vs)[0].permute(1, 0, 2).reshape(S, H * HD)
        rows["torch.nn.functional.scaled_dot_product_attention in bf16 (this device)"] = _rel(sd, ref)
    except Exception as e:  # not every build ships a bf16 SDPA kernel for this device
        print("  (torch bf16 SDPA unavailable:", repr(e)[:120], ")")
    vt = ops.transpose_v(v, S, 1, H)
    for label, variant in (("HIP one-wave kernel (w4b, default for long self-attention)", 11), ("HIP 8-wave kernel (folded softmax arithmetic)", 4), ("HIP 8-wave kernel, unfolded (v3)", 3)):
        rows[label] = _rel(ops.flash_attn(q, k, vt, S, S, 1, H, variant=variant), ref)
    print(f"[attention S={S} H={H} d=128, N(0,1) operands] rel-L2 vs fp32 softmax:")
    for kname, val in rows.items():
        print(f"    {val:.3e}  {kname}")
    ref_prec = rows["oracle TE restatement in bf16 (P -> bf16, fp32 accumulate)"]
    assert rows["HIP one-wave kernel (w4b, default for long self-attention)"] <= SLACK * ref_prec
    assert rows["HIP 8-wave kernel (folded softmax arithmetic)"] <= SLACK * ref_prec
    # w4b (the default, and the least accurate bf16 variant here: its folded softmax arithmetic trades ~25 % more error for speed) is
    # pinned next to the vendor-class arithmetic - a later speed tweak cannot silently widen the gap. Today 2.88e-3 vs 2.30e-3 = 1.25x.
    sdpa_key = "torch.nn.functional.scaled_dot_product_attention in bf16 (this device)"
    if sdpa_key in rows:
        ratio = rows["HIP one-wave kernel (w4b, default for long self-attention)"] / rows[sdpa_key]
        print(f"    w4b / torch bf16 SDPA error ratio = {ratio:.3f} (bound 1.3)")
        assert ratio <= 1.3


def test_full_size_block_hip_vs_reference_precision_oracle():
    """The configuration of test_fullsize_gpu.py::test_dit_full_size_single_block_vs_fp32_oracle with a third evaluation: the oracle in bf16."""
    from gen3c_amd.dit import VideoExtendGeneralDIT
    from oracle import dit_oracle
    dev = torch.device("cuda:0")
    net = VideoExtendGeneralDIT(in_channels=81, rope_t_extrapolation_ratio=2.0, num_blocks=1, device=dev, init_weights=False)
    net.initialize_weights(randomize_adaln=True, seed=17)
    B, T, H, W, M = 1, 16, 88, 160, 512
    g = torch.Generator(device=dev).manual_seed(23)
    rnd = lambda *s_: torch.randn(*s_, device=dev, generator=g)
    x = rnd(B, 16, T, H, W).to(torch.bfloat16)
    mask = torch.zeros(B, 1, T, H, W, dtype=torch.bfloat16, device=dev)
    mask[:, :, :1] = 1
    pose = (0.5 * rnd(B, 64, T, H, W)).to(torch.bfloat16)
    ctx = (0.2 * rnd(B, M, 1024)).to(torch.bfloat16)
    ctx[:, 64:] = 0
    ts = torch.tensor([0.3], dtype=torch.bfloat16, device=dev)
    pad = torch.zeros(B, 1, 8 * H, 8 * W, dtype=torch.bfloat16, device=dev)
    fps = torch.tensor([24.0], device=dev)
    y = net(x=x, timesteps=ts, crossattn_emb=ctx, crossattn_mask=None, fps=fps, padding_mask=pad, condition_video_indicator=mask[:, :, :, :1, :1],
            condition_video_input_mask=mask, condition_video_pose=pose)
    torch.cuda.synchronize()
    with torch.no_grad():
        sd32 = {k_: v_.detach().float() for k_, v_ in net.state_dict().items()}
        y32 = dit_oracle.dit_forward(sd32, x.float(), ts.float(), ctx.float(), mask.float(), pose.float(), pad.float(), fps, num_blocks=1, num_heads=32)
        del sd32
        sd16 = {k_: (v_.detach() if k_ == "pos_embedder.seq" else v_.detach().to(torch.bfloat16)) for k_, v_ in net.state_dict().items()}
        y16 = dit_oracle.dit_forward(sd16, x, ts, ctx, mask, pose, pad, fps, num_blocks=1, num_heads=32)
    r_hip, r_ref = _rel(y, y32), _rel(y16, y32)
    print(f"[dit D=4096 H=32, 1 block, 56320 tokens] vs the fp32 oracle: HIP {r_hip:.3e} | the oracle in bf16 (reference rounding points, torch kernels of this device) {r_ref:.3e}"
          f" | HIP vs bf16 oracle {_rel(y, y16):.3e}")
    assert r_hip <= SLACK * r_ref


def test_fp8_qk_emulation_study():
    """VERDICT r3 #10 (north_star names "MFMA bf16/fp8 QK^T"): can an e4m3 QK^T hold the attention tolerance (rel-L2 <= 1e-2 vs fp32)? Decided BEFORE writing a
    kernel, by emulating exactly what v_mfma_f32_32x32x64_f8f6f4 would compute: Q and K quantised to OCP e4m3 with one scale per (head, tensor) (amax -> 448),
    products exact, fp32 accumulation; softmax in fp32, P and V in bf16 as today. Expectation from the format: 3 mantissa bits = 2^-4 relative rounding per
    element -> logit noise ~0.044 -> P perturbed by ~4 % per key, which zero-mean V does not average away. Measured and recorded (profiles/r4_parity_measured.txt);
    the assertion documents the outcome: fp8 QK^T is >= 3x outside the tolerance the bf16 kernels meet with margin, so no fp8 kernel is shipped."""
    from oracle import dit_oracle
    dev = torch.device("cuda:0")
    HD = 128
    if not hasattr(torch, "float8_e4m3fn"):
        pytest.skip("torch build without float8_e4m3fn")
    out = {}
    for S, H in ((8192, 4), (56320, 1)):
        g = torch.Generator(device=dev).manual_seed(41 + S)
        q, k, v = (torch.randn(S, 1, H, HD, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
        rows = torch.randint(0, S, (512,), device=dev, generator=g)

        def quant(t):  # per (head, tensor) scale, as a kernel would apply on the way into LDS
            amax = t.float().abs().amax(dim=(0, 1, 3), keepdim=True)
            sc = 448.0 / amax
            return (t.float() * sc).to(torch.float8_e4m3fn).float() / sc

        def attn(qq, kk):
            sc = torch.einsum("sbhd,tbhd->bhst", qq[rows].float(), kk.float()) / math.sqrt(HD)
            p = torch.softmax(sc, dim=-1)
            return torch.einsum("bhst,tbhd->sbhd", p.to(torch.bfloat16).float(), v.float())

        ref = attn(q, k)
        out[S] = (_rel(attn(quant(q), quant(k)), ref), _rel(attn(quant(q), k), ref))
        print(f"[fp8 QK^T emulation, S={S}, N(0,1) operands, {rows.numel()} sampled rows] rel-L2 vs fp32 attention: e4m3 Q and K {out[S][0]:.3e} | e4m3 Q only {out[S][1]:.3e} "
              f"(bf16 kernels: ~3e-3; tolerance 1e-2)")
    assert out[56320][0] > 1e-2, "e4m3 QK^T met the attention tolerance on this data - revisit the decision not to build it"


def test_fp8_pv_emulation_study():
    """VERDICT r4 #7 / north_star "MFMA bf16/fp8 QK^T.V": the P.V half. Emulates what v_mfma_scale_f32_32x32x64_f8f6f4 would compute with QK^T and the softmax
    as today (bf16 operands, fp32 scores) and P and V quantised to OCP e4m3: V with one scale per (head) (amax -> 448), P = exp(s - m) in (0, 1] scaled by 256
    (a power of two: exact, keeps exp(-17) above the e4m3 subnormal floor); products exact, fp32 accumulation; the row sum taken over the QUANTISED P (what
    the kernel's MFMA would see). Two operand sets per size: N(0,1) (score std 1, near-uniform attention over all keys) and a peaked one (q scaled x4: score
    std 4, a handful of keys carry the mass). e4m3 = 3 mantissa bits = 2^-4 relative rounding per element (rms 3.6 %); zero-mean V does not average the
    P errors away, and V's own rounding is another 3.6 % of each term. The assertion documents the outcome (numbers: profiles/r5_parity_measured.txt)."""
    dev = torch.device("cuda:0")
    HD = 128
    if not hasattr(torch, "float8_e4m3fn"):
        pytest.skip("torch build without float8_e4m3fn")
    e4 = lambda t: t.to(torch.float8_e4m3fn).float()
    out = {}
    for S, H in ((8192, 4), (56320, 1)):
        for label, qscale in (("N(0,1)", 1.0), ("peaked", 4.0)):
            g = torch.Generator(device=dev).manual_seed(43 + S)
            q, k, v = (torch.randn(S, 1, H, HD, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
            q = (q.float() * qscale).to(torch.bfloat16)
            rows = torch.randint(0, S, (512,), device=dev, generator=g)
            sc = torch.einsum("sbhd,tbhd->bhst", q[rows].float(), k.float()) / math.sqrt(HD)
            e = torch.exp(sc - sc.amax(dim=-1, keepdim=True))  # (0, 1]: what the kernel holds before the deferred normalisation
            vf = v.float()
            vamax = vf.abs().amax(dim=(0, 1, 3), keepdim=True)
            v8 = e4(vf * (448.0 / vamax)) * (vamax / 448.0)
            e8 = e4(e * 256.0) / 256.0
            e16 = e.to(torch.bfloat16).float()

            def pv(pp, vv):  # sum over keys / row sum of the same (rounded) weights
                return torch.einsum("bhst,tbhd->sbhd", pp, vv) / pp.sum(dim=-1).permute(2, 0, 1)[..., None]

            ref = pv(e, vf)
            r = dict(bf16=_rel(pv(e16, vf), ref), p8=_rel(pv(e8, vf), ref), v8=_rel(pv(e16, v8), ref), p8v8=_rel(pv(e8, v8), ref))
            out[(S, label)] = r
            print(f"[fp8 P.V emulation, S={S}, {label} (score std {float(sc.std()):.2f}), 512 sampled rows] rel-L2 vs fp32 P.V: bf16 P (today) {r['bf16']:.3e} | "
                  f"e4m3 P only {r['p8']:.3e} | e4m3 V only {r['v8']:.3e} | e4m3 P and V {r['p8v8']:.3e}   (tolerance 1e-2)")
    worst = max(r["p8v8"] for r in out.values())
    best = min(r["p8v8"] for r in out.values())
    print(f"[fp8 P.V emulation] e4m3 P and V: best case {best:.3e}, worst case {worst:.3e}")
    # north_star's fp8 clause is closed on these numbers: an opt-in kernel would need every case <= 1e-2
    assert worst > 1e-2, "e4m3 P.V met the attention tolerance on every operand set - build it behind G3_ATTN_FP8_PV (VERDICT r4 #7)"
