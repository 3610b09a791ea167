"""One GPU plays ONE RANK of a context-parallel run of the full-size denoise step: the PRODUCT code path - Gen3CDenoiser.denoise_step ->
VideoExtendGeneralDIT.forward with context parallelism enabled -> parallel.ContextParallelAttention, every kernel at the per-rank shapes of cp = 1 / 2 / 4 / 8
(M = 2 x S_local rows, attention Sq = S_local against all 56 320 keys in head groups on two streams) - with torch.distributed's three calls the path makes
(get_world_size / get_rank / all_gather_into_tensor) replaced by an in-process stand-in that plays rank `--rank` of `cp` ranks: the "exchange" is a
device copy of this rank's shard into every slot of the gathered buffer on a separate stream (the bytes a real exchange would deliver into HBM; the values of the
other ranks' keys are this rank's own - irrelevant for timing). What this measures: the COMPUTE side of a rank's step with today's kernels, launch gaps included.
What it cannot: the xGMI exchange itself - the model printed at the end adds it under stated assumptions.

  python tools/cp_rank_emulate.py [--cps 1,2,4,8] [--steps 3] [--blocks 28] [--configs 4,auto,gather_first;4,w4b,gather_first;...] [--out gpurun_out/r6_cp_rank_shapes.json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import _lib, ops  # noqa: E402
from gen3c_amd.dit import VideoExtendGeneralDIT  # noqa: E402
from gen3c_amd.sampler import Gen3CDenoiser, VideoExtendCondition, add_condition_video_indicator_and_video_input_mask  # noqa: E402

PEAK = 2500.0


class _Work:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)
        return True


class FakeRank:
    """Stand-in for the three torch.distributed calls on the CP path: plays rank `rank` of `world`."""

    def __init__(self, world, rank, dev):
        self.world, self.rank = world, rank
        self.comm = torch.cuda.Stream(device=dev)
        self.bytes = 0
        self._orig = {}

    def __enter__(self):
        for name in ("get_world_size", "get_rank", "all_gather_into_tensor"):
            self._orig[name] = getattr(dist, name)
        dist.get_world_size = lambda group=None: self.world
        dist.get_rank = lambda group=None: self.rank
        dist.all_gather_into_tensor = self._all_gather
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(dist, name, fn)
        return False

    def _all_gather(self, out, inp, group=None, async_op=False):
        cur = torch.cuda.current_stream()
        self.comm.wait_stream(cur)  # the send buffer is produced on the launch stream
        with torch.cuda.stream(self.comm):
            out.view(self.world, -1).copy_(inp.reshape(1, -1).expand(self.world, -1))
            ev = self.comm.record_event()
        self.bytes += (self.world - 1) * inp.numel() * inp.element_size()
        w = _Work(ev)
        if not async_op:
            w.wait()
        return w


def run_cp(cp, rank, configs, steps, blocks, dev, net):
    T, Hl, Wl = 16, 88, 160
    rs = np.random.RandomState(1)
    normal = lambda shape, std: torch.from_numpy((rs.standard_normal(shape) * std).astype(np.float32)).to(torch.bfloat16).to(dev)
    den = Gen3CDenoiser(net, state_shape=(16, T, Hl, Wl))
    den.scheduler.set_timesteps(35)
    xt_full = normal((1, 16, T, Hl, Wl), den.scheduler.init_noise_sigma)
    gt, pose, ctx = normal((1, 16, T, Hl, Wl), 0.5), normal((1, 64, T, Hl, Wl), 0.5), normal((1, 512, 1024), 0.2)
    ctx[:, 64:] = 0
    pad = torch.zeros(1, 1, 8 * Hl, 8 * Wl, device=dev, dtype=torch.bfloat16)
    fps = torch.tensor([24.0], device=dev)

    def make_cond(p):
        c = VideoExtendCondition(crossattn_emb=ctx, crossattn_mask=None, padding_mask=pad, fps=fps, video_cond_bool=True, condition_video_pose=p)
        return add_condition_video_indicator_and_video_input_mask(gt, c, 1)

    cond, uncond = make_cond(pose), make_cond(torch.zeros_like(pose))
    results = []
    lib = _lib.load()
    with FakeRank(cp, rank, dev) as fr:
        if cp > 1:
            net.enable_context_parallel("fake-cp-group")
            from gen3c_amd.parallel import split_inputs_cp
            xt0 = split_inputs_cp(xt_full, 2, net.cp_group)
        else:
            net.disable_context_parallel()
            xt0 = xt_full
        for cfg in (configs if cp > 1 else [None]):
            if cfg is not None:
                G, kern, sched = cfg
                net._cp_attn.configure(head_groups=G, kernel=kern, schedule=sched)
                net._cp_attn.stats = None
            xt = xt0
            xt = den.denoise_step(xt, 0, cond, uncond, 1.0, 0.001, 1)  # warm-up (tables, workspaces, cross-attention K / V)
            torch.cuda.synchronize()
            ops.enable_kernel_timers(True)
            if cp > 1:
                net._cp_attn.stats = []
            fr.bytes = 0
            t0 = time.perf_counter()
            for i in range(steps):
                xt = den.denoise_step(xt, 1 + i, cond, uncond, 1.0, 0.001, 1)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            timers = ops.collected_kernel_timers()
            ops.enable_kernel_timers(False)
            assert bool(torch.isfinite(xt.float()).all())
            cls = {}
            for name, meta, tm in timers:
                if name == "gemm_nt":
                    key = ("gemm", meta["epilogue"], meta["M"], meta["N"], meta["K"])
                    fl = 2.0 * meta["M"] * meta["N"] * meta["K"]
                elif name == "flash_attn_fwd":
                    key = ("attn", meta["Sq"], meta["Skv"], meta["B"], meta["H"], meta.get("kernel", "?"))
                    fl = 4.0 * meta["Sq"] * meta["Skv"] * meta["H"] * 128 * meta["B"]
                else:
                    continue
                c = cls.setdefault(key, [0, 0.0, 0.0])
                c[0] += 1
                c[1] += tm.elapsed_ms()
                c[2] += fl
            table = []
            for key, (n, tms, fl) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
                if tms <= 0:
                    continue
                d = dict(kind=key[0], launches_per_step=n / steps, ms_per_step=round(tms / steps, 3), tflops=round(fl / tms / 1e9, 1))
                if key[0] == "gemm":
                    d.update(epilogue=key[1], M=key[2], N=key[3], K=key[4], kernel=lib.g3_gemm_kernel_name(key[2], key[3], key[4], key[1]).decode(),
                             tile_rounds=round(((key[2] + 255) // 256) * ((key[3] + 255) // 256) / 256.0, 3))
                else:
                    d.update(Sq=key[1], Skv=key[2], B=key[3], H=key[4], kernel=key[5])
                table.append(d)
            wait_ms = None
            if cp > 1 and net._cp_attn.stats:
                wait_ms = round(sum(tm.elapsed_ms() for (_k, _g, tm) in net._cp_attn.stats) / steps, 3)
                net._cp_attn.stats = None
            eff = dict(net._cp_attn.effective) if cp > 1 and net._cp_attn.effective else None
            res = dict(cp=cp, rank=rank, config=None if cfg is None else dict(head_groups=cfg[0], kernel=cfg[1], schedule=cfg[2]), effective=eff,
                       ms_per_step=round(ms, 2), gathered_bytes_per_step=fr.bytes // steps, emulated_exchange_wait_ms_per_step=wait_ms, classes=table)
            results.append(res)
            a = sum(c["ms_per_step"] for c in table if c["kind"] == "attn" and c["Skv"] > 2048)
            g = sum(c["ms_per_step"] for c in table if c["kind"] == "gemm")
            print(f"cp={cp} rank={rank} cfg={cfg} eff={eff}: {ms:.1f} ms/step  (self-attn launches {a:.1f} ms [overlapping streams are summed], block GEMMs {g:.1f} ms, "
                  f"emulated exchange wait {wait_ms})", flush=True)
    if cp > 1:
        net.disable_context_parallel()
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cps", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=28)
    ap.add_argument("--rank", type=int, default=-1, help="-1: a middle rank (cp // 2)")
    ap.add_argument("--configs", default="4,auto,gather_first;4,w4b,gather_first;2,auto,gather_first;8,auto,gather_first;4,auto,local_first;1,auto,gather_first")
    ap.add_argument("--out", default="gpurun_out/r6_cp_rank_shapes.json")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    configs = [(int(c.split(",")[0]), c.split(",")[1], c.split(",")[2]) for c in args.configs.split(";")]
    net = VideoExtendGeneralDIT(in_channels=16 + 16 * 4 + 1, rope_t_extrapolation_ratio=2.0, num_blocks=args.blocks, device=dev, init_weights=False)
    net.initialize_weights(randomize_adaln=True, seed=1234)
    net.cross_attention_skip_zero_context = False  # the dense workload bench.py times (the record profiles/r6_cp_rank_shapes.* was taken with the product default, ~17 / cp ms less per step)
    allres = []
    for cp in [int(c) for c in args.cps.split(",")]:
        rank = (cp // 2 if args.rank < 0 else args.rank) if cp > 1 else 0
        allres += run_cp(cp, rank, configs, args.steps, args.blocks, dev, net)
    # ---- summary: best configuration per cp, per-class ratios to cp = 1, and the speed-up model
    base = next(r for r in allres if r["cp"] == 1) if any(r["cp"] == 1 for r in allres) else None
    summary = {}
    for cp in sorted({r["cp"] for r in allres}):
        # per-class rates are read from the run with ONE head group where there is one: with G > 1 the groups' launches overlap on two streams and their
        # summed durations would understate the attention rate
        runs = [r for r in allres if r["cp"] == cp]
        one = [r for r in runs if (r["effective"] or {}).get("head_groups") == 1]
        summary[cp] = one[0] if one else min(runs, key=lambda r: r["ms_per_step"])
    lines = []
    if base is not None:
        def rate(r, pred):
            sel = [c for c in r["classes"] if pred(c)]
            fl = sum(c["tflops"] * c["ms_per_step"] for c in sel)
            ms = sum(c["ms_per_step"] for c in sel)
            return fl / ms if ms else None
        names = [("self-attention", lambda c: c["kind"] == "attn" and c["Skv"] > 2048), ("cross-attention", lambda c: c["kind"] == "attn" and c["Skv"] <= 2048),
                 ("GEMM N>=8192 K=4096 (QKV / KV, MLP-up)", lambda c: c["kind"] == "gemm" and c["N"] >= 8192 and c["K"] == 4096 and c["M"] > 4096),
                 ("GEMM N=4096 K=4096 (Q, out-projections)", lambda c: c["kind"] == "gemm" and c["N"] == 4096 and c["K"] == 4096 and c["M"] > 4096),
                 ("GEMM N=4096 K=16384 (MLP-down)", lambda c: c["kind"] == "gemm" and c["N"] == 4096 and c["K"] == 16384)]
        for nm, pred in names:
            b = rate(base, pred)
            row = [f"{nm}: cp=1 {b:.0f} TF/s" if b else f"{nm}: -"]
            for cp, r in summary.items():
                if cp == 1:
                    continue
                v = rate(r, pred)
                if v and b:
                    row.append(f"cp={cp} {v:.0f} ({v / b * 100:.0f} %)")
            lines.append("  ".join(row))
        lines.append("")
        # ---- exchange model. A rank RECEIVES gathered_bytes_per_step per step; assumed sustained all-gather rate INTO one GPU over its 7 xGMI links: 250 / 375 / 500 GB/s
        # (the guide: ~153 GB/s per link peak; 375 = 35 % of 7 links' bidirectional peak - unmeasured here, hence three values). Per layer the schedule of parallel.py is
        # replayed: the G head groups' collectives go out back to back on RCCL's stream when K | V exist; gather_first: Q projection + its norm pass (q_ms), then group g's
        # attention as soon as group g's exchange has landed; local_first: every group's partial over this rank's own shard first (1 / cp of the attention, no wait), then
        # group g's remote part once its exchange has landed. Attention wall time per layer = this run's step time minus its GEMM time minus the HBM-bound rest (the
        # cp = 1 rest scaled by 1 / cp) - launches on two streams overlap, so their summed durations overstate it. RCCL's own CU use while a collective is in flight is
        # NOT modelled (no second GPU here); the driver's --gpus 8 run measures all of it (bench.py `cp` object).
        base_attn = sum(c["ms_per_step"] for c in base["classes"] if c["kind"] == "attn")
        base_gemm = sum(c["ms_per_step"] for c in base["classes"] if c["kind"] == "gemm")
        base_rest = base["ms_per_step"] - base_attn - base_gemm
        layers = args.blocks
        best_pred = {}
        for r in allres:
            cp = r["cp"]
            if cp == 1:
                continue
            eff = r["effective"] or {}
            G, sched = eff.get("head_groups", 4), eff.get("schedule", "gather_first")
            gemm = sum(c["ms_per_step"] for c in r["classes"] if c["kind"] == "gemm")
            cross = sum(c["ms_per_step"] for c in r["classes"] if c["kind"] == "attn" and c["Skv"] <= 2048)
            attn_layer = max(0.0, r["ms_per_step"] - gemm - cross - base_rest / cp) / layers
            q_ms = 0.0
            if sched == "gather_first":  # the self-attention Q projection: one of the two plain N = 4096 launches per block (the other is the cross-attention Q)
                q_ms = sum(c["ms_per_step"] for c in r["classes"] if c["kind"] == "gemm" and c["N"] == 4096 and c["K"] == 4096 and c["epilogue"] == 0) / 2 / layers
            for bw in (250.0, 375.0, 500.0):
                ex_g = r["gathered_bytes_per_step"] / layers / G / (bw * 1e9) * 1e3
                if sched == "gather_first":
                    t = q_ms
                    for g_ in range(G):
                        t = max(t, (g_ + 1) * ex_g) + attn_layer / G
                    exposed = t - (q_ms + attn_layer)
                else:
                    t = attn_layer / cp
                    for g_ in range(G):
                        t = max(t, (g_ + 1) * ex_g) + attn_layer * (cp - 1) / cp / G
                    exposed = t - attn_layer
                pred = r["ms_per_step"] + layers * exposed
                r.setdefault("predicted", {})[str(int(bw))] = dict(exchange_ms_per_layer=round(ex_g * G, 3), exposed_ms_per_layer=round(exposed, 3), ms_per_step=round(pred, 1),
                                                                   speedup=round(base["ms_per_step"] / pred, 3))
                key = (cp, bw)
                if key not in best_pred or pred < best_pred[key][0]:
                    best_pred[key] = (pred, r, exposed, ex_g * G, attn_layer, q_ms)
        for (cp, bw), (pred, r, exposed, ex_layer, attn_layer, q_ms) in sorted(best_pred.items()):
            lines.append(f"cp={cp} @ {bw:.0f} GB/s: best {r['effective']}: compute {r['ms_per_step']:.1f} ms/step (measured, one GPU as rank {r['rank']}); exchange "
                         f"{r['gathered_bytes_per_step'] / 1e9:.1f} GB/step = {ex_layer:.2f} ms/layer vs attention {attn_layer:.2f} ms/layer (+ Q {q_ms:.2f}) -> exposed {exposed:.2f} ms/layer; "
                         f"predicted {pred:.1f} ms/step = {base['ms_per_step'] / pred:.2f} x (compute only: {base['ms_per_step'] / r['ms_per_step']:.2f} x)")
    text = "\n".join(lines)
    print(text)
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps(dict(runs=allres, summary_text=text), indent=1))
    out.with_suffix(".txt").write_text(text + "\n")


if __name__ == "__main__":
    main()
