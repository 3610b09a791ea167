"""Per-operation breakdown of one full-size tokenizer encode + decode (121x704x1280, product configuration): every primitive call of
CausalVideoTokenizerNet is bracketed with a hipEvent pair and the calls are grouped by (op, geometry, shape) - ms, TFLOP/s, streamed GB/s.
Shows which convolution classes (long K / short K) and which HBM-bound passes the time goes to.

  python tools/tokenizer_breakdown.py [T H W]   ->  stdout table (gpurun_out/tokenizer_breakdown.txt when run by tools/gpu_r3.sh)
"""
import collections
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402
from gen3c_amd.tokenizer import _GEOM, CausalVideoTokenizerNet  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    T, H, W = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (121, 704, 1280)))
    net = CausalVideoTokenizerNet(channels=128, device=dev)
    net.init_random(seed=0)
    x = (torch.rand(1, 3, T, H, W, device=dev) * 2 - 1).to(torch.bfloat16)
    records = []
    phase = ["warm"]

    def timed(label_fn, fn):
        def wrapper(*a, **k):
            tm = ops.HipTimer()
            tm.start()
            out = fn(*a, **k)
            tm.stop()
            records.append((phase[0], label_fn(out, *a, **k), tm))
            return out
        return wrapper

    def conv_label(out, xin, name, kind, residual=None, stats=False):
        taps = _GEOM[kind][0] * _GEOM[kind][1] * _GEOM[kind][2]
        M, N, K = out.numel() // out.shape[-1], out.shape[-1], xin.shape[-1]
        return ("conv", kind + ("+res" if residual is not None else ""), M, N, K * taps, 2.0 * M * N * K * taps,
                2.0 * (xin.numel() + out.numel() * (2 if residual is not None else 1)))

    def gn_label(out, xin, name, swish):
        return ("groupnorm", "swish" if swish else "plain", xin.numel() // xin.shape[-1], xin.shape[-1], 0, 0.0, 2.0 * 3 * xin.numel())

    def rs_label(out, xin, mode):
        return ("resample", f"mode{mode}", out.numel() // out.shape[-1], out.shape[-1], 0, 0.0, 2.0 * (xin.numel() + out.numel()))

    inner_conv = net._conv
    net._conv = timed(conv_label, net._conv)
    net._gn = timed(gn_label, net._gn)
    net._resample = timed(rs_label, net._resample)
    # the attention blocks call _conv / _gn (timed above); their own kernels are timed as the remainder of the block
    for nm in ("_spatial_attn", "_temporal_attn"):
        net.__dict__[nm] = timed(lambda out, xin, name, nm=nm: (nm[1:], "block", xin.numel() // xin.shape[-1], xin.shape[-1], 0,
                                                                4.0 * xin.shape[0] * (xin.shape[1] * xin.shape[2]) ** 2 * xin.shape[3] if nm == "_spatial_attn" else 0.0,
                                                                0.0), getattr(net, nm))
    totals = {}
    z = None
    for rep in ("warm", "timed"):
        phase[0] = rep
        for name, fn in (("encode", net.encoder), ("decode", net.decoder)):
            arg = x if name == "encode" else z
            torch.cuda.synchronize()
            tm = ops.HipTimer()
            tm.start()
            marker = len(records)
            out = fn(arg)
            tm.stop()
            totals[(rep, name)] = (tm, marker, len(records))
            if name == "encode":
                z = out
    torch.cuda.synchronize()
    for name in ("encode", "decode"):
        tm, lo, hi = totals[("timed", name)]
        total = tm.elapsed_ms()
        print(f"\n=== {name} {T}x{H}x{W}: {total:.2f} ms")
        groups = collections.OrderedDict()
        blocks = 0.0
        for (_ph, lab, t) in records[lo:hi]:
            key = lab[:5]
            g = groups.setdefault(key, [0, 0.0, 0.0, 0.0])
            g[0] += 1; g[1] += t.elapsed_ms(); g[2] += lab[5]; g[3] += lab[6]
        inner = sum(g[1] for k, g in groups.items() if k[0] not in ("spatial_attn", "temporal_attn"))
        print(f"{'op':14s} {'geom':10s} {'M':>9s} {'N':>5s} {'K':>6s} {'calls':>5s} {'ms':>8s} {'%':>6s} {'TFLOP/s':>8s} {'GB/s':>7s}")
        for key, g in sorted(groups.items(), key=lambda kv: -kv[1][1]):
            tf = g[2] / (g[1] * 1e-3) / 1e12 if g[2] else 0.0
            gb = g[3] / (g[1] * 1e-3) / 1e9 if g[3] else 0.0
            print(f"{key[0]:14s} {key[1]:10s} {key[2]:9d} {key[3]:5d} {key[4]:6d} {g[0]:5d} {g[1]:8.3f} {100 * g[1] / total:6.1f} {tf:8.0f} {gb:7.0f}")
        print(f"(attention 'block' rows include the convs / norms listed separately; sum of non-block rows {inner:.2f} ms; event overhead is inside the total)")


if __name__ == "__main__":
    main()
