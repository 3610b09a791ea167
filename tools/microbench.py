"""Interleaved A/B timing of kernel variants at the benchmark shapes (hipEvents on the launch stream).
usage (GPU box): python tools/microbench.py [gemm] [attn]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402


def timeit(fn, iters):
    tm = ops.HipTimer()
    fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(iters):
        fn()
    tm.stop()
    return tm.elapsed_ms() / iters


def bench_gemm():
    dev = torch.device("cuda:0")
    shapes = [("qkv", 56320, 12288, 4096, 0), ("out", 56320, 4096, 4096, 2), ("w1", 56320, 16384, 4096, 1), ("w2", 56320, 4096, 16384, 2)]
    for name, M, N, K, epi in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        gate = torch.randn(1, N, device=dev).to(torch.bfloat16)
        res = torch.randn(M, N, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw = dict(gate=gate, residual=res) if epi == 2 else {}
        fl = 2.0 * M * N * K
        res_line = []
        outs = {}
        for rnd in range(3):
            for variant in (2, 3, "3e"):  # 2 = 8-wave ping-pong, 3 = one wave per SIMD, 3e = the same, persistent with the DEFERRED epilogue (gemm_w4e.hpp, round 5)
                ops.set_option("gemm_pingpong", 2 if variant == 2 else 3)
                ops.set_option("gemm_deferred", 1 if variant == "3e" else 0)
                ms = timeit(lambda: ops.gemm_nt(a, w, out=out, epilogue=epi, **kw), 5)
                res_line.append((variant, ms, fl / ms / 1e9))
                if rnd == 0:
                    outs[variant] = out.clone()
        ops.set_option("gemm_pingpong", 3)
        ops.set_option("gemm_deferred", 1)
        same = torch.equal(outs[2], outs[3]) and torch.equal(outs[3], outs["3e"])
        res_line.append(("bitwise-equal", 0.0 if same else -1.0, 1.0 if same else 0.0))
        if epi == 0:  # vendor-library yardstick for the plain GEMM (hipBLASLt through torch; not part of the product path)
            wt = w.t()
            ms = timeit(lambda: torch.matmul(a, wt, out=out), 5)
            res_line.append(("blaslt", ms, fl / ms / 1e9))
        print(f"gemm {name} {M}x{N}x{K} epi{epi}: " + "  ".join(f"[pp={v} {ms:.3f}ms {tf:.0f}TF]" for v, ms, tf in res_line), flush=True)
        del a, w, gate, res, out


def bench_attn():
    dev = torch.device("cuda:0")
    S, H = 56320, 8
    q = torch.randn(S, H * 128, device=dev).to(torch.bfloat16)
    k = torch.randn(S, H * 128, device=dev).to(torch.bfloat16)
    v = torch.randn(S, H * 128, device=dev).to(torch.bfloat16)
    vt = ops.transpose_v(v, S, 1, H)
    out = torch.empty_like(q)
    fl = 4.0 * S * S * 128 * H
    line = []
    variants = [int(x) for x in (os.environ.get("G3_MB_ATTN_VARIANTS") or "4,6,8").split(",")]
    ops.set_option("attn_variant", 0)
    ref = ops.flash_attn(q, k, vt, S, S, 1, H).float()
    for rnd in range(3):
        for variant in variants:
            ops.set_option("attn_variant", variant)
            ms = timeit(lambda: ops.flash_attn(q, k, vt, S, S, 1, H, out=out), 3)
            err = float((out.float() - ref).abs().max()) if rnd == 0 else 0.0
            line.append((variant, ms, fl / ms / 1e9, err))
    ops.set_option("attn_variant", 0)
    print(f"attn S={S} H={H}: " + "  ".join(f"[v{v} {ms:.2f}ms {tf:.0f}TF" + (f" maxdiff_vs_v4 {e:.1e}]" if e else "]") for v, ms, tf, e in line), flush=True)
    # cross attention shape
    kc = torch.randn(512, 32 * 128, device=dev).to(torch.bfloat16)
    vc = torch.randn(512, 32 * 128, device=dev).to(torch.bfloat16)
    qc = torch.randn(S, 32 * 128, device=dev).to(torch.bfloat16)
    vtc = ops.transpose_v(vc, 512, 1, 32)
    oc = torch.empty_like(qc)
    flc = 4.0 * S * 512 * 128 * 32
    line = []
    for variant in (4, 9, 11, 4, 9, 11):  # 8-wave kernel (the automatic choice for short contexts) vs the one-wave kernels
        ops.set_option("attn_variant", variant)
        ms = timeit(lambda: ops.flash_attn(qc, kc, vtc, S, 512, 1, 32, out=oc), 5)
        line.append((variant, ms, flc / ms / 1e9))
    ops.set_option("attn_variant", 0)
    print("cross-attn S=56320 M=512 H=32: " + "  ".join(f"[v{v} {ms:.3f}ms {tf:.0f}TF]" for v, ms, tf in line), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["gemm", "attn"]
    if "attn" in what:
        bench_attn()
    if "gemm" in what:
        bench_gemm()
