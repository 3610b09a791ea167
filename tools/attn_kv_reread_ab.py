"""does the self-attention kernel's K / V^T re-read (memory-side reads 4x the algorithmic bytes) cost wall time?
One process, one box, interleaved arms with EQUAL flops per launch (4 Sq Skv d H = 1.30e13) on flash_attn_fwd_w4b_kernel<true>:

  A  bench geometry   Sq = Skv = 56 320, H = 8 (one head per XCD), XCD-local heads: each XCD streams its head's 28.8 MB K / V^T panel
                      once per round of its 32 CUs (7 rounds over 220 query blocks)                      -> the re-read as shipped
  A8 same, attn_xcd_heads = 0: consecutive query blocks of a head go round-robin over the 8 XCDs, so EVERY XCD streams EVERY
                      head's panel                                                                        -> 8x the fabric reads of A
  B  L2-resident      Sq = 788 480, Skv = 4 096, H = 8: the head's K + V^T panel is 2 MB (an XCD's L2 holds 4 MB): after the first
                      round every K / V^T byte is an L2 hit                                               -> ~zero fabric re-read
  B2 half resident    Sq = 394 240, Skv = 8 192, H = 8: 4 MB panel (= the L2)

B / B2 pay more workgroup prologues + epilogues per flop (3 080 / 1 540 workgroups per head instead of 220); the per-workgroup fixed
cost is measured by arm C (Skv = 4 096 at Sq = 56 320: 1/13.75 of A's flops, same workgroup count as A) and taken out:
  t_fixed_per_wg = (t_C - t_A / 13.75) / n_wg, applied to B's workgroup count.
If zero re-read (B, corrected) is not faster than A, and 8x the re-read (A8) is not slower, no K / V^T sharing scheme can pay.

usage (GPU box): python tools/attn_kv_reread_ab.py           (timing, hipEvents on the launch stream)
                 G3_REREAD_PMC=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python tools/attn_kv_reread_ab.py   (2 launches per arm, no timing)"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import _lib, ops  # noqa: E402

PMC = bool(os.environ.get("G3_REREAD_PMC"))
dev = torch.device("cuda:0")
H, HD = 8, 128
ARMS = [("A", 56320, 56320, 1), ("A8", 56320, 56320, 0), ("B", 788480, 4096, 1), ("B2", 394240, 8192, 1), ("C", 56320, 4096, 1)]


def operands(Sq, Skv):
    g = torch.Generator(device=dev).manual_seed(Sq + Skv)
    q = torch.randn(Sq, H * HD, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(Skv, H * HD, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(Skv, H * HD, device=dev, generator=g).to(torch.bfloat16)
    return q, k, ops.transpose_v(v, Skv, 1, H), torch.empty_like(q)


def launch(arm, buf):
    _, Sq, Skv, xcd = arm
    q, k, vt, out = buf
    ops.set_option("attn_xcd_heads", xcd)
    ops.flash_attn(q, k, vt, Sq, Skv, 1, H, out=out, variant=11)


def main():
    bufs = {}
    for arm in ARMS:
        key = (arm[1], arm[2])
        if key not in bufs:
            bufs[key] = operands(*key)
    name = _lib.load().g3_flash_attn_kernel_name(56320, 56320, 1, H).decode()
    print(f"kernel for the bench geometry: {name}; PMC mode: {PMC}")
    times = {a[0]: [] for a in ARMS}
    rounds = 1 if PMC else 4
    for _ in range(rounds):
        for arm in ARMS:
            buf = bufs[(arm[1], arm[2])]
            launch(arm, buf)  # warm
            torch.cuda.synchronize()
            if PMC:
                launch(arm, buf)
                torch.cuda.synchronize()
                continue
            tm = ops.HipTimer()
            tm.start()
            for _i in range(3):
                launch(arm, buf)
            tm.stop()
            times[arm[0]].append(tm.elapsed_ms() / 3)
    ops.set_option("attn_xcd_heads", 1)
    if PMC:
        print("done (launch order per arm: warm, counted): " + " ".join(f"{a[0]}(Sq={a[1]},Skv={a[2]},xcd={a[3]})" for a in ARMS))
        return
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    fl = {a[0]: 4.0 * a[1] * a[2] * HD * H for a in ARMS}
    for a in ARMS:
        n = a[0]
        print(f"arm {n:3s} Sq={a[1]:7d} Skv={a[2]:6d} xcd_heads={a[3]}: {med[n]:8.3f} ms  {fl[n] / med[n] / 1e9:7.1f} TFLOP/s   runs " + " ".join(f"{t:.3f}" for t in times[n]))
    n_wg_A = (56320 // 256) * H
    fixed = (med["C"] - med["A"] * (4096 / 56320)) / n_wg_A  # ms per workgroup beyond the tile loop
    print(f"per-workgroup fixed cost (prologue + epilogue) from arm C: {fixed * 1e3:.2f} us x 256 CUs-worth of overlap (wall per workgroup slot: {fixed * 1e3 * 256:.1f} us)")
    for n, Sq in (("B", 788480), ("B2", 394240)):
        n_wg = (Sq // 256) * H
        corr = med[n] - fixed * (n_wg - n_wg_A)
        print(f"arm {n} with the extra workgroup turn-overs taken out: {corr:.3f} ms = {fl[n] / corr / 1e9:.1f} TFLOP/s  (A: {med['A']:.3f} ms = {fl['A'] / med['A'] / 1e9:.1f})")


if __name__ == "__main__":
    main()
