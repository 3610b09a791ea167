"""What the local-KV-first context-parallel schedule costs in COMPUTE on one rank, measured on one GPU at the per-rank shapes of cp = 2, 4, 8
(B = 2 batched cond / uncond forward, 32 heads in 4 head groups of 8): per head group, (a) gather-first = ONE attention launch over all
cp x S_local gathered keys, (b) local-first = partial over the rank's own shard + partials over the ranks before / after it + the merge
(g3_flash_attn_fwd_ex_bf16 / g3_attn_merge_partials_bf16), for a middle rank (3 parts) and an edge rank (2 parts). Both kernels.
The exchange itself (what the schedule is there to hide) cannot be measured on one GPU - the driver's multi-GPU run reports it (bench.py `cp`)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
S, B, H, G = 56320, 2, 32, 4
Hg = H // G
for cp in (2, 4, 8):
    Sl = S // cp
    rows = Sl * B
    g = torch.Generator(device=dev).manual_seed(cp)
    q = torch.randn(rows, Hg * 128, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(cp * rows, Hg * 128, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(cp * rows, Hg * 128, device=dev, generator=g).to(torch.bfloat16)
    vt = torch.stack([ops.transpose_v(v[r * rows:(r + 1) * rows], Sl, B, Hg) for r in range(cp)])  # [cp, B, Hg, 128, Sl]
    out = torch.empty(rows, Hg * 128, device=dev, dtype=torch.bfloat16)
    flops = 4.0 * Sl * S * 128 * Hg * B
    for variant, vname in ((11, "one-wave"), (4, "8-wave")):
        def gather_first():
            ops.flash_attn(q, k, vt, Sl, S, B, Hg, out=out, variant=variant)

        def local_first(rank):
            parts = []
            for (r0, r1) in ((rank, rank + 1), (0, rank), (rank + 1, cp)):
                if r1 > r0:
                    parts.append(ops.flash_attn(q, k[r0 * rows:r1 * rows], vt[r0:r1], Sl, (r1 - r0) * Sl, B, Hg, variant=variant, partial=True))
            ops.attn_merge(parts, Sl, B, Hg, out=out)

        cases = [("gather-first", gather_first), ("local-first edge rank", lambda: local_first(0))]
        if cp > 2:
            cases.append(("local-first middle rank", lambda: local_first(cp // 2)))
        for _ in range(2):
            for _n, fn in cases:
                fn()
        torch.cuda.synchronize()
        res = {}
        for rep in range(3):
            for name, fn in cases:
                tm = ops.HipTimer()
                tm.start()
                for _ in range(G):  # the 4 head groups of a layer, back to back on one stream
                    fn()
                tm.stop()
                res.setdefault(name, []).append(tm.elapsed_ms())
        base = sorted(res["gather-first"])[1]
        print(f"cp={cp} S_local={Sl} {vname:8s}: " + "  ".join(f"[{n}: {sorted(t)[1]:.3f} ms/layer = {G * flops / sorted(t)[1] / 1e9:.0f} TF/s, x{sorted(t)[1] / base:.3f}]"
                                                            for n, t in res.items()), flush=True)
