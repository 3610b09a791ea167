"""bench.py's cpu_baseline times ONE Cosmos-7B-width DiT block (fp32 oracle, host cores) and extrapolates by FLOPs. This times 1 and 2 blocks
once each on the same host to show the per-block cost is additive. ~80 s of CPU."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

threads = min(32, os.cpu_count() or 1)
r1 = bench.cpu_baseline(threads, blocks=1)
r2 = bench.cpu_baseline(threads, blocks=2)
print(f"host threads {threads}")
print(f"1 block : {r1['seconds']:.2f} s  -> {r1['value']:.6e} denoise-steps/s extrapolated")
print(f"2 blocks: {r2['seconds']:.2f} s  -> {r2['value']:.6e} denoise-steps/s extrapolated")
print(f"second block costs {r2['seconds'] - r1['seconds']:.2f} s = {(r2['seconds'] - r1['seconds']) / r1['seconds']:.3f} x the first (embedding / final layer are in both)")
