"""One warm + one timed encode / decode of a 121x704x1280 clip in the PRODUCT configuration (no A/B switches): the command behind
profiles/r3_tokenizer_kernel_stats.csv (a single-configuration rocprofv3 profile)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402
from gen3c_amd.tokenizer import CausalVideoTokenizerNet  # noqa: E402

dev = torch.device("cuda:0")
net = CausalVideoTokenizerNet(channels=128, device=dev)
net.init_random(seed=0)
x = (torch.rand(1, 3, 121, 704, 1280, device=dev) * 2 - 1).to(torch.bfloat16)
z = None
for name, fn, tflop in (("encode", net.encoder, 35.7), ("decode", net.decoder, 61.3)):
    arg = x if name == "encode" else z
    out = fn(arg)
    torch.cuda.synchronize()
    tm = ops.HipTimer()
    tm.start()
    out = fn(arg)
    tm.stop()
    ms = tm.elapsed_ms()
    print(f"tokenizer {name}: {ms:.2f} ms = {tflop / ms * 1e3:.0f} TFLOP/s ({tflop / ms * 1e3 / 2500 * 100:.1f} % of 2.5 PF)", flush=True)
    if name == "encode":
        z = out
