"""Full-size tokenizer timing: encode/decode of one 121x704x1280 clip (random weights, CV8x8x8-720p widths).
Algorithmic work: encode 35.7 TFLOP, decode 61.3 TFLOP."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import ops  # noqa: E402
from gen3c_amd.tokenizer import CausalVideoTokenizerNet  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    T, H, W = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (121, 704, 1280)))
    net = CausalVideoTokenizerNet(channels=128, device=dev)
    net.init_random(seed=0)
    x = (torch.rand(1, 3, T, H, W, device=dev) * 2 - 1).to(torch.bfloat16)
    for name, fn, arg, tflop in (("encode", net.encoder, x, 35.7), ("decode", net.decoder, None, 61.3)):
      if arg is None:
          arg = z
      for pp in (0, 2, 0, 2):  # conv GEMM kernel: 0 = one barrier per K tile, 2 = ping-pong (default)
        ops.set_option("gemm_pingpong", pp)
        out = fn(arg)
        torch.cuda.synchronize()
        tm = ops.HipTimer()
        tm.start()
        out = fn(arg)
        tm.stop()
        ms = tm.elapsed_ms()
        scale = (T * H * W) / (121 * 704 * 1280)
        print(f"tokenizer {name} {T}x{H}x{W} pingpong={pp}: {ms:.1f} ms  ~{tflop*scale/ms*1e3:.0f} TFLOP/s  out {tuple(out.shape)} finite={bool(torch.isfinite(out.float()).all())} "
              f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
      if name == "encode":
          z = out


if __name__ == "__main__":
    main()
