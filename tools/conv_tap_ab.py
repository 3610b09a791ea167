"""Round 5 A/B: the next tap's token addresses computed in the MFMA gaps of the barrier K step (conv_w4 = 1, default) vs by compiler code between two
statements (conv_w4 = 2): convolution shapes of the tokenizer, arms alternating in one process, then the tokenizer itself.  usage (GPU box): python tools/conv_tap_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
GEO = {"t3": (3, 1, 1, 1, 1, 1, -2, 0, 0), "s3": (1, 3, 3, 1, 1, 1, 0, -1, -1)}


def case(T, H, W, C, N, kind):
    geo = GEO[kind]
    kt, kh, kw = geo[:3]
    x = torch.randn(T, H, W, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(kt * kh * kw, N, C, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    o = [torch.empty(T, H, W, N, device=dev, dtype=torch.bfloat16) for _ in range(2)]

    def f(out):
        assert lib.g3_conv3d_cl_bf16(x.data_ptr(), C, w.data_ptr(), C, b.data_ptr(), None, N, out.data_ptr(), N, C, N, T, H, W, T, H, W, *geo, st) == 0

    times = {1: [], 2: []}
    for _ in range(4):
        for arm in (1, 2):
            ops.set_option("conv_w4", arm)
            f(o[arm - 1])
            torch.cuda.synchronize()
            tm = ops.HipTimer()
            tm.start()
            for _i in range(5):
                f(o[arm - 1])
            tm.stop()
            times[arm].append(tm.elapsed_ms() / 5)
    ops.set_option("conv_w4", 1)
    assert torch.equal(o[0], o[1]), "the two forms differ"
    fl = 2.0 * T * H * W * N * C * kt * kh * kw
    m = {a: sorted(v)[len(v) // 2] for a, v in times.items()}
    print(f"{kind} C={C:3d} N={N:3d} T={T:2d} {H}x{W}: in gaps {m[1]:7.3f} ms {fl / m[1] / 1e9:6.0f} TF/s | between statements {m[2]:7.3f} ms {fl / m[2] / 1e9:6.0f} TF/s | {100 * (m[2] / m[1] - 1):+.1f} %  (bitwise equal)", flush=True)


for T, H, W, C, kind in ((31, 176, 320, 256, "s3"), (31, 176, 320, 256, "t3"), (16, 88, 160, 512, "s3"), (16, 88, 160, 512, "t3"), (31, 352, 640, 128, "s3"), (31, 352, 640, 128, "t3")):
    case(T, H, W, C, C, kind)
