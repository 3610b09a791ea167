"""A/B of g3_groupnorm_apply_cl_bf16 (GroupNorm apply + swish, HBM-bound) between the product library and lib/libgen3c_hip_ab.so at the tokenizer's shapes."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from gen3c_amd import _lib, ops  # noqa: E402

base = _lib.load()
alt = C.CDLL(str(ROOT / "gen3c_amd" / "lib" / "libgen3c_hip_ab.so"))
for name, argtypes in _lib.SIGNATURES.items():
    getattr(alt, name).argtypes = argtypes
    getattr(alt, name).restype = _lib._RESTYPES.get(name, C.c_int)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for (T, HW, Cc) in ((31, 176 * 320, 256), (31, 176 * 320, 512), (16, 14080, 512), (31, 176 * 320, 128)):
    x = torch.randn(T, HW, Cc, device=dev).to(torch.bfloat16)
    g = torch.rand(Cc, device=dev).to(torch.bfloat16) + 0.5
    b = torch.randn(Cc, device=dev).to(torch.bfloat16) * 0.1
    stats = torch.stack([x.double().sum(dim=(1, 2)), (x.double() ** 2).sum(dim=(1, 2))], dim=1).contiguous()
    outs, res = {}, {"product": [], "ab": []}
    for rep in range(6):
        for name, lib in (("product", base), ("ab", alt)):
            o = torch.empty_like(x)
            tm = ops.HipTimer()
            tm.start()
            rc = lib.g3_groupnorm_apply_cl_bf16(x.data_ptr(), Cc, g.data_ptr(), b.data_ptr(), stats.data_ptr(), o.data_ptr(), Cc, T, HW, Cc, 1e-6, 1, st)
            tm.stop()
            assert rc == 0
            if rep:
                res[name].append(tm.elapsed_ms())
            outs[name] = o
    gb = 2 * x.numel() * 2 / 1e9
    ref = (x.float() - x.float().mean(dim=(1, 2), keepdim=True)) / (x.float().var(dim=(1, 2), unbiased=False, keepdim=True) + 1e-6).sqrt() * g.float() + b.float()
    ref = ref * torch.sigmoid(ref)
    for name, ms in res.items():
        m = sum(ms) / len(ms)
        print(f"[{T}x{HW}x{Cc}] {name:8s} {m:.4f} ms  {gb / m * 1e3 / 1e3:.2f} TB/s   vs fp32 reference rel-L2 {float((outs[name].float() - ref).norm() / ref.norm()):.2e}")
    print(f"    product vs ab: max |diff| {float((outs['product'].float() - outs['ab'].float()).abs().max()):.3e}")
