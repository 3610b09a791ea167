"""Debug helper: per-row error of one attention variant against the fp32 reference on the rescale test inputs."""
import math, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gen3c_amd import _lib
if os.environ.get('DBG_LIB'):
    _lib._LIB_PATH = Path(os.environ['DBG_LIB']).resolve()
from gen3c_amd import ops
dev = torch.device("cuda:0")
variant = int(os.environ.get("V", "10"))
Sq, Skv, B, H = 64, 320, 1, 1
for case in range(4):
    g = torch.Generator(device=dev).manual_seed(11)
    q = torch.randn(Sq, 128, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(Skv, 128, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(Skv, 128, device=dev, generator=g).to(torch.bfloat16)
    if case in (0, 1): k[200] = (q[5].float() * 3.0).to(torch.bfloat16)
    if case in (0, 2): k[300] = (q[17].float() * 6.0).to(torch.bfloat16)
    if case in (0, 3): k[130] = (q[40].float() * 0.5).to(torch.bfloat16)
    vt = ops.transpose_v(v, Skv, B, H)
    ops.set_option("attn_variant", variant)
    out = ops.flash_attn(q, k, vt, Sq, Skv, B, H).float()
    ops.set_option("attn_variant", 0)
    ref = torch.softmax(q.float() @ k.float().T / math.sqrt(128), -1) @ v.float()
    err = (out - ref).abs().amax(1)
    bad = (err > 0.05).nonzero().flatten().tolist()
    print(f"case {case}: bad rows {bad}  max err {float(err.max()):.3f}")
    if bad:
        r = bad[0]
        print("   ratio out/ref row", r, (out[r, :6] / ref[r, :6]).tolist())
