import sys, math, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from gen3c_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
def run(T, H, W, C, N, kind, res):
    geo = {"t3": (3,1,1,1,1,1,-2,0,0), "s3": (1,3,3,1,1,1,0,-1,-1), "p1": (1,1,1,1,1,1,0,0,0)}[kind]
    kt,kh,kw = geo[:3]
    x = torch.randn(T,H,W,C, device=dev).to(torch.bfloat16); w = (torch.randn(kt*kh*kw, N, C, device=dev)*0.02).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16); r = torch.randn(T,H,W,N, device=dev).to(torch.bfloat16) if res else None
    o = torch.empty(T,H,W,N, device=dev, dtype=torch.bfloat16)
    def f():
        rc = lib.g3_conv3d_cl_bf16(x.data_ptr(), C, w.data_ptr(), C, b.data_ptr(), r.data_ptr() if res else None, N, o.data_ptr(), N, C, N, T,H,W, T,H,W, *geo, st)
        assert rc == 0
    for _ in range(2): f()
    torch.cuda.synchronize(); tm = ops.HipTimer(); tm.start()
    for _ in range(5): f()
    tm.stop(); ms = tm.elapsed_ms()/5
    fl = 2.0*T*H*W*N*C*kt*kh*kw
    print(f"{kind}{'+res' if res else '    '} C={C} N={N} T={T:3d} {H}x{W}: {ms:7.3f} ms {fl/ms/1e9:6.0f} TF/s   in {x.numel()*2/1e6:.0f} MB", flush=True)
for T in (3, 8, 31):
    for res in (False, True):
        run(T, 176, 320, 256, 256, "t3", res)
for T in (2, 16):
    run(T, 88, 160, 512, 512, "t3", True)
    run(T, 88, 160, 512, 512, "p1", False)
    run(T, 88, 160, 512, 512, "s3", False)

# the same products as plain GEMMs (no gather, no per-tap state): what of a short-K convolution's cost is the implicit-GEMM machinery?
def gemm(M, N, K, epi):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.randn(1, N, device=dev).to(torch.bfloat16); o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm_nt(a, w, out=o, epilogue=epi, **({"gate": b} if epi == 3 else {}))
    for _ in range(2): f()
    torch.cuda.synchronize(); tm = ops.HipTimer(); tm.start()
    for _ in range(5): f()
    tm.stop(); ms = tm.elapsed_ms() / 5
    print(f"gemm epi{epi} M={M} N={N} K={K}: {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:6.0f} TF/s", flush=True)
for (M, N, K) in ((1745920, 256, 768), (1745920, 256, 2304), (225280, 512, 1536), (225280, 512, 512), (225280, 512, 4608)):
    gemm(M, N, K, 3)
    gemm(M, N, K, 0)
