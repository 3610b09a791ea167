#!/bin/bash
# Sample board power / clocks (rocm-smi) while the self-attention or GEMM kernel loops: is the matrix pipe's clock power-managed?
cd /root/repo
python - <<'PY' &
import sys, time, torch
sys.path.insert(0, '/root/repo')
from gen3c_amd import ops
dev = torch.device('cuda:0')
S, H = 56320, 32
q = torch.randn(S, H*128, device=dev).to(torch.bfloat16); k = torch.randn(S, H*128, device=dev).to(torch.bfloat16)
v = torch.randn(S, H*128, device=dev).to(torch.bfloat16); vt = ops.transpose_v(v, S, 1, H); out = torch.empty_like(q)
a = torch.randn(S, 4096, device=dev).to(torch.bfloat16); w = (torch.randn(12288, 4096, device=dev)*0.02).to(torch.bfloat16); o2 = torch.empty(S, 12288, device=dev, dtype=torch.bfloat16)
torch.cuda.synchronize(); print('PHASE idle', time.time(), flush=True); time.sleep(3)
print('PHASE attention', time.time(), flush=True)
t0 = time.time()
while time.time() - t0 < 8:
    for _ in range(5): ops.flash_attn(q, k, vt, S, S, 1, H, out=out)
    torch.cuda.synchronize()
print('PHASE gemm', time.time(), flush=True)
t0 = time.time()
while time.time() - t0 < 8:
    for _ in range(40): ops.gemm_nt(a, w, out=o2)
    torch.cuda.synchronize()
print('PHASE done', time.time(), flush=True)
PY
PID=$!
for i in $(seq 1 50); do
  echo "T $(date +%s.%N)"; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | head -8
  sleep 0.5
  kill -0 $PID 2>/dev/null || break
done
wait $PID
