"""Compile one HIP source for gfx950 and print a per-kernel VGPR / spill / scratch table.  usage: python tools/kernel_resources.py gen3c_amd/csrc/gemm.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Iinclude", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd="/root/repo").stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        cur = re.sub(r"\(.*", "", cur).replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
    if "error" in line:
        print(line)
for k, v in rows.items():
    if flt in k:
        print(f"{k:60s} vgpr {v.get('VGPRs', -1):3d} agpr {v.get('AGPRs', -1):3d} spill {v.get('VGPRs Spill', -1):3d} scratch {v.get('ScratchSize', -1):4d} occ {v.get('Occupancy', -1)}")
