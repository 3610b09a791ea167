"""Audit of the hand-laid inline-asm kernels (attention MW / w4 / w4b, GEMM w4): what the compiler was not told, checked in its output.

Originally: audit of the hand-counted LDS reads in the MW attention kernels (cdna_hip_programming.md 5.7, item 1 / form (ii)).

An inline-asm `ds_read_b128` destination counts as written for the compiler the moment the statement ends, so under register pressure
hipcc may spill, copy or reuse that register BEFORE the data has landed (silent garbage). This script compiles csrc/attention.hip to
gfx950 assembly and walks every kernel: between an asm `ds_read_b128` and the asm `s_waitcnt lgkmcnt(N)` that retires it (reads retire in
order: all but the N youngest), no other instruction may name the destination registers. It also reports scratch (spill) reloads inside
the tile loop: their compiler-inserted `s_waitcnt vmcnt(0)` would also wait for the tile's LDS-DMA.

  python tools/asm_audit.py        -> exit code 1 on any finding (run by tests/test_abi.py-style CPU checks and by hand after edits)
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(asm_text: str):
    funcs, cur, name_full = {}, None, {}
    for ln in asm_text.split("\n"):
        m = re.match(r"^(_ZN12_GLOBAL__N_1\d+(flash_attn_fwd\w*kernel)(?:I(\w+)EEv|E)NS_10AttnParamsE):", ln)
        if m:
            cur = m.group(2).replace("flash_attn_fwd_", "") + "<" + (m.group(3) or "") + ">"
            name_full[cur] = m.group(2)
            funcs[cur] = []
        elif cur is not None:
            funcs[cur].append(ln)
            if ln.startswith(".Lfunc_end"):
                cur = None
    findings = []
    for name, v in funcs.items():
        n_asm_reads = 0
        pending = {}
        in_asm = False
        owned_agprs = "w4" in name_full.get(name, "")
        owned_limit = 256 if "w4b" in name_full.get(name, "") else 192  # w4b also owns the fragment ring a[192:255]
        hdr0 = [i for i, l in enumerate(v) if "Inner Loop Header" in l]
        loop_lo = hdr0[0] if hdr0 else len(v)
        loop_hi = max([i for i, l in enumerate(v) if "in Loop: Header" in l] + [0]) + 400
        last_trans = None  # (dst register, line) of a compiler-generated transcendental with no instruction after it yet
        for i, l in enumerate(v):
            t = l.strip()
            if t and t[0] not in ";." and "ASM" not in t:
                # hipcc pads a transcendental result read by the NEXT VALU instruction (gfx940+ trans forwarding hazard) only when it
                # generated the reader itself; a reader inside an inline-asm statement gets a stale value
                if last_trans is not None and in_asm and re.search(r"\b" + last_trans[0] + r"\b", t.split(None, 1)[1] if " " in t else ""):
                    findings.append(f"{name}: line {i}: asm `{t}` reads {last_trans[0]} straight after the compiler-generated transcendental that writes it")
                m_tr = re.match(r"(v_exp_|v_log_|v_rcp_|v_rsq_|v_sqrt_|v_sin_|v_cos_)\S*\s+(v\d+)", t)
                last_trans = (m_tr.group(2), i) if (m_tr and not in_asm) else None
            if "ASMSTART" in t:
                in_asm = True
                continue
            if "ASMEND" in t:
                in_asm = False
                continue
            if not t or t[0] in ";.":
                continue
            if in_asm and t.startswith("ds_read_b128"):
                n_asm_reads += 1
                for r in regs(t.split()[1].rstrip(",")):
                    pending[r] = i
                continue
            if in_asm and t.startswith("s_waitcnt lgkmcnt"):
                n = int(re.search(r"lgkmcnt\((\d+)\)", t).group(1))
                lines = sorted(set(pending.values()))
                keep = set(lines[len(lines) - n:]) if n > 0 else set()
                pending = {r: li for r, li in pending.items() if li in keep}
                continue
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                pending = {}
            for tok in re.findall(r"v\[\d+:\d+\]|v\d+", t):
                if regs(tok) & set(pending):
                    findings.append(f"{name}: line {i}: `{t}` touches an in-flight ds_read destination")
                    break
            if owned_agprs and not in_asm:  # the accumulator file a0..a191 belongs to the kernel's asm statements
                for m in re.finditer(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", t):
                    lo = int(m.group(1) or m.group(3))
                    if lo < owned_limit:
                        findings.append(f"{name}: line {i}: compiler-generated `{t}` touches an asm-owned AGPR")
                        break
            if owned_agprs and loop_lo <= i <= loop_hi and re.match(r"s_(buffer_)?load_", t):
                # the hand-counted lgkmcnt(N) waits assume LDS reads are the only LGKM operations in flight (SMEM returns out of order)
                findings.append(f"{name}: line {i}: scalar memory load near the tile loop of a kernel with hand-counted lgkmcnt waits: `{t}`")
            if owned_agprs and t.startswith("scratch_"):
                findings.append(f"{name}: line {i}: scratch access in a kernel with asm-owned AGPRs: `{t}`")
        hdr = [i for i, l in enumerate(v) if "Inner Loop Header" in l]
        if hdr and n_asm_reads:
            end = max(i for i, l in enumerate(v) if "in Loop: Header" in l)
            for i in range(hdr[0], end + 12):
                if "scratch_load" in v[i]:  # performance, not correctness: reported, does not fail the audit
                    print(f"note: {name}: line {i}: scratch reload inside the tile loop: `{v[i].strip()}`")
        if n_asm_reads:
            print(f"{name}: {n_asm_reads} hand-counted ds_reads audited" + (f" + asm-owned AGPRs a0..a{owned_limit - 1}" if owned_agprs else ""))
    return findings


def audit_gemm_w4(asm_text: str):
    """gemm_w4.hpp: inside the K loop (first to last MFMA of the kernel) the accumulators a0..a255 and the fragment buffers v192..v255 belong
    to the asm statements - no compiler-generated instruction may name them there (the epilogue, behind the loop, is free to); no scratch."""
    findings, cur, body = [], None, []
    funcs = {}
    for ln in asm_text.split("\n"):
        m = re.match(r"^(_ZN12_GLOBAL__N_1\d+gemm_bf16_nt_w4(_conv)?_kernelILi(\d+)E(?:Lb([01])E)?EEvNS_10GemmParamsE):", ln)
        if m:
            cur = f"gemm_w4{m.group(2) or ''}<{m.group(3)}{', persistent' if m.group(4) == '1' else ''}>"
            funcs[cur] = []
        elif cur is not None:
            funcs[cur].append(ln)
            if ln.startswith(".Lfunc_end"):
                cur = None
    for name, v in funcs.items():
        mf = [i for i, l in enumerate(v) if "v_mfma" in l]
        if not mf:
            findings.append(f"{name}: no MFMA found")
            continue
        lo, hi = mf[0], mf[-1]
        in_asm = False
        # loop depth of every line (block comments "... Depth=N"); the K-tile loop is the innermost one (depth 1, or 2 inside the persistent
        # tile loop). A scratch access THERE stalls every K tile on vmcnt(0) together with the tile's LDS-DMA: finding. Outside it (kernel
        # set-up, between the K loop and the last K tile, epilogue) it costs one wait per OUTPUT tile: reported as a note.
        depth, depths = 0, []
        for li, l in enumerate(v):
            if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
                m_d = re.search(r"Depth=(\d+)", l)
                depth = int(m_d.group(1)) if m_d else 0
                for l2 in v[li + 1:li + 4]:  # a loop header's own depth is on the continuation lines ("=>This Inner Loop Header: Depth=N")
                    m_h = re.search(r"Loop Header: Depth=(\d+)", l2)
                    if m_h and l2.lstrip().startswith(";"):
                        depth = int(m_h.group(1))
            depths.append(depth)
        max_depth = max(depths[i] for i in mf)  # depth of the K-tile loop = where the bulk of the MFMAs sits (the last K tiles are peeled: one less)
        for i, l in enumerate(v):
            t = l.strip()
            if "ASMSTART" in t:
                in_asm = True
                continue
            if "ASMEND" in t:
                in_asm = False
                continue
            if not t or t[0] in ";.":
                continue
            if t.startswith("scratch_"):
                if max_depth > 0 and depths[i] == max_depth and lo <= i <= hi:
                    findings.append(f"{name}: line {i}: scratch access inside the K-tile loop `{t}`")
                elif lo <= i <= hi:
                    print(f"note: {name}: line {i}: scratch access between K tiles (once per output tile): `{t}`")
            if in_asm or not (lo <= i <= hi):
                continue
            for m in re.finditer(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b", t):
                kind = m.group(1) or m.group(4)
                first = int(m.group(2) or m.group(5))
                last = int(m.group(3) or m.group(5))
                if kind == "a" or last >= 192:
                    findings.append(f"{name}: line {i}: compiler-generated `{t}` touches an asm-owned register inside the K loop")
                    break
        print(f"{name}: K loop lines {lo}..{hi} audited (asm-owned a0..a255, v192..v255)")
    return findings


def audit_gemm_w4e(asm_text: str):
    """gemm_w4e.hpp (round 5): from the first fragment read to the end of the kernel the accumulators a0..a255, the fragment buffers v192..v255, the
    drained tile v64..v191 and the gate vectors v56..v63 belong to the generated asm statements (gemm_w4e_gen.hpp) - the finished tile LIVES in
    v64..v191 across compiler-generated code (pointer arithmetic, the unit switch), which therefore may not name any of them anywhere in the kernel;
    no scratch; no scalar memory load behind the first MFMA (the gated-residual statements wait with hand-counted lgkmcnt(N): SMEM returns out of order);
    every period statement keeps its 16 MFMAs."""
    findings, cur, funcs = [], None, {}
    for ln in asm_text.split("\n"):
        m = re.match(r"^(_ZN12_GLOBAL__N_1\d+gemm_bf16_nt_w4e_kernelILi(\d+)ELb([01])EEEvNS_10GemmParamsE):", ln)
        if m:
            cur = f"gemm_w4e<{m.group(2)}, {'tokens' if m.group(3) == '1' else 'weights'} first>"
            funcs[cur] = []
        elif cur is not None:
            funcs[cur].append(ln)
            if ln.startswith(".Lfunc_end"):
                cur = None
    if len(funcs) != 6:
        findings.append(f"gemm_w4e: expected 6 kernel instances (epilogues 0, 1, 2 x piece order), found {sorted(funcs)}")
    for name, v in funcs.items():
        in_asm, seen_mfma, n_mfma_stmt, n_stmt = False, False, 0, 0
        for i, l in enumerate(v):
            t = l.strip()
            if "ASMSTART" in t:
                in_asm, n_mfma_stmt = True, 0
                continue
            if "ASMEND" in t:
                in_asm = False
                if n_mfma_stmt:
                    n_stmt += 1
                    if n_mfma_stmt != 16:
                        findings.append(f"{name}: line {i}: an asm statement with {n_mfma_stmt} MFMAs (a K step has 16)")
                continue
            if not t or t[0] in ";.":
                continue
            if in_asm:
                if t.startswith("v_mfma"):
                    n_mfma_stmt += 1
                    seen_mfma = True
                continue
            if t.startswith("scratch_"):
                findings.append(f"{name}: line {i}: scratch access `{t}`")
            if seen_mfma and re.match(r"s_(buffer_)?load_", t):
                findings.append(f"{name}: line {i}: scalar memory load behind the first MFMA (hand-counted lgkmcnt waits): `{t}`")
            for m in re.finditer(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b", t):
                kind = m.group(1) or m.group(4)
                last = int(m.group(3) or m.group(5))
                if kind == "a" or last >= 56:
                    findings.append(f"{name}: line {i}: compiler-generated `{t}` touches an asm-owned register")
                    break
        print(f"{name}: {n_stmt} K-step statements audited (asm-owned a0..a255, v56..v255 over the whole kernel, no scratch, no SMEM behind the first MFMA)")
    return findings


def audit_attn_d512(asm_text: str):
    """attention_d512.hip (round 4): all 256 AGPRs hold the O^T accumulators and belong to the asm statements from the zeroing prologue to the
    epilogue's reads - NO compiler-generated instruction of the kernel may name an AGPR (hipcc, left alone, parks the score accumulators in
    a[0:31]: exactly where block (0, 0) lives); no scratch anywhere (a reload's compiler-inserted vmcnt(0) would also drain the tile's LDS-DMA);
    between a hand-issued ds_read_b128 and the hand-counted lgkmcnt wait that retires it nothing else may name its destination registers."""
    findings, body, cur = [], [], False
    for ln in asm_text.split("\n"):
        if re.match(r"^_ZN12_GLOBAL__N_1\d+spatial_attn_d512_kernel\w*:", ln):
            cur = True
        elif cur:
            body.append(ln)
            if ln.startswith(".Lfunc_end"):
                break
    if not body:
        return ["attention_d512: kernel not found in the assembly"]
    in_asm, pending, n_reads, n_mfma = False, {}, 0, 0
    for i, l in enumerate(body):
        t = l.strip()
        if "ASMSTART" in t:
            in_asm = True
            continue
        if "ASMEND" in t:
            in_asm = False
            continue
        if not t or t[0] in ";.":
            continue
        n_mfma += t.startswith("v_mfma")
        if t.startswith("scratch_"):
            findings.append(f"attention_d512: line {i}: scratch access `{t}`")
        if in_asm and t.startswith("ds_read_b128"):
            n_reads += 1
            for r in regs(t.split()[1].rstrip(",")):
                pending[r] = i
            continue
        if in_asm and t.startswith("s_waitcnt lgkmcnt"):
            n = int(re.search(r"lgkmcnt\((\d+)\)", t).group(1))
            lines = sorted(set(pending.values()))
            keep = set(lines[len(lines) - n:]) if n > 0 else set()
            pending = {r: li for r, li in pending.items() if li in keep}
            continue
        if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
            pending = {}
        for tok in re.findall(r"v\[\d+:\d+\]|v\d+", t):
            if regs(tok) & set(pending):
                findings.append(f"attention_d512: line {i}: `{t}` touches an in-flight ds_read destination")
                break
        if not in_asm and re.search(r"\ba\[\d+:\d+\]|\ba\d+\b", t):
            findings.append(f"attention_d512: line {i}: compiler-generated `{t}` touches an asm-owned AGPR")
    print(f"attention_d512: {n_reads} hand-counted ds_reads, {n_mfma} MFMAs audited (asm-owned a0..a255, no scratch)")
    if n_mfma != 128 or n_reads < 64:
        findings.append(f"attention_d512: expected 128 MFMAs per tile body and >= 64 hand-issued fragment reads, found {n_mfma} / {n_reads}")
    return findings


def _compile_to_asm(td: Path, stem: str, extra=()):
    out = td / (stem + ".s")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", *extra, f"-I{ROOT / 'include'}", "-S", "--cuda-device-only",
                        str(ROOT / "gen3c_amd" / "csrc" / (stem + ".hip")), "-o", str(out)], capture_output=True, text=True)
    return r, out


def main():
    # the three translation units are compiled side by side (gemm.hip alone takes minutes: it bounds the audit's wall time), with the flags of
    # gen3c_amd/build.py (PER_FILE_FLAGS: attention.hip without the SLP vectoriser)
    from concurrent.futures import ThreadPoolExecutor
    jobs = (("attention", ("-fno-slp-vectorize",)), ("gemm", ()), ("attention_d512", ()))
    with tempfile.TemporaryDirectory() as td:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            done = list(ex.map(lambda j: _compile_to_asm(Path(td), *j), jobs))
        for r, _ in done:
            if r.returncode != 0:
                print(r.stderr)
                sys.exit(2)
        (_, a_attn), (_, a_gemm), (_, a_d512) = done
        findings = audit(a_attn.read_text())
        gemm_text = a_gemm.read_text()
        findings += audit_gemm_w4(gemm_text)
        findings += audit_gemm_w4e(gemm_text)
        findings += audit_attn_d512(a_d512.read_text())
    for f in findings:
        print("FINDING:", f)
    print("asm audit:", "clean" if not findings else f"{len(findings)} finding(s)")
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
