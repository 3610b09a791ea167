"""Builds gen3c_amd/lib/libgen3c_hip.so (the C-ABI library of include/gen3c_hip.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as the "does it build" check and the
resulting .so travels with the tree to the MI355X box. One translation unit per .hip file, compiled in parallel,
recompiled only when the source (or a header) is newer than its object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIBDIR = ROOT / "lib"
OBJDIR = LIBDIR / "obj"
LIB = LIBDIR / "libgen3c_hip.so"
ARCH = "gfx950"
# attention.hip: the SLP vectoriser packs the softmax row-sum adds into v_pk_add_f32 and, in the hand-placed one-wave-per-SIMD kernel, collects
# them into one serial chain away from where the source puts them (packed fp32 VALU beside MFMAs is slower anyway, MI355X_MICROARCH.md)
PER_FILE_FLAGS = {"attention.hip": ("-fno-slp-vectorize",)}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain is required to build the gen3c_amd HIP library)")


def _newer(src: Path, dst: Path, extra: list[Path]) -> bool:
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [src, *extra])


AUDITED = ("attention", "gemm", "attention_d512")  # translation units with hand-laid inline-asm kernels (tools/asm_audit.py)


def _asm_audit(hipcc: str, flags: list, objdir: Path, verbose: bool) -> None:
    """The hand-laid kernels keep live state in registers the compiler is not told about (gemm_w4e: the drained tile in v64..v191 across compiler code);
    whether hipcc left them alone is a property of ITS output, so the build checks it: every audited unit is compiled to gfx950 assembly with the
    product flags (side by side with the objects) and tools/asm_audit.py walks it. Findings fail the build - a different compiler version must not
    silently corrupt the default GEMM path. Skipped (with a note) only where tools/ did not travel with the package."""
    root = ROOT.parent
    if not (root / "tools" / "asm_audit.py").exists():
        print("gen3c_amd.build: tools/asm_audit.py not found - inline-asm register audit skipped", file=sys.stderr)
        return
    if str(root) not in sys.path:
        sys.path.insert(0, str(root))
    from tools import asm_audit

    def to_asm(stem: str) -> str:
        out = objdir / (stem + ".audit.s")
        cmd = [hipcc, *flags, *PER_FILE_FLAGS.get(stem + ".hip", ()), "-S", "--cuda-device-only", str(CSRC / (stem + ".hip")), "-o", str(out)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc -S failed for {stem}.hip:\n{r.stderr}")
        text = out.read_text()
        out.unlink()
        return text

    with ThreadPoolExecutor(max_workers=len(AUDITED)) as ex:
        attn, gemm, d512 = ex.map(to_asm, AUDITED)
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        findings = asm_audit.audit(attn) + asm_audit.audit_gemm_w4(gemm) + asm_audit.audit_gemm_w4e(gemm) + asm_audit.audit_attn_d512(d512)
    if verbose:
        print(buf.getvalue())
    if findings:
        raise RuntimeError("inline-asm register audit failed (tools/asm_audit.py):\n" + "\n".join(findings))


def build(verbose: bool = False, force: bool = False, extra_flags: tuple = (), suffix: str = "", audit: bool = True) -> Path:
    """suffix / extra_flags build an A/B copy of the library (lib/libgen3c_hip<suffix>.so) next to the product one."""
    hipcc = _hipcc()
    objdir = LIBDIR / ("obj" + suffix)
    lib = LIBDIR / f"libgen3c_hip{suffix}.so"
    objdir.mkdir(parents=True, exist_ok=True)
    headers = sorted(CSRC.glob("*.hpp")) + sorted((ROOT.parent / "include").glob("*.h"))
    sources = sorted(CSRC.glob("*.hip"))
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT.parent / 'include'}", "-Wall",
             "-Wno-unused-function", *extra_flags]

    stamp = objdir / "asm_audit.ok"
    stale = [src.stem for src in sources if force or _newer(src, objdir / (src.stem + ".o"), headers)]
    audit_future = None
    audit_pool = None
    if audit and not suffix and (not stamp.exists() or any(s_ in AUDITED for s_ in stale)):
        stamp.unlink(missing_ok=True)
        audit_pool = ThreadPoolExecutor(max_workers=1)
        audit_future = audit_pool.submit(_asm_audit, hipcc, flags, objdir, verbose)

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        if src.stem in stale:
            cmd = [hipcc, *flags, *PER_FILE_FLAGS.get(src.name, ()), "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(sources))) as ex:
        objs = list(ex.map(compile_one, sources))

    if audit_future is not None:
        audit_future.result()  # raises on findings: no library is linked from objects that failed the audit
        audit_pool.shutdown()
        stamp.write_text("tools/asm_audit.py: clean\n")

    if force or not lib.exists() or any(o.stat().st_mtime > lib.stat().st_mtime for o in objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
