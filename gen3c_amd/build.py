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


def build(verbose: bool = False, force: bool = False, extra_flags: tuple = (), suffix: str = "") -> Path:
    """suffix / extra_flags build an A/B copy of the library (lib/libgen3c_hip<suffix>.so) next to the product one."""
    hipcc = _hipcc()
    objdir = LIBDIR / ("obj" + suffix)
    lib = LIBDIR / f"libgen3c_hip{suffix}.so"
    objdir.mkdir(parents=True, exist_ok=True)
    headers = sorted(CSRC.glob("*.hpp")) + sorted((ROOT.parent / "include").glob("*.h"))
    sources = sorted(CSRC.glob("*.hip"))
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{ROOT.parent / 'include'}", "-Wall",
             "-Wno-unused-function", *extra_flags]

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        if force or _newer(src, obj, headers):
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
