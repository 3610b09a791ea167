"""CPU check of the hand-laid inline-asm kernels: tools/asm_audit.py cross-compiles attention.hip and gemm.hip to gfx950 assembly and verifies
what the compiler was not told - no compiler-generated instruction touches the asm-owned accumulator / fragment registers, nothing spills
to scratch, no scalar loads next to the hand-counted lgkmcnt waits, no in-flight LDS destination is touched before its wait, no transcendental
result is consumed by the very next instruction inside an asm statement. Needs hipcc (no GPU)."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_asm_audit_clean():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "asm_audit.py")], capture_output=True, text=True, timeout=1500)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "asm audit: clean" in r.stdout
