"""CPU: the C-ABI library builds/loads and exports every symbol include/gen3c_hip.h declares (no compute calls)."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "gen3c_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(g3_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from gen3c_amd import _lib, build
    build.build()
    syms = _declared_symbols()
    assert len(syms) >= 10
    lib = ctypes.CDLL(str(_lib.lib_path()))
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in gen3c_hip.h but not exported by libgen3c_hip.so"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in gen3c_amd/_lib.py"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in the header"


def test_loader_sets_signatures():
    from gen3c_amd import _lib
    lib = _lib.load()
    assert lib.g3_abi_version() == 1
    assert lib.g3_last_error() is not None


def test_no_cpu_fallback():
    import pytest
    import torch
    from gen3c_amd import _lib, ops
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(_lib.Gen3cHipError):
        ops.gemm_nt(a, a)


def test_product_code_never_imports_oracle():
    for p in (ROOT / "gen3c_amd").rglob("*.py"):
        src = p.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f"{p} imports the oracle"
