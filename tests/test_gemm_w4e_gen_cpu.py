"""CPU checks of the generated instruction streams of gemm_bf16_nt_w4e_kernel (tools/gen_gemm_w4e.py -> gen3c_amd/csrc/gemm_w4e_gen.hpp):
the committed header is what the generator produces, and the data path of the deferred epilogue (accumulator register -> packed pair ->
LDS slice X in MFMA layout -> transposing read -> 16-byte store) is the identity map of a wave's 128 x 128 quadrant."""
import importlib.util
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _gen():
    spec = importlib.util.spec_from_file_location("gen_gemm_w4e", ROOT / "tools" / "gen_gemm_w4e.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_header_is_up_to_date():
    gen = _gen()
    assert gen.OUT.read_text() == gen.generate(), "gemm_w4e_gen.hpp is stale: run python tools/gen_gemm_w4e.py"


def test_layout_model_is_the_identity_of_the_quadrant():
    gen = _gen()
    m = gen.layout_model()
    assert len(m) == 128 * 128
    for (row, col), (tok, feat) in m.items():
        assert (row, col) == (tok, feat)


def test_x_slice_accesses_are_conflict_free_where_the_hardware_allows():
    """ds_read_b128 is served in 4 groups of 16 lanes (MI355X_MICROARCH.md, LDS table); within a group the 16-byte slots (address mod 256) must differ."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for ch in range(4):
        for grp in groups:
            slots = set()
            for lane in grp:
                rr, c = lane >> 3, lane & 7
                addr = rr * 128 + ((c ^ rr) << 4) + 1024 * ch
                slots.add((addr % 256) // 16)
            assert len(slots) == 16


def test_every_period_statement_has_its_16_mfmas_and_balanced_gaps():
    gen = _gen()
    text = gen.OUT.read_text()
    fns = re.findall(r"G3_DEVICE void (gw4e_(?:none|gelu|gated)_k\d+)\(const GW4EOps& o\) \{(.*?)\n\}", text, re.S)
    assert len(fns) == 48
    for name, body in fns:
        lines = [re.sub(r'^GW4E_AB_\w\("', "", l.strip()).strip('")').replace("\\n\\t", "") for l in body.split("\n")
                 if l.strip().startswith('"') or l.strip().startswith("GW4E_")]
        idx = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        assert len(idx) == 16, name
        gaps = [idx[k + 1] - idx[k] - 1 for k in range(15)] + [len(lines) - idx[-1] - 1]
        # at most 5 single-issue instructions hide under one 32-cycle MFMA at one wave per SIMD (MI355X_MICROARCH.md); the barrier step carries its wait + barrier behind
        assert max(gaps[:15]) <= 5, (name, gaps)
