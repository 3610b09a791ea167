"""Generator of gen3c_amd/csrc/gemm_w4e_gen.hpp: the instruction streams of gemm_bf16_nt_w4e_kernel (csrc/gemm_w4e.hpp), the one-wave-per-SIMD block GEMM
whose epilogue rides INSIDE the next output tile's K loop.

Why a generator: at one wave per SIMD nothing but the wave's own instruction order hides anything, so every K step is one asm statement of 16 MFMAs with the
other instructions placed in the gaps between them (gemm_w4.hpp). The deferred epilogue adds, per K step, a different slice of epilogue work (one packed
register of GELU / gated-residual arithmetic, a transposing LDS read, a residual LDS-DMA piece, a store) - 16 different statements per epilogue class.
Writing those by hand as macro strings is where mistakes hide; here each statement is assembled from a list of micro-ops and spread over the gaps by one rule.

Register ownership (literal names, as gemm_w4.hpp):   a[0:255]  accumulators, block (i, j) = features 32 i.., tokens 32 j.. at a[16 (4 j + i) : +15]
                                                       v[192:255] two buffers of 8 operand fragments
                                                       v[64:191]  P: the finished tile's accumulators as packed bf16 pairs (drain), unit u at v[64 + 16 u : +15]
                                                       v[56:63]   gate vectors of the finished tile (gated residual only): current half in v[56:59]
A unit = 32 tokens x 64 features of the wave's 128 x 128 quadrant (token block J = u >> 1, feature half ih = u & 1), 16 registers in the MFMA layout
(lane = token row l31, 4 consecutive features per (i, q)). It is transposed through a private 4 KiB LDS slice X ([32 rows][128 B], 16-byte chunk index
XOR row & 7) into W = v[64:79] in row layout: lane (rr = lane >> 3, c = lane & 7) holds, for chunk ch = 0..3, row 8 ch + rr, features 8 c .. 8 c + 7 in
W[4 ch : 4 ch + 3] - one 16-byte store per chunk, 8 lanes = one 128-byte line.

Period of 16 K steps (kappa = 0..15) per unit u, K tiles 1 + 4 u .. 4 + 4 u of the NEXT output tile (K tile 0 = preamble):
   every kappa          the epilogue arithmetic of register W[kappa]  (none / GELU / residual + gate * x with the residual word read just in time from LDS slice Y)
   kappa = 2, 6, 10, 14 (the BARRIER step: no operand LDS-DMA in it) store of chunk 3 of unit u - 1 / chunks 0, 1, 2 of unit u, and the residual LDS-DMA piece
                        (u,2) (u,3) (u+1,0) (u+1,1) -> Y slot; both are the last VMEM operations in front of the barrier, whose vmcnt(n) leaves them in flight
   kappa = 3            transposing read of unit u's chunk 3 into W[12:15] (stored one step earlier); kappa = 7, 11, 15: chunks 0, 1, 2 of unit u + 1
   between kappa = 3 and 4 (C++ side): unit u + 1's registers -> X  (gw4e_xwrite_<u>: the only statement that names a unit's registers)

usage: python tools/gen_gemm_w4e.py [--check]     (--check: exit 1 if the committed header differs from what this script generates)"""
from __future__ import annotations

import struct
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gen3c_amd" / "csrc" / "gemm_w4e_gen.hpp"

FR0, P0, G0 = 192, 64, 56
STAGE = 65536
T_OFF = 32768
EPI_NONE, EPI_GELU, EPI_GATED = 0, 1, 2
EPI_NAME = {EPI_NONE: "none", EPI_GELU: "gelu", EPI_GATED: "gated"}


def f32_bits(x) -> str:
    return "0x%08x" % struct.unpack("<I", struct.pack("<f", float(np.float32(x))))[0]


# ---- constants of gelu_erf_fast (common.hpp), folded in float arithmetic exactly as the C++ compiler folds them
f = np.float32
GELU_C0 = f(0.3275911) * f(0.70710678118654752440)
GELU_K = [f(0.5) * f(1.061405429), f(0.5) * f(-1.453152027), f(0.5) * f(1.421413741), f(0.5) * f(-0.284496736), f(0.5) * f(0.254829592)]
GELU_E = f(-0.5) * f(1.44269504088896340736)


# ---------------------------------------------------------------------------------------------------------------- micro-ops
class Op:
    """One instruction of a statement. kind: valu / trans / lds / vmem / salu / wait. pin: gap index it must sit in (None: spread)."""

    def __init__(self, text: str, kind: str = "valu", pin: int | None = None, tag: str = "", last: bool = False, ab: str = ""):
        self.text, self.kind, self.pin, self.tag, self.last = text, kind, pin, tag, last  # last: pinned op that goes BEHIND the spread ops of its gap
        self.ab = ab  # timing-ablation class (macro GW4E_AB_<ab>): "M" epilogue arithmetic, "S" stores, "L" epilogue LDS traffic + residual pieces


def frag(buf: int, fi: int) -> str:
    b = FR0 + 32 * buf + 4 * fi
    return f"v[{b}:{b + 3}]"


def accblk(i: int, j: int) -> str:
    b = 16 * (4 * j + i)
    return f"a[{b}:{b + 15}]"


def mfma(i: int, j: int, cur: int, init: bool) -> str:
    return f"v_mfma_f32_32x32x16_bf16 {accblk(i, j)}, {frag(cur, i)}, {frag(cur, 4 + j)}, {'0' if init else accblk(i, j)}"


def frag_read(nxt: int, fi: int) -> Op:
    ad, off = ("%[adw]", 4096 * fi) if fi < 4 else ("%[adt]", T_OFF + 4096 * (fi - 4))
    return Op(f"ds_read_b128 {frag(nxt, fi)}, {ad} offset:{off}", "lds")


def dma_m0(q: int) -> Op:
    return Op(f"s_mov_b32 m0, %[m{q}]", "salu", tag=f"m0:{q}")


def dma_ld(q: int, role: str = "") -> Op:
    """role: "F" / "S" = the piece belongs to the first / second operand of the piece order (gemm_w4e.hpp: TF) - emitted with the cache-policy macro
    GW4E_CPOL_<role> behind the instruction (A/B builds: non-temporal hint on one operand's stream; empty in the product)."""
    return Op(f"global_load_lds_dwordx4 %[vo{q}], %[sb{q}]" + (f"@@{role}" if role else ""), "vmem", tag=f"ld:{q}")


# base filler layout of a K step (gap -> ops), as tuned in gemm_w4.hpp: READ: the 8 fragment reads of the next K step (token fragments first: they were
# consumed first); np pieces: s_mov m0 one gap ahead of its load, one piece per ~3 MFMAs
def base_layout(ks: int, read: bool, np_: int) -> list[list[Op]]:
    nxt = (ks & 1) ^ 1
    gaps: list[list[Op]] = [[] for _ in range(16)]
    if read:
        for g, fi in enumerate((4, 5, 6, 7, 0, 1, 2, 3)):
            gaps[g].append(frag_read(nxt, fi))
    if np_ == 5:
        for q, (gm, gl) in enumerate(((0, 1), (3, 4), (6, 7), (9, 10), (12, 13))):
            gaps[gm].append(dma_m0(q))
            gaps[gl].append(dma_ld(q, ("F" if q < 2 else "S") if ks == 0 else "S"))  # K step 0: first operand's rows 6, 7 + second's 0..2; K step 1: second's 3..7
    elif np_ == 6:
        for q, (gm, gl) in enumerate(((0, 1), (3, 4), (5, 6), (8, 9), (11, 12), (13, 14))):
            gaps[gm].append(dma_m0(q))
            gaps[gl].append(dma_ld(q, "F"))  # K step 3: the first operand's rows 0..5 of K tile t + 2
    elif np_ != 0:
        raise ValueError(np_)
    return gaps


def spread(gaps: list[list[Op]], ops: list[Op], first_gap: int = 0) -> None:
    """Places `ops` IN ORDER into the gaps: pinned ops where they say, the others water-filled so that the gap totals are as even as the order allows."""
    for op in ops:
        if op.pin is not None and not op.last:
            gaps[op.pin].append(op)
    free = [op for op in ops if op.pin is None]
    tail = [op for op in ops if op.pin is not None and op.last]
    if not free:
        for op in tail:
            gaps[op.pin].append(op)
        return
    n = len(free)
    load = [len(g) for g in gaps]
    level = 0
    while sum(max(0, level - load[g]) for g in range(first_gap, 16)) < n:
        level += 1
    g = first_gap
    room = max(0, level - load[g])
    for op in free:
        while room == 0:
            g += 1
            room = max(0, level - load[g])
        gaps[g].append(op)
        room -= 1
    for op in tail:
        assert op.pin >= g, "a trailing pinned op would overtake the ops it depends on"
        gaps[op.pin].append(op)


def check_m0_pairs(seq: list[Op]) -> None:
    """An LDS-DMA load must follow ITS s_mov m0 with at least one instruction between and no other m0 write between."""
    last_m0, last_idx = None, -1
    for k, op in enumerate(seq):
        if op.tag.startswith("m0:"):
            last_m0, last_idx = op.tag[3:], k
        elif op.tag.startswith("ld:"):
            assert last_m0 == op.tag[3:], f"LDS-DMA piece {op.tag} issued under m0 of piece {last_m0}"
            assert k - last_idx >= 2, f"piece {op.tag}: no instruction between s_mov m0 and the load"


def check_trans(seq: list[Op]) -> None:
    """The result of a transcendental must not be read by the next instruction (hipcc does not pad inside an asm statement)."""
    for k, op in enumerate(seq[:-1]):
        if op.kind == "trans":
            dst = op.text.split()[1].rstrip(",")
            nxt = seq[k + 1].text
            assert dst not in nxt.split(None, 1)[1] if " " in nxt else True, f"transcendental result {dst} consumed by the next instruction: {nxt}"


def fix_waits(seq: list[Op]) -> None:
    """`s_waitcnt lgkmcnt(@tag)`: wait for the LDS read tagged `tag` = number of LDS instructions issued after it, up to the wait."""
    for k, op in enumerate(seq):
        if op.kind == "wait" and "lgkmcnt(@" in op.text:
            tag = op.text.split("@")[1].rstrip(")")
            src = max(i for i in range(k) if seq[i].tag == tag)
            n = sum(1 for i in range(src + 1, k) if seq[i].kind == "lds")
            op.text = f"s_waitcnt lgkmcnt({n})"


def assemble(ks: int, init: bool, gaps: list[list[Op]], bar: bool, mfmas: bool = True, trailing: int = 0) -> list[str]:
    """trailing: number of VMEM operations at the END of a barrier step (store, residual piece) that the barrier's wait leaves in flight - vmcnt retires in
    order, so `vmcnt(n)` still certifies every LDS-DMA piece issued before them."""
    cur = ks & 1
    seq: list[Op] = [Op("s_waitcnt lgkmcnt(0)", "wait")]
    order = [(i, j) for j in range(4) for i in range(4)]
    for g, (i, j) in enumerate(order):
        if mfmas:
            seq.append(Op(mfma(i, j, cur, init), "mfma"))
        seq.extend(gaps[g])
    if bar:
        seq.append(Op(f"@GW4E_BARWAIT_{trailing}", "wait"))  # macro: vmcnt(trailing [+ 1 with the leader's L2 prefetch load behind them])
        seq.append(Op("s_barrier", "salu"))
    fix_waits(seq)
    check_m0_pairs(seq)
    check_trans(seq)
    return [ab_text(op) for op in seq]


def ab_text(op: Op) -> str:
    return f"#{op.ab}#{op.text}" if op.ab else op.text


def prefetch_ops() -> list[Op]:
    """K step 2 (the barrier step, no LDS-DMA pieces): one plain load per lane that touches the 128-byte line of ONE row of the token (weight) slice of
    K tile t + D - 64 lanes x 4 waves = the 256 rows of the slice - so that the slice sits in the XCD's L2 when its LDS-DMA pieces ask for it. The
    data goes to a register nobody reads (v55). Issued last before the barrier, whose wait leaves exactly these loads in flight (macro GW4E_BARWAIT)."""
    return [Op("@GW4E_PFA", "vmem", pin=15, last=True)]


# ---------------------------------------------------------------------------------------------------------------- epilogue arithmetic of one register
def w_reg(kappa: int) -> str:
    return f"v{P0 + kappa}"


def math_ops(epi: int, kappa: int, flush: bool = False) -> list[Op]:
    ops = _math_ops(epi, kappa, flush)
    for op in ops:
        op.ab = "L" if op.kind == "lds" else ("" if op.kind == "wait" else "M")
    return ops


def _math_ops(epi: int, kappa: int, flush: bool = False) -> list[Op]:
    w = w_reg(kappa)
    if epi == EPI_NONE:
        return []
    if epi == EPI_GELU:
        ops: list[Op] = [Op(f"v_lshlrev_b32 %[x0], 16, {w}"), Op(f"v_and_b32 %[x1], 0xffff0000, {w}")]
        chains = []
        for s in ("0", "1"):
            X, T, P, E = f"%[x{s}]", f"%[t{s}]", f"%[p{s}]", f"%[e{s}]"
            chains.append([
                Op(f"v_fma_f32 {T}, |{X}|, %[c0], 1.0"),
                Op(f"v_rcp_f32 {T}, {T}", "trans"),
                Op(f"v_mov_b32 {P}, {f32_bits(GELU_K[0])}"),
                Op(f"v_fmaak_f32 {P}, {P}, {T}, {f32_bits(GELU_K[1])}"),
                Op(f"v_fmaak_f32 {P}, {P}, {T}, {f32_bits(GELU_K[2])}"),
                Op(f"v_fmaak_f32 {P}, {P}, {T}, {f32_bits(GELU_K[3])}"),
                Op(f"v_fmaak_f32 {P}, {P}, {T}, {f32_bits(GELU_K[4])}"),
                Op(f"v_mul_f32 {E}, {X}, {X}"),
                Op(f"v_mul_f32 {E}, {f32_bits(GELU_E)}, {E}"),
                Op(f"v_exp_f32 {E}, {E}", "trans"),
                Op(f"v_mul_f32 {P}, {P}, {T}"),
                Op(f"v_mul_f32 {P}, {P}, {E}"),
                Op(f"v_mul_f32_e64 {P}, {P}, |{X}|"),
                Op(f"v_max_f32_e64 {X}, {X}, 0"),
                Op(f"v_sub_f32 {X}, {X}, {P}"),
            ])
        for a, b in zip(*chains):
            ops += [a, b]
        ops.append(Op(f"v_cvt_pk_bf16_f32 {w}, %[x0], %[x1]"))
        return ops
    if epi == EPI_GATED:
        j, ch = kappa & 3, kappa >> 2
        gate = f"v{G0 + j}"
        ops = [Op(f"ds_read_b32 %[r], %[yb] offset:{1024 * ch + 4 * j}", "lds", pin=None if flush else 0, tag="resid"),
               Op(f"v_lshlrev_b32 %[x0], 16, {w}"), Op(f"v_and_b32 %[x1], 0xffff0000, {w}"),
               Op(f"v_lshlrev_b32 %[g0], 16, {gate}"), Op(f"v_and_b32 %[g1], 0xffff0000, {gate}"),
               Op("v_mul_f32 %[x0], %[g0], %[x0]"), Op("v_mul_f32 %[x1], %[g1], %[x1]"),
               Op("s_waitcnt lgkmcnt(@resid)", "wait"),
               Op("v_lshlrev_b32 %[g0], 16, %[r]"), Op("v_and_b32 %[g1], 0xffff0000, %[r]"),
               Op("v_add_f32 %[x0], %[g0], %[x0]"), Op("v_add_f32 %[x1], %[g1], %[x1]"),
               Op(f"v_cvt_pk_bf16_f32 {w}, %[x0], %[x1]")]
        return ops
    raise ValueError(epi)


TEMPS = {EPI_NONE: [], EPI_GELU: ["x0", "x1", "t0", "t1", "p0", "p1", "e0", "e1"], EPI_GATED: ["x0", "x1", "g0", "g1", "r"]}


def x_read(ch: int) -> Op:
    b = P0 + 4 * ch
    return Op(f"ds_read_b128 v[{b}:{b + 3}], %[xr] offset:{1024 * ch}", "lds", ab="L")


def store_chunk(ch: int, pin: int | None = 15) -> Op:
    b = P0 + 4 * ch
    return Op(f"global_store_dwordx4 %[coff], v[{b}:{b + 3}], %[cb]", "vmem", pin=pin, last=pin is not None, ab="S")


def resid_piece(pin_m0: int | None, pin_ld: int | None) -> list[Op]:
    return [Op("s_mov_b32 m0, %[ym]", "salu", pin=pin_m0, tag="m0:y", ab="L"), Op("global_load_lds_dwordx4 %[roff], %[rb]", "vmem", pin=pin_ld, tag="ld:y", ab="L")]


def carry_ops(epi: int, kappa: int, store: bool = True) -> tuple[list[Op], int]:
    """Epilogue micro-ops of period K step kappa and the number of trailing VMEM operations its barrier leaves in flight (K step 2 only).
    Stores and residual pieces ride in the BARRIER step (it carries no operand LDS-DMA), as the last VMEM operations in front of the barrier: the
    timing ablation priced a store behind K step 3's six pieces at 3.5-5 % of a launch (profiles/r5_gemm_w4e_ablation.txt)."""
    ops: list[Op] = []
    ks = kappa & 3
    trailing = 0
    if ks == 3:  # transposing read into the registers whose chunk was stored one step earlier: chunk 3 of THIS unit at kappa = 3, chunks 0..2 of the next
        ops.append(Op(x_read({3: 3, 7: 0, 11: 1, 15: 2}[kappa]).text, "lds", pin=10, ab="L"))
    if ks == 2:
        if epi == EPI_GATED:
            ops += resid_piece(10, 11)
            trailing += 1
    ops += math_ops(epi, kappa)
    if ks == 2 and store:
        ops.append(store_chunk({2: 3, 6: 0, 10: 1, 14: 2}[kappa], pin=14))
        trailing += 1
    return ops, trailing


# ---------------------------------------------------------------------------------------------------------------- C++ emission
def operand_lists(texts: list[str], extra_clobber_mem: bool = True):
    """Named operands used by the statement -> (outputs, inputs) constraint strings."""
    import re
    names = []
    for t in texts:
        for m in re.finditer(r"%\[(\w+)\]", t):
            if m.group(1) not in names:
                names.append(m.group(1))
    outs, ins = [], []
    for n in names:
        if n in ("x0", "x1", "t0", "t1", "p0", "p1", "e0", "e1", "g0", "g1", "r", "xt", "xu"):
            outs.append(f'[{n}] "=&v"({n})')
        elif n in ("adw", "adt", "xr", "yb", "coff", "roff", "xw", "pfo", "pfwo"):
            ins.append(f'[{n}] "v"(o.{n})')
        elif n.startswith("vo"):
            ins.append(f'[{n}] "v"(o.vo[{n[2:]}])')
        elif n.startswith("sb"):
            ins.append(f'[{n}] "s"(o.sb[{n[2:]}])')
        elif n[0] == "m" and n[1:].isdigit():
            ins.append(f'[{n}] "s"(o.m[{n[1:]}])')
        elif n in ("cb", "rb", "ym", "c0", "rb1", "ym1", "gb0", "gb1", "goff", "pfb", "pfwb", "pfxa", "pfxw"):
            cons = "v" if n == "goff" else "s"
            ins.append(f'[{n}] "{cons}"(o.{n})')
        else:
            raise KeyError(n)
    return outs, ins


def emit_fn(name: str, texts: list[str], comment: str = "") -> str:
    outs, ins = operand_lists(texts)
    temps = [o.split("(")[1].rstrip(")") for o in outs]
    body = []
    if comment:
        body.append(f"// {comment}")
    body.append(f"G3_DEVICE void {name}(const GW4EOps& o) {{")
    if temps:
        body.append("    uint32_t " + ", ".join(temps) + ";")
    body.append("    asm volatile(")
    for t in texts:
        if t.startswith("@"):
            body.append(f"        {t[1:]}")
        elif t.startswith("#"):
            _, cls, txt = t.split("#", 2)
            body.append(f'        GW4E_AB_{cls}("{txt}\\n\\t")')
        elif "@@" in t:
            txt, role = t.split("@@")
            body.append(f'        "{txt}" GW4E_CPOL_{role} "\\n\\t"')
        else:
            body.append(f'        "{t}\\n\\t"')
    body.append("        : " + ", ".join(outs))
    # the prefetch operands exist only in the A/B build that prefetches (macro): an unused asm input would still make the compiler compute it per K tile
    body.append("        : " + ", ".join(ins) + (" GW4E_PF_OPERANDS" if any(t == "@GW4E_PFA" for t in texts) else ""))
    body.append('        : GW4E_OWNED, "memory");')
    if temps:
        body.append("    " + " ".join(f"(void){t};" for t in temps))
    body.append("}")
    return "\n".join(body) + "\n"


def gen_plain() -> str:
    out = []
    variants = [
        ("gw4e_ks0", 0, False, True, 5, False), ("gw4e_ks0_init", 0, True, True, 5, False), ("gw4e_ks1", 1, False, True, 5, False),
        ("gw4e_ks2_bar", 2, False, True, 0, True), ("gw4e_ks3", 3, False, True, 6, False),
        # tail of the workgroup's LAST output tile: no further K tile / output tile to fetch
        ("gw4e_ks3_nodma", 3, False, True, 0, False), ("gw4e_ks0_nodma", 0, False, True, 0, False), ("gw4e_ks1_nodma", 1, False, True, 0, False),
        ("gw4e_ks2_nobar", 2, False, True, 0, False), ("gw4e_ks3_last", 3, False, False, 0, False),
    ]
    for name, ks, init, read, np_, bar in variants:
        gaps = base_layout(ks, read, np_)
        if bar:
            spread(gaps, prefetch_ops())
        out.append(emit_fn(name, assemble(ks, init, gaps, bar), f"plain K step {ks}" + (" (first of an output tile: C = 0)" if init else "") + (", barrier" if bar else "")))
    # K tile 33 (the first behind the eight periods): its barrier step stores the last unit's chunk 3
    gaps = base_layout(2, True, 0)
    spread(gaps, [store_chunk(3, pin=14)] + prefetch_ops())
    out.append(emit_fn("gw4e_ks2_bar_store3", assemble(2, False, gaps, True, trailing=1), "plain K step 2 + store of W[12:15] (chunk 3 of the last unit), barrier"))
    # preamble K tile of a carried epilogue: k-step 1 also reads unit 0 back from X (all four chunks)
    gaps = base_layout(1, True, 5)
    spread(gaps, [Op(x_read(ch).text, "lds", pin=8 + 2 * ch, ab="L") for ch in range(4)])
    out.append(emit_fn("gw4e_ks1_pre", assemble(1, False, gaps, False), "preamble K tile, K step 1: + transposing reads of unit 0 (chunks 0..3) into W"))
    # gated residual: the finished tile's two gate vectors (feature halves) into v[56:59] / v[60:63], behind K step 3 of K tile nk - 3
    gaps = base_layout(3, True, 6)
    spread(gaps, [Op(f"global_load_dwordx4 v[{G0}:{G0 + 3}], %[goff], %[gb0]", "vmem", pin=2), Op(f"global_load_dwordx4 v[{G0 + 4}:{G0 + 7}], %[goff], %[gb1]", "vmem", pin=7)])
    out.append(emit_fn("gw4e_ks3_gate", assemble(3, False, gaps, False), "K step 3 + the two gate vector loads (certified by the next barrier's vmcnt(0))"))
    return "\n".join(out)


def gen_carry(epi: int) -> str:
    out = []
    for kappa in range(16):
        ks = kappa & 3
        gaps = base_layout(ks, True, (5, 5, 0, 6)[ks])
        ops, trailing = carry_ops(epi, kappa)
        spread(gaps, ops + (prefetch_ops() if ks == 2 else []))
        out.append(emit_fn(f"gw4e_{EPI_NAME[epi]}_k{kappa}", assemble(ks, False, gaps, ks == 2, trailing=trailing),
                           f"{EPI_NAME[epi]}: period K step kappa = {kappa} (K step {ks}" + (", barrier" if ks == 2 else "") + ")"))
    # kappa = 2 of the FIRST period: there is no previous unit whose chunk 3 could be stored (and W[12:15] still holds unit 0's raw chunk 3)
    gaps = base_layout(2, True, 0)
    ops, trailing = carry_ops(epi, 2, store=False)
    spread(gaps, ops + prefetch_ops())
    out.append(emit_fn(f"gw4e_{EPI_NAME[epi]}_k2f", assemble(2, False, gaps, True, trailing=trailing), f"{EPI_NAME[epi]}: period K step kappa = 2 of unit 0 (no store), barrier"))
    # flush of one unit without a K loop under it (the workgroup's last output tile): X -> W, residual pieces -> Y, arithmetic, stores
    # (the residual pieces of the unit are issued from C++ by gw4e_resid_piece before this statement, whose closing wait covers them)
    seq: list[Op] = [Op("s_waitcnt lgkmcnt(0)", "wait")] + [x_read(ch) for ch in range(4)] + [Op("s_waitcnt vmcnt(0) lgkmcnt(0)", "wait")]
    out.append(emit_fn(f"gw4e_{EPI_NAME[epi]}_flush_load", [ab_text(op) for op in seq], f"{EPI_NAME[epi]}: flush, unit registers back from X (and the residual pieces landed)"))
    for ch in range(4):
        seq = []
        for kappa in range(4 * ch, 4 * ch + 4):
            ops = math_ops(epi, kappa, flush=True)
            seq += ops
        seq.append(store_chunk(ch, pin=None))
        fix_waits(seq)
        check_trans(seq)
        out.append(emit_fn(f"gw4e_{EPI_NAME[epi]}_flush_c{ch}", [ab_text(op) for op in seq], f"{EPI_NAME[epi]}: flush, arithmetic + store of chunk {ch}"))
    return "\n".join(out)


def gen_misc() -> str:
    out = []
    # drain: accumulators -> packed bf16 pairs in P. Temporaries: fragment buffer 1 (free behind the last K step; buffer 0 holds the next tile's first fragments)
    texts = ["s_nop 7", "s_nop 3"]
    tmp = [FR0 + 32 + k for k in range(4)]
    n = 0
    for j in range(4):
        for i in range(4):
            for k in range(8):
                a = 16 * (4 * j + i) + 2 * k
                pr = P0 + 32 * j + 8 * i + k
                t = tmp[n & 3]
                n += 1
                texts += [f"v_accvgpr_read_b32 v{pr}, a{a}", f"v_accvgpr_read_b32 v{t}, a{a + 1}", f"v_cvt_pk_bf16_f32 v{pr}, v{pr}, v{t}"]
    body = ["// drain: the 256 accumulators of the finished tile -> 128 packed bf16 registers v[64:191] (feature pairs; the Linear's own rounding to bf16)",
            "G3_DEVICE void gw4e_drain() {", "    asm volatile("]
    body += [f'        "{t}\\n\\t"' for t in texts]
    body += ['        : : : GW4E_OWNED, "memory");', "}"]
    out.append("\n".join(body) + "\n")
    # X write of unit u (MFMA layout -> [32 token rows][64 features] bf16, chunk XOR row & 7): lane (l31, g) holds features 32 il + 8 q + 4 g .. + 3 of row l31
    for u in range(8):
        texts = []
        for il in range(2):
            for q in range(4):
                k = 4 * il + q
                r = P0 + 16 * u + 8 * il + 2 * q
                if k == 0:
                    texts.append(f"#L#ds_write_b64 %[xw], v[{r}:{r + 1}]")
                else:
                    xt = "%[xt]" if k & 1 else "%[xu]"  # two address temporaries in turn: a write's address register is not rewritten straight behind it
                    texts += [f"#L#v_xor_b32 {xt}, {16 * k}, %[xw]", f"#L#ds_write_b64 {xt}, v[{r}:{r + 1}]"]
        out.append(emit_fn(f"gw4e_xwrite_{u}", texts, f"unit {u} (token block {u >> 1}, feature half {u & 1}): registers v[{P0 + 16 * u}:{P0 + 16 * u + 15}] -> LDS slice X"))
    # one residual LDS-DMA piece outside a K step (behind the drain; flush)
    out.append(emit_fn("gw4e_resid_piece", ["#L#s_mov_b32 m0, %[ym]", "s_nop 0", "#L#global_load_lds_dwordx4 %[roff], %[rb]"], "one residual piece -> Y slot (outside a K step)"))
    # gate vectors change halves between two units
    texts = [f"v_swap_b32 v{G0 + k}, v{G0 + 4 + k}" for k in range(4)]
    body = ["// the other feature half's gate vector becomes the current one (v[56:59])", "G3_DEVICE void gw4e_gate_swap() {", "    asm volatile("]
    body += [f'        "{t}\\n\\t"' for t in texts]
    body += ['        : : : GW4E_OWNED);', "}"]
    out.append("\n".join(body) + "\n")
    # gate loads outside a K step (flush of a workgroup whose only... never needed: nk >= 36 guarantees K tile nk - 3 exists) - not generated
    return "\n".join(out)


HEADER = '''// GENERATED by tools/gen_gemm_w4e.py - do not edit; tests/test_gemm_w4e_gen_cpu.py checks that it is up to date.
// Instruction streams of gemm_bf16_nt_w4e_kernel (gemm_w4e.hpp): see the generator for the schedule. Included inside gemm.hip's anonymous namespace.

#define GW4E_P_VGPRS @PLIST@
#define GW4E_OWNED GW4_OWNED, GW4E_P_VGPRS, "v55"

// timing ablations of the deferred epilogue (tools/gemm_ablate_w4e.py; results are garbage): -DG3_AB_GW4E_ABLATE=<bits>  1: no epilogue arithmetic, 2: no stores,
// 4: no epilogue LDS traffic / residual pieces, 8: no drain (gemm_w4e.hpp)
#ifndef G3_AB_GW4E_ABLATE
#define G3_AB_GW4E_ABLATE 0
#endif
#if G3_AB_GW4E_ABLATE & 1
#define GW4E_AB_M(x) ""
#else
#define GW4E_AB_M(x) x
#endif
#if G3_AB_GW4E_ABLATE & 2
#define GW4E_AB_S(x) ""
#else
#define GW4E_AB_S(x) x
#endif
#if G3_AB_GW4E_ABLATE & 4
#define GW4E_AB_L(x) ""
#else
#define GW4E_AB_L(x) x
#endif

// L2 prefetch of the operand slices of K tile t + G3_GW4E_PFD (gemm_w4e.hpp): measured SLOWER and off (profiles/r5_gemm_prefetch_ab.txt: token slices only -1..3 %,
// both -10..15 %: 64 scattered line requests per load on a vector memory path the operand staging already half fills - what round 3 had found for the non-persistent kernel).
// A/B builds: -DG3_AB_GW4E_PF=0 (default) none, 3 token slices by the leader workgroup only (+2 % on MLP-down, +-0 elsewhere; relies on an EXEC = 0 load still counting in vmcnt)
#ifndef G3_AB_GW4E_PF
#define G3_AB_GW4E_PF 0
#endif
#if G3_AB_GW4E_PF == 0
#define GW4E_PF_OPERANDS
#define GW4E_PFA ""
#define GW4E_BARWAIT_0 "s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\t"
#define GW4E_BARWAIT_1 "s_waitcnt vmcnt(1) lgkmcnt(0)\\n\\t"
#define GW4E_BARWAIT_2 "s_waitcnt vmcnt(2) lgkmcnt(0)\\n\\t"
#else  // 3: only the LEADER among the workgroups that share a token slice prefetches it (EXEC = 0 for the others: no request leaves the CU)
#define GW4E_PF_OPERANDS , [pfo] "v"(o.pfo), [pfb] "s"(o.pfb), [pfxa] "s"(o.pfxa)
#define GW4E_PFA "s_mov_b64 exec, %[pfxa]\\n\\tglobal_load_dword v55, %[pfo], %[pfb]\\n\\ts_mov_b64 exec, -1\\n\\t"
#define GW4E_BARWAIT_0 "s_waitcnt vmcnt(1) lgkmcnt(0)\\n\\t"
#define GW4E_BARWAIT_1 "s_waitcnt vmcnt(2) lgkmcnt(0)\\n\\t"
#define GW4E_BARWAIT_2 "s_waitcnt vmcnt(3) lgkmcnt(0)\\n\\t"
#endif

// cache policy of the operand LDS-DMA pieces by role (first / second operand of the piece order): empty in the product. A/B builds (profiles/r6_mlp_down_ab.txt):
// -DG3_AB_GW4E_CPOL=<bits>  1: non-temporal hint on the FIRST operand's pieces (the token stream where N <= 4096), 2: on the second's
#ifndef G3_AB_GW4E_CPOL
#define G3_AB_GW4E_CPOL 0
#endif
#if G3_AB_GW4E_CPOL & 1
#define GW4E_CPOL_F " nt"
#else
#define GW4E_CPOL_F ""
#endif
#if G3_AB_GW4E_CPOL & 2
#define GW4E_CPOL_S " nt"
#else
#define GW4E_CPOL_S ""
#endif

struct GW4EOps {
    uint32_t adw, adt;                          // fragment read addresses of the NEXT K step
    uint32_t m[6]; uint32_t vo[6]; const char* sb[6];  // operand LDS-DMA pieces: LDS destination, per-lane source offset, wave-uniform source base
    uint32_t xw, xr, yb;                        // per-lane LDS addresses: X write (MFMA layout), X transposing read, Y residual word
    uint32_t coff, roff, goff;                  // per-lane byte offsets: output store, residual piece, gate vector
    char* cb;                                   // store base of the chunk
    const char* rb; uint32_t ym;                // residual piece: source base, LDS destination
    const char* gb0; const char* gb1;           // gate vectors of the two feature halves
    float c0;                                   // GELU: 0.3275911 / sqrt(2)
    uint32_t pfo, pfwo; const char* pfb; const char* pfwb;  // L2 prefetch: per-lane row offsets into the token / weight panels, slice bases of K tile t + D
    uint64_t pfxa, pfxw;                        // EXEC masks of the prefetch loads (all ones for the leader workgroup of a slice, else 0)
};

'''


def generate() -> str:
    parts = [HEADER.replace("@PLIST@", ",".join(f'"v{k}"' for k in range(G0, FR0))), gen_misc(), gen_plain()]
    for epi in (EPI_NONE, EPI_GELU, EPI_GATED):
        parts.append(gen_carry(epi))
    return "\n".join(parts)


# ---------------------------------------------------------------------------------------------------------------- layout model (tests)
def layout_model():
    """Follows every accumulator element of a wave quadrant through drain -> X write -> X read -> store and returns {(row, col): (i, j, r, lane)}
    as the generated streams + the addresses of gemm_w4e.hpp compute them; the test checks it is the identity map of the quadrant."""
    xmem = {}
    out = {}
    for u in range(8):
        J, ih = u >> 1, u & 1
        xmem.clear()
        for lane in range(64):
            l31, g = lane & 31, lane >> 5
            xw = l31 * 128 + 8 * g + ((l31 & 7) << 4)
            for il in range(2):
                for q in range(4):
                    k = 4 * il + q
                    addr = xw ^ (16 * k)
                    i = 2 * ih + il
                    # registers P[16 u + 8 il + 2 q], +1: packed pairs k' = 2 q, 2 q + 1 of block (i, J): accumulator registers r = 4 q .. 4 q + 3
                    for e in range(4):
                        r = 4 * q + e
                        feat = 32 * i + 8 * (r >> 2) + 4 * g + (r & 3)
                        xmem[addr + 2 * e] = (feat, 32 * J + l31)
        for lane in range(64):
            rr, c = lane >> 3, lane & 7
            xr = rr * 128 + ((c ^ rr) << 4)
            for ch in range(4):
                for e in range(8):
                    feat, tok = xmem[xr + 1024 * ch + 2 * e]
                    row, col = 32 * J + 8 * ch + rr, 64 * ih + 8 * c + e
                    out[(row, col)] = (tok, feat)
    return out


def main():
    text = generate()
    if "--check" in sys.argv:
        if not OUT.exists() or OUT.read_text() != text:
            print(f"{OUT} is stale: run python tools/gen_gemm_w4e.py")
            sys.exit(1)
        print("up to date")
        return
    OUT.write_text(text)
    print(f"wrote {OUT} ({len(text)} bytes)")


if __name__ == "__main__":
    main()
