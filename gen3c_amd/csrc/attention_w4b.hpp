// Included by attention.hip after attention_w4.hpp, inside the same anonymous namespace.
//
// w4b: the one-wave-per-SIMD self-attention kernel (see attention_w4.hpp for the why) with the issue stream of a tile trimmed to what the
// SIMD can hide under its MFMAs (MI355X_MICROARCH.md: <= 5 single-issue instructions per 32-cycle MFMA gap at one wave per SIMD; the w4
// stream carried 5.7 per gap plus a 69-instruction tile head with the matrix pipe idle - timing ablations in profiles/r2_attn_ablation.txt:
// +7..10 % without the row-max head, +3..10 % without the in-stream LDS-DMA address arithmetic, +16 % without both).
//   * K / V^T operand fragments live in an asm-owned ACCUMULATOR-register ring a[192:255] (16 slots; ds_read_b128 writes AGPRs directly and
//     the MFMA takes srcA from there): 16 architectural VGPRs back, no compiler-visible register is ever the target of an in-flight LDS
//     read, and the read-ahead distance W4B_D is free to choose.
//   * the row-max chains of the NEXT tile's scores ride in the P.V region of this tile (two v_max3 per step, the lane^32 exchange in the last
//     two steps): the tile head is the rescale test only.
//   * LDS-DMA pieces are two instructions inside the step statement (s_mov m0 in front of the step's first MFMA, the load behind it): the
//     per-lane source offsets are eight loop-invariant VGPRs, the tile's K / V^T bases wave-uniform SGPR pairs advanced by SALU.
//   * one pair unit per step (four at the tile head, behind the first fragment reads whose latency they cover), so no step carries two.
//   * hot statements carry NO clobber list (hipcc pads every boundary between two asm statements that share a register - clobbers included -
//     with an s_nop) and alternate their temporaries / chains / accumulators by step parity; the statements outside the loop keep the lists,
//     so the AGPR file stays reserved, and tools/asm_audit.py verifies in the generated code that no compiler instruction touches it.
// Template flag XB (the default kernel, variant 11): the tile barrier moves into the P.V region (in front of step 10) and steps 10..15 read the
// NEXT tile's first six K fragments across it, so a tile starts with its operands already in the ring; the four head units become half units
// on the odd steps of region A and the LDS-DMA pieces sit on the even steps (one per 4 MFMAs); with a 1-D grid every XCD works through its
// own (batch, head) pairs. PMC: 39.9 (XB) vs 41.1 cycles per 32-cycle MFMA.
// Limits (checked by the launcher, which otherwise runs w4): S_kv a multiple of 64.

#define W4B_RING_AGPRS "a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
#define W4B_OWNED W4_OWNED_AGPRS, W4B_RING_AGPRS
constexpr int W4B_RING0 = 192;  // fragment n of a tile (K fragments 0..15, V^T fragments 16..31) sits in a[192 + 4 (n & 15) : +3]
constexpr int W4B_D = 6;        // fragment n is read W4B_D steps before the step that consumes it

#if G3_AB_ATTN_ABLATE & 32
#define W4B_READ ""
#else
#define W4B_READ "ds_read_b128 a[%c[ra]:%c[rb]], %[addr] offset:%c[off]\n\t"
#endif
#define W4B_QK(h) "v_mfma_f32_32x32x16_bf16 %[s" #h "], a[%c[fa]:%c[fb]], a[%c[q" #h "]:%c[e" #h "]], %[s" #h "]\n\t"
#define W4B_QK0(h) "v_mfma_f32_32x32x16_bf16 %[s" #h "], a[%c[fa]:%c[fb]], a[%c[q" #h "]:%c[e" #h "]], %[c" #h "]\n\t"
#define W4B_PV(h) "v_mfma_f32_32x32x16_bf16 a[%c[o" #h "]:%c[g" #h "]], a[%c[fa]:%c[fb]], %[pf" #h "], a[%c[o" #h "]:%c[g" #h "]]\n\t"
#if G3_AB_ATTN_ABLATE & 128
#define W4B_M0 ""
#define W4B_DMA ""
#else
#define W4B_M0 "s_mov_b32 m0, %[m0v]\n\t"
#define W4B_DMA "global_load_lds_dwordx4 %[voff], %[sbase]\n\t"
#endif
// row-max pair k of a half: one v_max3 on the chain of key block 0, one on the chain of key block 1 (chains start from 0: only the excess over
// the running maximum, to which the scores are relative, matters)
#if G3_AB_ATTN_ABLATE & 256
#define W4B_MAXP(k) ""
#define W4B_MAXP0(k) "v_mov_b32 %[mx], 0\n\tv_mov_b32 %[my], 0\n\t"
#else
#define W4B_MAXP(k) "v_max3_f32 %[mx], %[mx], %[ma" #k "], %[mb" #k "]\n\tv_max3_f32 %[my], %[my], %[mc" #k "], %[md" #k "]\n\t"
#define W4B_MAXP0(k) "v_max3_f32 %[mx], 0, %[ma" #k "], %[mb" #k "]\n\tv_max3_f32 %[my], 0, %[mc" #k "], %[md" #k "]\n\t"
#endif

// fragment read outside a step (tile head)
// the first W4B_D fragments of a tile (slots 0..D-1), one statement: fragments 2 i, 2 i + 1 share the address register ad<i> (K: the two key
// blocks of k-step i; V^T: output blocks 2 i, 2 i + 1 of slice 0 - then ad0 == ad1 == ad2)
template <int O0, int O1, int O2, int O3, int O4, int O5> G3_DEVICE void w4b_read_head(uint32_t ad0, uint32_t ad1, uint32_t ad2) {
    static_assert(W4B_D == 6, "w4b_read_head is laid out for a read-ahead of 6");
    if (!(G3_AB_ATTN_ABLATE & 32))
        asm volatile("ds_read_b128 a[192:195], %0 offset:%c3\n\tds_read_b128 a[196:199], %0 offset:%c4\n\tds_read_b128 a[200:203], %1 offset:%c5\n\t"
                     "ds_read_b128 a[204:207], %1 offset:%c6\n\tds_read_b128 a[208:211], %2 offset:%c7\n\tds_read_b128 a[212:215], %2 offset:%c8"
                     ::"v"(ad0), "v"(ad1), "v"(ad2), "n"(O0), "n"(O1), "n"(O2), "n"(O3), "n"(O4), "n"(O5) : W4B_OWNED);
}

// ---- region A step I: [m0] [read] wait ; S0 (+)= K.Q0 ; pair unit ; [LDS-DMA piece] ; S1 (+)= K.Q1
// EXTRA (steps without INIT / DMA only): half of a second pair unit behind the step - 1: exp2, exp2, first row-sum add; 2 (two steps later, so
// that no statement reads what its predecessor wrote): second add + cvt_pk
#define W4B_XA "v_exp_f32 %[a2], %[x2]\n\tv_exp_f32 %[b2], %[y2]\n\tv_add_f32 %[pe], %[pe], %[a2]\n\t"
#define W4B_XB "v_add_f32 %[pe], %[pe], %[b2]\n\tv_cvt_pk_bf16_f32 %[k2], %[a2], %[b2]\n\t"
template <int I, int ROFF, bool READ, int WN, bool INIT, bool DMA, int EXTRA = 0>
G3_DEVICE void w4b_step_qk(uint32_t addr, f32x16& s0, f32x16& s1, const f32x16& c0, const f32x16& c1, float x1, float y1, uint32_t& k1, float& p1, float& r1,
                           uint32_t m0v, uint32_t voff, const char* sbase, float& a1, float& b1, float x2 = 0.f, float y2 = 0.f, uint32_t* k2 = nullptr,
                           float* pe = nullptr, float* a2 = nullptr, float* b2 = nullptr) {
    constexpr int ks = I >> 1;
    constexpr int q0 = W4_QBASE + 4 * ks, q1 = W4_QBASE + 4 * (8 + ks);
    constexpr int fa = W4B_RING0 + 4 * (I & 15), ra = W4B_RING0 + 4 * ((I + W4B_D) & 15);
#define W4B_A_OUT [a1] "+v"(a1), [b1] "+v"(b1), [k1] "=&v"(k1), [p1] "+v"(p1), [r1] "+v"(r1)
#define W4B_A_IN [addr] "v"(addr), [x1] "v"(x1), [y1] "v"(y1), [off] "n"(ROFF), [wn] "n"(WN), [q0] "n"(q0), [e0] "n"(q0 + 3), [q1] "n"(q1), [e1] "n"(q1 + 3), \
                 [fa] "n"(fa), [fb] "n"(fa + 3), [ra] "n"(ra), [rb] "n"(ra + 3)
    static_assert(READ, "region A always reads ahead");
    if constexpr (INIT && DMA)
        asm volatile(W4B_M0 W4B_READ W4_WAIT W4B_QK0(0) W4_UNIT(1) W4B_DMA W4B_QK0(1)
                     : W4B_A_OUT, [s0] "=&v"(s0), [s1] "=&v"(s1) : W4B_A_IN, [c0] "v"(c0), [c1] "v"(c1), [m0v] "s"(m0v), [voff] "v"(voff), [sbase] "s"(sbase)
                     : "memory");
    else if constexpr (!INIT && DMA)
        asm volatile(W4B_M0 W4B_READ W4_WAIT W4B_QK(0) W4_UNIT(1) W4B_DMA W4B_QK(1)
                     : W4B_A_OUT, [s0] "+v"(s0), [s1] "+v"(s1) : W4B_A_IN, [m0v] "s"(m0v), [voff] "v"(voff), [sbase] "s"(sbase) : "memory");
    else if constexpr (INIT && !DMA && EXTRA == 1)
        asm volatile(W4B_READ W4_WAIT W4B_QK0(0) W4_UNIT(1) W4B_QK0(1) W4B_XA
                     : W4B_A_OUT, [s0] "=&v"(s0), [s1] "=&v"(s1), [a2] "=&v"(*a2), [b2] "=&v"(*b2), [pe] "+v"(*pe)
                     : W4B_A_IN, [c0] "v"(c0), [c1] "v"(c1), [x2] "v"(x2), [y2] "v"(y2));
    else if constexpr (INIT && !DMA)
        asm volatile(W4B_READ W4_WAIT W4B_QK0(0) W4_UNIT(1) W4B_QK0(1)
                     : W4B_A_OUT, [s0] "=&v"(s0), [s1] "=&v"(s1) : W4B_A_IN, [c0] "v"(c0), [c1] "v"(c1));
    else if constexpr (EXTRA == 1)
        asm volatile(W4B_READ W4_WAIT W4B_QK(0) W4_UNIT(1) W4B_QK(1) W4B_XA
                     : W4B_A_OUT, [s0] "+v"(s0), [s1] "+v"(s1), [a2] "=&v"(*a2), [b2] "=&v"(*b2), [pe] "+v"(*pe) : W4B_A_IN, [x2] "v"(x2), [y2] "v"(y2));
    else if constexpr (EXTRA == 2)
        asm volatile(W4B_READ W4_WAIT W4B_QK(0) W4_UNIT(1) W4B_QK(1) W4B_XB
                     : W4B_A_OUT, [s0] "+v"(s0), [s1] "+v"(s1), [k2] "=&v"(*k2), [pe] "+v"(*pe) : W4B_A_IN, [a2] "v"(*a2), [b2] "v"(*b2));
    else
        asm volatile(W4B_READ W4_WAIT W4B_QK(0) W4_UNIT(1) W4B_QK(1) : W4B_A_OUT, [s0] "+v"(s0), [s1] "+v"(s1) : W4B_A_IN);
    static_assert(EXTRA == 0 || (!DMA && !(INIT && EXTRA == 2)), "w4b_step_qk: extra half units only on steps without an LDS-DMA piece");
#undef W4B_A_OUT
#undef W4B_A_IN
}

// ---- region B step I: [read] wait ; O(h0,d) += V^T.P0 ; [pair unit] ; O(h1,d) += V^T.P1 ; [row-max work of the next tile's scores]
// MX: 0 none, 1 first pair of a half (chains start), 2 one pair, 3 three pairs (one behind the first MFMA, two behind the second)
template <int I, int ROFF, bool READ, int WN, bool UNIT, int MX>
G3_DEVICE void w4b_step_pv(uint32_t addr, const u32x4& pf0, const u32x4& pf1, float x1, float y1, uint32_t& k1, float& p1, float& r1, float& mx, float& my,
                           float ma0, float mb0, float mc0, float md0, float ma1, float mb1, float mc1, float md1, float ma2, float mb2, float mc2, float md2,
                           float& a1, float& b1) {
    constexpr int D = I & 3;
    constexpr int o0 = 16 * D, o1 = 16 * (4 + D);
    constexpr int fa = W4B_RING0 + 4 * ((16 + I) & 15), ra = W4B_RING0 + 4 * ((16 + I + W4B_D) & 15);
#define W4B_B_IN [pf0] "v"(pf0), [pf1] "v"(pf1), [wn] "n"(WN), [o0] "n"(o0), [g0] "n"(o0 + 15), [o1] "n"(o1), [g1] "n"(o1 + 15), [fa] "n"(fa), [fb] "n"(fa + 3)
#define W4B_B_RD [addr] "v"(addr), [off] "n"(ROFF), [ra] "n"(ra), [rb] "n"(ra + 3)
#define W4B_B_UO [a1] "+v"(a1), [b1] "+v"(b1), [k1] "=&v"(k1), [p1] "+v"(p1), [r1] "+v"(r1)
#define W4B_B_UI [x1] "v"(x1), [y1] "v"(y1)
#define W4B_B_M0 [ma0] "v"(ma0), [mb0] "v"(mb0), [mc0] "v"(mc0), [md0] "v"(md0)
    if constexpr (READ && UNIT && MX == 0)
        asm volatile(W4B_READ W4_WAIT W4B_PV(0) W4_UNIT(1) W4B_PV(1) : W4B_B_UO : W4B_B_IN, W4B_B_RD, W4B_B_UI);
    else if constexpr (READ && UNIT && MX == 1)
        asm volatile(W4B_READ W4_WAIT W4B_PV(0) W4_UNIT(1) W4B_PV(1) W4B_MAXP0(0) : W4B_B_UO, [mx] "=&v"(mx), [my] "=&v"(my) : W4B_B_IN, W4B_B_RD, W4B_B_UI, W4B_B_M0);
    else if constexpr (READ && UNIT && MX == 2)
        asm volatile(W4B_READ W4_WAIT W4B_PV(0) W4_UNIT(1) W4B_PV(1) W4B_MAXP(0) : W4B_B_UO, [mx] "+v"(mx), [my] "+v"(my) : W4B_B_IN, W4B_B_RD, W4B_B_UI, W4B_B_M0);
    else if constexpr (!READ && UNIT && MX == 1)
        asm volatile(W4_WAIT W4B_PV(0) W4_UNIT(1) W4B_PV(1) W4B_MAXP0(0) : W4B_B_UO, [mx] "=&v"(mx), [my] "=&v"(my) : W4B_B_IN, W4B_B_UI, W4B_B_M0);
    else if constexpr (!READ && UNIT && MX == 2)
        asm volatile(W4_WAIT W4B_PV(0) W4_UNIT(1) W4B_PV(1) W4B_MAXP(0) : W4B_B_UO, [mx] "+v"(mx), [my] "+v"(my) : W4B_B_IN, W4B_B_UI, W4B_B_M0);
    else if constexpr (!READ && UNIT && MX == 0)
        asm volatile(W4_WAIT W4B_PV(0) W4_UNIT(1) W4B_PV(1) : W4B_B_UO : W4B_B_IN, W4B_B_UI);
    else if constexpr (!READ && !UNIT && MX == 3)
        asm volatile(W4_WAIT W4B_PV(0) W4B_MAXP(0) W4B_PV(1) W4B_MAXP(1) W4B_MAXP(2)
                     : [mx] "+v"(mx), [my] "+v"(my)
                     : W4B_B_IN, W4B_B_M0, [ma1] "v"(ma1), [mb1] "v"(mb1), [mc1] "v"(mc1), [md1] "v"(md1), [ma2] "v"(ma2), [mb2] "v"(mb2), [mc2] "v"(mc2), [md2] "v"(md2));
    else if constexpr (READ && !UNIT && MX == 3)
        asm volatile(W4B_READ W4_WAIT W4B_PV(0) W4B_MAXP(0) W4B_PV(1) W4B_MAXP(1) W4B_MAXP(2)
                     : [mx] "+v"(mx), [my] "+v"(my)
                     : W4B_B_IN, W4B_B_RD, W4B_B_M0, [ma1] "v"(ma1), [mb1] "v"(mb1), [mc1] "v"(mc1), [md1] "v"(md1), [ma2] "v"(ma2), [mb2] "v"(mb2), [mc2] "v"(mc2), [md2] "v"(md2));
    else if constexpr (READ && !UNIT && MX == 0)
        asm volatile(W4B_READ W4_WAIT W4B_PV(0) W4B_PV(1) : : W4B_B_IN, W4B_B_RD);
    else {
        static_assert(!READ && !UNIT && MX == 0, "w4b_step_pv: combination not laid out");
        asm volatile(W4_WAIT W4B_PV(0) W4B_PV(1) : : W4B_B_IN);
    }
#undef W4B_B_RD
#undef W4B_B_UO
#undef W4B_B_UI
#undef W4B_B_M0
}
// region B steps 14 / 15: the four chains of each half are folded and exchanged with lane ^ 32
// step 14: wait ; PV0 ; a_h = max3(a_h, b_h, c_h), a_h = max(a_h, d_h) (h = 0, 1) ; PV1 ; t_h = a_h
// step 15: wait ; PV0 ; swap upper half of t_h with lower half of a_h ; PV1 ; a_h = max(a_h, t_h)
template <int I, int WN, bool READ = false, int ROFF = 0>
G3_DEVICE void w4b_step_pv_fold(const u32x4& pf0, const u32x4& pf1, float& a0, float& b0, float& c0, float& d0, float& a1, float& b1, float& c1, float& d1, float& t0,
                                 float& t1, uint32_t addr = 0u) {
    constexpr int D = I & 3;
    constexpr int o0 = 16 * D, o1 = 16 * (4 + D);
    constexpr int fa = W4B_RING0 + 4 * ((16 + I) & 15), ra = W4B_RING0 + 4 * ((16 + I + W4B_D) & 15);
    if constexpr (READ && I == 14)
        asm volatile(W4B_READ W4_WAIT W4B_PV(0) "v_max3_f32 %[a0], %[a0], %[b0], %[c0]\n\tv_max3_f32 %[a1], %[a1], %[b1], %[c1]\n\tv_max_f32 %[a0], %[a0], %[d0]\n\tv_max_f32 %[a1], %[a1], %[d1]\n\t"
                     W4B_PV(1) "v_mov_b32 %[t0], %[a0]\n\tv_mov_b32 %[t1], %[a1]\n\t"
                     : [a0] "+v"(a0), [a1] "+v"(a1), [t0] "=&v"(t0), [t1] "=&v"(t1)
                     : W4B_B_IN, [b0] "v"(b0), [c0] "v"(c0), [d0] "v"(d0), [b1] "v"(b1), [c1] "v"(c1), [d1] "v"(d1), [addr] "v"(addr), [off] "n"(ROFF), [ra] "n"(ra), [rb] "n"(ra + 3));
    else if constexpr (READ)
        asm volatile(W4B_READ W4_WAIT W4B_PV(0) "v_permlane32_swap_b32 %[t0], %[a0]\n\tv_permlane32_swap_b32 %[t1], %[a1]\n\t" W4B_PV(1)
                     "v_max_f32 %[a0], %[a0], %[t0]\n\tv_max_f32 %[a1], %[a1], %[t1]\n\t"
                     : [a0] "+v"(a0), [a1] "+v"(a1), [t0] "+v"(t0), [t1] "+v"(t1) : W4B_B_IN, [addr] "v"(addr), [off] "n"(ROFF), [ra] "n"(ra), [rb] "n"(ra + 3));
    else if constexpr (I == 14)
        asm volatile(W4_WAIT W4B_PV(0) "v_max3_f32 %[a0], %[a0], %[b0], %[c0]\n\tv_max3_f32 %[a1], %[a1], %[b1], %[c1]\n\tv_max_f32 %[a0], %[a0], %[d0]\n\tv_max_f32 %[a1], %[a1], %[d1]\n\t"
                     W4B_PV(1) "v_mov_b32 %[t0], %[a0]\n\tv_mov_b32 %[t1], %[a1]\n\t"
                     : [a0] "+v"(a0), [a1] "+v"(a1), [t0] "=&v"(t0), [t1] "=&v"(t1)
                     : W4B_B_IN, [b0] "v"(b0), [c0] "v"(c0), [d0] "v"(d0), [b1] "v"(b1), [c1] "v"(c1), [d1] "v"(d1));
    else
        asm volatile(W4_WAIT W4B_PV(0) "v_permlane32_swap_b32 %[t0], %[a0]\n\tv_permlane32_swap_b32 %[t1], %[a1]\n\t" W4B_PV(1)
                     "v_max_f32 %[a0], %[a0], %[t0]\n\tv_max_f32 %[a1], %[a1], %[t1]\n\t"
                     : [a0] "+v"(a0), [a1] "+v"(a1), [t0] "+v"(t0), [t1] "+v"(t1) : W4B_B_IN);
}
#undef W4B_B_IN

G3_DEVICE void w4b_fence_acc() { asm volatile("s_nop 7\n\ts_nop 3" ::: W4B_OWNED); }
// in-place updates of the rescale branch: as plain C++ the branch's results get registers of their own and the common path pays the copies
template <int R> G3_DEVICE void w4b_sub_inplace(f32x16& x, float d) {
    float t = x[R];
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(t) : "v"(d));
    x[R] = t;
}
template <int R> G3_DEVICE void w4b_set_inplace(f32x16& x, float v) {
    float t = x[R];
    asm volatile("v_mov_b32 %0, %1" : "+v"(t) : "v"(v));
    x[R] = t;
}
G3_DEVICE void w4b_mul_inplace(float& x, float a) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a)); }

// XB: the tile barrier sits in the P.V region (before step 10) and steps 10..15 read the NEXT tile's first six K fragments across it, so a tile
// starts with its operands in the ring; the four pair units that covered the head reads move into region A as half units.
template <bool XB>
__global__ __launch_bounds__(W4_THREADS, 1) void flash_attn_fwd_w4b_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);  // [2][64][128]
    bf16_t* sV = sK + 2 * KVB * HD;                     // [2][128][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;
    int qblk = blockIdx.x, head = blockIdx.y, batch = blockIdx.z;
    if (p.xcd_heads) {  // 1-D grid: XCD x (= workgroup index % 8) works through the (batch, head) pairs x, x + 8, x + 16, ...
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int hb = xcd + 8 * (slot / p.grid_q);
        qblk = slot - (slot / p.grid_q) * p.grid_q;
        head = hb % p.n_heads;
        batch = hb / p.n_heads;
    }

    const bf16_t* Qb = p.Q + batch * p.q_batch + head * p.q_head;
    const bf16_t* Kb = p.K + batch * p.k_batch + head * p.k_head;
    const bf16_t* Vb = p.Vt + batch * p.vt_batch + head * p.vt_head;
    bf16_t* Ob = p.O + batch * p.o_batch + head * p.o_head;
    // split-KV output pointers: fetched from the kernel arguments HERE (opaque to the optimiser), not by a scalar load hipcc would otherwise
    // sink to the end of the tile loop, in front of the tail tiles whose lgkmcnt waits are hand-counted (tools/asm_audit.py)
    float* o32_base = p.O32;
    float* lse_base = p.LSE;
    asm volatile("" : "+s"(o32_base), "+s"(lse_base));

    // ---- Q fragments of both halves, pre-multiplied by scale * log2(e), into a[128:191]; O accumulators a[0:127] = 0 (as w4)
    static_for<0, 16>([&](auto fc) {
        constexpr int f = decltype(fc)::value, h = f >> 3, ks = f & 7;
        const int q_idx = qblk * W4_BQ + wave * 64 + 32 * h + l31;
        const bool q_ok = q_idx < p.Sq;
        bf16x8 qv = q_ok ? load_bf16x8(Qb + (int64_t)q_idx * p.q_row + 8 * g + 16 * ks) : zero_bf16x8();
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = f32_to_bf16((float)qv[e] * p.scale_log2);
        const u32x4 qw = __builtin_bit_cast(u32x4, qv);
        w4_acc_write<W4_QBASE + 4 * f + 0>(qw[0]);
        w4_acc_write<W4_QBASE + 4 * f + 1>(qw[1]);
        w4_acc_write<W4_QBASE + 4 * f + 2>(qw[2]);
        w4_acc_write<W4_QBASE + 4 * f + 3>(qw[3]);
    });
    static_for<0, 128>([&](auto rc) { w4_acc_zero<decltype(rc)::value>(); });

    // ---- LDS-DMA staging (slot layout as w4). The eight per-lane source byte offsets are loop invariants held in VGPRs; the tile's position
    // is carried by the wave-uniform base (SGPR pair) of the instruction.
    const int k_row0 = tid >> 4, k_src_chunk = (tid & 15) ^ (k_row0 & 15);
    const int v_row0 = tid >> 3, v_src_chunk = (tid & 7) ^ ((v_row0 >> 1) & 7);
    const char* Kbytes = reinterpret_cast<const char*>(Kb);
    const char* Vbytes = reinterpret_cast<const char*>(Vb);
    const uint32_t k_row_bytes = (uint32_t)p.k_row * 2u;
    const uint32_t v_row_bytes = (uint32_t)p.vt_row * 2u;
    uint32_t dma_off[8];  // [0..3] K piece j (rows k_row0 + 16 j of the tile), [4..7] V^T piece j (rows v_row0 + 32 j)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dma_off[j] = ((uint32_t)k_row0 + 16u * j) * k_row_bytes + (uint32_t)k_src_chunk * 16u;
        dma_off[4 + j] = ((uint32_t)v_row0 + 32u * j) * v_row_bytes + (uint32_t)v_src_chunk * 16u;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(dma_off[j]));  // opaque: keep them resident instead of re-deriving them per piece
    const uint32_t k_tile_bytes = (uint32_t)KVB * k_row_bytes;
    const uint32_t seg_len = (uint32_t)p.vt_seg_len, seg_bytes = (uint32_t)p.vt_seg_stride * 2u;
    auto v_tile_off = [&](uint32_t kv0) -> uint32_t {  // byte offset of key kv0 inside a V^T row (a 64-key tile never straddles segments)
        if (!seg_len) return kv0 * 2u;
        const uint32_t sg = kv0 / seg_len;
        return sg * seg_bytes + (kv0 - sg * seg_len) * 2u;
    };
    auto dma_tile_builtin = [&](const char* base, const uint32_t* off, bf16_t* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[i]),
                                             (__attribute__((address_space(3))) void*)(dst + wave * 64 * 8 + 256 * 8 * i), 16, 0, 0);
    };

    // ---- per-lane LDS byte addresses of the operand fragments inside slot 0 of each ring (same fragment mapping as v3 / w4)
    const int krow_perm = swap23(l31);
    const uint32_t lds_k0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)sK;
    const uint32_t lds_v0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)sV;
    uint32_t kaddr[8], vaddr[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kaddr[ks] = lds_k0 + 2u * (uint32_t)k_off(krow_perm, 2 * ks + g);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) vaddr[s4] = lds_v0 + 2u * (uint32_t)v_off(l31, 2 * s4 + g);

    float m_run[2], mx_cur[2];
    // row sums: PERSISTENT partial accumulators [half][step parity x lane of the pair] - consecutive step statements never name the same one
    float psum[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 4; ++k) psum[h][k] = 0.f;
    float ta[2] = {0.f, 0.f}, tb[2] = {0.f, 0.f};  // exp2 results of a pair unit, one set per step parity (see the note on statement boundaries)
    float pe[2][2] = {{0.f, 0.f}, {0.f, 0.f}};     // XB: row-sum accumulators of the half units (half 0; [step parity][lane of the pair])
    float xa[2], xb[2];                            // XB: exp2 results of a half unit between its two steps
    const int nt = p.Skv / KVB;                     // launcher: S_kv % 64 == 0

    // ---- prologue: K(0), V(0) (and K(1)) by LDS-DMA; scores of tile 0 with C = 0, then made relative to their exact row maximum
    dma_tile_builtin(Kbytes, dma_off, sK);
    dma_tile_builtin(Vbytes, dma_off + 4, sV);
    if (nt > 1) dma_tile_builtin(Kbytes + k_tile_bytes, dma_off, sK + KVB * HD);
    G3_JITTER(wave, blockIdx.x + 5);
    lds_dma_publish_barrier();
    G3_JITTER(wave + 2, blockIdx.x);
    f32x16 SA[2][2], SB[2][2];  // [half][32-key block]
    f32x16 negm[2];             // -m_run of the half in every element: C operand of the first QK^T MFMA of a block
    {
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        static_for<0, 16>([&](auto ic) {
            constexpr int i = decltype(ic)::value, mb = i >> 3, ks = i & 7;
            const bf16x8 kf = load_bf16x8(sK + 32 * mb * HD + k_off(krow_perm, 2 * ks + g));  // compiler-managed read + wait (prologue only)
            if constexpr (ks == 0) {
                w4_qk0<0, -1>(SA[0][mb], kf, zero);
                w4_qk0<8, -1>(SA[1][mb], kf, zero);
            } else {
                w4_qk<ks, -1>(SA[0][mb], kf);
                w4_qk<8 + ks, -1>(SA[1][mb], kf);
            }
        });
    }
    w4_fence_v(SA[0][0], SA[0][1], SA[1][0], SA[1][1]);
    __syncthreads();  // K(0)'s slot is the destination of the first LDS-DMA of the tile loop (K(2)): every wave must be done reading it
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float ma = max3(SA[h][0][0], SA[h][0][1], SA[h][0][2]);
        float mb2 = max3(SA[h][1][0], SA[h][1][1], SA[h][1][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) {
            ma = max3(ma, SA[h][0][r], SA[h][0][r + 1]);
            mb2 = max3(mb2, SA[h][1][r], SA[h][1][r + 1]);
        }
        m_run[h] = xor32_max(max3(ma, mb2, max3(SA[h][0][15], SA[h][1][15], SA[h][1][15])));
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) SA[h][mb][r] -= m_run[h];
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[h][r] = -m_run[h];
        mx_cur[h] = 0.f;
    }
    if (XB && nt > 1) {  // the first six K(1) fragments (slot 1 of the K ring, published by the prologue barrier): tile 0 starts with them in the ring
        constexpr int KS1 = KVB * HD * 2;
        w4b_read_head<KS1, KS1 + 32 * HD * 2, KS1, KS1 + 32 * HD * 2, KS1, KS1 + 32 * HD * 2>(kaddr[0], kaddr[1], kaddr[2]);
    }
    // wave-uniform source state of the in-stream LDS-DMA: V^T tile t+1 (byte offset inside a row, tiles left in its segment)
    uint32_t v_off_next = v_tile_off(KVB);
    const uint32_t seg_tiles = seg_len ? seg_len / KVB : 0x7fffffffu;
    uint32_t v_seg_pos = seg_len ? (1u % seg_tiles) : 1u;  // position of tile 1 inside its segment
    const uint32_t v_seg_jump = seg_len ? seg_bytes - seg_len * 2u : 0u;

    auto tile = [&](f32x16 (&S_cur)[2][2], f32x16 (&S_next)[2][2], int t, auto has_next_c, auto par_c) {
        constexpr bool has_next = decltype(has_next_c)::value;
        constexpr int par = decltype(par_c)::value;
        constexpr int D = W4B_D;
        constexpr int KS = (par ^ 1) * KVB * HD * 2;  // LDS byte offset of K(t+1)'s slot
        constexpr int VS = par * HD * KVB * 2;         // of V^T(t)'s slot
        // byte offset (immediate) and per-lane address register of fragment n of this tile
        auto frag_off = [](auto nc) constexpr { constexpr int n = decltype(nc)::value; return n < 16 ? KS + 32 * (n & 1) * HD * 2 : VS + 32 * ((n - 16) & 3) * KVB * 2; };
        auto frag_addr = [&](auto nc) -> uint32_t { constexpr int n = decltype(nc)::value; if constexpr (n < 16) return kaddr[(n >> 1) & 7]; else return vaddr[((n - 16) >> 2) & 3]; };
        G3_JITTER(wave + blockIdx.x, t);  // race screen only
        // ---- tile head: the first D fragment reads; their latency is covered by the rescale test and four pair units
        if constexpr (!(XB && has_next)) {
            constexpr int n0 = has_next ? 0 : 16;
            using N0 = std::integral_constant<int, n0>;
            using N1 = std::integral_constant<int, n0 + 1>;
            using N2 = std::integral_constant<int, n0 + 2>;
            using N3 = std::integral_constant<int, n0 + 3>;
            using N4 = std::integral_constant<int, n0 + 4>;
            using N5 = std::integral_constant<int, n0 + 5>;
            w4b_read_head<frag_off(N0{}), frag_off(N1{}), frag_off(N2{}), frag_off(N3{}), frag_off(N4{}), frag_off(N5{})>(frag_addr(N0{}), frag_addr(N2{}), frag_addr(N4{}));
        }
        if (__any(fmaxf(mx_cur[0], mx_cur[1]) > RESCALE_THR)) {  // rare: some row's maximum grew by more than 2^THR since its last rescale
            float alpha[2], delta[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                delta[h] = mx_cur[h];  // >= 0 (the chains start from 0)
                alpha[h] = __builtin_amdgcn_exp2f(-delta[h]);
                m_run[h] += delta[h];
            }
            // 12 wait states: O (MFMA results in AGPRs) -> v_accvgpr_read; it also separates the two v_exp above from the asm statements that
            // read alpha (hipcc does not pad a transcendental result consumed INSIDE an inline-asm statement: measured as a stale alpha; tools/asm_audit.py
            // looks for the pattern) - hence tied to them
            asm volatile("s_nop 7\n\ts_nop 3" : "+v"(alpha[0]), "+v"(alpha[1]) : : W4B_OWNED);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 4; ++k) w4b_mul_inplace(psum[h][k], alpha[h]);
                if (XB && h == 0) {
                    w4b_mul_inplace(pe[0][0], alpha[0]);
                    w4b_mul_inplace(pe[0][1], alpha[0]);
                    w4b_mul_inplace(pe[1][0], alpha[0]);
                    w4b_mul_inplace(pe[1][1], alpha[0]);
                }
                const float nm = -m_run[h], dl = delta[h];
                static_for<0, 16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    w4b_sub_inplace<r>(S_cur[h][0], dl);
                    w4b_sub_inplace<r>(S_cur[h][1], dl);
                    w4b_set_inplace<r>(negm[h], nm);
                });
            }
            static_for<0, 64>([&](auto rc) { w4_acc_scale<decltype(rc)::value>(alpha[0]); });
            static_for<64, 128>([&](auto rc) { w4_acc_scale<decltype(rc)::value>(alpha[1]); });
        }
        u32x4 pb[2][4];  // P fragments (bf16 pairs) of P.V slice sl, half h
        // pair unit u = 0..31 in consumption order: slice sl = u >> 3 (P.V steps 4 sl .. 4 sl + 3 need it), half h = (u >> 2) & 1, pair q = u & 3
        auto ux = [&](auto uc) -> float { constexpr int u = decltype(uc)::value, sl = u >> 3, h = (u >> 2) & 1, q = u & 3; return S_cur[h][sl >> 1][(sl & 1) * 8 + 2 * q]; };
        auto uy = [&](auto uc) -> float { constexpr int u = decltype(uc)::value, sl = u >> 3, h = (u >> 2) & 1, q = u & 3; return S_cur[h][sl >> 1][(sl & 1) * 8 + 2 * q + 1]; };
        auto uput = [&](auto uc, uint32_t pk) { constexpr int u = decltype(uc)::value, sl = u >> 3, h = (u >> 2) & 1, q = u & 3; pb[h][sl][q] = pk; };
        auto sm_unit = [&](auto uc) {  // a pair unit outside a step
            constexpr int u = decltype(uc)::value, h = (u >> 2) & 1, pa = 2 * (u & 1);
            float a, b;
            uint32_t pk;
            if (G3_AB_ATTN_ABLATE & 16) {
                pk = 0;
            } else {
                asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %3, %3, %0\n\tv_add_f32 %4, %4, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1"
                             : "=&v"(a), "=&v"(b), "=v"(pk), "+v"(psum[h][pa]), "+v"(psum[h][pa + 1]) : "v"(ux(uc)), "v"(uy(uc)));
            }
            uput(uc, pk);
        };
        constexpr int NHEAD = has_next ? (XB ? 0 : 4) : 20;  // pair units at the tile head (the last tile has no region A to carry units 4..19)
        static_for<0, NHEAD>([&](auto uc) { sm_unit(uc); });

        // ---- region A: S_next[h] = K(t+1).Q_h^T, step I: key block mb = I & 1, k-step ks = I >> 1 (alternating blocks), pair unit 4 + I,
        //      LDS-DMA piece I in steps 0..7: K(t+2) pieces 0..3 -> slot of K(t), V^T(t+1) pieces 0..3 -> slot of V^T(t-1)
        if constexpr (has_next) {
            const int t2 = min(t + 2, nt - 1);  // past the end: re-read the last tile (its slot is not consumed any more)
            const char* kbase = Kbytes + (uint32_t)t2 * k_tile_bytes;
            const char* vbase = Vbytes + v_off_next;
            static_for<0, 16>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                using NR = std::integral_constant<int, I + D>;
                using U = std::integral_constant<int, (XB ? 0 : 4) + I>;
                constexpr int hu = (U::value >> 2) & 1, pa = 2 * (I & 1);
                // LDS-DMA pieces: steps 0..7 (one per 2 MFMAs), or with XB every second step (one per 4 MFMAs: four waves then ask the vector
                // memory path for 32 instead of 64 B/clk/CU, its peak - gemm_w4.hpp measured what the bursts cost)
                constexpr bool dma = XB ? (I & 1) == 0 : I < 8;
                constexpr int j = XB ? (I >> 1) : (I & 7);
                const uint32_t m0v = dma ? (j < 4 ? lds_k0 + (uint32_t)(par * KVB * HD * 2 + 256 * j * 16) : lds_v0 + (uint32_t)((par ^ 1) * HD * KVB * 2 + 256 * (j - 4) * 16)) + (uint32_t)wave * 1024u : 0u;
                uint32_t k1 = 0;
                if constexpr (XB && (I & 1)) {
                    // half units of pair units 16..19 (all half 0) on the odd steps: first half of unit 16 + x in step 4 x + 1, second half in 4 x + 3
                    constexpr int ex = (I & 2) ? 2 : 1;
                    constexpr int xu = 16 + (I >> 2);
                    using XU = std::integral_constant<int, xu>;
                    uint32_t k2 = 0;
                    w4b_step_qk<I, frag_off(NR{}), true, D, (I >> 1) == 0, false, ex>(frag_addr(NR{}), S_next[0][I & 1], S_next[1][I & 1], negm[0], negm[1], ux(U{}), uy(U{}), k1,
                                                                                      psum[hu][pa], psum[hu][pa + 1], 0u, 0u, nullptr, ta[I & 1], tb[I & 1], ux(XU{}), uy(XU{}), &k2,
                                                                                      &pe[0][ex - 1], &xa[0], &xb[0]);
                    if constexpr (ex == 2) uput(XU{}, k2);
                } else {
                    w4b_step_qk<I, frag_off(NR{}), true, D, (I >> 1) == 0, dma>(frag_addr(NR{}), S_next[0][I & 1], S_next[1][I & 1], negm[0], negm[1], ux(U{}), uy(U{}), k1,
                                                                               psum[hu][pa], psum[hu][pa + 1], m0v, dma_off[j], j < 4 ? kbase : vbase, ta[I & 1], tb[I & 1]);
                }
                uput(U{}, k1);
            });
            // V^T source of the next tile's pieces
            const bool wrap = (v_seg_pos + 1u == seg_tiles);
            v_off_next += wrap ? (uint32_t)(KVB * 2) + v_seg_jump : (uint32_t)(KVB * 2);
            v_seg_pos = wrap ? 0u : v_seg_pos + 1u;
        }
        G3_JITTER(wave + blockIdx.x + 3, t);
        // ---- region B: O_h^T += V^T(t).P_h^T, step I: slice s = I >> 2, output block d = I & 3; pair units 20 + I in steps 0..11; the row-max
        //      chains of S_next: half 0 pairs 0..7 in steps 2..9, half 1 pairs 0, 1 in steps 10, 11 and 2..7 in steps 12, 13; fold in 14, 15
        {
            float cx[2][2], cy[2][2], tx[2];  // chains [half][step parity] of key block 0 / 1, exchange temporaries
            auto sn = [&](auto hc, auto mbc, auto rc) -> float { return S_next[decltype(hc)::value][decltype(mbc)::value][decltype(rc)::value]; };
            // XB: fragment n >= 32 is fragment n - 32 of the NEXT tile: K(t+2), published by the barrier in front of step 10, slot `par` of the K ring
            constexpr int KSN = par * KVB * HD * 2;
            auto rd_off = [](auto nc) constexpr { constexpr int n = decltype(nc)::value; return n < 32 ? VS + 32 * ((n - 16) & 3) * KVB * 2 : KSN + 32 * (n & 1) * HD * 2; };
            auto rd_addr = [&](auto nc) -> uint32_t { constexpr int n = decltype(nc)::value; if constexpr (n < 32) return vaddr[((n - 16) >> 2) & 3]; else return kaddr[((n - 32) >> 1) & 7]; };
            static_for<0, 14>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                constexpr bool rd = (I + D < 16) || (XB && has_next);
                using NR = std::integral_constant<int, rd ? 16 + I + D : 16>;
                constexpr bool unit = I < 12;
                using U = std::integral_constant<int, unit ? 20 + I : 20>;
                constexpr int hu = (U::value >> 2) & 1, pa = 2 * (I & 1);
                // row-max work: half 0 pairs 0..7 in steps 2..9 (chain set = step parity, so pairs 0, 1 start the chains), half 1 pairs 0, 1 in
                // steps 10, 11 (start) and three more per chain set in steps 12, 13
                constexpr int mxk = !has_next ? 0 : (I < 2 ? 0 : (I == 2 || I == 3 || I == 10 || I == 11) ? 1 : I < 12 ? 2 : 3);
                constexpr int mh = I < 10 ? 0 : 1;
                constexpr int k0 = I < 2 ? 0 : I < 10 ? I - 2 : I < 12 ? I - 10 : I == 12 ? 2 : 5;  // first pair of the step
                constexpr int k1i = mxk == 3 ? k0 + 1 : k0, k2i = mxk == 3 ? k0 + 2 : k0;
                using H = std::integral_constant<int, mh>;
#define W4B_SN(mb, r) (has_next ? sn(H{}, std::integral_constant<int, mb>{}, std::integral_constant<int, (r)>{}) : 0.f)
                if constexpr (XB && has_next && I == 10) {
                    // every wave has issued its last read of V^T(t) (step 9) and of K(t+1) (region A): drain this wave's LDS-DMA and publish
                    // K(t+2) / V^T(t+1). No operands: the statement shares no register with its neighbours (no pad).
                    if (!(G3_AB_ATTN_ABLATE & 64)) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                }
                uint32_t k1 = 0;
                w4b_step_pv<I, rd_off(NR{}), rd, rd ? D : 15 - I, unit, mxk>(rd_addr(NR{}), pb[0][I >> 2], pb[1][I >> 2], ux(U{}), uy(U{}), k1, psum[hu][pa], psum[hu][pa + 1],
                                                                          cx[mh][I & 1], cy[mh][I & 1], W4B_SN(0, 2 * k0), W4B_SN(0, 2 * k0 + 1), W4B_SN(1, 2 * k0),
                                                                          W4B_SN(1, 2 * k0 + 1), W4B_SN(0, 2 * k1i), W4B_SN(0, 2 * k1i + 1), W4B_SN(1, 2 * k1i),
                                                                          W4B_SN(1, 2 * k1i + 1), W4B_SN(0, 2 * k2i), W4B_SN(0, 2 * k2i + 1), W4B_SN(1, 2 * k2i),
                                                                          W4B_SN(1, 2 * k2i + 1), ta[I & 1], tb[I & 1]);
#undef W4B_SN
                if constexpr (unit) uput(U{}, k1);
            });
            if constexpr (has_next) {
                if constexpr (XB) {
                    using N14 = std::integral_constant<int, 16 + 14 + D>;
                    using N15 = std::integral_constant<int, 16 + 15 + D>;
                    w4b_step_pv_fold<14, D, true, rd_off(N14{})>(pb[0][3], pb[1][3], cx[0][0], cy[0][0], cx[0][1], cy[0][1], cx[1][0], cy[1][0], cx[1][1], cy[1][1], tx[0], tx[1],
                                                                 rd_addr(N14{}));
                    w4b_step_pv_fold<15, D, true, rd_off(N15{})>(pb[0][3], pb[1][3], cx[0][0], cy[0][0], cx[0][1], cy[0][1], cx[1][0], cy[1][0], cx[1][1], cy[1][1], tx[0], tx[1],
                                                                 rd_addr(N15{}));
                } else {
                    w4b_step_pv_fold<14, 1>(pb[0][3], pb[1][3], cx[0][0], cy[0][0], cx[0][1], cy[0][1], cx[1][0], cy[1][0], cx[1][1], cy[1][1], tx[0], tx[1]);
                    w4b_step_pv_fold<15, 0>(pb[0][3], pb[1][3], cx[0][0], cy[0][0], cx[0][1], cy[0][1], cx[1][0], cy[1][0], cx[1][1], cy[1][1], tx[0], tx[1]);
                }
                mx_cur[0] = cx[0][0];
                mx_cur[1] = cx[1][0];
            } else {
                float d0 = 0.f, d1 = 0.f;
                uint32_t kd = 0;
                w4b_step_pv<14, 0, false, 1, false, 0>(0u, pb[0][3], pb[1][3], 0.f, 0.f, kd, d0, d1, d0, d1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, d0, d1);
                w4b_step_pv<15, 0, false, 0, false, 0>(0u, pb[0][3], pb[1][3], 0.f, 0.f, kd, d0, d1, d0, d1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, d0, d1);
            }
        }
        if (!XB && has_next && !(G3_AB_ATTN_ABLATE & 64)) lds_dma_publish_barrier();  // drains the LDS-DMA (vmcnt(0)) and publishes K(t+2) / V(t+1)
    };

    using True = std::integral_constant<bool, true>;
    using False = std::integral_constant<bool, false>;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 2 < nt; t += 2) {
        tile(SA, SB, t, True{}, P0{});
        tile(SB, SA, t + 1, True{}, P1{});
    }
    if (t + 1 < nt) {
        tile(SA, SB, t, True{}, P0{});
        tile(SB, SA, t + 1, False{}, P1{});
    } else {
        tile(SA, SB, t, False{}, P0{});
    }

    w4b_fence_acc();
    static_for<0, 2>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        const float extra = (XB && h == 0) ? (pe[0][0] + pe[0][1]) + (pe[1][0] + pe[1][1]) : 0.f;
        const float inv = 1.0f / xor32_sum(((psum[h][0] + psum[h][1]) + (psum[h][2] + psum[h][3])) + extra);
        const int q_idx = qblk * W4_BQ + wave * 64 + 32 * h + l31;
        if (o32_base) {  // split-KV part (see AttnParams): fp32 normalised partial + log-sum-exp in the log2 domain
            float* prow = o32_base + (int64_t)batch * p.o_batch + (int64_t)head * p.o_head + (int64_t)q_idx * p.o_row;
            static_for<0, 16>([&](auto cc) {
                constexpr int d = decltype(cc)::value >> 2, q4 = decltype(cc)::value & 3;
                constexpr int R = 16 * (4 * h + d) + 4 * q4;
                f32x4 o;
                o[0] = w4_acc_read<R + 0>() * inv;
                o[1] = w4_acc_read<R + 1>() * inv;
                o[2] = w4_acc_read<R + 2>() * inv;
                o[3] = w4_acc_read<R + 3>() * inv;
                if (q_idx < p.Sq) *reinterpret_cast<f32x4*>(prow + 32 * d + 8 * q4 + 4 * g) = o;
            });
            if (g == 0 && q_idx < p.Sq) lse_base[((int64_t)batch * p.n_heads + head) * p.Sq + q_idx] = m_run[h] - __builtin_amdgcn_logf(inv);
            return;
        }
        bf16_t* orow = Ob + (int64_t)q_idx * p.o_row;
        static_for<0, 16>([&](auto cc) {  // (d, q4): 4 consecutive output dims per store
            constexpr int d = decltype(cc)::value >> 2, q4 = decltype(cc)::value & 3;
            constexpr int R = 16 * (4 * h + d) + 4 * q4;
            bf16x4 o;
            o[0] = f32_to_bf16(w4_acc_read<R + 0>() * inv);
            o[1] = f32_to_bf16(w4_acc_read<R + 1>() * inv);
            o[2] = f32_to_bf16(w4_acc_read<R + 2>() * inv);
            o[3] = f32_to_bf16(w4_acc_read<R + 3>() * inv);
            if (q_idx < p.Sq) *reinterpret_cast<bf16x4*>(orow + 32 * d + 8 * q4 + 4 * g) = o;
        });
    });
}
