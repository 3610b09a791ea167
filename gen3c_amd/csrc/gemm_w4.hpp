// Included by gemm.hip inside its anonymous namespace (shares GemmParams, lds_off, store_tile_lds, the XCD-aware tile order).
//
// gemm_bf16_nt_w4_kernel: the block GEMM with ONE wave per SIMD. Workgroup = 4 waves = one 256 (token) x 256 (feature) tile, each wave a
// 128 x 128 quadrant: 16 accumulator blocks of 32 x 32 = all 256 AGPRs, every operand fragment feeds FOUR MFMAs (the 8-wave ping-pong kernel:
// two) - half the LDS operand traffic per flop, which is what the kernel pays for at the 1 400 W cap. Nothing hides a wave's own LDS-DMA issue
// cost or LDS latency at one wave per SIMD, so the instruction stream of a K tile is laid out by hand, as in csrc/attention_w4b.hpp:
//   * asm-owned registers, by literal name: a[0:255] accumulators (block (i, j): features 32 i.., tokens 32 j.., at a[16 (4 j + i) : +15]),
//     v[192:255] two buffers of 8 operand fragments (W_0..3, T_0..3 of one 16-wide k-step);
//   * a K tile (64) = 4 k-step statements of 16 MFMAs; the fragments of k-step s + 1 are read while k-step s multiplies (s = 3: the first
//     k-step of the NEXT tile, from the other LDS stage), so no LDS latency is exposed;
//   * 2 LDS stages of 64 KiB ([256 feature rows][64 k] + [256 token rows][64 k], chunk index XOR-swizzled as in the other kernels). ONE barrier
//     per K tile, after k-step 2, behind `s_waitcnt vmcnt(0) lgkmcnt(0)`: it publishes tile t + 1 (LDS-DMA issued from k-step 3 of tile t - 1
//     through k-step 1 of tile t) and certifies that every wave holds the k-step-3 fragments of tile t in registers, which frees tile t's stage:
//     its refill with tile t + 2 starts in k-step 3 and continues through k-steps 0 and 1 of tile t + 1 (6 + 5 + 5 pieces, one per ~3 MFMAs:
//     a rate the vector memory path absorbs; k-step 2, the one in front of the barrier, issues none);
//   * an LDS-DMA piece is `s_mov m0` one MFMA gap ahead + `global_load_lds_dwordx4` behind an MFMA; its 16 per-lane source offsets are
//     loop-invariant VGPRs (rows clamped to the matrix, so M / N tails read valid memory and are never stored), the K position is the
//     wave-uniform base.
// Needs K % 64 == 0, K >= 128 and the full-line epilogue's alignment (host: launch()); everything else runs on the ping-pong kernel.

#define GW4_ACC_AGPRS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
#define GW4_FRAG_VGPRS "v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255"
#define GW4_OWNED GW4_ACC_AGPRS, GW4_FRAG_VGPRS

constexpr int GW4_THREADS = 256;
constexpr int GW4_FRAG0 = 192;               // buffer b, fragment f (0..3 weights, 4..7 tokens) at v[192 + 32 b + 4 f : +3]
constexpr int GW4_STAGE_BYTES = 65536;       // [W tile 32 KiB][T tile 32 KiB]
constexpr int GW4_T_OFF = 32768;

template <int R> G3_DEVICE void gw4_acc_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"n"(R) : GW4_OWNED); }
template <int R> G3_DEVICE float gw4_acc_read() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "n"(R) : GW4_ACC_AGPRS);
    return v;
}

// timing ablations (tools/gemm_ablate_w4.py; results are garbage): -DG3_AB_GW4_ABLATE=<bits>  1: the barrier does not wait for the LDS-DMA,
// 2: no barrier, 4: no LDS-DMA pieces, 8: no fragment reads, 32: no epilogue
#ifndef G3_AB_GW4_ABLATE
#define G3_AB_GW4_ABLATE 0
#endif
// fragment read: buffer B, fragment F <- LDS [stage-and-k-step address + row-block offset]
#if G3_AB_GW4_ABLATE & 8
#define GW4_RD(B, F) ""
#else
#define GW4_RD(B, F) "ds_read_b128 v[%c[r" #B #F "]:%c[e" #B #F "]], %[ad" #F "] offset:%c[o" #F "]\n\t"
#endif
// MFMA (i, j) on buffer B:  a[16 (4 j + i)] += W_i . T_j^T
#define GW4_MM(I, J) "v_mfma_f32_32x32x16_bf16 a[%c[d" #I #J "]:%c[z" #I #J "]], v[%c[w" #I "]:%c[x" #I "]], v[%c[t" #J "]:%c[u" #J "]], a[%c[d" #I #J "]:%c[z" #I #J "]]\n\t"
// LDS-DMA piece Q of a statement: its LDS destination goes into M0 one gap earlier (GW4_M0), the load itself sits behind the next MFMA (GW4_LD)
#if G3_AB_GW4_ABLATE & 4
#define GW4_M0(Q) ""
#define GW4_LD(Q) ""
#else
#define GW4_M0(Q) "s_mov_b32 m0, %[m" #Q "]\n\t"
#define GW4_LD(Q) "global_load_lds_dwordx4 %[vo" #Q "], %[sb" #Q "]\n\t"
#endif

struct GW4Pieces {  // up to 6 pieces per k-step statement: LDS destination, per-lane source byte offset, wave-uniform source base
    uint32_t m[6];
    uint32_t vo[6];
    const char* sb[6];
    uint64_t va[6];  // implicit-GEMM convolution (gemm_w4_conv.hpp): token pieces carry a full per-lane address instead of base + offset
};
// LDS-DMA piece whose source is a per-lane 64-bit address (gathered activation rows of a convolution tap, or the zero page)
#if G3_AB_GW4_ABLATE & 4
#define GW4_LDV(Q) ""
#else
#define GW4_LDV(Q) "global_load_lds_dwordx4 %[va" #Q "], off\n\t"
#endif

// One k-step: wait for this k-step's fragments (buffer KS & 1), 16 MFMAs; between them the 8 fragment reads of the next k-step (other buffer)
// when READ, and NP (0, 5 or 6) LDS-DMA pieces spread evenly over the MFMA gaps: four waves issuing one 1 KiB piece per ~3 MFMAs ask the
// vector memory path for ~43 B/clk/CU (its peak is 64) - issued back to back (8 per k-step, round 2's first form) they queue up and the
// issue stall lands on the only wave that can feed the matrix pipe.
// BAR: behind the MFMAs wait for all LDS reads and LDS-DMA of this wave and join the workgroup barrier.
// PT (NP == 5 only): 0 = every piece base + offset; 1 = pieces 0, 1 base + offset (weights), 2..4 per-lane addresses (conv token rows); 2 = all five
// per-lane addresses.
template <int KS, bool READ, int NP, bool BAR, int PT = 0>
G3_DEVICE void gw4_kstep(uint32_t adw, uint32_t adt, const GW4Pieces& pc) {
    constexpr int cur = GW4_FRAG0 + 32 * (KS & 1), nxt = GW4_FRAG0 + 32 * ((KS & 1) ^ 1);
#define GW4_OPS_MM(I, J) [d##I##J] "n"(16 * (4 * J + I)), [z##I##J] "n"(16 * (4 * J + I) + 15)
#define GW4_OPS_ALLMM GW4_OPS_MM(0, 0), GW4_OPS_MM(1, 0), GW4_OPS_MM(2, 0), GW4_OPS_MM(3, 0), GW4_OPS_MM(0, 1), GW4_OPS_MM(1, 1), GW4_OPS_MM(2, 1), GW4_OPS_MM(3, 1), \
                      GW4_OPS_MM(0, 2), GW4_OPS_MM(1, 2), GW4_OPS_MM(2, 2), GW4_OPS_MM(3, 2), GW4_OPS_MM(0, 3), GW4_OPS_MM(1, 3), GW4_OPS_MM(2, 3), GW4_OPS_MM(3, 3)
#define GW4_OPS_FR [w0] "n"(cur), [x0] "n"(cur + 3), [w1] "n"(cur + 4), [x1] "n"(cur + 7), [w2] "n"(cur + 8), [x2] "n"(cur + 11), [w3] "n"(cur + 12), [x3] "n"(cur + 15), \
                   [t0] "n"(cur + 16), [u0] "n"(cur + 19), [t1] "n"(cur + 20), [u1] "n"(cur + 23), [t2] "n"(cur + 24), [u2] "n"(cur + 27), [t3] "n"(cur + 28), [u3] "n"(cur + 31)
// reads: fragment F of the next buffer; F = 0..3 weight row blocks (address adw, + 4 KiB each), 4..7 token row blocks (adt, T tile)
#define GW4_OPS_RD [rN0] "n"(nxt), [eN0] "n"(nxt + 3), [rN1] "n"(nxt + 4), [eN1] "n"(nxt + 7), [rN2] "n"(nxt + 8), [eN2] "n"(nxt + 11), [rN3] "n"(nxt + 12), [eN3] "n"(nxt + 15), \
                   [rN4] "n"(nxt + 16), [eN4] "n"(nxt + 19), [rN5] "n"(nxt + 20), [eN5] "n"(nxt + 23), [rN6] "n"(nxt + 24), [eN6] "n"(nxt + 27), [rN7] "n"(nxt + 28), [eN7] "n"(nxt + 31), \
                   [ad0] "v"(adw), [ad1] "v"(adw), [ad2] "v"(adw), [ad3] "v"(adw), [ad4] "v"(adt), [ad5] "v"(adt), [ad6] "v"(adt), [ad7] "v"(adt), \
                   [o0] "n"(0), [o1] "n"(4096), [o2] "n"(8192), [o3] "n"(12288), [o4] "n"(GW4_T_OFF), [o5] "n"(GW4_T_OFF + 4096), [o6] "n"(GW4_T_OFF + 8192), \
                   [o7] "n"(GW4_T_OFF + 12288)
#define GW4_OPS_P(Q) [m##Q] "s"(pc.m[Q]), [vo##Q] "v"(pc.vo[Q]), [sb##Q] "s"(pc.sb[Q])
#define GW4_OPS_PV(Q) [m##Q] "s"(pc.m[Q]), [va##Q] "v"(pc.va[Q])
#if G3_AB_GW4_ABLATE & 2
#define GW4_BARRIER "s_waitcnt lgkmcnt(0)\n\t"
#elif G3_AB_GW4_ABLATE & 1
#define GW4_BARRIER "s_waitcnt lgkmcnt(0)\n\ts_barrier\n\t"
#else
#define GW4_BARRIER "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\t"
#endif
    // token fragments of the previous k-step were last read by its MFMAs (j-major order: T_0 first), so they are overwritten first
    if constexpr (READ && NP == 6 && !BAR)
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     GW4_MM(0, 0) GW4_RD(N, 4) GW4_M0(0) GW4_MM(1, 0) GW4_RD(N, 5) GW4_LD(0) GW4_MM(2, 0) GW4_RD(N, 6) GW4_MM(3, 0) GW4_RD(N, 7) GW4_M0(1)
                     GW4_MM(0, 1) GW4_RD(N, 0) GW4_LD(1) GW4_MM(1, 1) GW4_RD(N, 1) GW4_M0(2) GW4_MM(2, 1) GW4_RD(N, 2) GW4_LD(2) GW4_MM(3, 1) GW4_RD(N, 3)
                     GW4_MM(0, 2) GW4_M0(3) GW4_MM(1, 2) GW4_LD(3) GW4_MM(2, 2) GW4_MM(3, 2) GW4_M0(4)
                     GW4_MM(0, 3) GW4_LD(4) GW4_MM(1, 3) GW4_M0(5) GW4_MM(2, 3) GW4_LD(5) GW4_MM(3, 3)
                     : : GW4_OPS_ALLMM, GW4_OPS_FR, GW4_OPS_RD, GW4_OPS_P(0), GW4_OPS_P(1), GW4_OPS_P(2), GW4_OPS_P(3), GW4_OPS_P(4), GW4_OPS_P(5) : GW4_OWNED, "memory");
    else if constexpr (READ && NP == 5 && !BAR && PT == 0)
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     GW4_MM(0, 0) GW4_RD(N, 4) GW4_M0(0) GW4_MM(1, 0) GW4_RD(N, 5) GW4_LD(0) GW4_MM(2, 0) GW4_RD(N, 6) GW4_MM(3, 0) GW4_RD(N, 7) GW4_M0(1)
                     GW4_MM(0, 1) GW4_RD(N, 0) GW4_LD(1) GW4_MM(1, 1) GW4_RD(N, 1) GW4_MM(2, 1) GW4_RD(N, 2) GW4_M0(2) GW4_MM(3, 1) GW4_RD(N, 3) GW4_LD(2)
                     GW4_MM(0, 2) GW4_MM(1, 2) GW4_M0(3) GW4_MM(2, 2) GW4_LD(3) GW4_MM(3, 2)
                     GW4_MM(0, 3) GW4_M0(4) GW4_MM(1, 3) GW4_LD(4) GW4_MM(2, 3) GW4_MM(3, 3)
                     : : GW4_OPS_ALLMM, GW4_OPS_FR, GW4_OPS_RD, GW4_OPS_P(0), GW4_OPS_P(1), GW4_OPS_P(2), GW4_OPS_P(3), GW4_OPS_P(4) : GW4_OWNED, "memory");
    else if constexpr (READ && NP == 5 && !BAR && PT == 1)  // same stream, pieces 2..4 gathered by per-lane address
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     GW4_MM(0, 0) GW4_RD(N, 4) GW4_M0(0) GW4_MM(1, 0) GW4_RD(N, 5) GW4_LD(0) GW4_MM(2, 0) GW4_RD(N, 6) GW4_MM(3, 0) GW4_RD(N, 7) GW4_M0(1)
                     GW4_MM(0, 1) GW4_RD(N, 0) GW4_LD(1) GW4_MM(1, 1) GW4_RD(N, 1) GW4_MM(2, 1) GW4_RD(N, 2) GW4_M0(2) GW4_MM(3, 1) GW4_RD(N, 3) GW4_LDV(2)
                     GW4_MM(0, 2) GW4_MM(1, 2) GW4_M0(3) GW4_MM(2, 2) GW4_LDV(3) GW4_MM(3, 2)
                     GW4_MM(0, 3) GW4_M0(4) GW4_MM(1, 3) GW4_LDV(4) GW4_MM(2, 3) GW4_MM(3, 3)
                     : : GW4_OPS_ALLMM, GW4_OPS_FR, GW4_OPS_RD, GW4_OPS_P(0), GW4_OPS_P(1), GW4_OPS_PV(2), GW4_OPS_PV(3), GW4_OPS_PV(4) : GW4_OWNED, "memory");
    else if constexpr (READ && NP == 5 && !BAR && PT == 2)  // all five gathered by per-lane address
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     GW4_MM(0, 0) GW4_RD(N, 4) GW4_M0(0) GW4_MM(1, 0) GW4_RD(N, 5) GW4_LDV(0) GW4_MM(2, 0) GW4_RD(N, 6) GW4_MM(3, 0) GW4_RD(N, 7) GW4_M0(1)
                     GW4_MM(0, 1) GW4_RD(N, 0) GW4_LDV(1) GW4_MM(1, 1) GW4_RD(N, 1) GW4_MM(2, 1) GW4_RD(N, 2) GW4_M0(2) GW4_MM(3, 1) GW4_RD(N, 3) GW4_LDV(2)
                     GW4_MM(0, 2) GW4_MM(1, 2) GW4_M0(3) GW4_MM(2, 2) GW4_LDV(3) GW4_MM(3, 2)
                     GW4_MM(0, 3) GW4_M0(4) GW4_MM(1, 3) GW4_LDV(4) GW4_MM(2, 3) GW4_MM(3, 3)
                     : : GW4_OPS_ALLMM, GW4_OPS_FR, GW4_OPS_RD, GW4_OPS_PV(0), GW4_OPS_PV(1), GW4_OPS_PV(2), GW4_OPS_PV(3), GW4_OPS_PV(4) : GW4_OWNED, "memory");
    else if constexpr (READ && NP == 0 && !BAR)
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     GW4_MM(0, 0) GW4_RD(N, 4) GW4_MM(1, 0) GW4_RD(N, 5) GW4_MM(2, 0) GW4_RD(N, 6) GW4_MM(3, 0) GW4_RD(N, 7)
                     GW4_MM(0, 1) GW4_RD(N, 0) GW4_MM(1, 1) GW4_RD(N, 1) GW4_MM(2, 1) GW4_RD(N, 2) GW4_MM(3, 1) GW4_RD(N, 3)
                     GW4_MM(0, 2) GW4_MM(1, 2) GW4_MM(2, 2) GW4_MM(3, 2) GW4_MM(0, 3) GW4_MM(1, 3) GW4_MM(2, 3) GW4_MM(3, 3)
                     : : GW4_OPS_ALLMM, GW4_OPS_FR, GW4_OPS_RD : GW4_OWNED, "memory");
    else if constexpr (READ && NP == 0 && BAR)
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     GW4_MM(0, 0) GW4_RD(N, 4) GW4_MM(1, 0) GW4_RD(N, 5) GW4_MM(2, 0) GW4_RD(N, 6) GW4_MM(3, 0) GW4_RD(N, 7)
                     GW4_MM(0, 1) GW4_RD(N, 0) GW4_MM(1, 1) GW4_RD(N, 1) GW4_MM(2, 1) GW4_RD(N, 2) GW4_MM(3, 1) GW4_RD(N, 3)
                     GW4_MM(0, 2) GW4_MM(1, 2) GW4_MM(2, 2) GW4_MM(3, 2) GW4_MM(0, 3) GW4_MM(1, 3) GW4_MM(2, 3) GW4_MM(3, 3) GW4_BARRIER
                     : : GW4_OPS_ALLMM, GW4_OPS_FR, GW4_OPS_RD : GW4_OWNED, "memory");
    else {
        static_assert(!READ && NP == 0 && !BAR, "gw4_kstep: combination not laid out");
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                     GW4_MM(0, 0) GW4_MM(1, 0) GW4_MM(2, 0) GW4_MM(3, 0) GW4_MM(0, 1) GW4_MM(1, 1) GW4_MM(2, 1) GW4_MM(3, 1)
                     GW4_MM(0, 2) GW4_MM(1, 2) GW4_MM(2, 2) GW4_MM(3, 2) GW4_MM(0, 3) GW4_MM(1, 3) GW4_MM(2, 3) GW4_MM(3, 3)
                     : : GW4_OPS_ALLMM, GW4_OPS_FR : GW4_OWNED, "memory");
    }
}

// K step 2 (the barrier step: no LDS-DMA pieces) of the implicit-GEMM convolution, which also moves the eight per-lane token addresses on to the next 64-channel
// tile (+ 128 bytes) in the MFMA gaps behind the fragment reads: as compiler code between two statements the eight 64-bit adds were exposed once per K tile.
G3_DEVICE void gw4_kstep2_bar_advance(uint32_t adw, uint32_t adt, uint64_t (&ta)[8], uint64_t step) {
    constexpr int KS = 2;
    constexpr int cur = GW4_FRAG0 + 32 * (KS & 1), nxt = GW4_FRAG0 + 32 * ((KS & 1) ^ 1);
#define GW4_ADV(Q) "v_lshl_add_u64 %[ta" #Q "], %[ta" #Q "], 0, %[stp]\n\t"
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 GW4_MM(0, 0) GW4_RD(N, 4) GW4_MM(1, 0) GW4_RD(N, 5) GW4_MM(2, 0) GW4_RD(N, 6) GW4_MM(3, 0) GW4_RD(N, 7)
                 GW4_MM(0, 1) GW4_RD(N, 0) GW4_MM(1, 1) GW4_RD(N, 1) GW4_MM(2, 1) GW4_RD(N, 2) GW4_MM(3, 1) GW4_RD(N, 3)
                 GW4_MM(0, 2) GW4_ADV(0) GW4_MM(1, 2) GW4_ADV(1) GW4_MM(2, 2) GW4_ADV(2) GW4_MM(3, 2) GW4_ADV(3)
                 GW4_MM(0, 3) GW4_ADV(4) GW4_MM(1, 3) GW4_ADV(5) GW4_MM(2, 3) GW4_ADV(6) GW4_MM(3, 3) GW4_ADV(7) GW4_BARRIER
                 : [ta0] "+v"(ta[0]), [ta1] "+v"(ta[1]), [ta2] "+v"(ta[2]), [ta3] "+v"(ta[3]), [ta4] "+v"(ta[4]), [ta5] "+v"(ta[5]), [ta6] "+v"(ta[6]), [ta7] "+v"(ta[7])
                 : GW4_OPS_ALLMM, GW4_OPS_FR, GW4_OPS_RD, [stp] "s"(step) : GW4_OWNED, "memory");
#undef GW4_ADV
}

// K step 2 of the implicit-GEMM convolution at a TAP CHANGE: the eight per-lane token addresses of the next tap (gemm_w4_conv.hpp: set_tap) are
// computed HERE, 13 VALU operations per piece spread over the 16 MFMA gaps, instead of ~110 compiler instructions (eight 64-bit multiply-adds among
// them) exposed between two statements - once per 64-channel tile pair at 128 channels. 32-bit arithmetic: the caller guarantees input rows < 2^24
// and an activation tensor below 4 GiB. Per piece q (E = q & 1):
//   ti = max(bt + dt, 0); ok = ti < Ti && bit sp of smask; row = ti * frame_rows + yx + dyx; off = row * lda2 + chunk_E;
//   address = ok ? a_base + off : zero page + chunk_E        (low / high words tl / th)
struct GW4Tap {  // wave-uniform values of the tap (scalar registers)
    int dt, sp, dyx, Ti, frame_rows;
    uint32_t lda2, a_lo;
};
G3_DEVICE void gw4_kstep2_bar_settap(uint32_t adw, uint32_t adt, uint32_t (&tl)[8], uint32_t (&th)[8], const int (&bt)[8], const int (&yx)[8],
                                     const uint32_t (&sm)[8], uint32_t chunk_even, uint32_t chunk_odd, uint32_t zel, uint32_t zeh, uint32_t zol,
                                     uint32_t zoh, uint32_t a_hi, const GW4Tap& tp) {
    constexpr int KS = 2;
    constexpr int cur = GW4_FRAG0 + 32 * (KS & 1), nxt = GW4_FRAG0 + 32 * ((KS & 1) ^ 1);
    uint32_t ti, b, row;
    uint64_t cy;
// the 13 operations of piece Q (chunk / zero page operands of its parity E), as three groups that go into consecutive gaps
#define GW4_TAP_A(Q) "v_add_u32 %[ti], %[sdt], %[bt" #Q "]\n\tv_max_i32 %[ti], 0, %[ti]\n\tv_cmp_gt_i32 vcc, %[sTi], %[ti]\n\tv_bfe_u32 %[b], %[sm" #Q "], %[ssp], 1\n\t" \
                     "v_cndmask_b32 %[b], 0, %[b], vcc\n\tv_mad_u32_u24 %[row], %[ti], %[sfr], %[yx" #Q "]\n\t"
#define GW4_TAP_B(Q, E) "v_cmp_eq_u32 vcc, 1, %[b]\n\tv_add_u32 %[row], %[sdyx], %[row]\n\tv_mad_u32_u24 %[row], %[row], %[slda], %[ch" #E "]\n\t" \
                        "v_add_co_u32 %[tl" #Q "], %[cy], %[sal], %[row]\n\tv_addc_co_u32 %[th" #Q "], %[cy], 0, %[ahi], %[cy]\n\t" \
                        "v_cndmask_b32 %[tl" #Q "], %[zl" #E "], %[tl" #Q "], vcc\n\tv_cndmask_b32 %[th" #Q "], %[zh" #E "], %[th" #Q "], vcc\n\t"
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 GW4_MM(0, 0) GW4_RD(N, 4) GW4_TAP_A(0) GW4_MM(1, 0) GW4_RD(N, 5) GW4_TAP_B(0, 0) GW4_MM(2, 0) GW4_RD(N, 6) GW4_TAP_A(1) GW4_MM(3, 0) GW4_RD(N, 7) GW4_TAP_B(1, 1)
                 GW4_MM(0, 1) GW4_RD(N, 0) GW4_TAP_A(2) GW4_MM(1, 1) GW4_RD(N, 1) GW4_TAP_B(2, 0) GW4_MM(2, 1) GW4_RD(N, 2) GW4_TAP_A(3) GW4_MM(3, 1) GW4_RD(N, 3) GW4_TAP_B(3, 1)
                 GW4_MM(0, 2) GW4_TAP_A(4) GW4_MM(1, 2) GW4_TAP_B(4, 0) GW4_MM(2, 2) GW4_TAP_A(5) GW4_MM(3, 2) GW4_TAP_B(5, 1)
                 GW4_MM(0, 3) GW4_TAP_A(6) GW4_MM(1, 3) GW4_TAP_B(6, 0) GW4_MM(2, 3) GW4_TAP_A(7) GW4_MM(3, 3) GW4_TAP_B(7, 1) GW4_BARRIER
                 : [tl0] "=&v"(tl[0]), [tl1] "=&v"(tl[1]), [tl2] "=&v"(tl[2]), [tl3] "=&v"(tl[3]), [tl4] "=&v"(tl[4]), [tl5] "=&v"(tl[5]), [tl6] "=&v"(tl[6]), [tl7] "=&v"(tl[7]),
                   [th0] "=&v"(th[0]), [th1] "=&v"(th[1]), [th2] "=&v"(th[2]), [th3] "=&v"(th[3]), [th4] "=&v"(th[4]), [th5] "=&v"(th[5]), [th6] "=&v"(th[6]), [th7] "=&v"(th[7]),
                   [ti] "=&v"(ti), [b] "=&v"(b), [row] "=&v"(row), [cy] "=&s"(cy)
                 : GW4_OPS_ALLMM, GW4_OPS_FR, GW4_OPS_RD,
                   [bt0] "v"(bt[0]), [bt1] "v"(bt[1]), [bt2] "v"(bt[2]), [bt3] "v"(bt[3]), [bt4] "v"(bt[4]), [bt5] "v"(bt[5]), [bt6] "v"(bt[6]), [bt7] "v"(bt[7]),
                   [yx0] "v"(yx[0]), [yx1] "v"(yx[1]), [yx2] "v"(yx[2]), [yx3] "v"(yx[3]), [yx4] "v"(yx[4]), [yx5] "v"(yx[5]), [yx6] "v"(yx[6]), [yx7] "v"(yx[7]),
                   [sm0] "v"(sm[0]), [sm1] "v"(sm[1]), [sm2] "v"(sm[2]), [sm3] "v"(sm[3]), [sm4] "v"(sm[4]), [sm5] "v"(sm[5]), [sm6] "v"(sm[6]), [sm7] "v"(sm[7]),
                   [ch0] "v"(chunk_even), [ch1] "v"(chunk_odd), [zl0] "v"(zel), [zh0] "v"(zeh), [zl1] "v"(zol), [zh1] "v"(zoh), [ahi] "v"(a_hi),
                   [sdt] "s"(tp.dt), [ssp] "s"(tp.sp), [sdyx] "s"(tp.dyx), [sTi] "s"(tp.Ti), [sfr] "s"(tp.frame_rows), [slda] "s"(tp.lda2), [sal] "s"(tp.a_lo)
                 : GW4_OWNED, "vcc", "memory");
#undef GW4_TAP_A
#undef GW4_TAP_B
}

// PERSIST (g3_set_option("gemm_persistent", 1); OFF by default - measured equal to 8 % slower in round 3, profiles/r3_gemm_persistent_ab.txt: the
// hardware's workgroup turnover was never the cost, and the residual epilogues lose their full prefetch): one workgroup per CU walks output
// tiles L = blockIdx.x, + gridDim.x, ... (same XCD-aware order). Between two tiles nothing of the
// K loop changes; what moves is the dead time around it: the next tile's first LDS stage is requested BEFORE the current tile's epilogue
// (which then transposes through the idle second stage instead of the first), so neither a workgroup dispatch nor the first stage's
// LDS-DMA latency sits between two K loops - only the epilogue itself, whose stores drain under the next tile's accumulator clear.
template <int EPI, bool PERSIST = false>
__global__ __launch_bounds__(GW4_THREADS, 1) void gemm_bf16_nt_w4_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // [stage 2][W tile 256 x 64 | T tile 256 x 64] bf16
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;

    // XCD-aware tile order (as gemm_bf16_nt_pp_kernel): logical tile L -> (m0, n0)
    const int nblk = p.tiles_m * p.tiles_n;
    const int wn = wave & 1;   // feature half of the block tile
    const int wm = wave >> 1;  // token half
    auto tile_origin = [&](int bid, int& m0_out, int& n0_out) {
        {
            const int q = nblk >> 3, r = nblk & 7;
            const int xcd = bid & 7, slot = bid >> 3;
            const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
            bid = base + slot;
        }
        int tile_m, tile_n;
        if (p.tile_order_rowmajor == 1) {
            tile_m = bid / p.tiles_n;
            tile_n = bid - tile_m * p.tiles_n;
        } else {
            const int GM = p.tile_order_rowmajor >= 2 ? p.tile_order_rowmajor : 4;  // option values >= 2: super-row height for A/B runs
            const int per_group = GM * p.tiles_n;
            const int grp = bid / per_group;
            const int within = bid - grp * per_group;
            const int gm = min(GM, p.tiles_m - grp * GM);
            tile_n = within / gm;
            tile_m = grp * GM + (within - tile_n * gm);
        }
        m0_out = tile_m * BM;
        n0_out = tile_n * BN;
    };
    int L = blockIdx.x;
    int m0, n0;
    tile_origin(L, m0, n0);

    // ---- LDS-DMA: wave w stages rows [64 w, 64 w + 64) of both tiles, 8 pieces of 8 rows each; lane -> row 8 q + (lane >> 3), physical chunk
    // lane & 7 holding logical chunk (lane & 7) ^ ((row >> 1) & 7). Per-lane byte offsets from the tile's first row, rows clamped to the matrix.
    uint32_t vo_w[8], vo_t[8];
    const char* w_tile;
    const char* t_tile;
    auto setup_sources = [&](int m0s, int n0s) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = wave * 64 + 8 * q + (lane >> 3);
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);
            const int nrow = min(n0s + r, p.N - 1) - n0s, mrow = min(m0s + r, p.M - 1) - m0s;  // may be negative only if the tile is empty (never launched)
            vo_w[q] = (uint32_t)((int64_t)nrow * p.ldw * 2 + chunk * 16);
            vo_t[q] = (uint32_t)((int64_t)mrow * p.lda * 2 + chunk * 16);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(vo_w[q]), "+v"(vo_t[q]));  // keep them resident
        w_tile = reinterpret_cast<const char*>(p.W + (int64_t)n0s * p.ldw);
        t_tile = reinterpret_cast<const char*>(p.A + (int64_t)m0s * p.lda);
    };
    setup_sources(m0, n0);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const uint32_t m0_w = lds0 + (uint32_t)wave * 8192u, m0_t = lds0 + GW4_T_OFF + (uint32_t)wave * 8192u;  // + stage * 64 KiB

    if (lds0 & 127u) __builtin_trap();  // the XOR form needs the tiles 128-byte aligned (they are: no static LDS in this kernel)

    const int nk = p.K / BK;
    // ---- prologue pieces: K tile 0 complete (stage 0), the weight rows 0..5 of K tile 1 (stage 1: what k-step 3 of "tile -1" would have issued)
    auto dma8 = [&](const char* base, const uint32_t (&vo)[8], uint32_t dst) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + vo[q]),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(dst + 1024u * q), 16, 0, 0);
    };
    auto issue_stage0 = [&]() {
        dma8(w_tile, vo_w, m0_w);
        dma8(t_tile, vo_t, m0_t);
    };
    auto issue_w1 = [&]() {
#pragma unroll
        for (int q = 0; q < 6; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_tile + 128 + vo_w[q]),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(m0_w + GW4_STAGE_BYTES + 1024u * q), 16, 0, 0);
    };
    issue_stage0();
    issue_w1();
  for (;;) {  // output tiles of this workgroup (one unless PERSIST)
    // ---- fragment read addresses (stage 0, row block 0): row * 128 + ((2 ks + g) ^ ((row >> 1) & 7)) * 16 = address(ks = 0) ^ (ks << 5)
    // one set per stage: a ds_read immediate offset has 16 bits and the second stage starts at 64 KiB. (Re)derived per output tile from an
    // opaque copy of the LDS base: in the persistent form they must not stay live across the epilogue, which needs the registers.
    uint32_t adw[2][4], adt[2][4];
    {
        uint32_t ldsb = lds0;
        asm volatile("" : "+s"(ldsb));
        const uint32_t c0 = (uint32_t)((g ^ ((l31 >> 1) & 7)) << 4);
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                adw[st][ks] = ((ldsb + (uint32_t)((wn * 128 + l31) * 128) + c0) ^ (uint32_t)(ks << 5)) + (uint32_t)(st * GW4_STAGE_BYTES);
                adt[st][ks] = ((ldsb + (uint32_t)((wm * 128 + l31) * 128) + c0) ^ (uint32_t)(ks << 5)) + (uint32_t)(st * GW4_STAGE_BYTES);
            }
    }
    {
        // the 256 accumulator writes run while the first tile's LDS-DMA is in flight (one wave per SIMD: nothing else would cover its latency;
        // PERSIST: under the drain of the previous tile's stores - vmcnt counts them too)
        static_for<0, 256>([&](auto rc) { gw4_acc_zero<decltype(rc)::value>(); });
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __syncthreads();
        asm volatile("ds_read_b128 v[192:195], %0\n\tds_read_b128 v[196:199], %0 offset:4096\n\tds_read_b128 v[200:203], %0 offset:8192\n\t"
                     "ds_read_b128 v[204:207], %0 offset:12288\n\tds_read_b128 v[208:211], %1 offset:32768\n\tds_read_b128 v[212:215], %1 offset:36864\n\t"
                     "ds_read_b128 v[216:219], %1 offset:40960\n\tds_read_b128 v[220:223], %1 offset:45056"
                     ::"v"(adw[0][0]), "v"(adt[0][0]) : GW4_OWNED, "memory");
    }

    // K tile t in stage S = t & 1. The 16 LDS-DMA pieces of a tile (8 weight, 8 token row blocks of this wave) are issued over three k-steps:
    // weight pieces 0..5 of tile t + 2 -> this stage in k-step 3 (behind the barrier that frees it), weight pieces 6, 7 + token pieces 0..2
    // of tile t + 1 -> the other stage in k-step 0, token pieces 3..7 in k-step 1; k-step 2 issues none, so every piece has >= 1 k-step
    // (the weight pieces >= 3) before the barrier's vmcnt(0). (Token pieces first measured worse: MLP-up -5 %, the others +-1 %.)
    // DMA_N: tile t + 1 exists; DMA_W: tile t + 2 exists; NEXT = DMA_N.
    auto ktile = [&](auto sc, auto dma_n_c, auto dma_w_c, auto next_c, int t) {
        constexpr int S = decltype(sc)::value;
        constexpr bool DMA_N = decltype(dma_n_c)::value, DMA_W = decltype(dma_w_c)::value, NEXT = decltype(next_c)::value;
        constexpr int SO = S * GW4_STAGE_BYTES, SN = (S ^ 1) * GW4_STAGE_BYTES;
        G3_JITTER(wave + blockIdx.x, t);
        const char* wn1 = w_tile + (int64_t)(t + 1) * 128;
        const char* tn1 = t_tile + (int64_t)(t + 1) * 128;
        const char* wn2 = w_tile + (int64_t)(t + 2) * 128;
        GW4Pieces p0{}, p1{}, p3{};
        if constexpr (DMA_N) {
#pragma unroll
            for (int q = 0; q < 2; ++q) p0.m[q] = m0_w + SN + 1024u * (6 + q), p0.vo[q] = vo_w[6 + q], p0.sb[q] = wn1;
#pragma unroll
            for (int q = 0; q < 3; ++q) p0.m[2 + q] = m0_t + SN + 1024u * q, p0.vo[2 + q] = vo_t[q], p0.sb[2 + q] = tn1;
#pragma unroll
            for (int q = 0; q < 5; ++q) p1.m[q] = m0_t + SN + 1024u * (3 + q), p1.vo[q] = vo_t[3 + q], p1.sb[q] = tn1;
        }
        if constexpr (DMA_W) {
#pragma unroll
            for (int q = 0; q < 6; ++q) p3.m[q] = m0_w + SO + 1024u * q, p3.vo[q] = vo_w[q], p3.sb[q] = wn2;
        }
        gw4_kstep<0, true, DMA_N ? 5 : 0, false>(adw[S][1], adt[S][1], p0);
        gw4_kstep<1, true, DMA_N ? 5 : 0, false>(adw[S][2], adt[S][2], p1);
        if constexpr (NEXT) {
            gw4_kstep<2, true, 0, true>(adw[S][3], adt[S][3], p3);
            gw4_kstep<3, true, DMA_W ? 6 : 0, false>(adw[S ^ 1][0], adt[S ^ 1][0], p3);
        } else {
            gw4_kstep<2, true, 0, false>(adw[S][3], adt[S][3], p3);
            gw4_kstep<3, false, 0, false>(0u, 0u, p3);
        }
    };
    // residual rows of the epilogue (EPI_GATED_RESIDUAL / EPI_BIAS_RESIDUAL), in the epilogue's read-back layout: block J (32 tokens), pass s8:
    // token row 32 J + 4 s8 + (lane >> 4), features 8 (lane & 15) .. + 7. Requested before the LAST K tile: 128 VGPRs that are idle until then.
    constexpr bool HAS_RES = (EPI == EPI_GATED_RESIDUAL || EPI == EPI_BIAS_RESIDUAL);
    const int rsub = lane >> 4, c2 = lane & 15;
    bf16x8 rpre[HAS_RES ? 4 : 1][HAS_RES ? 8 : 1];
    // token blocks [J0, J1). Non-persistent: all four before the last K tile. PERSIST: blocks 0, 1 there, blocks 2, 3 behind the K loop (their
    // latency hides under the epilogue of blocks 0, 1) - all four plus the next tile's source offsets do not fit (77 spills).
    auto prefetch_residual = [&](auto j0c, auto j1c) {
        if constexpr (HAS_RES) {
            const int n = n0 + wn * 128 + 8 * c2;
            static_for<decltype(j0c)::value, decltype(j1c)::value>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const int m = m0 + wm * 128 + 32 * J + 4 * s8 + rsub;
                    rpre[J][s8] = (m < p.M && n < p.N) ? load_bf16x8(p.R + (int64_t)m * p.ldr + n) : zero_bf16x8();
                }
            });
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using ISPLIT = std::integral_constant<int, PERSIST ? 2 : 4>;
    using I4 = std::integral_constant<int, 4>;
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 3 < nk; t += 2) {
        ktile(S0{}, T_{}, T_{}, T_{}, t);
        ktile(S1{}, T_{}, T_{}, T_{}, t + 1);
    }
    if (nk - t == 3) {
        ktile(S0{}, T_{}, T_{}, T_{}, t);
        ktile(S1{}, T_{}, F_{}, T_{}, t + 1);
        prefetch_residual(I0{}, ISPLIT{});
        ktile(S0{}, F_{}, F_{}, F_{}, t + 2);
    } else {  // 2 tiles left
        ktile(S0{}, T_{}, F_{}, T_{}, t);
        prefetch_residual(I0{}, ISPLIT{});
        ktile(S1{}, F_{}, F_{}, F_{}, t + 1);
    }
    prefetch_residual(ISPLIT{}, I4{});

    // ---- epilogue: the accumulators leave the AGPRs one 32-token block (64 registers) at a time and go through the full-line LDS transpose of
    // the other kernels (store_tile_lds: private 16 KiB fp32 slice per wave, a lane then owns 8 consecutive features of one token row). At one
    // wave per SIMD nothing covers the latency of the residual rows, so they were requested before the last K tile (rpre).
    asm volatile("s_nop 7\n\ts_nop 3" ::: GW4_OWNED);  // last MFMA results -> v_accvgpr_read
    __syncthreads();                                     // the operand stages are idle once every wave is past its last fragment read
    if (G3_AB_GW4_ABLATE & 32) return;  // timing ablation: no epilogue at all
    // PERSIST: the next tile's K tile 0 goes into stage 0 NOW, under the epilogue, which transposes through stage 1 instead
    const int m0e = m0, n0e = n0;  // this tile's origin, for the epilogue
    const int Lnext = L + (int)gridDim.x;
    const bool has_next = PERSIST && Lnext < nblk;
    if (has_next) {
        tile_origin(Lnext, m0, n0);
        setup_sources(m0, n0);
        issue_stage0();
    }
    {
        const int m0 = m0e, n0 = n0e;  // (shadow: everything below addresses the finished tile)
        char* stage = smem_raw + (PERSIST ? GW4_STAGE_BYTES : 0) + wave * 16384;
        const int n = n0 + wn * 128 + 8 * c2;
        bf16x8 gv1 = zero_bf16x8();
        if (EPI != EPI_NONE && EPI != EPI_GELU && EPI != EPI_QK_NORM_ROPE && p.gate_rows == 1 && n < p.N) gv1 = load_bf16x8(p.gate + n);
        int qk_type = 0, vt_head = -1;  // vt_head >= 0: this wave's 128 features are v head vt_head and go to V^T instead of C
        bf16x8 nwv = zero_bf16x8();
        if (EPI == EPI_QK_NORM_ROPE) {
            const int nh = n0 + wn * 128;  // first feature of this wave's head
            qk_type = nh < p.n_q ? 1 : (nh < p.n_q + p.n_k ? 2 : 0);
            if (qk_type != 0) nwv = load_bf16x8((qk_type == 1 ? p.nw_q : p.nw_k) + 8 * c2);
            if (qk_type == 0 && p.vt != nullptr && nh < p.N) vt_head = (nh - p.n_q - p.n_k) >> 7;
        }
        static_for<0, 4>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            // EPI_QK_NORM_ROPE: the cos / sin entries of this lane's 8 features for the 8 token rows of the block, requested before the
            // accumulators are moved (64 v_accvgpr_read + the LDS transpose cover most of their latency)
            f32x4 tcos[EPI == EPI_QK_NORM_ROPE ? 8 : 1][2], tsin[EPI == EPI_QK_NORM_ROPE ? 8 : 1][2];
            if (EPI == EPI_QK_NORM_ROPE && qk_type != 0 && p.rope_cos != nullptr) {
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const int m = min(m0 + wm * 128 + 32 * J + 4 * s8 + rsub, p.M - 1);
                    const int64_t o = (int64_t)(m / p.rope_B) * 128 + 8 * c2;
                    tcos[s8][0] = *reinterpret_cast<const f32x4*>(p.rope_cos + o);
                    tcos[s8][1] = *reinterpret_cast<const f32x4*>(p.rope_cos + o + 4);
                    tsin[s8][0] = *reinterpret_cast<const f32x4*>(p.rope_sin + o);
                    tsin[s8][1] = *reinterpret_cast<const f32x4*>(p.rope_sin + o + 4);
                }
            }
            f32x16 acc[4];
            static_for<0, 64>([&](auto rc) {
                constexpr int R = decltype(rc)::value;
                acc[R >> 4][R & 15] = gw4_acc_read<64 * J + R>();
            });
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * q4 + e];
                    *reinterpret_cast<f32x4*>(stage + l31 * 512 + (((8 * i + 2 * q4 + g) ^ l31) << 4)) = v;
                }
            if (EPI == EPI_QK_NORM_ROPE && vt_head >= 0) {
                // A v head with a V^T destination (V^T [B][H_v][128][vt_ld], what the attention kernel reads): the staged [32 token rows][128
                // features] fp32 block is read back COLUMN-wise - a lane gathers 8 consecutive key positions of one feature and batch item
                // (token rows b + B (8 gq + e), B in {1, 2, 4}) and stores them as one 16-byte piece of the V^T row; 4 lanes cover the 64
                // contiguous bytes a 32-row block contributes to it. Positions >= S are written as zeros (the V^T tail contract).
                const int B = p.rope_B, ng = 4 / B;            // groups of 8 key positions per batch item in a 32-row block
                const int mbase = m0 + wm * 128 + 32 * J;      // multiple of 32, hence of B
                const int bg = lane & 3, bb = bg / ng, gq = bg - bb * ng;
                const int s0 = mbase / B + 8 * gq;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int f = 16 * it + (lane >> 2);
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int row = bb + B * (8 * gq + e);
                        const float val = *reinterpret_cast<const float*>(stage + row * 512 + ((((f >> 2) ^ row) << 4) | ((f & 3) << 2)));
                        o[e] = f32_to_bf16((s0 + e < p.vt_S) ? val : 0.f);
                    }
                    if (s0 < p.vt_ld) store_bf16x8(p.vt + (int64_t)bb * p.vt_batch + (int64_t)vt_head * 128 * p.vt_ld + (int64_t)f * p.vt_ld + s0, o);
                }
                return;  // this block is done (static_for body)
            }
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const int row = 4 * s8 + rsub;
                const int m = m0 + wm * 128 + 32 * J + row;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * 512 + (((2 * c2) ^ row) << 4));
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * 512 + (((2 * c2 + 1) ^ row) << 4));
                if (m >= p.M || n >= p.N) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = lo[e];
                    v[4 + e] = hi[e];
                }
                if (EPI == EPI_QK_NORM_ROPE) {
                    // nn.Linear rounds to bf16; per-head RMSNorm in fp32 -> bf16 (te RMSNorm); RoPE in fp32 -> bf16: the rounding points of
                    // qk_rmsnorm_rope_kernel (norm_rope.hip). A head's 128 features of this row sit in the 16 lanes c2 = 0..15 of this lane's
                    // row group; feature d pairs with d +- 64 = lane c2 ^ 8 (rotate_half).
                    if (qk_type != 0) {  // wave-uniform: 0 = plain features (v), 1 = q head, 2 = k head
                        float ss = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            v[e] = (float)f32_to_bf16(v[e]);
                            ss += v[e] * v[e];
                        }
                        ss += __shfl_xor(ss, 1, 64);
                        ss += __shfl_xor(ss, 2, 64);
                        ss += __shfl_xor(ss, 4, 64);
                        ss += __shfl_xor(ss, 8, 64);
                        const float rinv = rsqrtf(ss * (1.0f / 128.0f) + p.rms_eps);
                        uint32_t pk[4];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (float)f32_to_bf16(v[e] * rinv * (float)nwv[e]);
                        if (p.rope_cos != nullptr) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                bf16x2 t2;
                                t2[0] = f32_to_bf16(v[2 * e]);
                                t2[1] = f32_to_bf16(v[2 * e + 1]);
                                pk[e] = __shfl_xor(__builtin_bit_cast(uint32_t, t2), 8, 64);
                            }
                            const float sgn = (c2 & 8) ? 1.f : -1.f;  // first half: t cos - t2 sin; second half: t2 cos + t sin
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const bf16x2 t2 = __builtin_bit_cast(bf16x2, pk[e >> 1]);
                                const float other = (float)t2[e & 1];
                                const float cs = e < 4 ? tcos[s8][0][e & 3] : tcos[s8][1][e & 3];
                                const float sn = e < 4 ? tsin[s8][0][e & 3] : tsin[s8][1][e & 3];
                                const float a = v[e] * cs, b = other * sn;
                                v[e] = sgn < 0.f ? a - b : a + b;
                            }
                        }
                    }
                } else if (EPI == EPI_GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_erf_fast((float)f32_to_bf16(v[e]));  // (the Linear's own rounding first: gemm.hip, store_tile)
                } else if (EPI != EPI_NONE) {
                    const bf16x8 gv = p.gate_rows == 1 ? gv1 : load_bf16x8(p.gate + (int64_t)(m % p.gate_rows) * p.ldg + n);
                    if (EPI == EPI_GATED_RESIDUAL) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (float)rpre[J][s8][e] + (float)gv[e] * (float)f32_to_bf16(v[e]);
                    } else if (EPI == EPI_BIAS) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)gv[e];
                    } else if (EPI == EPI_BIAS_RESIDUAL) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (v[e] + (float)gv[e]) + (float)rpre[J][s8][e];
                    }
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(v[e]);
                store_bf16x8(p.C + (int64_t)m * p.ldc + n, o);
            }
        });
    }
    if (!has_next) break;
    __syncthreads();  // every wave is done with its stage-1 transpose slice: the next tile's weight rows of K tile 1 may land there
    asm volatile("" : "+s"(m0), "+s"(n0));  // (opaque: the offsets are derived AGAIN instead of being kept alive across the epilogue)
    setup_sources(m0, n0);
    issue_w1();
    L = Lnext;
  }
}

template <int EPI>
int launch_w4(const GemmParams& p, hipStream_t stream, const char* what) {
    const size_t smem = 2 * GW4_STAGE_BYTES;
    static bool attr_set[64] = {};
    static int n_cu[64] = {};
    static std::mutex attr_mu;
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) return g3_set_error(G3_ERR_LAUNCH, "gemm: hipGetDevice failed");
    {
        std::lock_guard<std::mutex> lock(attr_mu);
        if (!attr_set[dev_id]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_w4_kernel<EPI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if constexpr (EPI != EPI_QK_NORM_ROPE)
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_w4_kernel<EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess || cus <= 0) cus = 256;
            n_cu[dev_id] = cus;
            attr_set[dev_id] = true;
        }
    }
    const int nblk = p.tiles_m * p.tiles_n;
    // persistent form: one workgroup per CU (a multiple of 8 keeps a workgroup's tiles on its XCD's band of the XCD-aware order); only worth it
    // when a workgroup gets at least two tiles
    const int grid_p = (n_cu[dev_id] / 8) * 8;
    if constexpr (EPI != EPI_QK_NORM_ROPE) {  // (the norm / RoPE epilogue's tables do not fit beside the tile state: no persistent form)
        if (g3_opt_gemm_persistent && grid_p >= 8 && nblk >= 2 * grid_p) {
            hipLaunchKernelGGL((gemm_bf16_nt_w4_kernel<EPI, true>), dim3(grid_p), dim3(GW4_THREADS), smem, stream, p);
            return g3_check_launch(what);
        }
    }
    hipLaunchKernelGGL((gemm_bf16_nt_w4_kernel<EPI, false>), dim3(nblk), dim3(GW4_THREADS), smem, stream, p);
    return g3_check_launch(what);
}
