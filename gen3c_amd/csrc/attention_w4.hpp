// Included by attention.hip INSIDE its anonymous namespace (shares AttnParams, the LDS tile layout helpers k_off / v_off / swap23, max3,
// xor32_sum, lds_read_frag / static_for and RESCALE_THR).
//
// w4: self-attention with ONE wave per SIMD and 64 query rows per wave.
//
// Why (profiles/r2_attn_ablation.txt): in the 8-wave kernel (v3) every MFMA takes one 1-KiB operand fragment out of LDS, and those
// ds_read_b128 cost three times what the whole softmax costs (+33 % with the reads ablated, +11 % with the softmax ablated). The only way to
// halve the reads per flop is to let one fragment feed TWO MFMAs, i.e. 64 query rows per wave - which needs O (128 regs) + two score sets
// (128) + Q (64) + P (32) per lane: the 512-register budget of a wave that owns its SIMD. So: workgroup = 4 waves = 256 query rows; each
// wave two 32-row halves h = 0, 1 that share every K / V^T fragment; same LDS tiles, swizzles, LDS-DMA staging and folded softmax
// arithmetic as v3.
//
// Register files. With one wave per SIMD hipcc selects the ACCUMULATOR-register form of every MFMA builtin (C/D in AGPRs), which parks the
// scores in AGPRs and costs one v_accvgpr_read per score before the softmax VALU can touch them (round 1: 0.64x). Letting it allocate
// "a"-constrained asm operands was no better: the 128 O registers were copied (v_accvgpr_mov) on every loop back edge or spilled. So the
// accumulator file is OWNED by this file's asm statements, by literal register name (cdna_hip_programming.md 5.7, item 4):
//     a[0:127]    O^T accumulators, block (h, d) at a[16 (4 h + d) : +15]
//     a[128:191]  Q fragments (B operand of QK^T), fragment (h, ks) at a[128 + 4 (8 h + ks) : +3]
// Every statement that names them lists all 192 as clobbers, so hipcc never keeps a value of its own there across one of them
// (tools/asm_audit.py checks that no compiler-generated v_accvgpr_* touches a0..a191 and that nothing spills to scratch). The scores stay
// ordinary C++ values in VGPRs ("+v" operands of the QK^T MFMAs): the 256 architectural VGPRs hold only what VALU instructions touch.
// hipcc neither counts nor pads an asm MFMA: the hazards that exist here are fenced by hand -
//   * scores (MFMA D in VGPRs) -> first VALU reader: w4_fence_v() (12 wait states, tied to the score registers) after the QK^T region;
//   * O (MFMA D in AGPRs) -> v_accvgpr_read in the rescale branch / epilogue: w4_fence_acc().
// Accumulate chains (D of one MFMA = C of the next) need no wait states; LDS fragments are guarded by hand-counted lgkmcnt waits that live
// in the same statement as the first MFMA consuming the fragment (as two statements hipcc pads the boundary with an s_nop).

constexpr int W4_WAVES = 4;
constexpr int W4_THREADS = 64 * W4_WAVES;
constexpr int W4_BQ = 64 * W4_WAVES;  // 256 query rows per workgroup
constexpr int W4_QBASE = 128;         // first AGPR of the Q fragments

#define W4_OWNED_AGPRS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191"

template <int R> G3_DEVICE void w4_acc_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"n"(R) : W4_OWNED_AGPRS); }
template <int R> G3_DEVICE void w4_acc_write(uint32_t v) { asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "n"(R) : W4_OWNED_AGPRS); }
template <int R> G3_DEVICE float w4_acc_read() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "n"(R) : W4_OWNED_AGPRS);
    return v;
}
template <int R> G3_DEVICE void w4_acc_scale(float alpha) {  // a[R] *= alpha
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c2, %0" : "=&v"(t) : "v"(alpha), "n"(R) : W4_OWNED_AGPRS);
}
G3_DEVICE void w4_fence_acc() { asm volatile("s_nop 7\n\ts_nop 3" ::: W4_OWNED_AGPRS); }  // 8-pass MFMA result -> v_accvgpr_read: 12 wait states
// 8-pass MFMA result in VGPRs -> any non-MFMA reader: 12 wait states. Tied to the registers so that no reader can be scheduled above the fence.
G3_DEVICE void w4_fence_v(f32x16& a, f32x16& b, f32x16& c, f32x16& d) { asm volatile("s_nop 7\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }

// S^T block (VGPRs) += K . Q_(QI)^T ;  WAIT >= 0: first wait until all but the WAIT youngest LDS reads have landed
template <int QI, int WAIT> G3_DEVICE void w4_qk(f32x16& S, const bf16x8& kfrag) {
    if constexpr (WAIT >= 0)
        asm volatile("s_waitcnt lgkmcnt(%c4)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0"
                     : "+v"(S) : "v"(kfrag), "n"(W4_QBASE + 4 * QI), "n"(W4_QBASE + 4 * QI + 3), "n"(WAIT) : W4_OWNED_AGPRS);
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(S) : "v"(kfrag), "n"(W4_QBASE + 4 * QI), "n"(W4_QBASE + 4 * QI + 3) : W4_OWNED_AGPRS);
}
// first MFMA of a chain: S = K . Q^T + c  (c = -m_run of the half in every element: scores arrive relative to the running maximum)
template <int QI, int WAIT> G3_DEVICE void w4_qk0(f32x16& S, const bf16x8& kfrag, const f32x16& c) {
    if constexpr (WAIT >= 0)
        asm volatile("s_waitcnt lgkmcnt(%c5)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%c3:%c4], %2"
                     : "=&v"(S) : "v"(kfrag), "v"(c), "n"(W4_QBASE + 4 * QI), "n"(W4_QBASE + 4 * QI + 3), "n"(WAIT) : W4_OWNED_AGPRS);
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c3:%c4], %2" : "=&v"(S) : "v"(kfrag), "v"(c), "n"(W4_QBASE + 4 * QI), "n"(W4_QBASE + 4 * QI + 3) : W4_OWNED_AGPRS);
}
// O^T block OB (AGPRs a[16 OB : +15]) += V^T . P^T
template <int OB, int WAIT> G3_DEVICE void w4_pv(const bf16x8& vfrag, const u32x4& pfrag) {
    if constexpr (WAIT >= 0)
        asm volatile("s_waitcnt lgkmcnt(%c4)\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vfrag), "v"(pfrag), "n"(16 * OB), "n"(16 * OB + 15), "n"(WAIT) : W4_OWNED_AGPRS);
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vfrag), "v"(pfrag), "n"(16 * OB), "n"(16 * OB + 15) : W4_OWNED_AGPRS);
}
// one LDS-DMA piece (1 KiB per wave: 64 lanes x 16 B), issued from INSIDE the MFMA stream: at one wave per SIMD the ~60-180 cycles an LDS-DMA
// instruction takes to issue are matrix-pipe idle time unless an MFMA is executing meanwhile, so the 8 pieces a wave contributes to the next
// tiles are spread over the first steps of region A instead of being issued back to back at the top of the tile. M0 (the LDS destination)
// is written in the same statement that reads it (hipcc does not preserve M0 around asm statements, so it does not expect it preserved either).
G3_DEVICE void w4_dma_piece(uint32_t lds_dst, const char* sbase, uint32_t voff) {
    if (G3_AB_ATTN_ABLATE & 128) return;  // timing ablation: no LDS-DMA in the stream (tiles keep their prologue contents)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}
// fragment read through an address derived on the spot (bit 7 flipped: K fragment ks from ks - 4, see lds_addr_flip128) - one statement, so
// that no pad separates the v_xor from the ds_read and the derived address never occupies a register across MFMAs
template <int OFF> G3_DEVICE void lds_read_frag_flip128(bf16x8& dst, uint32_t addr) {
    uint32_t tmp;
    asm volatile("v_xor_b32 %1, 0x80, %2\n\tds_read_b128 %0, %1 offset:%3" : "=v"(dst), "=&v"(tmp) : "v"(addr), "n"(OFF));
}

// ---- one asm statement per pipeline STEP. hipcc pads every boundary between two asm statements where the second names a register the
// first wrote (s_nop) and is free to move its own VALU across them; inside one statement the stream is exactly what is written here.
//   pair unit k: 2 exp2 + 2 row-sum adds + 1 cvt_pk of two scores -> one packed-bf16 dword of a P fragment
// (timing ablations, tools/attn_ablate.py: G3_AB_ATTN_ABLATE & 16 drops the pair units, & 32 the fragment reads, & 64 the tile barrier,
// & 128 the in-stream LDS-DMA pieces, & 256 the row-max chains)
#if G3_AB_ATTN_ABLATE & 16
#define W4_UNIT(k) "v_mov_b32 %[k" #k "], 0\n\t"
#else
#define W4_UNIT(k)                                                                                                                      \
    "v_exp_f32 %[a" #k "], %[x" #k "]\n\tv_exp_f32 %[b" #k "], %[y" #k "]\n\tv_add_f32 %[p" #k "], %[p" #k "], %[a" #k "]\n\t"      \
    "v_add_f32 %[r" #k "], %[r" #k "], %[b" #k "]\n\tv_cvt_pk_bf16_f32 %[k" #k "], %[a" #k "], %[b" #k "]\n\t"
#endif
#if G3_AB_ATTN_ABLATE & 32
#define W4_READ ""
#else
#define W4_READ "ds_read_b128 %[nf], %[addr] offset:%c[off]\n\t"
#endif
#ifdef G3_AB_ATTN_EXTRA_NOP  // A/B (round 5): one more issue slot per step statement = what the s_nop hipcc pads some statement boundaries with costs
#define W4_WAIT "s_nop 0\n\ts_waitcnt lgkmcnt(%c[wn])\n\t"
#else
#define W4_WAIT "s_waitcnt lgkmcnt(%c[wn])\n\t"
#endif
#define W4_QK(h) "v_mfma_f32_32x32x16_bf16 %[s" #h "], %[f], a[%c[q" #h "]:%c[e" #h "]], %[s" #h "]\n\t"
#define W4_QK0(h) "v_mfma_f32_32x32x16_bf16 %[s" #h "], %[f], a[%c[q" #h "]:%c[e" #h "]], %[c" #h "]\n\t"
#define W4_PV(h) "v_mfma_f32_32x32x16_bf16 a[%c[o" #h "]:%c[g" #h "]], %[f], %[pf" #h "], a[%c[o" #h "]:%c[g" #h "]]\n\t"

struct W4Unit {  // operands of one pair unit: two scores in, packed P dword out, the two running row-sum registers of the half
    float x, y;
};

// region A step: [read next fragment] wait; S0 (+)= K.Q0 ; unit 1 ; S1 (+)= K.Q1 ; [unit 2]
template <int QI0, int OFF, int WN, bool INIT, bool TWO>
G3_DEVICE void w4_step_qk(bf16x8& nf, uint32_t addr, const bf16x8& f, f32x16& s0, f32x16& s1, const f32x16& c0, const f32x16& c1, float x1, float y1,
                          uint32_t& k1, float& p1, float& r1, float x2, float y2, uint32_t& k2, float& p2, float& r2) {
    float a1, b1, a2, b2;
    constexpr int q0 = W4_QBASE + 4 * QI0, q1 = W4_QBASE + 4 * (8 + QI0);
#define W4_A_OUT_COMMON [nf] "=&v"(nf), [a1] "=&v"(a1), [b1] "=&v"(b1), [k1] "=&v"(k1), [p1] "+v"(p1), [r1] "+v"(r1)
#define W4_A_OUT_TWO , [a2] "=&v"(a2), [b2] "=&v"(b2), [k2] "=&v"(k2), [p2] "+v"(p2), [r2] "+v"(r2)
#define W4_A_IN_COMMON [f] "v"(f), [addr] "v"(addr), [x1] "v"(x1), [y1] "v"(y1), [off] "n"(OFF), [wn] "n"(WN), [q0] "n"(q0), [e0] "n"(q0 + 3), [q1] "n"(q1), [e1] "n"(q1 + 3)
    if constexpr (INIT && TWO)
        asm volatile(W4_READ W4_WAIT W4_QK0(0) W4_UNIT(1) W4_QK0(1) W4_UNIT(2)
                     : W4_A_OUT_COMMON W4_A_OUT_TWO, [s0] "=&v"(s0), [s1] "=&v"(s1)
                     : W4_A_IN_COMMON, [x2] "v"(x2), [y2] "v"(y2), [c0] "v"(c0), [c1] "v"(c1) : W4_OWNED_AGPRS);
    else if constexpr (INIT && !TWO)
        asm volatile(W4_READ W4_WAIT W4_QK0(0) W4_UNIT(1) W4_QK0(1)
                     : W4_A_OUT_COMMON, [s0] "=&v"(s0), [s1] "=&v"(s1) : W4_A_IN_COMMON, [c0] "v"(c0), [c1] "v"(c1) : W4_OWNED_AGPRS);
    else if constexpr (!INIT && TWO)
        asm volatile(W4_READ W4_WAIT W4_QK(0) W4_UNIT(1) W4_QK(1) W4_UNIT(2)
                     : W4_A_OUT_COMMON W4_A_OUT_TWO, [s0] "+v"(s0), [s1] "+v"(s1) : W4_A_IN_COMMON, [x2] "v"(x2), [y2] "v"(y2) : W4_OWNED_AGPRS);
    else
        asm volatile(W4_READ W4_WAIT W4_QK(0) W4_UNIT(1) W4_QK(1)
                     : W4_A_OUT_COMMON, [s0] "+v"(s0), [s1] "+v"(s1) : W4_A_IN_COMMON : W4_OWNED_AGPRS);
#undef W4_A_OUT_COMMON
#undef W4_A_OUT_TWO
#undef W4_A_IN_COMMON
}

// region B step: [read next fragment] wait; O(h0,d) += V^T.P0 ; [unit] ; O(h1,d) += V^T.P1
template <int D, int OFF, int WN, bool READ, bool UNIT>
G3_DEVICE void w4_step_pv(bf16x8& nf, uint32_t addr, const bf16x8& f, const u32x4& pf0, const u32x4& pf1, float x1, float y1, uint32_t& k1, float& p1,
                          float& r1) {
    float a1, b1;
    constexpr int o0 = 16 * D, o1 = 16 * (4 + D);
#define W4_B_IN_COMMON [f] "v"(f), [pf0] "v"(pf0), [pf1] "v"(pf1), [wn] "n"(WN), [o0] "n"(o0), [g0] "n"(o0 + 15), [o1] "n"(o1), [g1] "n"(o1 + 15)
    if constexpr (READ && UNIT)
        asm volatile(W4_READ W4_WAIT W4_PV(0) W4_UNIT(1) W4_PV(1)
                     : [nf] "=&v"(nf), [a1] "=&v"(a1), [b1] "=&v"(b1), [k1] "=&v"(k1), [p1] "+v"(p1), [r1] "+v"(r1)
                     : W4_B_IN_COMMON, [addr] "v"(addr), [off] "n"(OFF), [x1] "v"(x1), [y1] "v"(y1) : W4_OWNED_AGPRS);
    else if constexpr (READ && !UNIT)
        asm volatile(W4_READ W4_WAIT W4_PV(0) W4_PV(1) : [nf] "=&v"(nf) : W4_B_IN_COMMON, [addr] "v"(addr), [off] "n"(OFF) : W4_OWNED_AGPRS);
    else if constexpr (!READ && UNIT)
        asm volatile(W4_WAIT W4_PV(0) W4_UNIT(1) W4_PV(1)
                     : [a1] "=&v"(a1), [b1] "=&v"(b1), [k1] "=&v"(k1), [p1] "+v"(p1), [r1] "+v"(r1) : W4_B_IN_COMMON, [x1] "v"(x1), [y1] "v"(y1) : W4_OWNED_AGPRS);
    else
        asm volatile(W4_WAIT W4_PV(0) W4_PV(1) : : W4_B_IN_COMMON : W4_OWNED_AGPRS);
#undef W4_B_IN_COMMON
}

template <int CTX, int W4_RD = 4, int W4_KREGS = 8>
__global__ __launch_bounds__(W4_THREADS, 1) void flash_attn_fwd_w4_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);  // [2][64][128]
    bf16_t* sV = sK + 2 * KVB * HD;                     // [2][128][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;
    const int head = blockIdx.y;
    const int batch = blockIdx.z;

    const bf16_t* Qb = p.Q + batch * p.q_batch + head * p.q_head;
    const bf16_t* Kb = p.K + batch * p.k_batch + head * p.k_head;
    const bf16_t* Vb = p.Vt + batch * p.vt_batch + head * p.vt_head;
    bf16_t* Ob = p.O + batch * p.o_batch + head * p.o_head;

    // ---- Q fragments of both halves (B operand: column = q row, k = head dim), pre-multiplied by scale * log2(e) (fold), into a[128:191];
    //      O accumulators a[0:127] = 0
    static_for<0, 16>([&](auto fc) {
        constexpr int f = decltype(fc)::value, h = f >> 3, ks = f & 7;
        const int q_idx = blockIdx.x * W4_BQ + wave * 64 + 32 * h + l31;
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

    // ---- LDS-DMA staging: K tile = 1024 16-B slots (row = slot >> 4, chunk = slot & 15), V^T tile = 1024 slots (row = slot >> 3,
    // chunk = slot & 7). Thread tid fills slots tid + 256 i (i = 0..3) of each: K rows (tid >> 4) + 16 i, V^T rows (tid >> 3) + 32 i - the
    // swizzle term (row & 15 resp. (row >> 1) & 7) is the same for the four, so one source chunk per lane; 32-bit byte offsets from the
    // wave-uniform head bases (SGPR-base form of global_load_lds).
    const int k_row0 = tid >> 4, k_src_chunk = (tid & 15) ^ (k_row0 & 15);
    const int v_row0 = tid >> 3, v_src_chunk = (tid & 7) ^ ((v_row0 >> 1) & 7);
    const char* Kbytes = reinterpret_cast<const char*>(Kb);
    const char* Vbytes = reinterpret_cast<const char*>(Vb);
    const uint32_t k_row_bytes = (uint32_t)p.k_row * 2u;
    const uint32_t v_row_bytes = (uint32_t)p.vt_row * 2u;
    const uint32_t k_lane = (uint32_t)k_row0 * k_row_bytes + (uint32_t)k_src_chunk * 16u;
    const uint32_t v_lane = (uint32_t)v_row0 * v_row_bytes + (uint32_t)v_src_chunk * 16u;
    const uint32_t k_last = (uint32_t)(p.Skv - 1) * k_row_bytes + (uint32_t)k_src_chunk * 16u;  // clamp target for ragged tails
    auto dma_k = [&](int kv0, int slot) {
        bf16_t* d = sK + slot * KVB * HD + wave * 64 * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t o = min(k_lane + (uint32_t)(kv0 + 16 * i) * k_row_bytes, k_last);  // rows past S_kv-1 re-read the last row (masked later)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kbytes + o),
                                             (__attribute__((address_space(3))) void*)(d + 256 * 8 * i), 16, 0, 0);
        }
    };
    const uint32_t seg_len = (uint32_t)p.vt_seg_len, seg_bytes = (uint32_t)p.vt_seg_stride * 2u;
    auto dma_v = [&](int kv0, int slot) {
        bf16_t* d = sV + slot * HD * KVB + wave * 64 * 8;
        uint32_t tile_off = (uint32_t)kv0 * 2u;
        if (seg_len) {  // a 64-key tile never straddles segments (seg_len % 64 == 0)
            const uint32_t sg = (uint32_t)kv0 / seg_len;
            tile_off = sg * seg_bytes + ((uint32_t)kv0 - sg * seg_len) * 2u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t o = v_lane + tile_off + 32u * (uint32_t)i * v_row_bytes;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vbytes + o),
                                             (__attribute__((address_space(3))) void*)(d + 256 * 8 * i), 16, 0, 0);
        }
    };

    // ---- per-lane LDS byte addresses of the operand fragments inside slot 0 of each ring (same fragment mapping as v3)
    const int krow_perm = swap23(l31);
    const uint32_t lds_k0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)sK;
    const uint32_t lds_v0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)sV;
    uint32_t kaddr[8], vaddr[4];
    if (W4_KREGS < 8 && (lds_k0 & 255u)) __builtin_trap();  // lds_read_frag_flip128 needs the K ring 256-byte aligned (it is: no static LDS)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kaddr[ks] = lds_k0 + 2u * (uint32_t)k_off(krow_perm, 2 * ks + g);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) vaddr[s4] = lds_v0 + 2u * (uint32_t)v_off(l31, 2 * s4 + g);

    float m_run[2], mx_cur[2];
    // row sums: 8 partial accumulators per half, PERSISTENT over the tiles (zeroing 16 registers and folding them into l every tile costs 32
    // issue slots per tile that the one-wave stream does not have). [half][step parity x unit slot x lane of the pair]: consecutive
    // statements never name the same accumulator, and the two units of one step never name the same "+v" variable twice.
    float psum[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 8; ++k) psum[h][k] = 0.f;
    const int nt = (p.Skv + KVB - 1) / KVB;

    auto row_max = [&](const f32x16 (&S)[2]) -> float {  // two independent v_max3 chains (one per 32-kv block), then the lane^32 partner
        float ma = max3(S[0][0], S[0][1], S[0][2]);
        float mb2 = max3(S[1][0], S[1][1], S[1][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) {
            ma = max3(ma, S[0][r], S[0][r + 1]);
            mb2 = max3(mb2, S[1][r], S[1][r + 1]);
        }
        return xor32_max(max3(ma, mb2, max3(S[0][15], S[1][15], S[1][15])));
    };
    auto mask_tail = [&](f32x16 (&S)[2], int kv0) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + 32 * mb + 16 * (r >> 3) + 8 * g + (r & 7);
                if (kv >= p.Skv) S[mb][r] = -INFINITY;
            }
    };

    // ---- prologue: K(0), V(0) (and K(1)) by LDS-DMA; scores of tile 0 with C = 0, then made relative to their exact row maximum
    dma_k(0, 0);
    dma_v(0, 0);
    if (nt > 1) dma_k(KVB, 1);
    lds_dma_publish_barrier();
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
        if (nt == 1 && KVB > p.Skv) mask_tail(SA[h], 0);
        m_run[h] = row_max(SA[h]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) SA[h][mb][r] -= m_run[h];
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[h][r] = -m_run[h];
        mx_cur[h] = 0.f;
    }

    auto tile = [&](f32x16 (&S_cur)[2][2], f32x16 (&S_next)[2][2], int t, auto has_next_c, auto par_c) {
        constexpr bool has_next = decltype(has_next_c)::value;
        constexpr int par = decltype(par_c)::value;
        const int kv0 = t * KVB;
        // The first fragment reads of the tile go out FIRST: their LDS latency (exposed at one wave per SIMD: step 0 cannot start without
        // fragment 0) is covered by the row-max chains of this tile's scores, VALU work that needs no LDS operand.
        constexpr int RD = W4_RD;
        bf16x8 fr[RD];  // one fragment ring through both regions (fragment n in slot n % RD, read RD-1 steps before its use)
        if (has_next) {
            constexpr int KS = (par ^ 1) * KVB * HD * 2;
            static_for<0, RD - 1>([&](auto ic) { constexpr int n = decltype(ic)::value; lds_read_frag<KS + 32 * (n & 1) * HD * 2>(fr[n % RD], kaddr[n >> 1]); });
        } else {
            constexpr int VS = par * HD * KVB * 2;
            static_for<0, RD - 1>([&](auto ic) { constexpr int j = decltype(ic)::value; lds_read_frag<VS + 32 * (j & 3) * KVB * 2>(fr[(16 + j) % RD], vaddr[j >> 2]); });
        }
        if (G3_AB_ATTN_ABLATE & 256) {  // timing ablation: no row-max chains
            mx_cur[0] = mx_cur[1] = 0.f;
        } else if (!has_next && kv0 + KVB > p.Skv) {  // ragged tile can only be the last one: row max of the masked scores
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                mask_tail(S_cur[h], kv0);
                mx_cur[h] = row_max(S_cur[h]);
            }
        } else {
            // row max of S_cur[h] (relative to m_run) as one serial v_max3 chain over its 32 scores + the lane^32 exchange; one asm statement
            // per half: hipcc pads every asm statement whose result the next VALU reads with an s_nop, and the chain is serial
            static_for<0, 2>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                const f32x16& s0 = S_cur[h][0];
                const f32x16& s1 = S_cur[h][1];
                float m, a, b;
                asm volatile("v_max3_f32 %0, %3, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\tv_max3_f32 %0, %0, %10, %11\n\t"
                             "v_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %0, %0, %14, %15\n\tv_max3_f32 %0, %0, %16, %17\n\tv_max3_f32 %0, %0, %18, %18\n\t"
                             "v_max3_f32 %1, %19, %20, %21\n\tv_max3_f32 %1, %1, %22, %23\n\tv_max3_f32 %1, %1, %24, %25\n\t"
                             "v_max3_f32 %1, %1, %26, %27\n\tv_max3_f32 %1, %1, %28, %29\n\tv_max3_f32 %1, %1, %30, %31\n\tv_max3_f32 %1, %1, %32, %33\n\t"
                             "v_max3_f32 %0, %0, %1, %34\n\t"
                             "v_mov_b32 %1, %0\n\tv_mov_b32 %2, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %2\n\tv_max3_f32 %0, %1, %2, %2"
                             : "=&v"(m), "=&v"(a), "=&v"(b)
                             : "v"(s0[0]), "v"(s0[1]), "v"(s0[2]), "v"(s0[3]), "v"(s0[4]), "v"(s0[5]), "v"(s0[6]), "v"(s0[7]), "v"(s0[8]), "v"(s0[9]), "v"(s0[10]),
                               "v"(s0[11]), "v"(s0[12]), "v"(s0[13]), "v"(s0[14]), "v"(s0[15]),
                               "v"(s1[0]), "v"(s1[1]), "v"(s1[2]), "v"(s1[3]), "v"(s1[4]), "v"(s1[5]), "v"(s1[6]), "v"(s1[7]), "v"(s1[8]), "v"(s1[9]), "v"(s1[10]),
                               "v"(s1[11]), "v"(s1[12]), "v"(s1[13]), "v"(s1[14]), "v"(s1[15]));
                mx_cur[h] = m;
            });
        }
        // K(t+2) -> slot of K(t) (last read before the previous barrier), V(t+1) -> slot of V(t-1): 8 pieces per wave, issued one per step
        // inside region A (dma_piece below)
        uint32_t v_tile_off = 0;
        if (has_next) {
            const uint32_t kvn = (uint32_t)(kv0 + KVB);
            v_tile_off = kvn * 2u;
            if (seg_len) {
                const uint32_t sg = kvn / seg_len;
                v_tile_off = sg * seg_bytes + (kvn - sg * seg_len) * 2u;
            }
        }
        auto dma_piece = [&](auto jc) {  // j = 0..3: K piece j, j = 4..7: V^T piece j - 4
            constexpr int j = decltype(jc)::value;
            if constexpr (j < 4) {
                // unconditional (no branch in the stream): past the last tile the source rows are clamped to the last key row and the
                // destination slot is not read any more
                const uint32_t o = min(k_lane + (uint32_t)(kv0 + 2 * KVB + 16 * j) * k_row_bytes, k_last);
                w4_dma_piece(lds_k0 + (uint32_t)(par * KVB * HD * 2 + (wave * 64 + 256 * j) * 16), Kbytes, o);
            } else {
                const uint32_t o = v_lane + v_tile_off + 32u * (uint32_t)(j - 4) * v_row_bytes;
                w4_dma_piece(lds_v0 + (uint32_t)((par ^ 1) * HD * KVB * 2 + (wave * 64 + 256 * (j - 4)) * 16), Vbytes, o);
            }
        };
        if (__any(fmaxf(mx_cur[0], mx_cur[1]) > RESCALE_THR)) {  // rare: some row's maximum grew by more than 2^THR since its last rescale
            w4_fence_acc();
            float alpha[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float delta = fmaxf(mx_cur[h], 0.f);
                alpha[h] = __builtin_amdgcn_exp2f(-delta);
                m_run[h] += delta;
#pragma unroll
                for (int k = 0; k < 8; ++k) psum[h][k] *= alpha[h];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) S_cur[h][mb][r] -= delta;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[h][r] = -m_run[h];
            }
            static_for<0, 64>([&](auto rc) { w4_acc_scale<decltype(rc)::value>(alpha[0]); });
            static_for<64, 128>([&](auto rc) { w4_acc_scale<decltype(rc)::value>(alpha[1]); });
        }
        // ---- the instruction stream of one tile is laid out BY HAND: one wave owns the SIMD, so nothing hides a badly placed instruction
        // (MI355X_MICROARCH.md: <= 5 single-issue instructions fit under one 32-cycle MFMA). hipcc does not model the asm MFMAs (left alone it
        // issues all 128 exp2 of a tile in front of the first MFMA), hence a __builtin_amdgcn_sched_barrier(0) after every MFMA + VALU piece.
        // Work per tile and wave: 64 MFMA (32 steps x 2 halves), 32 fragment reads, 32 "pair units" (2 exp2 + 2 adds + 1 cvt_pk) and two
        // 19-instruction row-max chains.
        u32x4 pb[2][4];  // P fragments (bf16 pairs) of P.V step sl, half h
        // pair unit u = 0..31 in consumption order: slice sl = u >> 3 (P.V step that needs it), half h = (u >> 2) & 1, pair q = u & 3.
        // ONE asm statement: pinned as a block (the row-sum adds otherwise get sunk out of the tile into the next basic block, where 64 of
        // them run back to back with the matrix pipe idle), no padding between its instructions, and each exp2 result is first read two
        // instructions after it was issued (transcendental -> VALU forwarding needs one wait state).
        auto sm_unit = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int sl = u >> 3, h = (u >> 2) & 1, q = u & 3;
            constexpr int mb = sl >> 1, r = (sl & 1) * 8 + 2 * q;
            float a, b;
            uint32_t pk;
            asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %3, %3, %0\n\tv_add_f32 %4, %4, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1"
                         : "=&v"(a), "=&v"(b), "=v"(pk), "+v"(psum[h][0]), "+v"(psum[h][1])
                         : "v"(S_cur[h][mb][r]), "v"(S_cur[h][mb][r + 1]));
            pb[h][sl][q] = pk;
        };
        // operands of pair unit u
        auto ux = [&](auto uc) -> float { constexpr int u = decltype(uc)::value, sl = u >> 3, h = (u >> 2) & 1, q = u & 3; return S_cur[h][sl >> 1][(sl & 1) * 8 + 2 * q]; };
        auto uy = [&](auto uc) -> float { constexpr int u = decltype(uc)::value, sl = u >> 3, h = (u >> 2) & 1, q = u & 3; return S_cur[h][sl >> 1][(sl & 1) * 8 + 2 * q + 1]; };
        auto uput = [&](auto uc, uint32_t pk) { constexpr int u = decltype(uc)::value, sl = u >> 3, h = (u >> 2) & 1, q = u & 3; pb[h][sl][q] = pk; };

        // ---- region A: S_next[h] = K(t+1).Q_h^T: 16 fragments, each feeding both halves; step i works on key block mb = i & 1, k-step
        //      ks = i >> 1 (ALTERNATING blocks: two consecutive statements never touch the same accumulator block - no boundary pad)
        //      ||  pair units: one per step in steps 0..11 (units 0..11), two per step in steps 12..15 (units 12..19)
        if (has_next) {
            constexpr int KS = (par ^ 1) * KVB * HD * 2;
            constexpr int VS = par * HD * KVB * 2;
            static_for<0, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int n = i + RD - 1;  // fragment whose read is issued in this step: K fragment n, or V^T fragment n - 16 of region B
                constexpr bool two = i >= 12;
                constexpr int u0 = two ? 12 + 2 * (i - 12) : i;
                constexpr int u1 = two ? u0 + 1 : u0;
                constexpr int off = (n < 16) ? KS + 32 * (n & 1) * HD * 2 : VS + 32 * ((n - 16) & 3) * KVB * 2;
                const uint32_t addr = (n < 16) ? kaddr[(n >> 1) & 7] : vaddr[(n - 16) >> 2];
                using U0 = std::integral_constant<int, u0>;
                using U1 = std::integral_constant<int, u1>;
                constexpr int h0 = (u0 >> 2) & 1, h1 = (u1 >> 2) & 1, pa = 4 * (i & 1);
                uint32_t k1 = 0, k2 = 0;
                w4_step_qk<(i >> 1), off, RD - 1, (i >> 1) == 0, two>(fr[n % RD], addr, fr[i % RD], S_next[0][i & 1], S_next[1][i & 1], negm[0], negm[1], ux(U0{}),
                                                                      uy(U0{}), k1, psum[h0][pa], psum[h0][pa + 1], ux(U1{}), uy(U1{}), k2, psum[h1][pa + 2],
                                                                      psum[h1][pa + 3]);
                uput(U0{}, k1);
                if constexpr (two) uput(U1{}, k2);
                if constexpr (i < 8) dma_piece(std::integral_constant<int, i>{});
            });
            w4_fence_v(S_next[0][0], S_next[0][1], S_next[1][0], S_next[1][1]);
        } else {
            static_for<0, 20>([&](auto uc) { sm_unit(uc); });
        }
        // ---- region B: O_h^T += V^T(t).P_h^T (16 fragments: step s = i >> 2, output block d = i & 3, each feeding both halves)
        //      ||  pair units 20..31 in steps 0..11 (slice 2 complete before step 8, slice 3 before step 12)
        //      ||  the two row-max chains of tile t+1 in steps 12..15 (two pieces per step)
        {
            constexpr int VS = par * HD * KVB * 2;
            static_for<0, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int j = i + RD - 1;
                constexpr bool rd = j < 16;
                constexpr int jj = rd ? j : 0;
                constexpr bool unit = i < 12;
                using U = std::integral_constant<int, unit ? 20 + i : 20>;
                constexpr int hu = (U::value >> 2) & 1, pa = 4 * (i & 1);
                uint32_t k1 = 0;
                w4_step_pv<(i & 3), VS + 32 * (jj & 3) * KVB * 2, rd ? RD - 1 : (15 - i), rd, unit>(fr[(16 + jj) % RD], vaddr[jj >> 2], fr[(16 + i) % RD], pb[0][i >> 2],
                                                                                                 pb[1][i >> 2], ux(U{}), uy(U{}), k1, psum[hu][pa], psum[hu][pa + 1]);
                if constexpr (unit) uput(U{}, k1);
            });
        }
        if (has_next && !(G3_AB_ATTN_ABLATE & 64)) lds_dma_publish_barrier();  // drains the LDS-DMA (vmcnt(0)) and publishes K(t+2) / V(t+1)
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

    w4_fence_acc();
    static_for<0, 2>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        const float inv = 1.0f / xor32_sum(((psum[h][0] + psum[h][1]) + (psum[h][2] + psum[h][3])) + ((psum[h][4] + psum[h][5]) + (psum[h][6] + psum[h][7])));
        const int q_idx = blockIdx.x * W4_BQ + wave * 64 + 32 * h + l31;
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
