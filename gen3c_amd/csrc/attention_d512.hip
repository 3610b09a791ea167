// Single-head flash attention with head dim 512 for the tokenizer's CausalAttnBlock (bf16 in/out, fp32 softmax + accumulation).
//
// Replaces, per frame, the three kernels of round 1-3 (scores = q k^T GEMM -> 14 080 x 14 080 bf16 score matrix in HBM, 396 MB per
// frame -> row softmax in place -> P.V GEMM) behind `CausalAttnBlock.forward` (cosmos_predict1/tokenizer/modules/layers3d.py:345-383:
// time2batch, w = bmm(q^T, k) * C^-0.5, softmax(dim=2), h = bmm(v, w^T)). One pass, no score round trip, no separate softmax launch.
//
// Why this is not the DiT kernel with a bigger constant: at d = 512 the output accumulator of a 128-row query block is
// 128 x 512 fp32 = 256 KB = HALF the CU's register file, and a K / V^T tile of 64 keys is 64 KB each. A workgroup therefore is
//   4 waves, ONE per SIMD (512 registers each), 128 query rows, 64-key tiles, K and V^T single-buffered in LDS (2 x 64 KB) + P (16 KB);
//   phase A (QK^T): wave w owns query rows [32 w, 32 w + 32) with its Q slice resident in registers (32 rows x 512 dims = 128 VGPRs) and
//     computes S^T[64 keys x 32 queries] = K . Q^T over all 512 dims (64 MFMAs); online softmax (lane = one query column, as in
//     attention.hip); P goes to LDS as ready-made B fragments, the row's rescale factor next to it;
//   phase B (P.V): wave w owns OUTPUT DIMS [128 w, 128 w + 128) of all 128 query rows (16 accumulator blocks = 256 registers) and
//     multiplies V^T fragments (each feeds 4 MFMAs) with the 4 query blocks' P fragments (64 MFMAs);
//   two barriers per tile; K(t+1) streams into the K buffer (LDS-DMA) during phase B of tile t, V^T(t+1) during phase A of tile t+1;
//   deferred rescale: the running maximum moves only when a tile's maximum exceeds it by more than 8 (log2 domain), so the 256
//     accumulator registers are rescaled once or twice per query block instead of every tile; P <= 2^8, exact in bf16 / fp32.
// Arithmetic intensity per CU: 128 KB of K / V^T per 512 MFMAs = 32 B/clk at full matrix rate, the same the block GEMMs stream; phase A
// reads one 1-KiB K fragment from LDS per MFMA (the four waves read the same tile: LDS-bound there), phase B half a fragment per MFMA.
// Transposed scores with the bit-2/3 row permutation of attention.hip: the score accumulators are, in register order, the P.V B operand.
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int D5 = 512;
constexpr int BQ5 = 128;   // query rows per workgroup
constexpr int KV5 = 64;    // keys per tile
constexpr int NT5 = 256;   // threads (4 waves, one per SIMD)
constexpr float RESCALE_THR = 8.0f;
// timing ablations (tools/flash512_ab.py; results are garbage): -DG3_AB_D512_ABLATE=<bits>  1: no LDS-DMA pieces, 2: no softmax arithmetic (exp2 / sums / max),
// 4: phase A does not read its K fragments from LDS, 8: phase B does not read V^T / P fragments, 16: no barriers
#ifndef G3_AB_D512_ABLATE
#define G3_AB_D512_ABLATE 0
#endif

struct Attn512Params {
    const bf16_t* Q;   // [frames][hw][512]
    const bf16_t* K;   // [frames][hw][512]
    const bf16_t* Vt;  // [frames][512][ld_vt]  (ld_vt >= hw, zero tail not required: hw % 64 == 0)
    bf16_t* O;         // [frames][hw][512]
    int hw, frames, nqb;  // nqb = ceil(hw / 128)
    int64_t ld_vt, vt_frame;  // elements
    float scale_log2;
    int xcd_frames;  // > 0: frames % 8 == 0 -> workgroup L (XCD L % 8) works on that XCD's own frames
};

G3_DEVICE int k5_off(int row, int chunk) { return row * D5 + ((chunk ^ (row & 15)) << 3); }           // [64][512], 64 chunks per row
G3_DEVICE int v5_off(int row, int chunk) { return row * KV5 + ((chunk ^ ((row >> 1) & 7)) << 3); }   // [512][64], 8 chunks per row
G3_DEVICE int swap23_5(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// ---- O^T accumulators: ALL 256 AGPRs, owned by inline-asm statements by literal register name (block (db, qb) at a[16 (4 qb + db) : +15]).
// hipcc, given 16 accumulator blocks + a 128-register Q slice as C++ values, shuttles the accumulators between the two register files in
// every phase and spills ~500 registers (measured on the first form of this kernel); it is therefore told nothing about the AGPRs except
// through the clobber lists below, and tools/asm_audit.py-style inspection of the generated code (tests/test_asm_audit_cpu.py) checks that
// no compiler-generated instruction touches them.
#define A5_AGPRS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
template <int R> G3_DEVICE void a5_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"n"(R) : A5_AGPRS); }
template <int R> G3_DEVICE float a5_read() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "n"(R));
    return v;
}
template <int R> G3_DEVICE void a5_scale(float f) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a%c1\n\tv_mul_f32 %0, %0, %2\n\tv_accvgpr_write_b32 a%c1, %0" : "=&v"(v) : "n"(R), "v"(f));
}
// score MFMA with the accumulator in ARCHITECTURAL VGPRs ("+v"): as a builtin hipcc puts the scores into a[0:31] - on top of the O accumulators
G3_DEVICE void a5_mfma_vgpr(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// LDS fragment reads issued by hand (invisible to hipcc's lgkmcnt bookkeeping, which answers LDS-DMA in flight with lgkmcnt(0) everywhere) and
// waited for with explicit counts: DS operations complete in order, lgkmcnt(N) = "all but the N most recent have landed".
template <int OFF> G3_DEVICE void a5_lds_read(bf16x8& dst, uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N> G3_DEVICE void a5_lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// one LDS-DMA piece (1 KiB = 64 lanes x 16 B, lane-linear in LDS): destination in M0, source = wave-uniform base (SGPR pair) + 32-bit per-lane offset
G3_DEVICE void a5_dma_piece(uint32_t lds_dst, uint32_t lane_off, const char* base) {
    if (G3_AB_D512_ABLATE & 1) return;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(lane_off), "s"(base) : "memory");
}
// the 4 MFMAs of one (k-step, dim block DB) group: V^T fragment x the four query blocks' P fragments
template <int DB> G3_DEVICE void a5_pv_group(const bf16x8& vf, const bf16x8& p0, const bf16x8& p1, const bf16x8& p2, const bf16x8& p3) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c[d0]:%c[e0]], %[vf], %[p0], a[%c[d0]:%c[e0]]\n\t"
                 "v_mfma_f32_32x32x16_bf16 a[%c[d1]:%c[e1]], %[vf], %[p1], a[%c[d1]:%c[e1]]\n\t"
                 "v_mfma_f32_32x32x16_bf16 a[%c[d2]:%c[e2]], %[vf], %[p2], a[%c[d2]:%c[e2]]\n\t"
                 "v_mfma_f32_32x32x16_bf16 a[%c[d3]:%c[e3]], %[vf], %[p3], a[%c[d3]:%c[e3]]"
                 :: [vf] "v"(vf), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3),
                    [d0] "n"(16 * (0 + DB)), [e0] "n"(16 * (0 + DB) + 15), [d1] "n"(16 * (4 + DB)), [e1] "n"(16 * (4 + DB) + 15),
                    [d2] "n"(16 * (8 + DB)), [e2] "n"(16 * (8 + DB) + 15), [d3] "n"(16 * (12 + DB)), [e3] "n"(16 * (12 + DB) + 15)
                 : A5_AGPRS);
}

__global__ __launch_bounds__(NT5, 1) void spatial_attn_d512_kernel(Attn512Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem5[];
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem5);           // [64][512]
    bf16_t* sV = sK + KV5 * D5;                               // [512][64]
    bf16_t* sP = sV + D5 * KV5;                               // [4 k-steps][4 query blocks][64 lanes][8]
    float* sAlpha = reinterpret_cast<float*>(sP + 4 * 4 * 64 * 8);  // [128] rescale factor (later: 1 / row sum) per query row
    int* sFlag = reinterpret_cast<int*>(sAlpha + BQ5);        // [4] "query block qb changed its running maximum in this tile"

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int g = lane >> 5;

    int frame, qblk;
    if (p.xcd_frames > 0) {  // workgroups are dealt round-robin to the 8 XCDs: give each XCD whole frames (their K / V^T stay in its L2)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        frame = xcd * p.xcd_frames + j / p.nqb;
        qblk = j % p.nqb;
    } else {
        frame = blockIdx.x / p.nqb;
        qblk = blockIdx.x % p.nqb;
    }
    const bf16_t* Qf = p.Q + (int64_t)frame * p.hw * D5;
    const bf16_t* Kf = p.K + (int64_t)frame * p.hw * D5;
    const bf16_t* Vf = p.Vt + (int64_t)frame * p.vt_frame;
    bf16_t* Of = p.O + (int64_t)frame * p.hw * D5;

    // ---- Q slice of this wave (B operand of K . Q^T: column = query row, k = dims): 32 fragments = 128 registers, resident
    const int q_idx = qblk * BQ5 + wave * 32 + l31;
    const bool q_ok = q_idx < p.hw;
    bf16x8 qf[32];
    {
        const bf16_t* qrow = Qf + (int64_t)(q_ok ? q_idx : 0) * D5 + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) qf[ks] = q_ok ? load_bf16x8(qrow + 16 * ks) : zero_bf16x8();
    }

    // ---- LDS-DMA staging (global_load_lds_dwordx4: a wave instruction writes 64 lanes x 16 B = 1 KiB, lane-linear). The XOR swizzle of
    // the read side is applied to the SOURCE chunk. K: piece = one key row (64 chunks); V^T: piece = 8 dim rows x 8 chunks.
    // Addressing is kept OUT of long-lived registers (the Q slice owns half of them): a piece's source = wave-uniform base (SGPRs) + one
    // 32-bit per-lane offset. K: key row r of the tile is one piece, lane l fetches chunk l ^ (r & 15) = l ^ i (r = 16 wave + i): one v_xor per
    // piece, volatile so that hipcc does not hoist 16 loop-invariant offsets (it spilled them). V^T: pieces alternate between two lane patterns.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* Kbytes = reinterpret_cast<const char*>(Kf);
    const char* Vbytes = reinterpret_cast<const char*>(Vf);
    const uint32_t vrow_bytes = (uint32_t)p.ld_vt * 2u;
    const uint32_t lane16 = (uint32_t)lane << 4;
    // dim row of lane l in piece j: 8 j + (l >> 3); its swizzle term ((row >> 1) & 7) = (4 (j & 1) + (l >> 4)) & 7
    const uint32_t v_lane_even = (uint32_t)(lane >> 3) * vrow_bytes + (uint32_t)(((lane & 7) ^ ((lane >> 4) & 7)) << 4);
    const uint32_t v_lane_odd = (uint32_t)(lane >> 3) * vrow_bytes + (uint32_t)(((lane & 7) ^ ((4 + (lane >> 4)) & 7)) << 4);
    const uint32_t lds_k0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)sK;
    const uint32_t lds_v0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)sV;
    const uint32_t lds_p0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)sP;
    // piece i (0..15) of this wave: K row 16 wave + i of the tile / V^T dim rows 8 (16 wave + i) .. + 7
    auto dma_k_piece = [&](auto I, int kv0) {
        constexpr int i = decltype(I)::value;
        uint32_t off;
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(off) : "n"(i << 4), "v"(lane16));
        a5_dma_piece(lds_k0 + (uint32_t)(wave_u * 16 + i) * (D5 * 2), off, Kbytes + (int64_t)(kv0 + wave_u * 16 + i) * (D5 * 2));
    };
    auto dma_v_piece = [&](auto I, int kv0) {
        constexpr int i = decltype(I)::value;
        a5_dma_piece(lds_v0 + (uint32_t)(wave_u * 16 + i) * (8 * KV5 * 2), (i & 1) ? v_lane_odd : v_lane_even,
                     Vbytes + (int64_t)(wave_u * 16 + i) * 8 * vrow_bytes + (int64_t)kv0 * 2);
    };

    const int krow_perm = swap23_5(l31);
    static_for<0, 256>([&](auto R) { a5_zero<R.value>(); });  // O^T accumulators a[0:255]: block (dim block db, query block qb) at a[16 (4 qb + db)]
    float m_run = -1e30f;  // running maximum (log2 domain, scale included) of this lane's query row
    float l_run = 0.f;     // this lane's partial row sum (lane ^ 32 holds the rest)
    const float c = p.scale_log2;
    const int nt = p.hw / KV5;

    G3_JITTER(wave, blockIdx.x + 3);
    static_for<0, 16>([&](auto I) { dma_k_piece(I, 0); });
    if (tid < 4) sFlag[tid] = 0;
    // per-lane LDS byte addresses of the fragments (everything else is an immediate offset):
    //   K fragment (group grp, j): key row 32 (j & 1) + perm(l31), 16-B chunk 2 ks + g with ks = 2 grp + (j >> 1), swizzled by (row & 15) -> the XOR
    //   touches the low 4 chunk bits only: 8 addresses for (grp & 3, j >> 1), + 256 B per (grp >> 2), + 32 KiB for the second key block
    uint32_t kaddr[8], vaddr[4];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) kaddr[cc] = lds_k0 + (uint32_t)krow_perm * (D5 * 2) + (uint32_t)((((2 * cc + g) ^ (krow_perm & 15)) & 15) << 4);
    //   V^T fragment (k-step s, dim block db): dim row 128 wave + 32 db + l31, chunk 2 s + g swizzled by ((row >> 1) & 7) = (l31 >> 1) & 7
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) vaddr[s4] = lds_v0 + (uint32_t)(128 * wave + l31) * (KV5 * 2) + (uint32_t)(((2 * s4 + g) ^ ((l31 >> 1) & 7)) << 4);
    const uint32_t paddr = lds_p0 + lane16;  // P fragment (k-step s, query block qb) of this lane: + 1 KiB (4 s + qb)
    lds_dma_publish_barrier();

    for (int t = 0; t < nt; ++t) {
        // ================= phase A: S^T = K . Q^T for this wave's 32 query rows, online softmax, P -> LDS
        f32x16 S[2];  // (two more accumulator sets = four independent MFMA chains measured the same: 6.236 vs 6.227 ms, profiles/r4_flash512_ab.txt)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[mb][r] = 0.f;
        {
            // 16 groups of {2 dim steps x 2 key blocks} = 4 MFMAs; the next group's four K fragments are read while this group multiplies.
            // sched_barrier pins the group order: left alone hipcc hoists all 64 fragment reads (256 registers) and spills the Q slice.
#ifdef G3_AB_D512_PREFETCH2  // (A/B) K fragments two groups ahead (three buffers)
            constexpr int NKB = 3;
#else
            constexpr int NKB = 2;
#endif
            bf16x8 kf[NKB][4];
            auto rd = [&](auto GRP, bf16x8 (&dst)[4]) {
                constexpr int grp = decltype(GRP)::value;
                static_for<0, 4>([&](auto J) {
                    constexpr int j = J.value;
                    if (!(G3_AB_D512_ABLATE & 4) || grp == 0) a5_lds_read<(grp >> 2) * 256 + (j & 1) * 32768>(dst[j], kaddr[2 * (grp & 3) + (j >> 1)]);
                    else dst[j] = kf[0][j];
                });
            };
            rd(std::integral_constant<int, 0>{}, kf[0]);
            if constexpr (NKB == 3) rd(std::integral_constant<int, 1>{}, kf[1]);
            static_for<0, 16>([&](auto GRP) {
                constexpr int grp = GRP.value;
                constexpr int ahead = NKB - 1;
                if constexpr (grp + ahead < 16) {
                    rd(std::integral_constant<int, grp + ahead>{}, kf[(grp + ahead) % NKB]);
                    a5_lds_wait<4 * ahead>();  // this group's four fragments have landed (the younger ones are still in flight)
                } else {
                    a5_lds_wait<4 * (15 - grp)>();
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) a5_mfma_vgpr(S[j & 1], kf[grp % NKB][j], qf[2 * grp + (j >> 1)]);
                // V^T(t) streams into its buffer (free since the barrier that ended tile t - 1) under these MFMAs: two 1-KiB pieces per group in
                // the first half of the phase, so that the last one has most of a microsecond to land before barrier #1
#ifdef G3_AB_D512_DMA_SPREAD  // (A/B: one piece behind every group instead of two behind each of the first eight)
                dma_v_piece(std::integral_constant<int, grp>{}, t * KV5);
#else
                if constexpr (grp < 8) {
                    dma_v_piece(std::integral_constant<int, 2 * grp>{}, t * KV5);
                    dma_v_piece(std::integral_constant<int, 2 * grp + 1>{}, t * KV5);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(S[0]), "+v"(S[1]));  // MFMA results -> VALU (the statements above are opaque to hipcc's hazard recogniser)
        }
        // S[mb][r] belongs to key = 64 t + 32 mb + 16 (r >> 3) + 8 g + (r & 7), query = l31 (after the bit-2/3 row permutation)
        float mx = S[0][0];
        if (!(G3_AB_D512_ABLATE & 2)) {
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[1][r]);
            mx = fmaxf(mx, wave_xor_f32(mx, 32)) * c;
        }
        float alpha = 1.0f;
        if (mx > m_run + RESCALE_THR) {  // deferred rescale: the reference point moves only on a jump of more than 2^8
            alpha = __builtin_amdgcn_exp2f(m_run - mx);
            m_run = mx;
        }
        const bool moved = __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0;
        float psum = 0.f;
        bf16x8 pb[4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = (G3_AB_D512_ABLATE & 2) ? S[mb][r] : __builtin_amdgcn_exp2f(__builtin_fmaf(S[mb][r], c, -m_run));
                psum += pv;
                pb[2 * mb + (r >> 3)][r & 7] = f32_to_bf16(pv);
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int s = 0; s < 4; ++s) store_bf16x8(sP + ((s * 4 + wave) * 64 + lane) * 8, pb[s]);
        if (moved) {
            if (g == 0) sAlpha[wave * 32 + l31] = alpha;
            if (lane == 0) sFlag[wave] = 1;
        }
        G3_JITTER(wave + 1, t);
        if (!(G3_AB_D512_ABLATE & 16)) lds_dma_publish_barrier();  // P / alpha / flags visible; V^T(t) has landed; every wave is done reading K(t)
        const bool has_next = t + 1 < nt;

        // ================= phase B: O^T[this wave's 128 dims][128 queries] += V^T . P^T
        static_for<0, 4>([&](auto QB) {
            constexpr int qb = QB.value;
            if (__builtin_amdgcn_readfirstlane(sFlag[qb])) {  // (wave-uniform) query block qb moved its running maximum: bring its 64 accumulator registers to the new reference
                const float a = sAlpha[qb * 32 + l31];
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // (the previous tile's MFMAs into these registers have long retired; cheap insurance)
                static_for<0, 64>([&](auto R) { a5_scale<64 * qb + R.value>(a); });
                asm volatile("s_nop 7" ::: "memory");  // v_accvgpr_write -> MFMA SrcC
            }
        });
        {
            // 16 (k-step, dim block) groups of 4 MFMAs (one V^T fragment x the 4 query blocks' P fragments); the next group's V^T fragment
            // (and, at a k-step boundary, its P fragments) is read while this group multiplies.
            bf16x8 pf[2][4], vf[2];
            auto rd_p = [&](auto S_, bf16x8 (&dst)[4]) {
                static_for<0, 4>([&](auto QB) {
                    if (!(G3_AB_D512_ABLATE & 8) || decltype(S_)::value == 0) a5_lds_read<(decltype(S_)::value * 4 + QB.value) * 1024>(dst[QB.value], paddr);
                    else dst[QB.value] = pf[0][QB.value];
                });
            };
            a5_lds_read<0>(vf[0], vaddr[0]);
            rd_p(std::integral_constant<int, 0>{}, pf[0]);
            static_for<0, 16>([&](auto GRP) {
                constexpr int grp = GRP.value, s_ = grp >> 2, db = grp & 3;
                if constexpr (grp + 1 < 16) {
                    if (!(G3_AB_D512_ABLATE & 8)) a5_lds_read<((grp + 1) & 3) * 32 * KV5 * 2>(vf[(grp + 1) & 1], vaddr[(grp + 1) >> 2]);
                    else vf[(grp + 1) & 1] = vf[grp & 1];
                    if constexpr (db == 3) {
                        rd_p(std::integral_constant<int, s_ + 1>{}, pf[(s_ + 1) & 1]);
                        a5_lds_wait<5>();
                    } else {
                        a5_lds_wait<1>();
                    }
                } else {
                    a5_lds_wait<0>();
                }
                const bf16x8(&pp)[4] = pf[s_ & 1];
                a5_pv_group<db>(vf[grp & 1], pp[0], pp[1], pp[2], pp[3]);
                // K(t + 1) streams into the K buffer (free since barrier #1) under these MFMAs
#ifdef G3_AB_D512_DMA_SPREAD
                if (has_next) dma_k_piece(std::integral_constant<int, grp>{}, (t + 1) * KV5);
#else
                if constexpr (grp < 8) {
                    if (has_next) {
                        dma_k_piece(std::integral_constant<int, 2 * grp>{}, (t + 1) * KV5);
                        dma_k_piece(std::integral_constant<int, 2 * grp + 1>{}, (t + 1) * KV5);
                    }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        G3_JITTER(wave + 2, t);
        if (!(G3_AB_D512_ABLATE & 16)) lds_dma_publish_barrier();  // K(t+1) has landed and is visible; every wave is done with V^T(t), P(t), alpha(t)
        if (moved) {  // reset for the next tile (the owner wave alone writes its flag / factors; read again only after the next barrier #1)
            if (g == 0) sAlpha[wave * 32 + l31] = 1.0f;
            if (lane == 0) sFlag[wave] = 0;
        }
    }

    // ---- epilogue: 1 / row sum per query row through LDS, then each wave stores its 128 dims of every row
    const float l_tot = l_run + wave_xor_f32(l_run, 32);
    if (g == 0) sAlpha[wave * 32 + l31] = 1.0f / l_tot;
    __syncthreads();
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // last MFMAs -> v_accvgpr_read
    static_for<0, 4>([&](auto QB) {
        constexpr int qb = QB.value;
        const int q = qblk * BQ5 + qb * 32 + l31;
        if (q < p.hw) {
            const float inv = sAlpha[qb * 32 + l31];
            bf16_t* orow = Of + (int64_t)q * D5 + 128 * wave;
            // block (db, qb) register r: dim = 32 db + (r & 3) + 8 (r >> 2) + 4 g
            static_for<0, 4>([&](auto DB) {
                static_for<0, 4>([&](auto Q4) {
                    constexpr int base = 16 * (4 * qb + DB.value) + 4 * Q4.value;
                    bf16x4 o;
                    o[0] = f32_to_bf16(a5_read<base + 0>() * inv);
                    o[1] = f32_to_bf16(a5_read<base + 1>() * inv);
                    o[2] = f32_to_bf16(a5_read<base + 2>() * inv);
                    o[3] = f32_to_bf16(a5_read<base + 3>() * inv);
                    *reinterpret_cast<bf16x4*>(orow + 32 * DB.value + 8 * Q4.value + 4 * g) = o;
                });
            });
        }
    });
}

}  // namespace

// include/gen3c_hip.h
extern "C" int g3_spatial_attn_d512_bf16(const void* q, const void* k, const void* vt, int64_t ld_vt, int64_t vt_frame_stride, void* o, int frames, int hw,
                                         float softmax_scale, void* stream) {
    if (!q || !k || !vt || !o) return g3_set_error(G3_ERR_ARG, "g3_spatial_attn_d512_bf16: null operand");
    if (frames <= 0 || hw <= 0 || (hw % KV5) != 0) return g3_set_error(G3_ERR_ARG, "g3_spatial_attn_d512_bf16: frames > 0 and hw %% 64 == 0 required (hw = %d)", hw);
    if (ld_vt < hw || (ld_vt & 7) || (vt_frame_stride & 7)) return g3_set_error(G3_ERR_ARG, "g3_spatial_attn_d512_bf16: bad V^T leading dimension");
    // 32-bit quantities of the LDS-DMA addressing: a K piece's per-lane offset stays inside one key row; a V^T piece's per-lane offset spans
    // 8 dim rows ((lane >> 3) * row bytes + chunk), its base is a 64-bit pointer - so the V^T bound is 8 rows, not the 512 of a whole frame
    if ((int64_t)hw * D5 * 2 >= (1ll << 32) || (int64_t)8 * ld_vt * 2 >= (1ll << 32))
        return g3_set_error(G3_ERR_ARG, "g3_spatial_attn_d512_bf16: a frame exceeds the 32-bit byte offsets of the LDS-DMA addressing");
    Attn512Params p;
    p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.Vt = (const bf16_t*)vt; p.O = (bf16_t*)o;
    p.hw = hw; p.frames = frames; p.nqb = (hw + BQ5 - 1) / BQ5;
    p.ld_vt = ld_vt; p.vt_frame = vt_frame_stride;
    p.scale_log2 = softmax_scale * 1.44269504088896340736f;
    p.xcd_frames = (frames % 8 == 0) ? frames / 8 : 0;
    const size_t smem = (size_t)(KV5 * D5 + D5 * KV5 + 4 * 4 * 64 * 8) * sizeof(bf16_t) + BQ5 * sizeof(float) + 4 * sizeof(int);
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_attn_d512_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return g3_set_error(G3_ERR_RESOURCE, "g3_spatial_attn_d512_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(spatial_attn_d512_kernel, dim3((unsigned)(frames * p.nqb)), dim3(NT5), smem, (hipStream_t)stream, p);
    return g3_check_launch("g3_spatial_attn_d512_bf16");
}
