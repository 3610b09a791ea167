// Flash-attention forward (no mask, no dropout, head_dim 128, bf16 in/out, fp32 softmax + accumulation).
//
// Replaces TransformerEngine's DotProductAttention as used by the reference DiT
// (cosmos_predict1/diffusion/module/attention.py:228-238 constructed, :288 called: q,k,v "sbhd",
// attn_mask_type="no_mask", scale = 1/sqrt(head_dim)) for both self-attention (S_kv = S_q = 56 320 tokens) and
// cross-attention (S_kv = 512 zero-padded T5 tokens, UNMASKED: general_dit.py:407-410).
//
// MI355X-first design:
//   * workgroup = 8 wave64; each wave owns 32 query rows (block = 256 rows) and all 128 output dims;
//   * scores are computed TRANSPOSED  S^T[kv][q] = K . Q^T  with v_mfma_f32_32x32x16_bf16, so that one lane owns
//     one query column: row-max / row-sum are lane-local plus ONE cross-lane exchange (lane ^ 32);
//   * the K-tile rows are fed to the MFMA with bits 2 and 3 of the row index swapped. With that permutation the
//     accumulator registers a lane holds after QK^T are, in order, exactly the 8 consecutive kv positions the
//     lane must supply as B-operand of the PV product - P never leaves registers (no LDS round trip, no permute);
//   * V is consumed as V^T [head][128][S_kv] (written by g3_transpose_v_bf16 right after the QKV projection), so
//     the PV A-operand is a plain 16-byte LDS read like K's - no transposing reads;
//   * O is accumulated transposed (O^T[d][q]): the online-softmax rescale is a lane-local scalar multiply;
//   * K tile [64][128] is XOR-swizzled by (row & 15), V^T tile [128][64] by ((row>>1) & 7): every ds_read_b128
//     lane group covers all 64 banks (conflict-free);
//   * split staging (global->reg before the MFMAs, reg->LDS after) double-buffered, one barrier per KV tile;
//   * softmax in the exp2 domain; the 1/sqrt(d)*log2(e) factor is applied in fp32 AFTER the MFMA (no extra
//     bf16 rounding of Q);
//   * grid = (q_blocks, heads, batch): consecutive workgroups share a head, so all 8 XCD L2s stream the same
//     K/V^T panels.
#include "common.hpp"
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

constexpr int HD = 128;      // head dim
constexpr int QB = 32;       // q rows per wave
constexpr int NWAVES = 8;
constexpr int BQ = QB * NWAVES;  // 256 q rows per block
constexpr int KVB = 64;      // kv per tile
constexpr int NTHREADS = 64 * NWAVES;

struct AttnParams {
    const bf16_t* Q; int64_t q_row, q_batch, q_head;      // element strides
    const bf16_t* K; int64_t k_row, k_batch, k_head;
    const bf16_t* Vt; int64_t vt_row, vt_batch, vt_head;  // V^T[b][h][d][kv]; vt_row = leading dim (>= ceil64(S_kv))
    bf16_t* O; int64_t o_row, o_batch, o_head;
    int Sq, Skv;
    float scale_log2;  // softmax_scale * log2(e)
    int vt_seg_len;        // > 0: V^T is stored in key segments of this many keys (multiple of 64), segment s at Vt + s*vt_seg_stride
    int64_t vt_seg_stride;  // elements between segments (context parallel: rank-major all-gather of per-rank V^T shards)
    // w4b with a 1-D grid: workgroup L runs on XCD L % 8 (round-robin dispatch); xcd_heads > 0 gives every XCD its own (batch, head) pairs,
    // so a head's K / V^T panel is streamed through ONE L2 instead of all eight (grid_q query blocks per pair, n_hb = H * B pairs)
    int grid_q, n_hb, n_heads, xcd_heads;
    // split-KV ("partial") mode, O32 != nullptr: the launch covers only part of a row's keys. Instead of the bf16 output the kernel writes the
    // row's NORMALISED partial result in fp32 (element strides o_row / o_batch / o_head as O) and LSE[b][h][q] = m + log2(l) (log2 domain,
    // softmax scale included); g3_attn_merge_partials_bf16 combines the parts of a row: softmax over the union of their keys.
    float* O32;
    float* LSE;
    // zero-tail shortcut (v3 kernels; g3_cross_attn_fwd_bf16): 0 < kv_dense < Skv, a multiple of 64: the caller GUARANTEES that keys [kv_dense, Skv) have
    // all-zero K rows AND all-zero V^T columns (zero-padded T5 tokens: to_k / to_v have no bias and RMSNorm(0) = 0). Their scores are exactly 0 and their values
    // add nothing, so only the first kv_dense keys go through the tile loop and the epilogue adds the tail in closed form: m' = max(m, 0),
    // l' = l 2^(m - m') + (Skv - kv_dense) 2^(-m'), O scaled by 2^(m - m') - the same softmax over all Skv keys (they stay in the denominator,
    // general_dit.py:407-410). 0 = every key goes through the loop.
    int kv_dense;
    // per-head RMSNorm of Q inside the kernel's Q load (v3 kernels; g3_cross_attn_fwd_bf16): q_norm_w != nullptr -> every query row of a head is replaced by
    // bf16(q * rsqrt(mean(q^2) + eps) * w) before it is used (te.pytorch.RMSNorm of Attention.to_q[1], attention.py:130-131, 262-273) - the cross-attention's Q
    // then never makes the separate read + write pass of g3_qk_rmsnorm_rope_bf16.
    const bf16_t* q_norm_w;
    float q_norm_eps;
};

G3_DEVICE int k_off(int row, int chunk) { return row * HD + ((chunk ^ (row & 15)) << 3); }          // [64][128]
G3_DEVICE int v_off(int row, int chunk) { return row * KVB + ((chunk ^ ((row >> 1) & 7)) << 3); }  // [128][64]
G3_DEVICE int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// CTX: 0 = long key/value sequence (self-attention), 1 = short context (cross-attention to the 512 T5 tokens).
// Same code today; separate instantiations so profiles attribute the two very different launches separately.
template <int CTX>
__global__ __launch_bounds__(NTHREADS, 2) void flash_attn_fwd_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);  // [2][64][128]
    bf16_t* sV = sK + 2 * KVB * HD;                     // [2][128][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int g = lane >> 5;
    const int head = blockIdx.y;
    const int batch = blockIdx.z;

    const bf16_t* Qb = p.Q + batch * p.q_batch + head * p.q_head;
    const bf16_t* Kb = p.K + batch * p.k_batch + head * p.k_head;
    const bf16_t* Vb = p.Vt + batch * p.vt_batch + head * p.vt_head;
    bf16_t* Ob = p.O + batch * p.o_batch + head * p.o_head;

    // ---- Q fragments (B operand: column = q row, k = head dim), straight from global memory
    const int q_idx = blockIdx.x * BQ + wave * QB + l31;
    const bool q_ok = q_idx < p.Sq;
    bf16x8 qf[8];
    {
        const bf16_t* qrow = Qb + (int64_t)(q_ok ? q_idx : 0) * p.q_row + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = q_ok ? load_bf16x8(qrow + 16 * ks) : zero_bf16x8();
    }

    // ---- staging assignment (2 K chunks + 2 V^T chunks per thread per tile)
    bf16x8 rk[2], rv[2];
    auto stage_load = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NTHREADS * i;
            {
                const int row = idx >> 4, chunk = idx & 15;
                const bool ok = (kv0 + row) < p.Skv;
                rk[i] = ok ? load_bf16x8(Kb + (int64_t)(kv0 + row) * p.k_row + chunk * 8) : zero_bf16x8();
            }
            {
                const int row = idx >> 3, chunk = idx & 7;
                const bool ok = (kv0 + chunk * 8) < p.Skv;  // padding inside a valid chunk is zero-filled by the producer
                rv[i] = ok ? load_bf16x8(Vb + (int64_t)row * p.vt_row + kv0 + chunk * 8) : zero_bf16x8();
            }
        }
    };
    auto stage_write = [&](int buf) {
        bf16_t* dK = sK + buf * KVB * HD;
        bf16_t* dV = sV + buf * HD * KVB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NTHREADS * i;
            store_bf16x8(dK + k_off(idx >> 4, idx & 15), rk[i]);
            store_bf16x8(dV + v_off(idx >> 3, idx & 7), rv[i]);
        }
    };

    f32x16 accO[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) accO[d][r] = 0.f;
    float m_run = -1e30f;  // running max in the scaled (log2) domain
    float l_run = 0.f;     // this lane's partial row sum (partner lane^32 holds the rest)

    const int krow_perm = swap23(l31);
    const int nt = (p.Skv + KVB - 1) / KVB;
    stage_load(0);
    stage_write(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const int kv0 = t * KVB;
        if (t + 1 < nt) stage_load(kv0 + KVB);

        const bf16_t* cK = sK + buf * KVB * HD;
        const bf16_t* cV = sV + buf * HD * KVB;

        // ---- S^T = K . Q^T   (two 32-kv row blocks)
        f32x16 accS[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accS[mb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = load_bf16x8(cK + k_off(32 * mb + krow_perm, 2 * ks + g));
                accS[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], accS[mb], 0, 0, 0);
            }
        }
        // accS[mb][r] belongs to kv = kv0 + 32*mb + 16*(r>>3) + 8*g + (r&7)   (after the bit-2/3 row permutation)

        if (kv0 + KVB > p.Skv) {  // ragged last tile: mask out-of-range keys (wave-uniform branch)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + 32 * mb + 16 * (r >> 3) + 8 * g + (r & 7);
                    if (kv >= p.Skv) accS[mb][r] = -INFINITY;
                }
        }

        // ---- online softmax (one query column per lane)
        float mx = accS[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, accS[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, accS[1][r]);
        mx = fmaxf(mx, wave_xor_f32(mx, 32));
        const float m_new = fmaxf(m_run, mx * p.scale_log2);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;

        float psum = 0.f;
        bf16x8 pb[4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(accS[mb][r], p.scale_log2, -m_new));
                psum += pv;
                pb[2 * mb + (r >> 3)][r & 7] = f32_to_bf16(pv);
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) accO[d][r] *= alpha;

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bf16x8 vf = load_bf16x8(cV + v_off(32 * d + l31, 2 * s + g));
                accO[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[s], accO[d], 0, 0, 0);
            }
        }

        if (t + 1 < nt) stage_write(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: normalise and store. accO[d][r]: dim = 32d + (r&3) + 8*(r>>2) + 4*g ; query = l31
    const float l_tot = l_run + wave_xor_f32(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        bf16_t* orow = Ob + (int64_t)q_idx * p.o_row;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(accO[d][4 * q4 + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + 32 * d + 8 * q4 + 4 * g) = o;
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// v2: software-pipelined variant. Per KV tile t one straight-line region holds
//        QK^T of tile t+1 (16 MFMA)  ||  exp2 / row-sum / bf16 packing of tile t (VALU)  ->  P.V of tile t (16 MFMA)
// so the matrix pipe has independent work while the softmax VALU stream runs (the scores of tile t+1 are a second,
// statically named accumulator set). The row-max + (rare) rescale decision is taken BEFORE that region, on scores that
// are already complete, with a deferred-rescale threshold: O and l are only rescaled when some row max grew by more
// than 2^RESCALE_THR since the last rescale (P is then bounded by 2^RESCALE_THR instead of 1 - same relative bf16
// rounding, fp32 accumulation headroom is ample). K tiles run one tile ahead of V tiles in the LDS ring.
// ---------------------------------------------------------------------------------------------------------------
constexpr float RESCALE_THR = 8.0f;

G3_DEVICE float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));  // no canonicalising v_max on MFMA outputs
    return r;
}
// exchange with lane^32: after the swap `a` holds {own low | partner low}, `b` {partner high | own high}
G3_DEVICE void swap_halves(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
G3_DEVICE float xor32_max(float x) {
    float a = x, b = x;
    swap_halves(a, b);
    return max3(a, b, b);
}
G3_DEVICE float xor32_sum(float x) {
    float a = x, b = x;
    swap_halves(a, b);
    return a + b;
}

template <int CTX>
__global__ __launch_bounds__(NTHREADS, 2) void flash_attn_fwd_v2_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);  // [2][64][128]
    bf16_t* sV = sK + 2 * KVB * HD;                     // [2][128][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int g = lane >> 5;
    const int head = blockIdx.y;
    const int batch = blockIdx.z;

    const bf16_t* Qb = p.Q + batch * p.q_batch + head * p.q_head;
    const bf16_t* Kb = p.K + batch * p.k_batch + head * p.k_head;
    const bf16_t* Vb = p.Vt + batch * p.vt_batch + head * p.vt_head;
    bf16_t* Ob = p.O + batch * p.o_batch + head * p.o_head;

    const int q_idx = blockIdx.x * BQ + wave * QB + l31;
    const bool q_ok = q_idx < p.Sq;
    bf16x8 qf[8];
    {
        const bf16_t* qrow = Qb + (int64_t)(q_ok ? q_idx : 0) * p.q_row + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = q_ok ? load_bf16x8(qrow + 16 * ks) : zero_bf16x8();
    }

    bf16x8 rk[2], rv[2];
    auto load_k = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NTHREADS * i;
            const int row = idx >> 4, chunk = idx & 15;
            rk[i] = (kv0 + row) < p.Skv ? load_bf16x8(Kb + (int64_t)(kv0 + row) * p.k_row + chunk * 8) : zero_bf16x8();
        }
    };
    auto load_v = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NTHREADS * i;
            const int row = idx >> 3, chunk = idx & 7;
            rv[i] = (kv0 + chunk * 8) < p.Skv ? load_bf16x8(Vb + (int64_t)row * p.vt_row + kv0 + chunk * 8) : zero_bf16x8();
        }
    };
    auto write_k = [&](int buf) {
        bf16_t* dK = sK + buf * KVB * HD;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NTHREADS * i;
            store_bf16x8(dK + k_off(idx >> 4, idx & 15), rk[i]);
        }
    };
    auto write_v = [&](int buf) {
        bf16_t* dV = sV + buf * HD * KVB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + NTHREADS * i;
            store_bf16x8(dV + v_off(idx >> 3, idx & 7), rv[i]);
        }
    };

    const int krow_perm = swap23(l31);
    // per-lane LDS element offsets, computed once: the XOR swizzle depends on the lane's row only, so the 32-row block
    // (mb / d) and the ring slot become compile-time immediates of the ds_read
    int koff[8], voff[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = k_off(krow_perm, 2 * ks + g);   // (32*mb + row) & 15 == row & 15
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) voff[s4] = v_off(l31, 2 * s4 + g);          // ((32*d + row) >> 1) & 7 == (row >> 1) & 7
    auto qk = [&](f32x16 (&S)[2], int kbuf) {
        const bf16_t* cK = sK + kbuf * KVB * HD;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[mb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = load_bf16x8(cK + 32 * mb * HD + koff[ks]);
                S[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], S[mb], 0, 0, 0);
            }
        }
    };

    f32x16 accO[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) accO[d][r] = 0.f;
    float m_run = -1e30f;
    float l_run = 0.f;
    const float c = p.scale_log2;
    const int nt = (p.Skv + KVB - 1) / KVB;

    // ---- prologue: K0, V0 (and K1) resident; S of tile 0 computed
    load_k(0);
    load_v(0);
    write_k(0);
    write_v(0);
    if (nt > 1) {
        load_k(KVB);
        write_k(1);
    }
    __syncthreads();
    f32x16 SA[2], SB[2];
    qk(SA, 0);

    // one tile: softmax(S_cur) + PV(t), overlapped with S_next = QK^T(t+1)
    // has_next and par (= t & 1, the LDS ring slot of V(t); K(t+1) sits in slot par^1) are compile-time at every call
    // site, so ring-slot offsets fold into ds_read immediates and the ragged-tile masking exists only in the final tile.
    auto tile = [&](f32x16 (&S_cur)[2], f32x16 (&S_next)[2], int t, auto has_next_c, auto par_c) {
        constexpr bool has_next = decltype(has_next_c)::value;
        constexpr int par = decltype(par_c)::value;
        const int kv0 = t * KVB;
        if (!has_next && kv0 + KVB > p.Skv) {  // ragged tile can only be the last one
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + 32 * mb + 16 * (r >> 3) + 8 * g + (r & 7);
                    if (kv >= p.Skv) S_cur[mb][r] = -INFINITY;
                }
        }
        // row max (lane-local chain of v_max3 + one half-swap)
        float mx = max3(S_cur[0][0], S_cur[0][1], S_cur[0][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = max3(mx, S_cur[0][r], S_cur[0][r + 1]);
        mx = max3(mx, S_cur[0][15], S_cur[1][0]);
#pragma unroll
        for (int r = 1; r < 15; r += 2) mx = max3(mx, S_cur[1][r], S_cur[1][r + 1]);
        mx = fmaxf(mx, S_cur[1][15]);
        mx = xor32_max(mx) * c;
        if (__any(mx - m_run > RESCALE_THR)) {  // wave-uniform, rare after the first tiles
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) accO[d][r] *= alpha;
        }

        // ---- straight-line region
        if (has_next) {
            if (t + 2 < nt) load_k(kv0 + 2 * KVB);
            load_v(kv0 + KVB);
            __builtin_amdgcn_s_setprio(1);
            qk(S_next, par ^ 1);
            __builtin_amdgcn_s_setprio(0);
        }
        const float neg_m = -m_run;
        float psum = 0.f;
        bf16x8 pb[4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(S_cur[mb][r], c, neg_m));
                psum += pv;
                pb[2 * mb + (r >> 3)][r & 7] = f32_to_bf16(pv);
            }
        l_run += psum;
        const bf16_t* cV = sV + par * HD * KVB;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const bf16x8 vf = load_bf16x8(cV + 32 * d * KVB + voff[s]);
                accO[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[s], accO[d], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (has_next) {
            if (t + 2 < nt) write_k(par);     // K(t+2) replaces K(t)   (last read in iteration t-1)
            write_v(par ^ 1);                 // V(t+1) replaces V(t-1) (last read in iteration t-1)
            __syncthreads();
        }
    };

    using True = std::integral_constant<bool, true>;
    using False = std::integral_constant<bool, false>;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int t = 0;  // always even here
    for (; t + 2 < nt; t += 2) {
        tile(SA, SB, t, True{}, P0{});
        tile(SB, SA, t + 1, True{}, P1{});
    }
    if (t + 1 < nt) {  // two tiles left
        tile(SA, SB, t, True{}, P0{});
        tile(SB, SA, t + 1, False{}, P1{});
    } else {           // one tile left
        tile(SA, SB, t, False{}, P0{});
    }

    const float l_tot = xor32_sum(l_run);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        bf16_t* orow = Ob + (int64_t)q_idx * p.o_row;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(accO[d][4 * q4 + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + 32 * d + 8 * q4 + 4 * g) = o;
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// v3: v2's algorithm with an explicitly pipelined instruction stream.
//   * K / V^T tiles go HBM/L2 -> LDS by global_load_lds_dwordx4 (LDS-DMA): no staging VGPRs, no ds_write pass; the XOR
//     swizzle is applied to the per-lane SOURCE chunk (the LDS image a wave writes is lane-linear);
//   * every MFMA's LDS operand is read 3 MFMAs ahead through a 4-deep fragment ring in registers;
//   * the softmax VALU of tile t is sliced behind the 16 QK^T MFMAs of tile t+1 and the first 8 P.V MFMAs of tile t
//     (each P slice is ready just before the P.V step that consumes it); the row-max chain of tile t+1 rides behind the
//     last 8 P.V MFMAs; __builtin_amdgcn_sched_group_barrier pins that interleave in the emitted stream;
//   * no per-cluster s_setprio (it fences the scheduler); the second-dispatched half of the workgroup gets static prio 1.
// Needs vt_row >= ceil64(S_kv) (true for g3_transpose_v_bf16's output): the V^T tail is read, not guarded.
// ---------------------------------------------------------------------------------------------------------------
// ---- hand-counted LDS fragment reads (MW variants). With LDS-DMA (global_load_lds) in flight hipcc (ROCm 7.2) no longer counts its
// own ds_reads: every wait in the v3 tile loop comes out as `s_waitcnt lgkmcnt(0)` (22 of 22), i.e. each MFMA group also waits for
// the fragment read issued just before it and the LDS latency the 4-deep ring was built to hide is exposed 11 times per tile. The MW
// kernels therefore issue the fragment reads themselves (inline asm, invisible to the compiler's counter bookkeeping) and wait with
// explicit counts: DS operations complete in order, so `lgkmcnt(N)` = "all but the N most recent reads have landed". The wait
// statement names the fragment it guards as "+v", so the MFMA that consumes the fragment cannot be scheduled above it
// (cdna_hip_programming.md 5.7, form (ii)).
template <int OFF> G3_DEVICE void lds_read_frag(bf16x8& dst, uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// address of K fragment ks from the one of ks - 4: the swizzled chunk index (2 ks + g) ^ (row & 15) differs in bit 3 only, i.e. the byte
// address in bit 7 (the LDS ring base is 256-byte aligned - checked at kernel entry). volatile: must not be hoisted out of the tile
// loop, the point is NOT to hold these addresses in registers (3 VGPRs that otherwise spill into the loop).
G3_DEVICE uint32_t lds_addr_flip128(uint32_t a) {
    uint32_t r;
    asm volatile("v_xor_b32 %0, 0x80, %1" : "=v"(r) : "v"(a));
    return r;
}
template <int N> G3_DEVICE void lds_wait_frag(bf16x8& frag) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N)); }

#define G3_SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
constexpr int SGB_VALU = 0x2, SGB_MFMA = 0x8, SGB_DSR = 0x100, SGB_TRANS = 0x400;

// QA / QBV: VALU slots pinned behind each MFMA of region A / region B (sched_group_barrier quotas).
// Timing ablations (tools/attn_ablate.py; results are garbage): -DG3_AB_ATTN_ABLATE=<bits>  1: no exp2, 2: no row-sum adds, 4: no row-max
// chain / rescale test, 8: fragments are not read from LDS (one stale register set feeds every MFMA).
#ifndef G3_AB_ATTN_ABLATE
#define G3_AB_ATTN_ABLATE 0
#endif
template <int CTX, int QA, int QBV, bool FOLD, bool MW = false, int MW_RD = 4, int MW_KREGS = 8>
__global__ __launch_bounds__(NTHREADS, 2) void flash_attn_fwd_v3_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem_raw);  // [2][64][128]
    bf16_t* sV = sK + 2 * KVB * HD;                     // [2][128][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int g = lane >> 5;
    const int Skv = p.kv_dense > 0 ? p.kv_dense : p.Skv;  // keys that go through the tile loop (AttnParams::kv_dense: the rest is an all-zero tail)
    const int head = blockIdx.y;
    const int batch = blockIdx.z;

    const bf16_t* Qb = p.Q + batch * p.q_batch + head * p.q_head;
    const bf16_t* Kb = p.K + batch * p.k_batch + head * p.k_head;
    const bf16_t* Vb = p.Vt + batch * p.vt_batch + head * p.vt_head;
    bf16_t* Ob = p.O + batch * p.o_batch + head * p.o_head;

    const int q_idx = blockIdx.x * BQ + wave * QB + l31;
    const bool q_ok = q_idx < p.Sq;
    bf16x8 qf[8];
    {
        const bf16_t* qrow = Qb + (int64_t)(q_ok ? q_idx : 0) * p.q_row + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = q_ok ? load_bf16x8(qrow + 16 * ks) : zero_bf16x8();
        if (p.q_norm_w) {  // (wave-uniform) the row's 128 features sit in this lane's 64 elements + those of lane ^ 32
            float ss = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)qf[ks][e]; ss += f * f; }
            ss = xor32_sum(ss);
            const float rinv = rsqrtf(ss * (1.0f / 128.0f) + p.q_norm_eps);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 w8 = load_bf16x8(p.q_norm_w + 16 * ks + 8 * g);
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[ks][e] = f32_to_bf16((float)qf[ks][e] * rinv * (float)w8[e]);
            }
        }
        if (FOLD) {  // FOLD: the softmax scale (in the exp2 domain) rides on Q, the running maximum on the MFMA's C operand
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[ks][e] = f32_to_bf16((float)qf[ks][e] * p.scale_log2);
        }
    }

    // ---- LDS-DMA staging. K tile: 1024 16-B slots (row = slot>>4, chunk = slot&15); V^T tile: 1024 slots (row = slot>>3,
    // chunk = slot&7). Thread tid fills slots tid and tid+512 of each; the lane fetches the LOGICAL chunk that the
    // swizzled read side expects to find in its physical slot.
    // All per-lane source addresses are 32-bit BYTE offsets from the (wave-uniform) head base pointers: one v_add per DMA.
    const int k_row0 = tid >> 4, k_src_chunk = (tid & 15) ^ (k_row0 & 15);          // rows k_row0 and k_row0+32 share row&15
    const int v_row0 = tid >> 3, v_src_chunk = (tid & 7) ^ ((v_row0 >> 1) & 7);     // rows v_row0 and v_row0+64 share (row>>1)&7
    const char* Kbytes = reinterpret_cast<const char*>(Kb);
    const char* Vbytes = reinterpret_cast<const char*>(Vb);
    const uint32_t k_row_bytes = (uint32_t)p.k_row * 2u;
    const uint32_t k_lane = (uint32_t)k_row0 * k_row_bytes + (uint32_t)k_src_chunk * 16u;   // + kv0 * k_row_bytes (+ 32 rows)
    const uint32_t v_lane0 = (uint32_t)v_row0 * (uint32_t)p.vt_row * 2u + (uint32_t)v_src_chunk * 16u;  // + kv0 * 2
    const uint32_t v_lane1 = v_lane0 + 64u * (uint32_t)p.vt_row * 2u;
    const uint32_t k_last = (uint32_t)(Skv - 1) * k_row_bytes + (uint32_t)k_src_chunk * 16u;  // clamp target for ragged tails
    auto dma_k = [&](int kv0, int slot) {
        bf16_t* d = sK + slot * KVB * HD + wave * 64 * 8;
        uint32_t o0 = k_lane + (uint32_t)kv0 * k_row_bytes;
        uint32_t o1 = o0 + 32u * k_row_bytes;
        o0 = min(o0, k_last | 0u);  // rows past S_kv-1 re-read the last row (their scores are masked); chunk bits agree
        o1 = min(o1, k_last | 0u);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kbytes + o0),
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kbytes + o1),
                                         (__attribute__((address_space(3))) void*)(d + 512 * 8), 16, 0, 0);
    };
    const uint32_t seg_len = (uint32_t)p.vt_seg_len, seg_bytes = (uint32_t)p.vt_seg_stride * 2u;
    auto dma_v = [&](int kv0, int slot) {
        bf16_t* d = sV + slot * HD * KVB + wave * 64 * 8;
        uint32_t tile_off = (uint32_t)kv0 * 2u;
        if (seg_len) {  // a 64-key tile never straddles segments (seg_len % 64 == 0)
            const uint32_t sg = (uint32_t)kv0 / seg_len;
            tile_off = sg * seg_bytes + ((uint32_t)kv0 - sg * seg_len) * 2u;
        }
        // ONE 32-bit per-lane byte offset from the uniform head base, exactly like K: selects the SGPR-base form of global_load_lds.
        // (Written as base + lane + tile the compiler hoists two 64-bit per-lane pointers out of the loop: 4 VGPRs + 64-bit adds.)
        const uint32_t o0 = v_lane0 + tile_off, o1 = v_lane1 + tile_off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vbytes + o0),
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vbytes + o1),
                                         (__attribute__((address_space(3))) void*)(d + 512 * 8), 16, 0, 0);
    };

    const int krow_perm = swap23(l31);
    int koff[8], voff[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) koff[ks] = k_off(krow_perm, 2 * ks + g);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) voff[s4] = v_off(l31, 2 * s4 + g);
    // MW: per-lane LDS BYTE addresses of the fragments inside slot 0 of each ring (slot / block / step offsets are immediates)
    const uint32_t lds_k0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)sK;
    const uint32_t lds_v0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)sV;
    uint32_t kaddr[8], vaddr[4];
    if (MW && MW_KREGS < 8 && (lds_k0 & 255u)) __builtin_trap();  // lds_addr_flip128 needs the K ring 256-byte aligned (it is: no static LDS)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kaddr[ks] = lds_k0 + 2u * (uint32_t)koff[ks];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) vaddr[s4] = lds_v0 + 2u * (uint32_t)voff[s4];

    f32x16 accO[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) accO[d][r] = 0.f;
    float m_run = -1e30f;
    float l_run = 0.f;
    const float c = p.scale_log2;
    const int nt = (Skv + KVB - 1) / KVB;

    auto row_max = [&](const f32x16 (&S)[2]) -> float {  // two independent v_max3 chains (one per 32-kv block)
        if (G3_AB_ATTN_ABLATE & 4) return 0.f;
        float ma = max3(S[0][0], S[0][1], S[0][2]);
        float mb2 = max3(S[1][0], S[1][1], S[1][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) {
            ma = max3(ma, S[0][r], S[0][r + 1]);
            mb2 = max3(mb2, S[1][r], S[1][r + 1]);
        }
        return xor32_max(max3(ma, mb2, max3(S[0][15], S[1][15], S[1][15]))) * (FOLD ? 1.0f : c);
    };
    auto mask_tail = [&](f32x16 (&S)[2], int kv0) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + 32 * mb + 16 * (r >> 3) + 8 * g + (r & 7);
                if (kv >= Skv) S[mb][r] = -INFINITY;
            }
    };

    // ---- prologue
    G3_JITTER(wave, blockIdx.x + 5);
    dma_k(0, 0);
    dma_v(0, 0);
    if (nt > 1) dma_k(KVB, 1);
    lds_dma_publish_barrier();
    G3_JITTER(wave + 2, blockIdx.x);
    f32x16 SA[2], SB[2];
    {
        const bf16_t* cK = sK;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) SA[mb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                SA[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_bf16x8(cK + 32 * mb * HD + koff[ks]), qf[ks], SA[mb], 0, 0, 0);
        }
    }
    G3_JITTER(wave + 1, blockIdx.x);
    // K(0)'s slot is the destination of the first LDS-DMA of the tile loop (K(2)): every wave must be done reading it. Inside the
    // loop the end-of-tile barrier separates the reads of K(t) from the DMA of K(t+2); the prologue needs its own.
#ifndef G3_AB_OMIT_PROLOGUE_BARRIER  // (defined only to prove that tools/race_screen.py detects this race)
    __syncthreads();
#endif
    if (nt == 1 && KVB > Skv) mask_tail(SA, 0);
    float mx_cur = row_max(SA);
    f32x16 negm;  // FOLD: -m_run in every element: C operand of the first QK^T MFMA of a block, so scores arrive as s*c - m_run
    if (FOLD) {   // scores are kept RELATIVE to m_run from here on; the first tile fixes m_run to its exact maximum
        m_run = mx_cur;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) SA[mb][r] -= m_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_run;
        mx_cur = 0.f;
    }
#ifndef G3_AB_NO_ATTN_SETPRIO
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#endif

    auto tile = [&](f32x16 (&S_cur)[2], f32x16 (&S_next)[2], int t, auto has_next_c, auto par_c) {
        constexpr bool has_next = decltype(has_next_c)::value;
        constexpr int par = decltype(par_c)::value;
        const int kv0 = t * KVB;
        G3_JITTER(wave + blockIdx.x, t);
        if (!has_next && kv0 + KVB > Skv) {  // ragged tile can only be the last one: redo its row max on masked scores
            mask_tail(S_cur, kv0);
            mx_cur = row_max(S_cur);
        }
        if (has_next) {
            if (t + 2 < nt) dma_k(kv0 + 2 * KVB, par);  // K(t+2) -> slot of K(t)   (last read before the previous barrier)
            dma_v(kv0 + KVB, par ^ 1);                  // V(t+1) -> slot of V(t-1)
        }
        if (FOLD) {
            if (__any(mx_cur > RESCALE_THR)) {  // mx_cur is relative to m_run
                const float delta = fmaxf(mx_cur, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_run += delta;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accO[d][r] *= alpha;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) S_cur[mb][r] -= delta;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] = -m_run;
            }
        } else if (__any(mx_cur - m_run > RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mx_cur);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) accO[d][r] *= alpha;
        }
        const float neg_m = -m_run;
        float psum[4] = {0.f, 0.f, 0.f, 0.f};  // one short add chain per slice (no 32-deep dependency)
        bf16x8 pb[4];
        auto softmax_slice = [&](int sl) {  // 8 scores -> P fragment of P.V step sl
            const int mb = sl >> 1, r0 = (sl & 1) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sv = FOLD ? S_cur[mb][r0 + j] : __builtin_fmaf(S_cur[mb][r0 + j], c, neg_m);
                const float pv = (G3_AB_ATTN_ABLATE & 1) ? sv : __builtin_amdgcn_exp2f(sv);
                if (!(G3_AB_ATTN_ABLATE & 2)) psum[sl] += pv;
                pb[sl][j] = f32_to_bf16(pv);
            }
        };

        // MW: ONE fragment ring of RD registers through both regions: fragment n = 0..15 are region A's K fragments, n = 16 + j
        // region B's V^T fragments; fragment n lives in slot n % RD and is read RD-1 MFMAs before its use, so the first RD-1 V^T reads
        // are issued while region A's last MFMAs run. (RD = 3 where 4 does not fit the 256-VGPR budget without spilling into the loop -
        // a scratch reload there is fatal: its vmcnt(0) also waits for the tile's LDS-DMA.)
        constexpr int RD = MW_RD;
        bf16x8 fr[RD];
        // ---- region A: S_next = K(t+1).Q^T (16 MFMA, fragments RD-1 ahead)  ||  softmax slices 0 and 1
        if (MW && has_next) {
            constexpr int KS = (par ^ 1) * KVB * HD * 2;  // byte offset of the K slot read here
            constexpr int VS = par * HD * KVB * 2;        // ... and of the V^T slot region B reads
            auto k_addr = [&](auto ksc) -> uint32_t {  // ks <= KADDR_REGS-1: register; above: derived from ks - 4
                constexpr int ks = decltype(ksc)::value;
                if constexpr (ks < MW_KREGS) return kaddr[ks];
                else return lds_addr_flip128(kaddr[ks - 4]);
            };
            if (!FOLD) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) S_next[mb][r] = 0.f;
            }
            static_for<0, RD - 1>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_read_frag<KS + 32 * (i >> 3) * HD * 2>(fr[i % RD], k_addr(std::integral_constant<int, (i & 7)>{})); });
            static_for<0, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int n = i + RD - 1;  // fragment issued now
                if constexpr (n < 16) lds_read_frag<KS + 32 * (n >> 3) * HD * 2>(fr[n % RD], k_addr(std::integral_constant<int, (n & 7)>{}));
                else lds_read_frag<VS + 32 * ((n - 16) & 3) * KVB * 2>(fr[n % RD], vaddr[(n - 16) >> 2]);  // first V^T fragments of region B
                lds_wait_frag<RD - 1>(fr[i % RD]);  // the RD-1 younger reads (K, then V^T) may still be in flight
                S_next[i >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % RD], qf[i & 7], (FOLD && (i & 7) == 0) ? negm : S_next[i >> 3], 0, 0, 0);
                if (i == 3) softmax_slice(0);
                if (i == 11) softmax_slice(1);
            });
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                G3_SGB(SGB_MFMA, 1);
                G3_SGB(SGB_VALU, QA);
                G3_SGB(SGB_TRANS, 1);
            }
        } else if (has_next) {
            const bf16_t* cK = sK + (par ^ 1) * KVB * HD;
            bf16x8 kf[4];
#pragma unroll
            for (int i = 0; i < 3; ++i) kf[i] = load_bf16x8(cK + 32 * (i >> 3) * HD + koff[i & 7]);
            if (G3_AB_ATTN_ABLATE & 8) kf[3] = kf[0];
            if (!FOLD) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) S_next[mb][r] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + 3 < 16 && !(G3_AB_ATTN_ABLATE & 8)) kf[(i + 3) & 3] = load_bf16x8(cK + 32 * ((i + 3) >> 3) * HD + koff[(i + 3) & 7]);
                S_next[i >> 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i & 3], qf[i & 7], (FOLD && (i & 7) == 0) ? negm : S_next[i >> 3], 0, 0, 0);
                if (i == 3) softmax_slice(0);
                if (i == 11) softmax_slice(1);
            }
            G3_SGB(SGB_DSR, 3);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                G3_SGB(SGB_MFMA, 1);
                if (i + 3 < 16) G3_SGB(SGB_DSR, 1);
                G3_SGB(SGB_VALU, QA);
                G3_SGB(SGB_TRANS, 1);
            }
        } else {
            softmax_slice(0);
            softmax_slice(1);
        }

        G3_JITTER(wave + blockIdx.x + 3, t);
        // ---- region B: O^T += V^T(t).P^T (16 MFMA, fragments 3 ahead) || softmax slices 2,3 || row max of tile t+1
        if (MW && has_next) {  // (the last tile of a row block takes the compiler-managed path below: one tile in hundreds, and its
                               // different register pressure made hipcc spill an in-flight fragment register - see tools/asm_audit.py)
            constexpr int VS = par * HD * KVB * 2;
            softmax_slice(2);
            float mx_next = 0.f;
            static_for<0, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int j = i + RD - 1;  // V^T fragment issued now
                if constexpr (j < 16) lds_read_frag<VS + 32 * (j & 3) * KVB * 2>(fr[(16 + j) % RD], vaddr[j >> 2]);
                lds_wait_frag<(j < 16) ? RD - 1 : (15 - i)>(fr[(16 + i) % RD]);
                accO[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(16 + i) % RD], pb[i >> 2], accO[i & 3], 0, 0, 0);
                if (i == 3) softmax_slice(3);
                if (has_next && i == 7) mx_next = row_max(S_next);
            });
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                G3_SGB(SGB_MFMA, 1);
                G3_SGB(SGB_VALU, QBV);
                G3_SGB(SGB_TRANS, 1);
            }
            l_run += (psum[0] + psum[1]) + (psum[2] + psum[3]);
            mx_cur = mx_next;
        } else {
            const bf16_t* cV = sV + par * HD * KVB;
            bf16x8 vf[4];
            // MFMA order i: step s = i>>2, output block d = i&3
#pragma unroll
            for (int i = 0; i < 3; ++i) vf[i] = load_bf16x8(cV + 32 * (i & 3) * KVB + voff[i >> 2]);
            if (G3_AB_ATTN_ABLATE & 8) vf[3] = vf[0];
            softmax_slice(2);
            float mx_next = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + 3 < 16 && !(G3_AB_ATTN_ABLATE & 8)) vf[(i + 3) & 3] = load_bf16x8(cV + 32 * ((i + 3) & 3) * KVB + voff[(i + 3) >> 2]);
                accO[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i & 3], pb[i >> 2], accO[i & 3], 0, 0, 0);
                if (i == 3) softmax_slice(3);
                if (has_next && i == 7) mx_next = row_max(S_next);
            }
            G3_SGB(SGB_DSR, 3);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                G3_SGB(SGB_MFMA, 1);
                if (i + 3 < 16) G3_SGB(SGB_DSR, 1);
                G3_SGB(SGB_VALU, QBV);
                G3_SGB(SGB_TRANS, 1);
            }
            l_run += (psum[0] + psum[1]) + (psum[2] + psum[3]);
            mx_cur = mx_next;
        }
        if (has_next) lds_dma_publish_barrier();  // drains the LDS-DMA (vmcnt(0)) and publishes K(t+2) / V(t+1)
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

    float l_tot = xor32_sum(l_run);
    float o_scale = 1.0f;
    if (p.kv_dense > 0 && p.kv_dense < p.Skv) {  // the all-zero tail in closed form (wave-uniform branch): score 0 for each of its keys, nothing added to O
        const float m_new = fmaxf(m_run, 0.f);
        o_scale = __builtin_amdgcn_exp2f(m_run - m_new);
        l_tot = l_tot * o_scale + (float)(p.Skv - p.kv_dense) * __builtin_amdgcn_exp2f(-m_new);
        m_run = m_new;
    }
    const float inv = o_scale / l_tot;
    if (p.O32) {  // split-KV part: fp32 normalised partial + log-sum-exp (wave-uniform branch, after the loop)
        if (q_ok) {
            float* orow = p.O32 + (int64_t)blockIdx.z * p.o_batch + (int64_t)blockIdx.y * p.o_head + (int64_t)q_idx * p.o_row;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = accO[d][4 * q4 + e] * inv;
                    *reinterpret_cast<f32x4*>(orow + 32 * d + 8 * q4 + 4 * g) = o;
                }
            if (g == 0) p.LSE[((int64_t)blockIdx.z * p.n_heads + blockIdx.y) * p.Sq + q_idx] = m_run + __builtin_amdgcn_logf(l_tot);
        }
        return;
    }
    if (q_ok) {
        bf16_t* orow = Ob + (int64_t)q_idx * p.o_row;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(accO[d][4 * q4 + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + 32 * d + 8 * q4 + 4 * g) = o;
            }
    }
}



#include "attention_w4.hpp"
#include "attention_w4b.hpp"

}  // namespace

// requested: the per-call kernel choice of the *_ex entry points (0 = the process-wide "attn_variant" option, whose 0 = automatic)
static int attn_resolve_variant(int Sq, int Skv, int B, int H, int requested = 0) {
    int variant = requested ? requested : g3_opt_attn_variant;
    if (variant == 0) {
        const long n_wg = (long)((Sq + 255) / 256) * H * B;
        const long rounds = (n_wg + 255) / 256;
        const bool even_fill = n_wg * 100 >= rounds * 256 * 93;  // <= 7 % of the last round idle (one workgroup per CU)
        variant = (Skv > 2048 && (Skv % KVB) == 0 && even_fill) ? 11 : 4;
    }
    if ((variant == 10 || variant == 11) && (Skv % KVB)) variant = 9;  // w4b: whole 64-key tiles only
    return variant;
}

extern "C" const char* g3_flash_attn_kernel_name_ex(int Sq, int Skv, int B, int H, int variant_req) {
    const bool long_ctx = Skv > 2048;
    switch (attn_resolve_variant(Sq, Skv, B, H, variant_req)) {
        case 1: return long_ctx ? "flash_attn_fwd_kernel<0>" : "flash_attn_fwd_kernel<1>";
        case 2: return long_ctx ? "flash_attn_fwd_v2_kernel<0>" : "flash_attn_fwd_v2_kernel<1>";
        case 3: return long_ctx ? "flash_attn_fwd_v3_kernel<0, 6, 8, false>" : "flash_attn_fwd_v3_kernel<1, 6, 8, false>";
        case 5: return long_ctx ? "flash_attn_fwd_v3_kernel<0, 6, 8, true>" : "flash_attn_fwd_v3_kernel<1, 6, 8, true>";
        case 6: return long_ctx ? "flash_attn_fwd_v3_kernel<0, 6, 8, true, true, 4, 4>" : "flash_attn_fwd_v3_kernel<1, 6, 8, false, true>";
        case 7: return long_ctx ? "flash_attn_fwd_v3_kernel<0, 6, 8, true, true, 4, 4>" : "flash_attn_fwd_v3_kernel<1, 6, 8, true, true, 4, 4>";
        case 8: return long_ctx ? "flash_attn_fwd_v3_kernel<0, 6, 8, false, true>" : "flash_attn_fwd_v3_kernel<1, 6, 8, false, true>";
        case 9: return long_ctx ? "flash_attn_fwd_w4_kernel<0>" : "flash_attn_fwd_w4_kernel<1>";
        case 10: return "flash_attn_fwd_w4b_kernel<false>";
        case 11: return "flash_attn_fwd_w4b_kernel<true>";
        default: return long_ctx ? "flash_attn_fwd_v3_kernel<0, 6, 8, true>" : "flash_attn_fwd_v3_kernel<1, 6, 8, false>";
    }
}

extern "C" const char* g3_flash_attn_kernel_name(int Sq, int Skv, int B, int H) { return g3_flash_attn_kernel_name_ex(Sq, Skv, B, H, 0); }

static int flash_attn_launch(const void* q, int64_t q_row, int64_t q_batch, int64_t q_head, const void* k, int64_t k_row, int64_t k_batch,
                             int64_t k_head, const void* vt, int64_t vt_row, int64_t vt_batch, int64_t vt_head, int vt_seg_len,
                             int64_t vt_seg_stride, void* o, int64_t o_row, int64_t o_batch, int64_t o_head, int Sq, int Skv, int B, int H,
                             int head_dim, float softmax_scale, void* stream, int variant_req = 0, float* o_partial = nullptr, float* lse = nullptr, int kv_dense = 0,
                             const void* q_norm_w = nullptr, float q_norm_eps = 0.f) {
    if (!q || !k || !vt || (!o && !o_partial)) return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_bf16: null operand");
    if ((o_partial != nullptr) != (lse != nullptr)) return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_ex_bf16: o_partial and lse go together");
    if (o_partial && (((uintptr_t)o_partial & 15) || ((uintptr_t)lse & 3))) return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_ex_bf16: misaligned o_partial / lse");
    if (head_dim != HD) return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_bf16: head_dim %d unsupported (128 only)", head_dim);
    if (Sq <= 0 || Skv <= 0 || B <= 0 || H <= 0) return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_bf16: bad shape");
    if ((q_row & 7) || (k_row & 7) || (vt_row & 7) || (o_row & 3) || (q_batch & 7) || (k_batch & 7) || (vt_batch & 7) ||
        (q_head & 7) || (k_head & 7) || (vt_head & 7) || (o_batch & 3) || (o_head & 3))
        return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_bf16: strides must keep 16-byte (q,k,vt) / 8-byte (o) alignment");
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) || ((uintptr_t)o & 7))
        return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_bf16: misaligned pointer");
    const int kv_span = vt_seg_len > 0 ? vt_seg_len : Skv;  // keys addressed through one vt_row
    if (vt_seg_len < 0 || (vt_seg_len > 0 && ((vt_seg_len % KVB) || (Skv % vt_seg_len) || (vt_seg_stride & 7) || vt_seg_stride <= 0)))
        return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_kvseg_bf16: segments must be a multiple of 64 keys, tile S_kv exactly, 16-byte aligned stride");
    if (vt_row < ((kv_span + 7) & ~7)) return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_bf16: vt leading dim %lld < keys per row rounded to 8", (long long)vt_row);
    AttnParams p;
    p.Q = (const bf16_t*)q; p.q_row = q_row; p.q_batch = q_batch; p.q_head = q_head;
    p.K = (const bf16_t*)k; p.k_row = k_row; p.k_batch = k_batch; p.k_head = k_head;
    p.Vt = (const bf16_t*)vt; p.vt_row = vt_row; p.vt_batch = vt_batch; p.vt_head = vt_head;
    p.O = (bf16_t*)o; p.o_row = o_row; p.o_batch = o_batch; p.o_head = o_head;
    p.Sq = Sq; p.Skv = Skv;
    p.scale_log2 = softmax_scale * 1.4426950408889634f;
    p.vt_seg_len = vt_seg_len; p.vt_seg_stride = vt_seg_stride;
    p.grid_q = 0; p.n_hb = 0; p.n_heads = H; p.xcd_heads = 0;
    p.O32 = o_partial; p.LSE = lse;
    p.kv_dense = 0;
    if (kv_dense > 0) {
        const int kd = ((kv_dense + KVB - 1) / KVB) * KVB;  // whole tiles: the keys between kv_dense and the tile end are zero rows like the rest of the tail
        if (kv_dense > Skv) return g3_set_error(G3_ERR_ARG, "g3_cross_attn_fwd_bf16: kv_dense %d > S_kv %d", kv_dense, Skv);
        if (kd < Skv) p.kv_dense = kd;
    }
    p.q_norm_w = (const bf16_t*)q_norm_w; p.q_norm_eps = q_norm_eps;
    if (q_norm_w && ((uintptr_t)q_norm_w & 15)) return g3_set_error(G3_ERR_ARG, "g3_cross_attn_fwd_bf16: misaligned q_norm_weight");
    const size_t smem = (size_t)2 * (KVB * HD + HD * KVB) * sizeof(bf16_t);  // 64 KiB
    static bool attr_set[64] = {};  // per device: hipFuncSetAttribute applies to the current device only
    static std::mutex attr_mu;
    // 0 (default) = automatic: w4b with the cross-barrier prefetch (11) on long contexts made of whole 64-key tiles whose 256-row workgroups fill the chip's 256 CUs evenly,
    // else 4. Explicit values are kept for A/B runs and tests: 1 non-pipelined, 2 software-pipelined, 3 LDS-DMA + pinned interleave,
    // 4 = 3 + folded scale/max on long contexts, 5-8 test forms of 4, 9 = w4 (one wave per SIMD), 10 = w4b, 11 = w4b + cross-barrier prefetch.
    int variant = attn_resolve_variant(Sq, Skv, B, H, variant_req);
    if (o_partial && variant == 9) variant = 4;  // w4 (ragged S_kv) has no partial epilogue: the 8-wave kernel takes ragged tiles too
    if ((p.kv_dense || p.q_norm_w) && (variant < 3 || variant >= 9)) variant = 4;  // the zero-tail epilogue and the Q norm live in the v3 kernels
    if (variant >= 3 && vt_row < ((kv_span + KVB - 1) / KVB) * KVB) variant = 2;  // (also 6-8)  // v3 reads the whole last V^T tile unguarded
    if (variant >= 3) {
        // v3 addresses its K / V^T LDS-DMA sources with 32-bit BYTE offsets from the per-(batch, head) base pointers: the largest
        // offsets it forms must stay below 4 GiB, otherwise they wrap silently (e.g. a strided K view of a fused [S*B, 3*4096]
        // QKV buffer at B >= 4). Out-of-range problems run on v2 (64-bit addressing) when V^T is not segmented.
        const uint64_t k_max = (uint64_t)(Skv - 1 + 2 * KVB) * (uint64_t)k_row * 2u + 256u;
        const uint64_t n_seg = vt_seg_len > 0 ? (uint64_t)(Skv / vt_seg_len) : 1u;
        const uint64_t v_max = (uint64_t)(HD - 1) * (uint64_t)vt_row * 2u + (n_seg - 1) * (uint64_t)vt_seg_stride * 2u + ((uint64_t)kv_span + KVB) * 2u + 256u;
        if (k_max > 0xFFFFFFFFull || v_max > 0xFFFFFFFFull) {
            if (vt_seg_len > 0)
                return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_kvseg_bf16: K / V^T span exceeds the kernel's 32-bit byte offsets (k %llu, vt %llu bytes)",
                                    (unsigned long long)k_max, (unsigned long long)v_max);
            variant = 2;
        }
    }
    if (vt_seg_len > 0 && variant < 3)
        return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_kvseg_bf16: segmented V^T is implemented by the default (v3) kernel only");
    if (o_partial && variant < 3)
        return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_ex_bf16: split-KV partial outputs are implemented by the v3 / w4b kernels only (variant %d chosen)", variant);
    if ((p.kv_dense || p.q_norm_w) && variant < 3)  // (a V^T leading dimension or an operand span that forced the 64-bit-addressing kernel above)
        return g3_set_error(G3_ERR_ARG, "g3_cross_attn_fwd_bf16: the Q norm / zero-tail form is implemented by the v3 kernels only (variant %d chosen: V^T leading dimension below "
                                        "ceil64(S_kv), or operands beyond the kernel's 32-bit byte offsets)", variant);
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 64) return g3_set_error(G3_ERR_LAUNCH, "flash_attn: hipGetDevice failed");
    {
      std::lock_guard<std::mutex> attr_lock(attr_mu);
      if (!attr_set[dev_id]) {
        const void* fns[16] = {reinterpret_cast<const void*>(&flash_attn_fwd_w4b_kernel<false>), reinterpret_cast<const void*>(&flash_attn_fwd_w4b_kernel<true>),
                               reinterpret_cast<const void*>(&flash_attn_fwd_kernel<0>), reinterpret_cast<const void*>(&flash_attn_fwd_kernel<1>),
                               reinterpret_cast<const void*>(&flash_attn_fwd_v2_kernel<0>), reinterpret_cast<const void*>(&flash_attn_fwd_v2_kernel<1>),
                               reinterpret_cast<const void*>(&flash_attn_fwd_v3_kernel<0, 6, 8, false>), reinterpret_cast<const void*>(&flash_attn_fwd_v3_kernel<1, 6, 8, false>),
                               reinterpret_cast<const void*>(&flash_attn_fwd_v3_kernel<0, 6, 8, true>), reinterpret_cast<const void*>(&flash_attn_fwd_v3_kernel<1, 6, 8, true>),
                               reinterpret_cast<const void*>(&flash_attn_fwd_v3_kernel<0, 6, 8, true, true, 4, 4>), reinterpret_cast<const void*>(&flash_attn_fwd_v3_kernel<1, 6, 8, true, true, 4, 4>),
                               reinterpret_cast<const void*>(&flash_attn_fwd_v3_kernel<0, 6, 8, false, true>), reinterpret_cast<const void*>(&flash_attn_fwd_v3_kernel<1, 6, 8, false, true>),
                               reinterpret_cast<const void*>(&flash_attn_fwd_w4_kernel<0>), reinterpret_cast<const void*>(&flash_attn_fwd_w4_kernel<1>)};
        for (int i = 0; i < 16; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "flash_attn: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        attr_set[dev_id] = true;
      }
    }
    dim3 grid((Sq + BQ - 1) / BQ, H, B);
    const bool long_ctx = Skv > 2048;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 10 || variant == 11) {  // w4 with the trimmed issue stream (attention_w4b.hpp)
        dim3 grid4((Sq + W4_BQ - 1) / W4_BQ, H, B);
        p.grid_q = (int)grid4.x; p.n_hb = H * B; p.n_heads = H;
        p.xcd_heads = (g3_opt_attn_xcd_heads && (H * B) % 8 == 0) ? 1 : 0;
        if (p.xcd_heads) grid4 = dim3(grid4.x * H * B, 1, 1);
        if (variant == 11) hipLaunchKernelGGL(flash_attn_fwd_w4b_kernel<true>, grid4, dim3(W4_THREADS), smem, st, p);
        else hipLaunchKernelGGL(flash_attn_fwd_w4b_kernel<false>, grid4, dim3(W4_THREADS), smem, st, p);
        return g3_check_launch("g3_flash_attn_fwd_bf16");
    }
    if (variant == 9) {  // one wave per SIMD, 64 query rows per wave (attention_w4.hpp)
        dim3 grid4((Sq + W4_BQ - 1) / W4_BQ, H, B);
        if (long_ctx) hipLaunchKernelGGL(flash_attn_fwd_w4_kernel<0>, grid4, dim3(W4_THREADS), smem, st, p);
        else hipLaunchKernelGGL(flash_attn_fwd_w4_kernel<1>, grid4, dim3(W4_THREADS), smem, st, p);
        return g3_check_launch("g3_flash_attn_fwd_bf16");
    }
#define G3_LAUNCH_ATTN(KERNEL0, KERNEL1) do { if (long_ctx) hipLaunchKernelGGL(KERNEL0, grid, dim3(NTHREADS), smem, st, p); else hipLaunchKernelGGL(KERNEL1, grid, dim3(NTHREADS), smem, st, p); } while (0)
    // VALU quotas (6, 8) per MFMA measured best of {(4,4), (5,6), (6,8)} (profiles/r1_v5_attn_quota_ab.txt)
    if (variant == 1) G3_LAUNCH_ATTN(flash_attn_fwd_kernel<0>, flash_attn_fwd_kernel<1>);
    else if (variant == 2) G3_LAUNCH_ATTN(flash_attn_fwd_v2_kernel<0>, flash_attn_fwd_v2_kernel<1>);
    else if (variant == 3) G3_LAUNCH_ATTN((flash_attn_fwd_v3_kernel<0, 6, 8, false>), (flash_attn_fwd_v3_kernel<1, 6, 8, false>));
    else if (variant == 5) G3_LAUNCH_ATTN((flash_attn_fwd_v3_kernel<0, 6, 8, true>), (flash_attn_fwd_v3_kernel<1, 6, 8, true>));  // tests: folded arithmetic at every length
    else if (variant == 6) G3_LAUNCH_ATTN((flash_attn_fwd_v3_kernel<0, 6, 8, true, true, 4, 4>), (flash_attn_fwd_v3_kernel<1, 6, 8, false, true>));  // hand-counted LDS waits
    else if (variant == 7) G3_LAUNCH_ATTN((flash_attn_fwd_v3_kernel<0, 6, 8, true, true, 4, 4>), (flash_attn_fwd_v3_kernel<1, 6, 8, true, true, 4, 4>));   // tests: MW + fold at every length
    else if (variant == 8) G3_LAUNCH_ATTN((flash_attn_fwd_v3_kernel<0, 6, 8, false, true>), (flash_attn_fwd_v3_kernel<1, 6, 8, false, true>)); // tests: MW, unfolded
    else G3_LAUNCH_ATTN((flash_attn_fwd_v3_kernel<0, 6, 8, true>), (flash_attn_fwd_v3_kernel<1, 6, 8, false>));  // short contexts: the fold's prologue does not pay
#undef G3_LAUNCH_ATTN
    return g3_check_launch("g3_flash_attn_fwd_bf16");
}

extern "C" int g3_flash_attn_fwd_bf16(const void* q, int64_t q_row, int64_t q_batch, int64_t q_head, const void* k,
                                      int64_t k_row, int64_t k_batch, int64_t k_head, const void* vt, int64_t vt_row,
                                      int64_t vt_batch, int64_t vt_head, void* o, int64_t o_row, int64_t o_batch,
                                      int64_t o_head, int Sq, int Skv, int B, int H, int head_dim, float softmax_scale,
                                      void* stream) {
    return flash_attn_launch(q, q_row, q_batch, q_head, k, k_row, k_batch, k_head, vt, vt_row, vt_batch, vt_head, 0, 0, o, o_row, o_batch, o_head,
                             Sq, Skv, B, H, head_dim, softmax_scale, stream);
}

extern "C" int g3_flash_attn_fwd_kvseg_bf16(const void* q, int64_t q_row, int64_t q_batch, int64_t q_head, const void* k, int64_t k_row,
                                            int64_t k_batch, int64_t k_head, const void* vt, int64_t vt_row, int64_t vt_batch,
                                            int64_t vt_head, int vt_seg_len, int64_t vt_seg_stride, void* o, int64_t o_row, int64_t o_batch,
                                            int64_t o_head, int Sq, int Skv, int B, int H, int head_dim, float softmax_scale, void* stream) {
    if (vt_seg_len <= 0) return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_kvseg_bf16: vt_seg_len must be positive");
    return flash_attn_launch(q, q_row, q_batch, q_head, k, k_row, k_batch, k_head, vt, vt_row, vt_batch, vt_head, vt_seg_len, vt_seg_stride, o, o_row,
                             o_batch, o_head, Sq, Skv, B, H, head_dim, softmax_scale, stream);
}

/* ---- cross-attention form: optional per-head RMSNorm of Q in the kernel's Q load (AttnParams::q_norm_w) and optional all-zero key tail (AttnParams::kv_dense) ---- */
extern "C" int g3_cross_attn_fwd_bf16(const void* q, int64_t q_row, int64_t q_batch, int64_t q_head, const void* q_norm_weight, float q_norm_eps, const void* k,
                                      int64_t k_row, int64_t k_batch, int64_t k_head, const void* vt, int64_t vt_row, int64_t vt_batch, int64_t vt_head, void* o,
                                      int64_t o_row, int64_t o_batch, int64_t o_head, int Sq, int Skv, int kv_dense, int B, int H, int head_dim,
                                      float softmax_scale, void* stream) {
    if (kv_dense < 0) return g3_set_error(G3_ERR_ARG, "g3_cross_attn_fwd_bf16: kv_dense must be >= 0 (0 = every key through the loop)");
    return flash_attn_launch(q, q_row, q_batch, q_head, k, k_row, k_batch, k_head, vt, vt_row, vt_batch, vt_head, 0, 0, o, o_row, o_batch, o_head, Sq, Skv, B, H,
                             head_dim, softmax_scale, stream, 0, nullptr, nullptr, kv_dense, q_norm_weight, q_norm_eps);
}

/* ---- per-call kernel choice + split-KV partial outputs ------------------------------------------------------------------------------------ */
extern "C" int g3_flash_attn_fwd_ex_bf16(const void* q, int64_t q_row, int64_t q_batch, int64_t q_head, const void* k, int64_t k_row,
                                         int64_t k_batch, int64_t k_head, const void* vt, int64_t vt_row, int64_t vt_batch, int64_t vt_head,
                                         int vt_seg_len, int64_t vt_seg_stride, void* o, float* o_partial, float* lse, int64_t o_row,
                                         int64_t o_batch, int64_t o_head, int Sq, int Skv, int B, int H, int head_dim, float softmax_scale,
                                         int variant, void* stream) {
    if (variant < 0 || variant > 11) return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_ex_bf16: variant %d out of range", variant);
    if (o && o_partial) return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_ex_bf16: pass either o (bf16 result) or o_partial + lse, not both");
    if (o_partial && ((o_row & 3) || (o_batch & 3) || (o_head & 3))) return g3_set_error(G3_ERR_ARG, "g3_flash_attn_fwd_ex_bf16: o_partial strides must be multiples of 4");
    return flash_attn_launch(q, q_row, q_batch, q_head, k, k_row, k_batch, k_head, vt, vt_row, vt_batch, vt_head, vt_seg_len, vt_seg_stride, o, o_row,
                             o_batch, o_head, Sq, Skv, B, H, head_dim, softmax_scale, stream, variant, o_partial, lse);
}

namespace {
// out[row][h*128 + d] = sum_i w_i O_i[row][h*128 + d],  w_i = 2^(lse_i - max_j lse_j) / sum_j 2^(lse_j - max): the softmax over the union of the
// parts' keys. One thread per 8 output elements (two 16-byte loads per part, one 16-byte store); HBM-streaming.
constexpr int MERGE_MAX_PARTS = 8;
struct MergeParams {
    const float* o[MERGE_MAX_PARTS];
    const float* lse[MERGE_MAX_PARTS];
    int n_parts;
    int64_t p_row, p_batch, p_head;  // strides of the fp32 parts (elements)
    bf16_t* out; int64_t o_row, o_batch, o_head;
    int Sq, B, H;
};
__global__ __launch_bounds__(256) void attn_merge_partials_kernel(MergeParams p) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (q, b, h, c) with c = 8-element chunk of the head dim, c fastest
    const int c = (int)(idx & 15);
    int64_t r = idx >> 4;
    const int h = (int)(r % p.H); r /= p.H;
    const int b = (int)(r % p.B);
    const int64_t qi = r / p.B;
    if (qi >= p.Sq) return;
    const int64_t li = ((int64_t)b * p.H + h) * p.Sq + qi;
    float l[MERGE_MAX_PARTS], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MERGE_MAX_PARTS; ++i)
        if (i < p.n_parts) { l[i] = p.lse[i][li]; mx = fmaxf(mx, l[i]); }
    float wsum = 0.f;
#pragma unroll
    for (int i = 0; i < MERGE_MAX_PARTS; ++i)
        if (i < p.n_parts) { l[i] = __builtin_amdgcn_exp2f(l[i] - mx); wsum += l[i]; }
    const float inv = 1.0f / wsum;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t poff = qi * p.p_row + (int64_t)b * p.p_batch + (int64_t)h * p.p_head + 8 * c;
#pragma unroll
    for (int i = 0; i < MERGE_MAX_PARTS; ++i)
        if (i < p.n_parts) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.o[i] + poff);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.o[i] + poff + 4);
            const float w = l[i] * inv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += w * a[e]; acc[4 + e] += w * bb[e]; }
        }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(acc[e]);
    store_bf16x8(p.out + qi * p.o_row + (int64_t)b * p.o_batch + (int64_t)h * p.o_head + 8 * c, o);
}
}  // namespace

extern "C" int g3_attn_merge_partials_bf16(const float* const* o_parts, const float* const* lse_parts, int n_parts, int64_t p_row, int64_t p_batch,
                                           int64_t p_head, void* out, int64_t o_row, int64_t o_batch, int64_t o_head, int Sq, int B, int H,
                                           int head_dim, void* stream) {
    if (!o_parts || !lse_parts || !out) return g3_set_error(G3_ERR_ARG, "g3_attn_merge_partials_bf16: null operand");
    if (n_parts < 1 || n_parts > MERGE_MAX_PARTS) return g3_set_error(G3_ERR_ARG, "g3_attn_merge_partials_bf16: 1..%d parts", MERGE_MAX_PARTS);
    if (head_dim != HD || Sq <= 0 || B <= 0 || H <= 0) return g3_set_error(G3_ERR_ARG, "g3_attn_merge_partials_bf16: bad shape");
    if ((p_row & 3) || (p_batch & 3) || (p_head & 3) || (o_row & 7) || (o_batch & 7) || (o_head & 7) || ((uintptr_t)out & 15))
        return g3_set_error(G3_ERR_ARG, "g3_attn_merge_partials_bf16: strides / pointers must keep 16-byte alignment");
    MergeParams p;
    for (int i = 0; i < MERGE_MAX_PARTS; ++i) {
        p.o[i] = i < n_parts ? o_parts[i] : nullptr;
        p.lse[i] = i < n_parts ? lse_parts[i] : nullptr;
        if (i < n_parts && (!p.o[i] || !p.lse[i] || ((uintptr_t)p.o[i] & 15))) return g3_set_error(G3_ERR_ARG, "g3_attn_merge_partials_bf16: part %d null / misaligned", i);
    }
    p.n_parts = n_parts; p.p_row = p_row; p.p_batch = p_batch; p.p_head = p_head;
    p.out = (bf16_t*)out; p.o_row = o_row; p.o_batch = o_batch; p.o_head = o_head; p.Sq = Sq; p.B = B; p.H = H;
    const int64_t total = (int64_t)Sq * B * H * 16;
    hipLaunchKernelGGL(attn_merge_partials_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return g3_check_launch("g3_attn_merge_partials_bf16");
}
