// HBM-bound companions of the DiT block: AdaLN LayerNorm+modulate, per-head RMSNorm (+RoPE) on q/k,
// the V -> V^T re-layout consumed by the attention kernel, and small vector helpers.
// All kernels move 16 bytes per lane per access (bf16x8) and keep the math in fp32.
#include "common.hpp"

namespace {

// ----------------------------------------------------------------------------------------------------------
// h = LayerNorm(x) * (1 + scale) + shift          (no affine, eps inside rsqrt)
// Reference: DITBuildingBlock.norm_state = nn.LayerNorm(x_dim, elementwise_affine=False, eps=1e-6) and
// adaln_norm_state (cosmos_predict1/diffusion/module/blocks.py:339-341, 408); FinalLayer (blocks.py:204, 239).
// rows are (s, b) pairs with b fastest: modulation row = row % mod_rows.
// ----------------------------------------------------------------------------------------------------------
constexpr int LN_THREADS = 256;
constexpr int LN_MAX_CHUNKS = 4;  // D <= 256*8*4 = 8192

// POS: first x[row] += pos_emb(row) IN PLACE ("x = x + extra_per_block_pos_emb" at the top of every block, blocks.py:547-548), then the
// LayerNorm of the updated row - one pass over x instead of two, and the [S*B, D] embedding table is never materialised: it is rebuilt
// per row from the three per-axis tables with the reference's bf16 rounding points (position_embedding.py:218-233 + normalize,
// attention.py:108-124):  e = bf16( bf16( bf16(pe_t[t] + pe_h[h]) + pe_w[w] ) / norm[s] ),  x = bf16(x + e),
// norm[s] = bf16(1e-6 + ||pe_t[t]+pe_h[h]+pe_w[w]||_2 / sqrt(D)) precomputed per token; rows are (s, b), b fastest, s = (t*Hp + h)*Wp + w.
struct PosEmbArgs {
    const bf16_t* pe_t; const bf16_t* pe_h; const bf16_t* pe_w; const bf16_t* norm;
    int Hp, Wp, B;
};

// POS: 0 plain; 1 the position embedding is rebuilt per row from its three axis tables (+ norm); 2 it is read from ONE materialised
// table pe.pe_t [rows / B][D] (already summed, normalised and rounded: a row then costs one table read instead of three + a division per element)
template <int POS>
__global__ __launch_bounds__(LN_THREADS) void ln_modulate_kernel(bf16_t* __restrict__ x, int64_t ldx,
                                                                 const bf16_t* __restrict__ shift,
                                                                 const bf16_t* __restrict__ scale, int64_t ldmod,
                                                                 int mod_rows, bf16_t* __restrict__ out, int64_t ldo,
                                                                 int rows, int D, float eps, PosEmbArgs pe) {
    __shared__ float red[2][LN_THREADS / 64];
    const int row = blockIdx.x;
    if (row >= rows) return;
    const int tid = threadIdx.x;
    const int nchunk = D >> 3;
    bf16_t* xr = x + (int64_t)row * ldx;
    const bf16_t *pt = nullptr, *ph = nullptr, *pw = nullptr;
    float pnorm = 1.f;
    if (POS == 1) {
        const int sidx = row / pe.B;
        const int wq = sidx % pe.Wp, hq = (sidx / pe.Wp) % pe.Hp, tq = sidx / (pe.Wp * pe.Hp);
        pt = pe.pe_t + (int64_t)tq * D; ph = pe.pe_h + (int64_t)hq * D; pw = pe.pe_w + (int64_t)wq * D;
        pnorm = (float)pe.norm[sidx];
    } else if (POS == 2) {
        pt = pe.pe_t + (int64_t)(row / pe.B) * D;
    }

    float v[LN_MAX_CHUNKS][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = tid + c * LN_THREADS;
        if (ch < nchunk) {
            bf16x8 t = load_bf16x8(xr + ch * 8);
            if (POS == 2) {
                const bf16x8 em = load_bf16x8(pt + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = f32_to_bf16((float)t[e] + (float)em[e]);
                store_bf16x8(xr + ch * 8, t);
            } else if (POS == 1) {
                const bf16x8 a = load_bf16x8(pt + ch * 8), b = load_bf16x8(ph + ch * 8), cw = load_bf16x8(pw + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bf16_t t1 = f32_to_bf16((float)a[e] + (float)b[e]);
                    const bf16_t t2 = f32_to_bf16((float)t1 + (float)cw[e]);
                    const bf16_t em = f32_to_bf16((float)t2 / pnorm);
                    t[e] = f32_to_bf16((float)t[e] + (float)em);
                }
                store_bf16x8(xr + ch * 8, t);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[c][e] = (float)t[e]; s += v[c][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
        }
    }
    // block reduce (sum)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) red[0][tid >> 6] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < LN_THREADS / 64; ++w) tot += red[0][w];
    const float mean = tot / (float)D;

    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = tid + c * LN_THREADS;
        if (ch < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if ((tid & 63) == 0) red[1][tid >> 6] = sq;
    __syncthreads();
    float tot2 = 0.f;
#pragma unroll
    for (int w = 0; w < LN_THREADS / 64; ++w) tot2 += red[1][w];
    const float rstd = rsqrtf(tot2 / (float)D + eps);

    const int mrow = row % mod_rows;
    const bf16_t* sh = shift + (int64_t)mrow * ldmod;
    const bf16_t* sc = scale + (int64_t)mrow * ldmod;
    bf16_t* orow = out + (int64_t)row * ldo;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = tid + c * LN_THREADS;
        if (ch < nchunk) {
            const bf16x8 tsh = load_bf16x8(sh + ch * 8);
            const bf16x8 tsc = load_bf16x8(sc + ch * 8);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = f32_to_bf16((v[c][e] - mean) * rstd * (1.0f + (float)tsc[e]) + (float)tsh[e]);
            store_bf16x8(orow + ch * 8, o);
        }
    }
}

// One WAVE per row (round 6; D = 4096 exactly: 8 chunks of 8 per lane), four rows per workgroup: the row never leaves the wave's registers, the two reductions are
// six cross-lane steps each - no LDS round trip, no workgroup barrier (the form above pays two per row). Same two-pass arithmetic (mean, then the centred squares);
// the partial sums are formed per lane over 64 elements instead of per thread over 16, so mean / variance can differ in the last fp32 bit. POS 0 / 2 as above.
template <int POS>
__global__ __launch_bounds__(256) void ln_modulate_wave_kernel(bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ shift, const bf16_t* __restrict__ scale,
                                                               int64_t ldmod, int mod_rows, bf16_t* __restrict__ out, int64_t ldo, int rows, float eps, PosEmbArgs pe) {
    constexpr int D = 4096;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    bf16_t* xr = x + (int64_t)row * ldx;
    const bf16_t* pt = POS == 2 ? pe.pe_t + (int64_t)(row / pe.B) * D : nullptr;
    float v[8][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = lane + 64 * c;
        bf16x8 t = load_bf16x8(xr + ch * 8);
        if (POS == 2) {
            const bf16x8 em = load_bf16x8(pt + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = f32_to_bf16((float)t[e] + (float)em[e]);
            store_bf16x8(xr + ch * 8, t);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[c][e] = (float)t[e]; s += v[c][e]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq / (float)D + eps);
    const int mrow = row % mod_rows;
    const bf16_t* sh = shift + (int64_t)mrow * ldmod;
    const bf16_t* sc = scale + (int64_t)mrow * ldmod;
    bf16_t* orow = out + (int64_t)row * ldo;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = lane + 64 * c;
        const bf16x8 tsh = load_bf16x8(sh + ch * 8);
        const bf16x8 tsc = load_bf16x8(sc + ch * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16((v[c][e] - mean) * rstd * (1.0f + (float)tsc[e]) + (float)tsh[e]);
        store_bf16x8(orow + ch * 8, o);
    }
}

// ----------------------------------------------------------------------------------------------------------
// Per-head RMSNorm (learned weight, eps 1e-6, fp32 math) followed by non-interleaved RoPE.
// Reference: Attention.cal_qkv (cosmos_predict1/diffusion/module/attention.py:262-280): to_q[1]/to_k[1] are
// te.pytorch.RMSNorm(128) ("R" of qkv_norm="RRI", blocks.py:292), then apply_rotary_pos_emb(.., fused=True) on
// q and k of SELF-attention only. TE semantics (transformer-engine 1.12, restated in oracle/dit_oracle.py):
//   y = x * rsqrt(mean(x^2) + eps) * w  computed in fp32, rounded to bf16;
//   rope: out = t*cos(f) + rotate_half(t)*sin(f), rotate_half(t) = cat(-t[64:], t[:64]), fp32, rounded to bf16.
// cos/sin tables are fp32 [S][128] (row = sequence position, shared by all batches/heads); nullptr = no RoPE.
// One thread owns 8 dims of the first half and the matching 8 of the second half; 8 threads per head.
// ----------------------------------------------------------------------------------------------------------
// The arithmetic of one (row, head): this thread's 8 dims of the first rotary half (a) and the matching 8 of the second (b). ONE body for both kernels
// below, so that they are bitwise equal by construction (-ffp-contract=off: no contraction differences between instantiations).
struct RopeRow { f32x4 c0, c1, c2, c3, s0, s1, s2, s3; };  // cos / sin of this thread's 8 + 8 dims at the row's sequence position

G3_DEVICE RopeRow load_rope_row(const float* cos_t, const float* sin_t, int64_t spos, int sl) {
    const float* cr = cos_t + spos * 128;
    const float* sr = sin_t + spos * 128;
    RopeRow t;
    t.c0 = *reinterpret_cast<const f32x4*>(cr + sl * 8);
    t.c1 = *reinterpret_cast<const f32x4*>(cr + sl * 8 + 4);
    t.c2 = *reinterpret_cast<const f32x4*>(cr + 64 + sl * 8);
    t.c3 = *reinterpret_cast<const f32x4*>(cr + 64 + sl * 8 + 4);
    t.s0 = *reinterpret_cast<const f32x4*>(sr + sl * 8);
    t.s1 = *reinterpret_cast<const f32x4*>(sr + sl * 8 + 4);
    t.s2 = *reinterpret_cast<const f32x4*>(sr + 64 + sl * 8);
    t.s3 = *reinterpret_cast<const f32x4*>(sr + 64 + sl * 8 + 4);
    return t;
}

template <bool ROPE>
G3_DEVICE void rmsnorm_rope_head(const bf16x8& a, const bf16x8& b, const bf16x8& wa, const bf16x8& wb, const RopeRow& t, float eps, bf16x8& oa, bf16x8& ob) {
    float fa[8], fb[8];
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        fa[e] = (float)a[e]; fb[e] = (float)b[e];
        ss += fa[e] * fa[e] + fb[e] * fb[e];
    }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    ss += __shfl_xor(ss, 4, 64);
    const float rinv = rsqrtf(ss * (1.0f / 128.0f) + eps);
    // round the normalised value to bf16 exactly where the reference does (RMSNorm output dtype)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        fa[e] = (float)f32_to_bf16(fa[e] * rinv * (float)wa[e]);
        fb[e] = (float)f32_to_bf16(fb[e] * rinv * (float)wb[e]);
    }
    if (ROPE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ca = e < 4 ? t.c0[e & 3] : t.c1[e & 3];
            const float cb = e < 4 ? t.c2[e & 3] : t.c3[e & 3];
            const float sa = e < 4 ? t.s0[e & 3] : t.s1[e & 3];
            const float sb = e < 4 ? t.s2[e & 3] : t.s3[e & 3];
            oa[e] = f32_to_bf16(fa[e] * ca - fb[e] * sa);  // first half: t*cos + (-t2)*sin
            ob[e] = f32_to_bf16(fb[e] * cb + fa[e] * sb);  // second half: t2*cos + t1*sin
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { oa[e] = f32_to_bf16(fa[e]); ob[e] = f32_to_bf16(fb[e]); }
    }
}

// `in` and `out` are NOT __restrict__: the DiT runs these kernels IN PLACE on the q | k columns of the fused QKV buffer (out == in,
// gen3c_amd/dit.py). That is sound because a thread reads exactly the 16 elements it later writes (its own 8-element slices of the two
// rotary halves; the rotate-half partner b[] is this thread's own second slice) and all of a head's loads precede its stores - an invariant
// of this body, kept visible to the compiler by leaving the two pointers possibly-aliasing.
// (general form: any H; one 8-lane group per (row, head) - every group re-reads its row's 1 KiB of cos / sin and the weights)
__global__ __launch_bounds__(256) void qk_rmsnorm_rope_kernel(const bf16_t* in, int64_t ld_in,
                                                              const bf16_t* __restrict__ w,
                                                              const float* __restrict__ cos_t,
                                                              const float* __restrict__ sin_t, bf16_t* out,
                                                              int64_t ld_out, int64_t n_pairs, int H, int B, float eps) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pair = gid >> 3;  // (row, head)
    const int sl = (int)(gid & 7);
    if (pair >= n_pairs) return;  // whole 8-lane groups exit together (n_pairs*8 granularity)
    const int64_t row = pair / H;
    const int head = (int)(pair - row * H);
    const int64_t spos = row / B;  // rows are (s, b), b fastest

    const bf16_t* src = in + row * ld_in + head * 128;
    const bf16x8 a = load_bf16x8(src + sl * 8);
    const bf16x8 b = load_bf16x8(src + 64 + sl * 8);
    const bf16x8 wa = load_bf16x8(w + sl * 8);
    const bf16x8 wb = load_bf16x8(w + 64 + sl * 8);
    bf16x8 oa, ob;
    if (cos_t != nullptr) {
        const RopeRow t = load_rope_row(cos_t, sin_t, spos, sl);
        rmsnorm_rope_head<true>(a, b, wa, wb, t, eps, oa, ob);
    } else {
        RopeRow t{};
        rmsnorm_rope_head<false>(a, b, wa, wb, t, eps, oa, ob);
    }
    bf16_t* dst = out + row * ld_out + head * 128;
    store_bf16x8(dst + sl * 8, oa);
    store_bf16x8(dst + 64 + sl * 8, ob);
}

// Octet form (round 6): one 8-lane group owns 8 CONSECUTIVE heads of a row (2 KiB contiguous) and keeps the row's cos / sin (32 registers) and the
// norm weights (8 registers) for all of them. The general form above moves 128 B of table + 32 B of weights through the vector memory path for every
// 32 B of payload read - it ran at 3.7 TB/s of payload (28.3 ms of a 3.3 s step); here the table costs 16 B per 32 B. Two weight regions
// (heads [0, h_a) with w_a, heads [h_a, h_a + h_b) with w_b: q and k of the fused QKV buffer normalised by ONE launch); h_a, h_b multiples of 8 so an
// octet never straddles them. Same arithmetic body -> bitwise equal to the general form (tests/test_kernels_gpu.py).
template <bool ROPE>
__global__ __launch_bounds__(256) void qk_rmsnorm_rope_octet_kernel(const bf16_t* in, int64_t ld_in, const bf16_t* __restrict__ w_a, int h_a,
                                                                    const bf16_t* __restrict__ w_b, int h_b, const float* __restrict__ cos_t,
                                                                    const float* __restrict__ sin_t, bf16_t* out, int64_t ld_out, int64_t rows, int B,
                                                                    float eps) {
    const int sl = threadIdx.x & 7;
    const int noct = (h_a + h_b) >> 3;
    const int64_t gid = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);  // (row, octet), octet fastest: a wave's 8 groups walk 16 KiB of one or two rows
    const int64_t row = gid / noct;
    if (row >= rows) return;
    const int oct = (int)(gid - row * noct);
    const int head0 = oct * 8;
    const bf16_t* w = head0 < h_a ? w_a : w_b;
    const bf16x8 wa = load_bf16x8(w + sl * 8);
    const bf16x8 wb = load_bf16x8(w + 64 + sl * 8);
    RopeRow t{};
    if (ROPE) t = load_rope_row(cos_t, sin_t, row / B, sl);
    const bf16_t* src = in + row * ld_in + head0 * 128 + sl * 8;
    bf16_t* dst = out + row * ld_out + head0 * 128 + sl * 8;
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // 4 heads' loads in flight, then their stores (in place: a head's loads precede its own stores)
        bf16x8 a[4], b[4];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            a[hh] = load_bf16x8(src + (half * 4 + hh) * 128);
            b[hh] = load_bf16x8(src + (half * 4 + hh) * 128 + 64);
        }
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
            bf16x8 oa, ob;
            rmsnorm_rope_head<ROPE>(a[hh], b[hh], wa, wb, t, eps, oa, ob);
            store_bf16x8(dst + (half * 4 + hh) * 128, oa);
            store_bf16x8(dst + (half * 4 + hh) * 128 + 64, ob);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------
// V [S][B][H][128] (row stride ld_in)  ->  V^T [B][H][128][ldvt], zero-filling kv in [S, ldvt).
// One workgroup transposes a 64(kv) x 128(d) tile of one (batch, head) through LDS.
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ v, int64_t ld_in,
                                                          bf16_t* __restrict__ vt, int64_t ldvt, int S, int B, int H) {
    __shared__ bf16_t tile[64][128 + 2];  // +2 elements: 65-dword row stride -> conflict-free column reads
    const int tid = threadIdx.x;
    const int kv0 = blockIdx.x * 64;
    const int head = blockIdx.y;
    const int batch = blockIdx.z;
    // load 64 rows x 16 chunks
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx >> 4, c = idx & 15;
        const int s = kv0 + r;
        bf16x8 t = zero_bf16x8();
        if (s < S) t = load_bf16x8(v + ((int64_t)s * B + batch) * ld_in + head * 128 + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r][c * 8 + e] = t[e];
    }
    __syncthreads();
    // store 128 d-rows x 8 chunks of 8 kv
    bf16_t* dst = vt + ((int64_t)batch * H + head) * 128 * ldvt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int d = idx >> 3, c = idx & 7;
        const int kv = kv0 + c * 8;
        if (kv < ldvt) {  // ldvt % 8 == 0
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = tile[c * 8 + e][d];
            store_bf16x8(dst + (int64_t)d * ldvt + kv, t);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------
// Small-M linear:  out[m][n] = act_out( sum_k act_in(a[m][k]) * w[n][k] + add[m][n] ),  M <= 8.
// Used for the timestep MLP and the AdaLN-LoRA modulation vectors (blocks.py:60-80, 411-415, 442-447):
// weights are streamed once (HBM-bound GEMV), one wave per output feature, fp32 accumulate, bf16 out.
// ----------------------------------------------------------------------------------------------------------
constexpr int GEMV_MAX_M = 8;

template <int ACT_IN>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ a, int64_t lda,
                                                   const bf16_t* __restrict__ w, int64_t ldw,
                                                   const bf16_t* __restrict__ add, int64_t ldadd,
                                                   bf16_t* __restrict__ out, int64_t ldo, int M, int N, int K) {
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float acc[GEMV_MAX_M];
#pragma unroll
    for (int m = 0; m < GEMV_MAX_M; ++m) acc[m] = 0.f;
    const bf16_t* wr = w + (int64_t)n * ldw;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        const bf16x8 wv = load_bf16x8(wr + k);
        float wf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[e] = (float)wv[e];
#pragma unroll
        for (int m = 0; m < GEMV_MAX_M; ++m) {
            if (m < M) {
                const bf16x8 av = load_bf16x8(a + (int64_t)m * lda + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = (float)av[e];
                    if (ACT_IN == 1) x = (float)f32_to_bf16(silu(x));  // reference applies nn.SiLU in bf16
                    acc[m] += x * wf[e];
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < GEMV_MAX_M; ++m) {
        if (m < M) {
            float s = acc[m];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) {
                if (add) s = (float)f32_to_bf16(s) + (float)add[(int64_t)m * ldadd + n];  // Linear output is bf16, then "+ lora"
                out[(int64_t)m * ldo + n] = f32_to_bf16(s);
            }
        }
    }
}

// x[i] += y[i]   (bf16, n % 8 == 0): "x = x + extra_per_block_pos_emb" (blocks.py:547-548)
__global__ __launch_bounds__(256) void add_inplace_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                                          int64_t nchunks) {
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * 256) {
        const bf16x8 a = load_bf16x8(x + c * 8);
        const bf16x8 b = load_bf16x8(y + c * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16((float)a[e] + (float)b[e]);
        store_bf16x8(x + c * 8, o);
    }
}

}  // namespace

extern "C" int g3_layernorm_modulate_bf16(const void* x, int64_t ldx, const void* shift, const void* scale,
                                          int64_t ldmod, int mod_rows, void* out, int64_t ldo, int rows, int D,
                                          float eps, void* stream) {
    if (!x || !shift || !scale || !out) return g3_set_error(G3_ERR_ARG, "g3_layernorm_modulate_bf16: null operand");
    if (rows <= 0 || D <= 0 || (D & 7) || D > LN_THREADS * 8 * LN_MAX_CHUNKS)
        return g3_set_error(G3_ERR_ARG, "g3_layernorm_modulate_bf16: D=%d must be a multiple of 8 and <= %d", D, LN_THREADS * 8 * LN_MAX_CHUNKS);
    if ((ldx & 7) || (ldo & 7) || (ldmod & 7) || mod_rows <= 0)
        return g3_set_error(G3_ERR_ARG, "g3_layernorm_modulate_bf16: leading dims must be multiples of 8");
    if (D == 4096 && g3_opt_ln_wave_rows)
        hipLaunchKernelGGL(ln_modulate_wave_kernel<0>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (bf16_t*)const_cast<void*>(x), ldx,
                           (const bf16_t*)shift, (const bf16_t*)scale, ldmod, mod_rows, (bf16_t*)out, ldo, rows, eps, PosEmbArgs{});
    else
        hipLaunchKernelGGL(ln_modulate_kernel<0>, dim3(rows), dim3(LN_THREADS), 0, (hipStream_t)stream, (bf16_t*)const_cast<void*>(x), ldx,
                           (const bf16_t*)shift, (const bf16_t*)scale, ldmod, mod_rows, (bf16_t*)out, ldo, rows, D, eps, PosEmbArgs{});
    return g3_check_launch("g3_layernorm_modulate_bf16");
}

extern "C" int g3_posemb_layernorm_modulate_bf16(void* x, int64_t ldx, const void* pe_t, const void* pe_h, const void* pe_w,
                                                 const void* pos_norm, int T, int Hp, int Wp, int B, const void* shift,
                                                 const void* scale, int64_t ldmod, int mod_rows, void* out, int64_t ldo, int D,
                                                 float eps, void* stream) {
    const bool materialised = pe_t && !pe_h && !pe_w && !pos_norm;  // pe_t is then the finished embedding [T*Hp*Wp][D]
    if (!x || !pe_t || !shift || !scale || !out || (!materialised && (!pe_h || !pe_w || !pos_norm)))
        return g3_set_error(G3_ERR_ARG, "g3_posemb_layernorm_modulate_bf16: null operand");
    if (T <= 0 || Hp <= 0 || Wp <= 0 || B <= 0 || D <= 0 || (D & 7) || D > LN_THREADS * 8 * LN_MAX_CHUNKS)
        return g3_set_error(G3_ERR_ARG, "g3_posemb_layernorm_modulate_bf16: bad shape (T=%d Hp=%d Wp=%d B=%d D=%d)", T, Hp, Wp, B, D);
    if ((ldx & 7) || (ldo & 7) || (ldmod & 7) || mod_rows <= 0)
        return g3_set_error(G3_ERR_ARG, "g3_posemb_layernorm_modulate_bf16: leading dims must be multiples of 8");
    const int64_t rows64 = (int64_t)T * Hp * Wp * B;
    if (rows64 > 0x7fffffff) return g3_set_error(G3_ERR_ARG, "g3_posemb_layernorm_modulate_bf16: too many rows");
    const int rows = (int)rows64;
    PosEmbArgs pe{(const bf16_t*)pe_t, (const bf16_t*)pe_h, (const bf16_t*)pe_w, (const bf16_t*)pos_norm, Hp, Wp, B};
    if (materialised && D == 4096 && g3_opt_ln_wave_rows)
        hipLaunchKernelGGL(ln_modulate_wave_kernel<2>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ldx, (const bf16_t*)shift, (const bf16_t*)scale,
                           ldmod, mod_rows, (bf16_t*)out, ldo, rows, eps, pe);
    else if (materialised)
        hipLaunchKernelGGL(ln_modulate_kernel<2>, dim3(rows), dim3(LN_THREADS), 0, (hipStream_t)stream, (bf16_t*)x, ldx,
                           (const bf16_t*)shift, (const bf16_t*)scale, ldmod, mod_rows, (bf16_t*)out, ldo, rows, D, eps, pe);
    else
        hipLaunchKernelGGL(ln_modulate_kernel<1>, dim3(rows), dim3(LN_THREADS), 0, (hipStream_t)stream, (bf16_t*)x, ldx,
                           (const bf16_t*)shift, (const bf16_t*)scale, ldmod, mod_rows, (bf16_t*)out, ldo, rows, D, eps, pe);
    return g3_check_launch("g3_posemb_layernorm_modulate_bf16");
}

static int qk_rmsnorm_rope_launch(const char* what, const void* in, int64_t ld_in, const void* w_a, int h_a, const void* w_b, int h_b, const float* cos_table,
                                  const float* sin_table, void* out, int64_t ld_out, int S, int B, int head_dim, float eps, void* stream) {
    if (!in || !w_a || !out || (h_b > 0 && !w_b)) return g3_set_error(G3_ERR_ARG, "%s: null operand", what);
    if (head_dim != 128) return g3_set_error(G3_ERR_ARG, "%s: head_dim must be 128", what);
    if ((cos_table == nullptr) != (sin_table == nullptr)) return g3_set_error(G3_ERR_ARG, "%s: need both cos and sin tables or neither", what);
    if ((ld_in & 7) || (ld_out & 7)) return g3_set_error(G3_ERR_ARG, "%s: leading dims must be multiples of 8", what);
    if (S <= 0 || B <= 0 || h_a <= 0 || h_b < 0) return g3_set_error(G3_ERR_ARG, "%s: bad shape", what);
    const int64_t rows = (int64_t)S * B;
    if (g3_opt_norm_octets && (h_a % 8) == 0 && (h_b % 8) == 0) {
        const int64_t ngroups = rows * ((h_a + h_b) >> 3);
        const int64_t nblk = (ngroups + 31) / 32;
        if (nblk > 0x7fffffff) return g3_set_error(G3_ERR_ARG, "%s: too many rows", what);
        if (cos_table)
            hipLaunchKernelGGL(qk_rmsnorm_rope_octet_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, (const bf16_t*)w_a, h_a,
                               (const bf16_t*)w_b, h_b, cos_table, sin_table, (bf16_t*)out, ld_out, rows, B, eps);
        else
            hipLaunchKernelGGL(qk_rmsnorm_rope_octet_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in, (const bf16_t*)w_a, h_a,
                               (const bf16_t*)w_b, h_b, cos_table, sin_table, (bf16_t*)out, ld_out, rows, B, eps);
        return g3_check_launch(what);
    }
    // general form: one launch per weight region
    for (int region = 0; region < (h_b > 0 ? 2 : 1); ++region) {
        const int H = region ? h_b : h_a;
        const int64_t col0 = region ? (int64_t)h_a * 128 : 0;
        const int64_t n_pairs = rows * H;
        const int64_t nblk = (n_pairs * 8 + 255) / 256;
        hipLaunchKernelGGL(qk_rmsnorm_rope_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in + col0, ld_in,
                           (const bf16_t*)(region ? w_b : w_a), cos_table, sin_table, (bf16_t*)out + col0, ld_out, n_pairs, H, B, eps);
        const int rc = g3_check_launch(what);
        if (rc) return rc;
    }
    return G3_OK;
}

extern "C" int g3_qk_rmsnorm_rope_bf16(const void* in, int64_t ld_in, const void* weight, const float* cos_table,
                                       const float* sin_table, void* out, int64_t ld_out, int S, int B, int H,
                                       int head_dim, float eps, void* stream) {
    return qk_rmsnorm_rope_launch("g3_qk_rmsnorm_rope_bf16", in, ld_in, weight, H, nullptr, 0, cos_table, sin_table, out, ld_out, S, B, head_dim, eps, stream);
}

extern "C" int g3_qk_rmsnorm_rope_pair_bf16(const void* in, int64_t ld_in, const void* weight_q, int H_q, const void* weight_k, int H_k, const float* cos_table,
                                            const float* sin_table, void* out, int64_t ld_out, int S, int B, int head_dim, float eps, void* stream) {
    if (H_k <= 0) return g3_set_error(G3_ERR_ARG, "g3_qk_rmsnorm_rope_pair_bf16: H_k must be positive (one region: g3_qk_rmsnorm_rope_bf16)");
    return qk_rmsnorm_rope_launch("g3_qk_rmsnorm_rope_pair_bf16", in, ld_in, weight_q, H_q, weight_k, H_k, cos_table, sin_table, out, ld_out, S, B, head_dim, eps, stream);
}

extern "C" int g3_transpose_v_bf16(const void* v, int64_t ld_in, void* vt, int64_t ldvt, int S, int B, int H,
                                   int head_dim, void* stream) {
    if (!v || !vt) return g3_set_error(G3_ERR_ARG, "g3_transpose_v_bf16: null operand");
    if (head_dim != 128) return g3_set_error(G3_ERR_ARG, "g3_transpose_v_bf16: head_dim must be 128");
    if ((ld_in & 7) || (ldvt & 7) || ldvt < S) return g3_set_error(G3_ERR_ARG, "g3_transpose_v_bf16: bad leading dims");
    dim3 grid((unsigned)((ldvt + 63) / 64), H, B);
    hipLaunchKernelGGL(transpose_v_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v, ld_in, (bf16_t*)vt,
                       ldvt, S, B, H);
    return g3_check_launch("g3_transpose_v_bf16");
}

extern "C" int g3_gemv_bf16(const void* a, int64_t lda, const void* w, int64_t ldw, const void* add, int64_t ldadd,
                            void* out, int64_t ldo, int M, int N, int K, int act_in, void* stream) {
    if (!a || !w || !out) return g3_set_error(G3_ERR_ARG, "g3_gemv_bf16: null operand");
    if (M <= 0 || M > GEMV_MAX_M || N <= 0 || K <= 0 || (K & 7) || (lda & 7) || (ldw & 7))
        return g3_set_error(G3_ERR_ARG, "g3_gemv_bf16: need 1<=M<=8, K,lda,ldw %% 8 == 0 (M=%d N=%d K=%d)", M, N, K);
    dim3 grid((N + 3) / 4);
    if (act_in == 0)
        hipLaunchKernelGGL(gemv_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)w, ldw,
                           (const bf16_t*)add, ldadd, (bf16_t*)out, ldo, M, N, K);
    else if (act_in == 1)
        hipLaunchKernelGGL(gemv_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)w, ldw,
                           (const bf16_t*)add, ldadd, (bf16_t*)out, ldo, M, N, K);
    else
        return g3_set_error(G3_ERR_ARG, "g3_gemv_bf16: unknown act_in %d", act_in);
    return g3_check_launch("g3_gemv_bf16");
}

extern "C" int g3_add_inplace_bf16(void* x, const void* y, int64_t n, void* stream) {
    if (!x || !y) return g3_set_error(G3_ERR_ARG, "g3_add_inplace_bf16: null operand");
    if (n <= 0 || (n & 7)) return g3_set_error(G3_ERR_ARG, "g3_add_inplace_bf16: n must be a positive multiple of 8");
    const int64_t nchunks = n >> 3;
    int64_t nblk = (nchunks + 255) / 256;
    if (nblk > 256 * 16) nblk = 256 * 16;
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, (const bf16_t*)y, nchunks);
    return g3_check_launch("g3_add_inplace_bf16");
}
