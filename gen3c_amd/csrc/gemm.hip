// bf16 "NT" GEMM for the DiT linears:  C[M,N] = epi( A[M,K] . W[N,K]^T )
//
// Replaces the nn.Linear(bias=False) calls of the reference's DiT blocks
// (cosmos_predict1/diffusion/module/attention.py:61-62,207-223; blocks.py:160-162,205-206) together with
// the elementwise tail each of them feeds (GELU: attention.py:94-99; gated residual: blocks.py:455-471).
//
// MI355X design (not a port of any CUDA tiling):
//   * one workgroup = 8 wave64 = 512 threads computes a 256(token) x 256(feature) tile, K-step 64;
//   * v_mfma_f32_32x32x16_bf16, operands swapped (A-operand = weight rows, B-operand = token rows) so that a
//     lane ends up owning 4 CONSECUTIVE output features of one token row -> 8-byte stores, and per-feature
//     epilogue vectors (gate) are lane-local;
//   * both operands have K contiguous, so every fragment is one 16-byte LDS read; both MFMA inputs use the
//     same lane->k mapping, so the contraction is independent of the hardware's k order inside a fragment;
//   * LDS tiles are [rows][64] bf16 (128-B rows) with the 16-B chunk index XOR-ed by (row>>1)&7, which makes
//     every 16-lane ds_read_b128 group hit 16 distinct 16-B slots of the 256-B bank row (conflict-free);
//   * global->register->LDS staging is split (issue the loads for tile t+1 before the MFMAs of tile t, write
//     them to the other LDS buffer afterwards) so HBM/L2 latency hides under 32 MFMAs per wave; one barrier/tile;
//   * blockIdx is remapped so that the 8 XCDs each own a contiguous band of token tiles (private L2 reuse of
//     the weight panel).
#include "common.hpp"
#include "gen3c_hip.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

constexpr int BM = 256;  // tokens per block tile
constexpr int BN = 256;  // features per block tile
constexpr int BK = 64;
constexpr int NTHREADS = 512;

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_GATED_RESIDUAL = 2, EPI_BIAS = 3, EPI_BIAS_RESIDUAL = 4,
       EPI_QK_NORM_ROPE = 5 };  // 5: per-head RMSNorm (+ RoPE) on the q / k feature ranges (gemm_w4.hpp only; g3_gemm_qk_norm_rope_bf16)

// Implicit-GEMM convolution geometry (channels-last activations [T][H][W][C], one batch item):
// output row m = (to, yo, xo); tap (dt, dy, dx) reads input position
//   ti = max(to*st + ot + dt, 0)   (causal: the first frame is replicated in front - CausalConv3d._replication_pad)
//   yi = yo*sh + oh + dy, xi = xo*sw + ow + dx   (outside [0,Hi) x [0,Wi) -> zero padding)
// and multiplies with the tap's [N][K] weight slab.
struct ConvGeom {
    int To, Ho, Wo, Ti, Hi, Wi;
    int kt, kh, kw, st, sh, sw, ot, oh, ow;
    int ntaps;
    int64_t w_tap_stride;  // elements between consecutive tap slabs of W
    int tap_gaps;          // gemm_w4_conv.hpp: compute a new tap's token addresses in the MFMA gaps of the barrier K step (option conv_w4 = 2: off, for A/B runs)
};
__device__ __attribute__((aligned(16))) unsigned g3_zero_page[2048];  // 8 KiB of zeros: source of padded taps on the LDS-DMA path (the one-wave kernel walks up to K * 2 bytes into it)

struct GemmParams {
    int tile_order_rowmajor;
    int wide_store;  // C / residual / gate rows are 16-byte addressable: epilogue goes through the LDS transpose (full-line stores)
    const bf16_t* A; int64_t lda;
    const bf16_t* W; int64_t ldw;
    bf16_t* C; int64_t ldc;
    int M, N, K;
    const bf16_t* gate; int gate_rows; int64_t ldg;  // gate[(m % gate_rows)][n]  (EPI_GATED_RESIDUAL) / bias (EPI_BIAS)
    const bf16_t* R; int64_t ldr;                     // residual rows
    int tiles_m, tiles_n;
    ConvGeom cv;
    // EPI_QK_NORM_ROPE: features [0, n_q) are q heads (weight nw_q), [n_q, n_q + n_k) k heads (nw_k), the rest is stored as is
    const bf16_t* nw_q; const bf16_t* nw_k; const float* rope_cos; const float* rope_sin; int n_q, n_k, rope_B; float rms_eps;
    bf16_t* vt; int64_t vt_ld, vt_batch; int vt_S;  // optional V^T destination of the remaining (v) heads: [B][H_v][128][vt_ld], S valid positions
    // gemm_w4_conv.hpp: optional GroupNorm statistics of the output ([frames][2] doubles: sum, sum of squares; frame = gn_rows consecutive rows)
    double* gn_stats = nullptr; int gn_rows = 0;
};

G3_DEVICE int lds_off(int row, int chunk) {  // element offset in a [rows][64] bf16 tile
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

// ---- epilogue shared by the kernels below. acc[i][j][r]: feature = nw + 32 i + (r & 3) + 8 (r >> 2) + 4 g ; token = mw + 32 j + l31
template <int EPI>
G3_DEVICE void store_tile(const GemmParams& p, f32x16 (&acc)[4][2], int mw, int nw, int l31, int g) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = mw + 32 * j + l31;
        if (m >= p.M) continue;
        bf16_t* crow = p.C + (int64_t)m * p.ldc;
        const bf16_t* rrow = (EPI == EPI_GATED_RESIDUAL || EPI == EPI_BIAS_RESIDUAL) ? (p.R + (int64_t)m * p.ldr) : nullptr;
        const bf16_t* grow = (EPI == EPI_GATED_RESIDUAL || EPI == EPI_BIAS || EPI == EPI_BIAS_RESIDUAL)
                                 ? (p.gate + (int64_t)(m % p.gate_rows) * p.ldg)
                                 : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int n = nw + 32 * i + 8 * q4 + 4 * g;
                if (n >= p.N) continue;  // N % 4 == 0 is required by the host wrapper
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q4 + e];
                // GELU / gated residual act on the Linear's OUTPUT, which nn.Linear rounds to bf16 (attention.py:94-99, blocks.py:455-471); every GEMM
                // kernel rounds here since round 5 (gemm_w4e.hpp keeps a finished tile as packed bf16): bitwise equal outputs across the kernels
                if (EPI == EPI_GELU || EPI == EPI_GATED_RESIDUAL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)f32_to_bf16(v[e]);
                }
                if (EPI == EPI_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf_fast(v[e]);
                } else if (EPI == EPI_GATED_RESIDUAL) {
                    const bf16x4 gv = *reinterpret_cast<const bf16x4*>(grow + n);
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rrow + n);
                    // reference order (blocks.py:456): block output is rounded to bf16 by its Linear, then
                    // gate*out and x+.. ; we keep fp32 until the single final rounding.
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)rv[e] + (float)gv[e] * v[e];
                } else if (EPI == EPI_BIAS) {
                    const bf16x4 gv = *reinterpret_cast<const bf16x4*>(grow + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)gv[e];
                } else if (EPI == EPI_BIAS_RESIDUAL) {
                    const bf16x4 gv = *reinterpret_cast<const bf16x4*>(grow + n);
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rrow + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (v[e] + (float)gv[e]) + (float)rv[e];
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(v[e]);
                *reinterpret_cast<bf16x4*>(crow + n) = o;
            }
        }
    }
}

// Full-line epilogue: the MFMA layout gives a lane 4 features of 32 different token rows, so direct stores touch 32 cache
// lines with 16 bytes each per instruction (measured ~2.5 TB/s of C traffic). Instead each wave transposes its tile through
// a PRIVATE 16 KiB slice of the (now idle) operand LDS, one 32-token half at a time, as fp32 [32 tokens][128 features]
// with the 16-byte chunk index XOR-ed by the row, and reads it back row-major: a lane then owns 8 consecutive features
// (one 16-byte bf16 store), 16 lanes cover a token row's 256 bytes, and residual/gate loads coalesce the same way.
// No barrier: LDS operations of one wave execute in order and the slice is not shared. Same arithmetic as store_tile.
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
template <int EPI>
G3_DEVICE void store_tile_lds(const GemmParams& p, f32x16 (&acc)[4][2], int mw, int nw, int lane, char* stage) {
    const int l31 = lane & 31, g = lane >> 5;
    const int rsub = lane >> 4, c2 = lane & 15;
    const int n = nw + 8 * c2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q4 + e];
                *reinterpret_cast<f32x4*>(stage + l31 * 512 + (((8 * i + 2 * q4 + g) ^ l31) << 4)) = v;
            }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int row = 4 * s + rsub;
            const int m = mw + 32 * j + row;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * 512 + (((2 * c2) ^ row) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * 512 + (((2 * c2 + 1) ^ row) << 4));
            if (m >= p.M || n >= p.N) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = lo[e];
                v[4 + e] = hi[e];
            }
            if (EPI == EPI_GELU || EPI == EPI_GATED_RESIDUAL) {  // the Linear's own rounding to bf16 (see store_tile)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)f32_to_bf16(v[e]);
            }
            if (EPI == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_erf_fast(v[e]);
            } else if (EPI != EPI_NONE) {
                const bf16x8 gv = load_bf16x8(p.gate + (int64_t)(p.gate_rows == 1 ? 0 : m % p.gate_rows) * p.ldg + n);
                if (EPI == EPI_GATED_RESIDUAL) {
                    const bf16x8 rv = load_bf16x8(p.R + (int64_t)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)rv[e] + (float)gv[e] * v[e];
                } else if (EPI == EPI_BIAS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)gv[e];
                } else if (EPI == EPI_BIAS_RESIDUAL) {
                    const bf16x8 rv = load_bf16x8(p.R + (int64_t)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (v[e] + (float)gv[e]) + (float)rv[e];
                }
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(v[e]);
            store_bf16x8(p.C + (int64_t)m * p.ldc + n, o);
        }
    }
}

// STAGE_GLDS = true : tiles go HBM/L2 -> LDS directly with global_load_lds_dwordx4 (no staging VGPRs, no ds_write
//                     pass); the LDS image is lane-linear per wave, so the XOR swizzle is applied to the per-lane
//                     SOURCE address (same involution as the read side). Needs K % 64 == 0 (no zero-fill on this path;
//                     M/N tails read a clamped valid row whose results are never stored).
// STAGE_GLDS = false: global -> VGPR -> ds_write_b128 with zero-fill guards (any K % 8 == 0).
// CONV = true: A rows are gathered per tap according to p.cv (implicit GEMM); K is the per-tap channel count.
template <int EPI, bool STAGE_GLDS, bool CONV, bool PIN>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_nt_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem_raw);  // [2][BM][BK]
    bf16_t* sW = sA + 2 * BM * BK;                      // [2][BN][BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int g = lane >> 5;

    // XCD-aware tile order: dispatch puts block b on XCD b%8; give each XCD a contiguous run of a global tile order
    // (bijective for any tile count). The order itself is built for the private 4 MiB L2s: token tiles are taken in
    // super-rows of GM=4 and swept feature-tile by feature-tile (token tile fastest), so the ~32 workgroups an XCD
    // runs concurrently form an ~(4 token x 8 feature) block: 12 operand panels feed 32 tiles instead of 33.
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + slot;
    }
    int tile_m, tile_n;
    if (p.tile_order_rowmajor) {
        tile_m = bid / p.tiles_n;
        tile_n = bid - tile_m * p.tiles_n;
    } else {
        constexpr int GM = 4;
        const int per_group = GM * p.tiles_n;
        const int grp = bid / per_group;
        const int within = bid - grp * per_group;
        const int gm = min(GM, p.tiles_m - grp * GM);  // last super-row may be shorter
        tile_n = within / gm;
        tile_m = grp * GM + (within - tile_n * gm);
    }
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- staging assignment: 4 chunks of A and 4 of W per thread per K tile
    const int ld_chunk = tid & 7;
    const int ld_row = tid >> 3;  // + 64*i
    const bf16_t* a_ptr[4];
    const bf16_t* w_ptr[4];
    bool a_ok[4], w_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = ld_row + 64 * i;
        a_ok[i] = (m0 + row) < p.M;
        w_ok[i] = (n0 + row) < p.N;
        a_ptr[i] = p.A + (int64_t)(a_ok[i] ? (m0 + row) : 0) * p.lda + ld_chunk * 8;
        w_ptr[i] = p.W + (int64_t)(w_ok[i] ? (n0 + row) : 0) * p.ldw + ld_chunk * 8;
    }

    // conv: per-row base input coordinates (before adding the tap offset)
    int cbt[4], cby[4], cbx[4];
    if (CONV) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ld_row + 64 * i;
            const int mm = m < p.M ? m : 0;
            const int to = mm / (p.cv.Ho * p.cv.Wo);
            const int rem = mm - to * (p.cv.Ho * p.cv.Wo);
            const int yo = rem / p.cv.Wo;
            const int xo = rem - yo * p.cv.Wo;
            cbt[i] = to * p.cv.st + p.cv.ot;
            cby[i] = yo * p.cv.sh + p.cv.oh;
            cbx[i] = xo * p.cv.sw + p.cv.ow;
        }
    }
    // source row of slot i for tap `tap`, or -1 for a zero (padded / out-of-range) row
    auto conv_src_row = [&](int i, int tap) -> int64_t {
        const int khw = p.cv.kh * p.cv.kw;
        const int dt = tap / khw;
        const int r2 = tap - dt * khw;
        const int dy = r2 / p.cv.kw;
        const int dx = r2 - dy * p.cv.kw;
        int ti = cbt[i] + dt;
        ti = ti < 0 ? 0 : ti;
        const int yi = cby[i] + dy, xi = cbx[i] + dx;
        const bool ok = a_ok[i] && ti < p.cv.Ti && yi >= 0 && yi < p.cv.Hi && xi >= 0 && xi < p.cv.Wi;
        return ok ? ((int64_t)ti * p.cv.Hi + yi) * p.cv.Wi + xi : (int64_t)-1;
    };
    const int nkc = (p.K + BK - 1) / BK;  // K tiles per tap

    bf16x8 ra[4], rw[4];
    auto stage_load = [&](int it) {
        const int tap = CONV ? it / nkc : 0;
        const int k0 = (CONV ? it - tap * nkc : it) * BK;
        const bool k_ok = (k0 + ld_chunk * 8) < p.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (CONV) {
                const int64_t src = conv_src_row(i, tap);
                ra[i] = (src >= 0 && k_ok) ? load_bf16x8(p.A + src * p.lda + ld_chunk * 8 + k0) : zero_bf16x8();
                rw[i] = (w_ok[i] && k_ok) ? load_bf16x8(w_ptr[i] + tap * p.cv.w_tap_stride + k0) : zero_bf16x8();
            } else {
                ra[i] = (a_ok[i] && k_ok) ? load_bf16x8(a_ptr[i] + k0) : zero_bf16x8();
                rw[i] = (w_ok[i] && k_ok) ? load_bf16x8(w_ptr[i] + k0) : zero_bf16x8();
            }
        }
    };
    auto stage_write = [&](int buf) {
        bf16_t* dA = sA + buf * BM * BK;
        bf16_t* dW = sW + buf * BN * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ld_row + 64 * i;
            store_bf16x8(dA + lds_off(row, ld_chunk), ra[i]);
            store_bf16x8(dW + lds_off(row, ld_chunk), rw[i]);
        }
    };
    // direct-to-LDS staging: this lane fills physical slot (row = ld_row + 64 i, chunk = ld_chunk) of the tile, i.e. it
    // must fetch LOGICAL chunk ld_chunk ^ ((row >> 1) & 7) = ld_chunk ^ ((tid >> 4) & 7)  (64 i does not touch bits 1..3).
    const int src_chunk = ld_chunk ^ ((tid >> 4) & 7);
    const bf16_t* ga[4];
    const bf16_t* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = ld_row + 64 * i;
        const int ra_row = (m0 + row) < p.M ? (m0 + row) : (p.M - 1);
        const int rw_row = (n0 + row) < p.N ? (n0 + row) : (p.N - 1);
        ga[i] = p.A + (int64_t)ra_row * p.lda + src_chunk * 8;
        gw[i] = p.W + (int64_t)rw_row * p.ldw + src_chunk * 8;
    }
    auto stage_glds = [&](int it, int buf) {
        const int tap = CONV ? it / nkc : 0;
        const int k0 = (CONV ? it - tap * nkc : it) * BK;
        bf16_t* dA = sA + buf * BM * BK + wave * 64 * 8;  // wave-uniform base; the hardware adds lane*16 bytes
        bf16_t* dW = sW + buf * BN * BK + wave * 64 * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* asrc;
            const bf16_t* wsrc;
            if (CONV) {
                const int64_t src = conv_src_row(i, tap);
                asrc = src >= 0 ? p.A + src * p.lda + src_chunk * 8 + k0 : reinterpret_cast<const bf16_t*>(g3_zero_page) + src_chunk * 8;
                wsrc = gw[i] + tap * p.cv.w_tap_stride + k0;
            } else {
                asrc = ga[i] + k0;
                wsrc = gw[i] + k0;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)asrc,
                                             (__attribute__((address_space(3))) void*)(dA + i * 512 * 8), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc,
                                             (__attribute__((address_space(3))) void*)(dW + i * 512 * 8), 16, 0, 0);
        }
    };

    // ---- wave tile: 128 features x 64 tokens
    const int wn = wave & 1;
    const int wm = wave >> 1;
    const int n_w0 = wn * 128;
    const int m_w0 = wm * 64;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = CONV ? nkc * p.cv.ntaps : nkc;
    if (STAGE_GLDS) {
        stage_glds(0, 0);
    } else {
        stage_load(0);
        stage_write(0);
    }
    lds_dma_publish_barrier();

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        G3_JITTER(wave + blockIdx.x, t);
        if (t + 1 < nk) {
            if (STAGE_GLDS) stage_glds(t + 1, buf ^ 1);  // buf^1 was last read in iteration t-1 (barrier passed)
            else stage_load(t + 1);
        }

        const bf16_t* cA = sA + buf * BM * BK;
        const bf16_t* cW = sW + buf * BN * BK;
        // register double-buffered fragments: the ds_reads of k-step ks+1 are issued before the MFMAs of k-step ks
        bf16x8 wf[2][4], af[2][2];
        auto load_frags = [&](int ks, int slot) {
            const int chunk = 2 * ks + g;
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[slot][i] = load_bf16x8(cW + lds_off(n_w0 + 32 * i + l31, chunk));
#pragma unroll
            for (int j = 0; j < 2; ++j) af[slot][j] = load_bf16x8(cA + lds_off(m_w0 + 32 * j + l31, chunk));
        };
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) load_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][i], af[ks & 1][j], acc[i][j], 0, 0, 0);
        }
        // pin the stream: the 6 fragment reads of k-step ks+1 ride behind the first 6 of the 8 MFMAs of k-step ks, so
        // only the first k-step after the barrier waits on LDS latency (0x8 = MFMA, 0x100 = DS read)
        if (PIN) {
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    if (ks + 1 < 4 && m < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }

        if (!STAGE_GLDS && t + 1 < nk) stage_write(buf ^ 1);
        lds_dma_publish_barrier();  // tile t+1 has landed for every wave
    }

    if (p.wide_store) store_tile_lds<EPI>(p, acc, m0 + m_w0, n0 + n_w0, lane, smem_raw + wave * 16384);
    else store_tile<EPI>(p, acc, m0 + m_w0, n0 + n_w0, l31, g);
}


// ---------------------------------------------------------------------------------------------------------------
// Ping-pong variant of the plain GEMM (no conv, K % 64 == 0) - the default for the DiT linears.
//
// Same 256 x 256 x 64 block tile, same 128(feature) x 64(token) wave tile and accumulator layout as the kernel above, but
// the K loop is cut into 4 PHASES per K tile, one accumulator quadrant (64 features x 32 tokens, 8 MFMAs = 256 MFMA
// cycles) each, and the two waves that share a SIMD run half a phase apart:
//
//     load section : fragment ds_reads of this phase (W half: 8, token half: 4) + ONE half-tile of LDS-DMA (2 per lane)
//     s_waitcnt vmcnt(8) ; s_barrier
//     MFMA section : s_setprio 1 ; 8 MFMAs ; s_setprio 0
//     s_barrier
//
// Waves 4..7 execute one extra s_barrier up front, so while waves 0..3 (one per SIMD) are in their MFMA section waves
// 4..7 are in their load section and vice versa: the 60-180 cycle issue cost of each LDS-DMA instruction and the LDS
// latency hide behind the other wave's MFMAs instead of stalling the matrix pipe.
//
// Operand tiles are staged as HALF-TILES of 128 rows x 64 k (16 KiB: W rows with bit 6 = h, token rows with bit 5 = h),
// so that they can be retired and restaged one at a time. Half-tiles are numbered q = 4 t + {0: W0, 1: T0, 2: T1, 3: W1}
// in the order they are needed; phase f = 4 t + P issues q = f + 6 and reads (P0: q = 4t, 4t+1; P1: 4t+2; P2: 4t+3),
// LDS holds two K tiles (2 x 4 x 16 KiB = 128 KiB), and the vmcnt before a phase's first barrier leaves the 4 newest
// half-tiles in flight - everything the NEXT phase reads has landed for every wave once that barrier (plus the stagger
// barrier) is passed; a buffer is restaged no earlier than two phases after its last ds_read.
// ---------------------------------------------------------------------------------------------------------------
#ifdef G3_AB_NO_GEMM_SETPRIO
#define G3_PP_SETPRIO(x) ((void)0)
#else
#define G3_PP_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
template <int N> G3_DEVICE void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
G3_DEVICE void wait_vmcnt_rt(int halftiles) {  // tail phases: the number of half-tiles that may stay in flight
    switch (halftiles) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<2>(); break;
        case 2: wait_vmcnt<4>(); break;
        case 3: wait_vmcnt<6>(); break;
        default: wait_vmcnt<8>(); break;
    }
}
G3_DEVICE void phase_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int EPI, int PH, bool CONV>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_nt_pp_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // [slot 2][region 4][128 rows][64] bf16
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int g = lane >> 5;

    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + slot;
    }
    int tile_m, tile_n;
    if (p.tile_order_rowmajor) {
        tile_m = bid / p.tiles_n;
        tile_n = bid - tile_m * p.tiles_n;
    } else {
        constexpr int GM = 4;
        const int per_group = GM * p.tiles_n;
        const int grp = bid / per_group;
        const int within = bid - grp * per_group;
        const int gm = min(GM, p.tiles_m - grp * GM);
        tile_n = within / gm;
        tile_m = grp * GM + (within - tile_n * gm);
    }
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int wn = wave & 1;   // feature half of the block tile
    const int wm = wave >> 1;  // token quarter of the block tile
    const int n_w0 = wn * 128, m_w0 = wm * 64;

    // ---- LDS-DMA sources. A half-tile is 1024 16-B slots, 2 per lane: slot = j*512 + tid -> local row j*64 + (tid >> 3),
    // physical chunk tid & 7, which holds logical chunk (tid & 7) ^ ((row >> 1) & 7).
    //   W half h : local row r <-> feature n0 + (r >> 6)*128 + h*64 + (r & 63)
    //   T half h : local row r <-> token   m0 + (r >> 5)*64  + h*32 + (r & 31)
    const int src_chunk = (tid & 7) ^ ((tid >> 4) & 7);
    const bf16_t* src[4][2];  // [region: W0, T0, T1, W1][j]   (CONV: only the W regions; token rows are gathered per tap)
    int crd_t[2][2], crd_yx[2][2];  // CONV: per (token half, j) base input coordinates: t, and (y << 16) | (x & 0xffff)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = j * 64 + (tid >> 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nrow = min(n0 + (r >> 6) * 128 + h * 64 + (r & 63), p.N - 1);
            const int mrow = min(m0 + (r >> 5) * 64 + h * 32 + (r & 31), p.M - 1);
            src[h ? 3 : 0][j] = p.W + (int64_t)nrow * p.ldw + src_chunk * 8;
            if (CONV) {
                const int to = mrow / (p.cv.Ho * p.cv.Wo);
                const int rem = mrow - to * (p.cv.Ho * p.cv.Wo);
                const int yo = rem / p.cv.Wo;
                const int xo = rem - yo * p.cv.Wo;
                crd_t[h][j] = to * p.cv.st + p.cv.ot;
                crd_yx[h][j] = ((yo * p.cv.sh + p.cv.oh) << 16) | ((xo * p.cv.sw + p.cv.ow) & 0xffff);
                src[h ? 2 : 1][j] = nullptr;
            } else {
                src[h ? 2 : 1][j] = p.A + (int64_t)mrow * p.lda + src_chunk * 8;
            }
        }
    }
    // CONV: K tile `tile` = (tap, channel tile kc); every region is issued once per K tile in increasing order, so each keeps
    // its own (kc, dt, dy, dx) odometer in scalars instead of dividing. Weights of tap (dt,dy,dx) start at W + tap*w_tap_stride.
    const int nkc = CONV ? p.K / BK : 1;
    int od_kc[4] = {0, 0, 0, 0}, od_tap[4] = {0, 0, 0, 0}, od_dt[4] = {0, 0, 0, 0}, od_dy[4] = {0, 0, 0, 0}, od_dx[4] = {0, 0, 0, 0};
    auto issue = [&](int region, int tile) __attribute__((always_inline)) {  // region compile-time after inlining
        char* dst = smem_raw + ((tile & 1) << 16) + region * 16384 + wave * 1024;
        if (!CONV) {
            const int k0 = tile * BK;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[region][j] + k0),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, 0, 0);
            return;
        }
        const int k0 = od_kc[region] * BK;
        const bool is_w = region == 0 || region == 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16_t* a;
            if (is_w) {
                a = src[region][j] + (int64_t)od_tap[region] * p.cv.w_tap_stride + k0;
            } else {
                const int h = region - 1;
                int ti = crd_t[h][j] + od_dt[region];
                ti = ti < 0 ? 0 : ti;  // causal: the first frame is replicated in front
                const int yi = (crd_yx[h][j] >> 16) + od_dy[region];
                const int xi = (int)(short)(crd_yx[h][j] & 0xffff) + od_dx[region];
                const bool ok = ti < p.cv.Ti && yi >= 0 && yi < p.cv.Hi && xi >= 0 && xi < p.cv.Wi;
                a = ok ? p.A + (((int64_t)ti * p.cv.Hi + yi) * p.cv.Wi + xi) * p.lda + src_chunk * 8 + k0
                       : reinterpret_cast<const bf16_t*>(g3_zero_page) + src_chunk * 8;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                             (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, 0, 0);
        }
        if (++od_kc[region] == nkc) {
            od_kc[region] = 0;
            ++od_tap[region];
            if (++od_dx[region] == p.cv.kw) {
                od_dx[region] = 0;
                if (++od_dy[region] == p.cv.kh) {
                    od_dy[region] = 0;
                    ++od_dt[region];
                }
            }
        }
    };

    // ---- fragment read addresses (bytes inside a slot): row*128 + ((2 ks + g) ^ ((row >> 1) & 7))*16
    unsigned koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = (unsigned)(l31 * 128 + (((2 * ks + g) ^ ((l31 >> 1) & 7)) << 4));
    const unsigned w_base = (unsigned)(wn * 64 * 128);  // + region*16384 + i*32*128
    const unsigned t_base = (unsigned)(wm * 32 * 128);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 wf[2][4];  // current W half: [row block][k-step]
    bf16x8 tf[2][4];  // both token halves: [half][k-step]

    const int nk = CONV ? nkc * p.cv.ntaps : p.K / BK;
    const int nq = 4 * nk;

    auto read_w = [&](const char* sl, int region) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                wf[i][ks] = *reinterpret_cast<const bf16x8*>(sl + region * 16384 + w_base + i * 4096 + koff[ks]);
    };
    auto read_t = [&](const char* sl, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) tf[h][ks] = *reinterpret_cast<const bf16x8*>(sl + (1 + h) * 16384 + t_base + koff[ks]);
    };

    // PH == 4: one accumulator quadrant (8 MFMAs) per phase, one half-tile issued per phase.
    auto phase4 = [&](auto PC, auto FULL, int t) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;
        G3_JITTER(wave + blockIdx.x, 4 * t + P);
        constexpr bool full = decltype(FULL)::value;
        const char* sl = smem_raw + ((t & 1) << 16);
        if (P == 0) {
            read_w(sl, 0);
            read_t(sl, 0);
        } else if (P == 1) {
            read_t(sl, 1);
        } else if (P == 2) {
            read_w(sl, 3);
        }
        const int f = 4 * t + P;
        if (full) {
            issue((P + 2) & 3, t + (P < 2 ? 1 : 2));
            wait_vmcnt<8>();
        } else {
            if (f + 6 < nq) issue((P + 2) & 3, t + (P < 2 ? 1 : 2));
            const int newest = min(f + 6, nq - 1);
            wait_vmcnt_rt(max(newest - (f + 2), 0));
        }
        phase_barrier();
        constexpr int ih = (P >= 2) ? 2 : 0;
        constexpr int jh = (P == 1 || P == 2) ? 1 : 0;
        G3_PP_SETPRIO(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[ih + i][jh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][ks], tf[jh][ks], acc[ih + i][jh], 0, 0, 0);
        G3_PP_SETPRIO(0);
        phase_barrier();
    };
    // PH == 2: one W half x both token halves (16 MFMAs) per phase, two half-tiles issued per phase. A buffer is restaged
    // one phase after its last read here, which is safe because the reads are retired (lgkmcnt(0)) BEFORE the reading
    // phase's first barrier - also for the wave group that runs a barrier behind.
    auto phase2 = [&](auto PC, auto FULL, int t) __attribute__((always_inline)) {
        constexpr int P = decltype(PC)::value;
        G3_JITTER(wave + blockIdx.x, 2 * t + P);
        constexpr bool full = decltype(FULL)::value;
        const char* sl = smem_raw + ((t & 1) << 16);
        if (P == 0) {
            read_w(sl, 0);
            read_t(sl, 0);
            read_t(sl, 1);
        } else {
            read_w(sl, 3);
        }
        const int q0 = 4 * t + (P == 0 ? 6 : 8);     // first of the two half-tiles this phase issues
        const int need = 4 * t + (P == 0 ? 3 : 6);   // newest half-tile the NEXT phase reads
        if (full) {
            issue(P == 0 ? 2 : 0, t + (P == 0 ? 1 : 2));
            issue(P == 0 ? 3 : 1, t + (P == 0 ? 1 : 2));
            wait_vmcnt<(P == 0 ? 8 : 6)>();
        } else {
            if (q0 < nq) {
                issue(P == 0 ? 2 : 0, t + (P == 0 ? 1 : 2));
                issue(P == 0 ? 3 : 1, t + (P == 0 ? 1 : 2));
            }
            const int newest = min(q0 + 1, nq - 1);
            wait_vmcnt_rt(max(newest - need, 0));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        phase_barrier();
        constexpr int ih = P * 2;
        G3_PP_SETPRIO(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ih + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][ks], tf[j][ks], acc[ih + i][j], 0, 0, 0);
        G3_PP_SETPRIO(0);
        phase_barrier();
    };

    // ---- prologue: half-tiles 0..5; what the first phase reads has landed before anyone starts
    {
        const int n0q = min(6, nq);
#pragma unroll
        for (int q = 0; q < 6; ++q)
            if (q < n0q) issue(q & 3, q >> 2);
        wait_vmcnt_rt(n0q - (PH == 4 ? 2 : 3));
        phase_barrier();
        if (wave >= 4) phase_barrier();  // stagger: waves 4..7 run half a phase behind waves 0..3
    }
    using std::integral_constant;
    using T_ = integral_constant<bool, true>;
    using F_ = integral_constant<bool, false>;
    int t = 0;
    if (PH == 4) {
        for (; t + 2 < nk; ++t) {
            phase4(integral_constant<int, 0>{}, T_{}, t);
            phase4(integral_constant<int, 1>{}, T_{}, t);
            phase4(integral_constant<int, 2>{}, T_{}, t);
            phase4(integral_constant<int, 3>{}, T_{}, t);
        }
        for (; t < nk; ++t) {
            phase4(integral_constant<int, 0>{}, F_{}, t);
            phase4(integral_constant<int, 1>{}, F_{}, t);
            phase4(integral_constant<int, 2>{}, F_{}, t);
            phase4(integral_constant<int, 3>{}, F_{}, t);
        }
    } else {
        for (; t + 2 < nk; ++t) {
            phase2(integral_constant<int, 0>{}, T_{}, t);
            phase2(integral_constant<int, 1>{}, T_{}, t);
        }
        for (; t < nk; ++t) {
            phase2(integral_constant<int, 0>{}, F_{}, t);
            phase2(integral_constant<int, 1>{}, F_{}, t);
        }
    }
    if (wave < 4) phase_barrier();

    if (p.wide_store) store_tile_lds<EPI>(p, acc, m0 + m_w0, n0 + n_w0, lane, smem_raw + wave * 16384);
    else store_tile<EPI>(p, acc, m0 + m_w0, n0 + n_w0, l31, g);
}

template <int EPI, int PH, bool CONV>
int launch_pp(const GemmParams& p, hipStream_t stream, const char* what) {
    const size_t smem = (size_t)2 * (BM + BN) * BK * sizeof(bf16_t);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_pp_kernel<EPI, PH, CONV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_nt_pp_kernel<EPI, PH, CONV>), dim3(p.tiles_m * p.tiles_n), dim3(NTHREADS), smem, stream, p);
    return g3_check_launch(what);
}

template <int EPI, bool STAGE_GLDS, bool CONV, bool PIN>
int launch_variant(const GemmParams& p, hipStream_t stream, const char* what) {
    const size_t smem = (size_t)2 * (BM + BN) * BK * sizeof(bf16_t);  // 128 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_kernel<EPI, STAGE_GLDS, CONV, PIN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int nblk = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI, STAGE_GLDS, CONV, PIN>), dim3(nblk), dim3(NTHREADS), smem, stream, p);
    return g3_check_launch(what);
}

#include "gemm_w4.hpp"
#include "gemm_w4_conv.hpp"
#include "gemm_w4e.hpp"

static int device_cu_count() {  // cached per device (sizes the persistent grid of gemm_w4e.hpp)
    static int n_cu[64] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(mu);
    if (!n_cu[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n_cu[dev] = cus;
    }
    return n_cu[dev];
}

// CONV on the one-wave-per-SIMD kernel (gemm_w4_conv.hpp): whole 64-channel tiles, >= 2 K tiles, K * 2 bytes inside the zero page,
// <= 32 spatial taps (bit masks), full-line epilogue, activation span addressable with 32-bit row * lda products
static bool conv_w4_applies(const GemmParams& p) {
    const int64_t in_rows = (int64_t)p.cv.Ti * p.cv.Hi * p.cv.Wi;
    return g3_opt_conv_w4 && (p.K % BK) == 0 && !g3_opt_gemm_regstage && p.wide_store && (p.K / BK) * p.cv.ntaps >= 2 && p.K * 2 <= G3_ZERO_PAGE_BYTES &&
           p.cv.kh * p.cv.kw <= 32 && in_rows < (1ll << 31) && p.lda * 2 < (1ll << 31) && p.cv.w_tap_stride * 2 < (1ll << 32);
}

template <int EPI, bool CONV>
int launch(const GemmParams& p, hipStream_t stream, const char* what) {
    const bool glds = (p.K % BK) == 0 && !g3_opt_gemm_regstage;
    if constexpr (CONV && (EPI == EPI_NONE || EPI == EPI_BIAS || EPI == EPI_BIAS_RESIDUAL)) {
        if (conv_w4_applies(p)) return launch_w4_conv<EPI>(p, stream, what);
    }
    if constexpr (!CONV && (EPI == EPI_NONE || EPI == EPI_GELU || EPI == EPI_GATED_RESIDUAL)) {  // persistent form with the deferred epilogue (gemm_w4e.hpp)
        if (glds && g3_opt_gemm_pingpong == 3) {
            const int n_cu = device_cu_count();
            if (w4e_applies(p, EPI, n_cu)) return launch_w4e<EPI>(p, stream, what, n_cu);
        }
    }
    if constexpr (!CONV) {  // one wave per SIMD (gemm_w4.hpp): whole 64-wide K tiles, at least two, full-line epilogue
        if (glds && g3_opt_gemm_pingpong == 3 && p.wide_store && p.K >= 2 * BK) return launch_w4<EPI>(p, stream, what);
    }
    if (glds && g3_opt_gemm_pingpong >= 2) return launch_pp<EPI, 2, CONV>(p, stream, what);
    if constexpr (!CONV) {
        if (glds && g3_opt_gemm_pingpong) return launch_pp<EPI, 4, false>(p, stream, what);
    }
    if (g3_opt_gemm_unpinned) return glds ? launch_variant<EPI, true, CONV, false>(p, stream, what) : launch_variant<EPI, false, CONV, false>(p, stream, what);
    return glds ? launch_variant<EPI, true, CONV, true>(p, stream, what) : launch_variant<EPI, false, CONV, true>(p, stream, what);
}

}  // namespace

extern "C" int g3_gemm_bf16_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int M,
                               int N, int K, int epilogue, const void* gate, int gate_rows, int64_t ldg,
                               const void* residual, int64_t ldr, void* stream) {
    if (!A || !W || !C) return g3_set_error(G3_ERR_ARG, "g3_gemm_bf16_nt: null operand");
    if (M <= 0 || N <= 0 || K <= 0) return g3_set_error(G3_ERR_ARG, "g3_gemm_bf16_nt: bad shape M=%d N=%d K=%d", M, N, K);
    if ((K & 7) || (lda & 7) || (ldw & 7) || (N & 3) || (ldc & 3))
        return g3_set_error(G3_ERR_ARG, "g3_gemm_bf16_nt: need K,lda,ldw %% 8 == 0 and N,ldc %% 4 == 0 (K=%d lda=%lld ldw=%lld N=%d ldc=%lld)",
                            K, (long long)lda, (long long)ldw, N, (long long)ldc);
    if (((uintptr_t)A | (uintptr_t)W) & 15) return g3_set_error(G3_ERR_ARG, "g3_gemm_bf16_nt: A/W must be 16-byte aligned");
    if ((uintptr_t)C & 7) return g3_set_error(G3_ERR_ARG, "g3_gemm_bf16_nt: C must be 8-byte aligned");
    GemmParams p;
    p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.C = (bf16_t*)C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.gate = (const bf16_t*)gate; p.gate_rows = gate_rows > 0 ? gate_rows : 1; p.ldg = ldg;
    p.R = (const bf16_t*)residual; p.ldr = ldr;
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    p.tile_order_rowmajor = g3_opt_gemm_rowmajor_tiles;
    p.wide_store = g3_opt_gemm_wide_store && !(N & 7) && !(ldc & 7) && !((uintptr_t)C & 15) &&
                   (!gate || (!(ldg & 7) && !((uintptr_t)gate & 15))) && (!residual || (!(ldr & 7) && !((uintptr_t)residual & 15)));
    p.cv = ConvGeom{};
    hipStream_t s = (hipStream_t)stream;
    const char* what = "g3_gemm_bf16_nt";
    switch (epilogue) {
        case EPI_NONE: return launch<EPI_NONE, false>(p, s, what);
        case EPI_GELU: return launch<EPI_GELU, false>(p, s, what);
        case EPI_GATED_RESIDUAL:
            if (!gate || !residual || (ldg & 3) || (ldr & 3) || (((uintptr_t)gate | (uintptr_t)residual) & 7))
                return g3_set_error(G3_ERR_ARG, "g3_gemm_bf16_nt: gated-residual epilogue needs 8-byte aligned gate+residual");
            return launch<EPI_GATED_RESIDUAL, false>(p, s, what);
        case EPI_BIAS:
            if (!gate || (ldg & 3) || ((uintptr_t)gate & 7)) return g3_set_error(G3_ERR_ARG, "g3_gemm_bf16_nt: bias epilogue needs bias");
            return launch<EPI_BIAS, false>(p, s, what);
        case EPI_BIAS_RESIDUAL:
            if (!gate || !residual || (ldg & 3) || (ldr & 3) || (((uintptr_t)gate | (uintptr_t)residual) & 7))
                return g3_set_error(G3_ERR_ARG, "g3_gemm_bf16_nt: bias-residual epilogue needs 8-byte aligned bias+residual");
            return launch<EPI_BIAS_RESIDUAL, false>(p, s, what);
        default: return g3_set_error(G3_ERR_ARG, "g3_gemm_bf16_nt: unknown epilogue %d", epilogue);
    }
}

// Name of the kernel family g3_gemm_bf16_nt / g3_gemm_qk_norm_rope_bf16 launch for this shape under the options in force (16-byte aligned,
// 8-element-strided operands assumed - what the DiT passes): for profilers and bench.py's roofline_gemm line, never needed to run the op.
extern "C" const char* g3_gemm_kernel_name(int M, int N, int K, int epilogue) {
    const bool glds = (K % BK) == 0 && !g3_opt_gemm_regstage;
    const bool wide = g3_opt_gemm_wide_store && !(N & 7);
    if (glds && g3_opt_gemm_pingpong == 3 && wide && (epilogue == EPI_NONE || epilogue == EPI_GELU || epilogue == EPI_GATED_RESIDUAL)) {
        GemmParams p{};
        p.M = M; p.N = N; p.K = K; p.wide_store = 1; p.gate_rows = 1; p.ldc = p.ldr = N; p.ldg = N;
        p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
        if (w4e_applies(p, epilogue, device_cu_count())) return "gemm_bf16_nt_w4e_kernel<EPI>";
    }
    if (glds && g3_opt_gemm_pingpong == 3 && wide && K >= 2 * BK) return "gemm_bf16_nt_w4_kernel<EPI>";
    if (glds && g3_opt_gemm_pingpong >= 2) return "gemm_bf16_nt_pp_kernel<EPI, 2, false>";
    if (glds && g3_opt_gemm_pingpong) return "gemm_bf16_nt_pp_kernel<EPI, 4, false>";
    return glds ? "gemm_bf16_nt_kernel<EPI, true, false, PIN>" : "gemm_bf16_nt_kernel<EPI, false, false, PIN>";
}

// Q / K projection with the per-head RMSNorm (+ RoPE) of the reference's Attention.cal_qkv (attention.py:247-280) in the epilogue:
//   C[:, 0:n_q]         = rope(rmsnorm(A W^T, norm_q))     C[:, n_q:n_q+n_k] = rope(rmsnorm(A W^T, norm_k))     C[:, n_q+n_k:] = A W^T
// With vt != NULL the remaining (v) heads are written TRANSPOSED into vt [B][H_v][128][vt_ld] (g3_transpose_v_bf16's layout, zero beyond S)
// and their columns of C are left untouched.
// Row m is token (s = m / B, b = m % B); cos / sin are fp32 [S][128] (NULL: no RoPE - cross-attention). Same rounding points as
// g3_gemm_bf16_nt followed by g3_qk_rmsnorm_rope_bf16 in place, which is also what runs when the one-wave-per-SIMD kernel does not apply.
extern "C" int g3_gemm_qk_norm_rope_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                                         int n_q, int n_k, const void* norm_q, const void* norm_k, const float* cos_table,
                                         const float* sin_table, int B, float eps, void* vt, int64_t vt_ld, void* stream) {
    if (!A || !W || !C) return g3_set_error(G3_ERR_ARG, "g3_gemm_qk_norm_rope_bf16: null operand");
    if (M <= 0 || N <= 0 || K <= 0 || B <= 0 || (M % B)) return g3_set_error(G3_ERR_ARG, "g3_gemm_qk_norm_rope_bf16: bad shape M=%d N=%d K=%d B=%d", M, N, K, B);
    if (n_q < 0 || n_k < 0 || (n_q % 128) || (n_k % 128) || n_q + n_k > N || (N % 128))
        return g3_set_error(G3_ERR_ARG, "g3_gemm_qk_norm_rope_bf16: q / k feature ranges must be whole 128-wide heads inside N (n_q=%d n_k=%d N=%d)", n_q, n_k, N);
    if ((n_q && !norm_q) || (n_k && !norm_k)) return g3_set_error(G3_ERR_ARG, "g3_gemm_qk_norm_rope_bf16: missing norm weight");
    if ((cos_table == nullptr) != (sin_table == nullptr)) return g3_set_error(G3_ERR_ARG, "g3_gemm_qk_norm_rope_bf16: need both cos and sin tables or neither");
    if ((K & 7) || (lda & 7) || (ldw & 7) || (ldc & 7)) return g3_set_error(G3_ERR_ARG, "g3_gemm_qk_norm_rope_bf16: K, lda, ldw, ldc must be multiples of 8");
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) || (((uintptr_t)norm_q | (uintptr_t)norm_k) & 15) || (((uintptr_t)cos_table | (uintptr_t)sin_table) & 15))
        return g3_set_error(G3_ERR_ARG, "g3_gemm_qk_norm_rope_bf16: operands must be 16-byte aligned");
    const int S = M / B, n_v = N - n_q - n_k;
    if (vt && (n_v <= 0 || (vt_ld & 7) || vt_ld < S || ((uintptr_t)vt & 15)))
        return g3_set_error(G3_ERR_ARG, "g3_gemm_qk_norm_rope_bf16: V^T destination needs v heads, vt_ld %% 8 == 0, vt_ld >= S and 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const bool fused = (K % BK) == 0 && K >= 2 * BK && !g3_opt_gemm_regstage && g3_opt_gemm_pingpong == 3 && g3_opt_gemm_wide_store;
    const bool vt_fused = fused && vt && (B == 1 || B == 2 || B == 4);
    int rc;
    if (fused) {
        GemmParams p;
        p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.C = (bf16_t*)C; p.ldc = ldc;
        p.M = M; p.N = N; p.K = K;
        p.gate = nullptr; p.gate_rows = 1; p.ldg = 0; p.R = nullptr; p.ldr = 0;
        p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
        p.tile_order_rowmajor = g3_opt_gemm_rowmajor_tiles;
        p.wide_store = 1;
        p.cv = ConvGeom{};
        p.nw_q = (const bf16_t*)norm_q; p.nw_k = (const bf16_t*)norm_k; p.rope_cos = cos_table; p.rope_sin = sin_table;
        p.n_q = n_q; p.n_k = n_k; p.rope_B = B; p.rms_eps = eps;
        p.vt = vt_fused ? (bf16_t*)vt : nullptr; p.vt_ld = vt_ld; p.vt_batch = (int64_t)(n_v / 128) * 128 * vt_ld; p.vt_S = S;
        rc = launch_w4<EPI_QK_NORM_ROPE>(p, s, "g3_gemm_qk_norm_rope_bf16");
    } else {
        rc = g3_gemm_bf16_nt(A, lda, W, ldw, C, ldc, M, N, K, EPI_NONE, nullptr, 1, 0, nullptr, 0, stream);
        bf16_t* c = (bf16_t*)C;
        if (rc == G3_OK && n_q) rc = g3_qk_rmsnorm_rope_bf16(c, ldc, norm_q, cos_table, sin_table, c, ldc, S, B, n_q / 128, 128, eps, stream);
        if (rc == G3_OK && n_k) rc = g3_qk_rmsnorm_rope_bf16(c + n_q, ldc, norm_k, cos_table, sin_table, c + n_q, ldc, S, B, n_k / 128, 128, eps, stream);
    }
    // V^T where the epilogue did not write it: the standalone transpose of C's v columns (which the fused form leaves unwritten)
    if (rc == G3_OK && vt && !vt_fused) rc = g3_transpose_v_bf16((const bf16_t*)C + n_q + n_k, ldc, vt, vt_ld, S, B, n_v / 128, 128, stream);
    return rc;
}

// Causal 3-D convolution as an implicit GEMM over channels-last activations.
//   in  [Ti*Hi*Wi][ld_in]  (C_in = K valid channels per position), w [kt*kh*kw][N][ldw] (tap-major, K contiguous),
//   out [To*Ho*Wo][ld_out]; bias [N] (may be NULL), residual [To*Ho*Wo][ldr] (may be NULL) added after the bias.
extern "C" int g3_groupnorm_stats_cl_bf16(const void* x, int64_t ld, void* stats_f64, int frames, int rows_per_frame, int C, void* stream);

static int conv3d_cl(const void* in, int64_t ld_in, const void* w, int64_t ldw, const void* bias, const void* residual,
                     int64_t ldr, void* out, int64_t ld_out, int K, int N, int Ti, int Hi, int Wi, int To, int Ho,
                     int Wo, int kt, int kh, int kw, int st, int sh, int sw, int ot, int oh, int ow, double* gn_stats, int gn_rows, void* stream) {
    if (!in || !w || !out) return g3_set_error(G3_ERR_ARG, "g3_conv3d_cl_bf16: null operand");
    if (K <= 0 || N <= 0 || (K & 7) || (ld_in & 7) || (ldw & 7) || (N & 3) || (ld_out & 3))
        return g3_set_error(G3_ERR_ARG, "g3_conv3d_cl_bf16: need K, ld_in, ldw %% 8 == 0 and N, ld_out %% 4 == 0 (K=%d N=%d)", K, N);
    if (Ti <= 0 || Hi <= 0 || Wi <= 0 || To <= 0 || Ho <= 0 || Wo <= 0 || kt <= 0 || kh <= 0 || kw <= 0 || st <= 0 || sh <= 0 || sw <= 0)
        return g3_set_error(G3_ERR_ARG, "g3_conv3d_cl_bf16: bad geometry");
    if ((int64_t)To * Ho * Wo > 0x7fffffffLL) return g3_set_error(G3_ERR_ARG, "g3_conv3d_cl_bf16: too many output positions");
    if ((((uintptr_t)in | (uintptr_t)w) & 15) || ((uintptr_t)out & 7)) return g3_set_error(G3_ERR_ARG, "g3_conv3d_cl_bf16: misaligned pointer");
    if (residual && ((ldr & 3) || ((uintptr_t)residual & 7))) return g3_set_error(G3_ERR_ARG, "g3_conv3d_cl_bf16: misaligned residual");
    if (residual && !bias) return g3_set_error(G3_ERR_ARG, "g3_conv3d_cl_bf16: residual without bias is not instantiated");
    GemmParams p;
    p.A = (const bf16_t*)in; p.lda = ld_in; p.W = (const bf16_t*)w; p.ldw = ldw; p.C = (bf16_t*)out; p.ldc = ld_out;
    p.M = To * Ho * Wo; p.N = N; p.K = K;
    p.gate = (const bf16_t*)bias; p.gate_rows = 1; p.ldg = 0;
    p.R = (const bf16_t*)residual; p.ldr = ldr;
    p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    p.tile_order_rowmajor = g3_opt_gemm_rowmajor_tiles;
    p.wide_store = g3_opt_gemm_wide_store && !(N & 7) && !(ld_out & 7) && !((uintptr_t)out & 15) && (!bias || !((uintptr_t)bias & 15)) &&
                   (!residual || (!(ldr & 7) && !((uintptr_t)residual & 15)));
    p.cv = ConvGeom{To, Ho, Wo, Ti, Hi, Wi, kt, kh, kw, st, sh, sw, ot, oh, ow, kt * kh * kw, (int64_t)N * ldw, g3_opt_conv_w4 != 2};
    hipStream_t s = (hipStream_t)stream;
    const char* what = "g3_conv3d_cl_bf16";
    // GroupNorm statistics of the output: in the one-wave kernel's epilogue when it runs (and a frame is at least one wave quadrant of rows),
    // else by the statistics pass over the finished output
    const bool fuse_stats = gn_stats && gn_rows >= 128 && conv_w4_applies(p);
    if (fuse_stats) { p.gn_stats = gn_stats; p.gn_rows = gn_rows; }
    int rc;
    if (residual) rc = launch<EPI_BIAS_RESIDUAL, true>(p, s, what);
    else if (bias) rc = launch<EPI_BIAS, true>(p, s, what);
    else rc = launch<EPI_NONE, true>(p, s, what);
    if (rc == G3_OK && gn_stats && !fuse_stats) {
        if (gn_rows <= 0 || (p.M % gn_rows)) return g3_set_error(G3_ERR_ARG, "g3_conv3d_cl_gnstats_bf16: gn_rows_per_frame must divide the output rows");
        rc = g3_groupnorm_stats_cl_bf16(out, ld_out, gn_stats, p.M / gn_rows, gn_rows, N, stream);
    }
    return rc;
}

extern "C" int g3_conv3d_cl_bf16(const void* in, int64_t ld_in, const void* w, int64_t ldw, const void* bias, const void* residual,
                                 int64_t ldr, void* out, int64_t ld_out, int K, int N, int Ti, int Hi, int Wi, int To, int Ho,
                                 int Wo, int kt, int kh, int kw, int st, int sh, int sw, int ot, int oh, int ow, void* stream) {
    return conv3d_cl(in, ld_in, w, ldw, bias, residual, ldr, out, ld_out, K, N, Ti, Hi, Wi, To, Ho, Wo, kt, kh, kw, st, sh, sw, ot, oh, ow, nullptr, 0, stream);
}

// Same convolution; additionally ADDS, per output frame (gn_rows_per_frame consecutive output rows = Ho * Wo), the sum and the sum of squares of
// the stored bf16 outputs to gn_stats[frame][0 / 1] (doubles; the caller zeroes them): the statistics CausalNormalize needs of this tensor
// (tokenizer/modules/utils.py:58-83), so that the following g3_groupnorm_apply_cl_bf16 does not have to read the tensor twice.
extern "C" int g3_conv3d_cl_gnstats_bf16(const void* in, int64_t ld_in, const void* w, int64_t ldw, const void* bias, const void* residual,
                                         int64_t ldr, void* out, int64_t ld_out, int K, int N, int Ti, int Hi, int Wi, int To, int Ho,
                                         int Wo, int kt, int kh, int kw, int st, int sh, int sw, int ot, int oh, int ow, void* gn_stats_f64,
                                         int gn_rows_per_frame, void* stream) {
    if (!gn_stats_f64 || gn_rows_per_frame <= 0 || ((uintptr_t)gn_stats_f64 & 7)) return g3_set_error(G3_ERR_ARG, "g3_conv3d_cl_gnstats_bf16: bad statistics buffer");
    if (((int64_t)To * Ho * Wo) % gn_rows_per_frame) return g3_set_error(G3_ERR_ARG, "g3_conv3d_cl_gnstats_bf16: gn_rows_per_frame must divide the output rows");
    return conv3d_cl(in, ld_in, w, ldw, bias, residual, ldr, out, ld_out, K, N, Ti, Hi, Wi, To, Ho, Wo, kt, kh, kw, st, sh, sw, ot, oh, ow,
                     (double*)gn_stats_f64, gn_rows_per_frame, stream);
}
