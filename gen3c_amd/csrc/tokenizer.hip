// Memory-bound companions of the causal video tokenizer (Cosmos-Tokenize1-CV8x8x8): everything except its
// convolutions / attention GEMMs, which run on the MFMA kernel of gemm.hip (g3_conv3d_cl_bf16, g3_gemm_bf16_nt).
//
// Activations are channels-last bf16: [T][H][W][C] (one batch item). Reference modules (tokenizer/modules):
//   CausalNormalize (GroupNorm, 1 group, per frame) + swish       utils.py:58-83, layers3d.py nonlinearity
//   Patcher3D / UnPatcher3D, Haar, patch_size 4                    patching.py:111-175, 250-311
//   avg_pool3d / repeat_interleave of the hybrid down/up-sampling  layers3d.py:135-234
//   softmax of CausalAttnBlock, CausalTemporalAttnBlock            layers3d.py:345-427
#include "common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm(1 group) statistics per frame: sum and sum of squares over rows_per_frame x C values, fp64 accumulation
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, int64_t ld, int rows_per_frame, int C,
                                                       double* __restrict__ stats) {
    const int frame = blockIdx.y;
    const int cpr = C >> 3;  // 16-byte chunks per row
    const int64_t nchunks = (int64_t)rows_per_frame * cpr;
    const bf16_t* xf = x + (int64_t)frame * rows_per_frame * ld;
    float s = 0.f, q = 0.f;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * 256) {
        const int64_t row = c / cpr;
        const int col = (int)(c - row * cpr) * 8;
        const bf16x8 v = load_bf16x8(xf + row * ld + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s += f; q += f * f; }
    }
    double ds = s, dq = q;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o, 64); dq += __shfl_xor(dq, o, 64); }
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ds; red[1][threadIdx.x >> 6] = dq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(stats + frame * 2 + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(stats + frame * 2 + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// Normalise (+ swish) given finished statistics. A thread keeps ONE 8-channel column for the whole launch (the grid stride is a multiple of the chunks per
// row: 256 threads, C / 8 <= 64 chunks): gamma / beta are loaded once and the row index advances by a constant - no 64-bit division per chunk (round 4; the
// first form recomputed row = chunk / (C / 8) and reloaded gamma / beta for every chunk: 4.0-5.0 TB/s; this form 5.4-5.7). Four rows in flight per iteration.
template <bool SWISH>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, int64_t ld, const bf16_t* __restrict__ gamma,
                                                       const bf16_t* __restrict__ beta, const double* __restrict__ stats,
                                                       bf16_t* __restrict__ out, int64_t ldo, int rows_per_frame, int C, float eps) {
    const int frame = blockIdx.y;
    const double n = (double)rows_per_frame * C;
    const double mean_d = stats[frame * 2] / n;
    const double var_d = stats[frame * 2 + 1] / n - mean_d * mean_d;
    const float mean = (float)mean_d;
    const float rstd = rsqrtf((float)(var_d > 0 ? var_d : 0) + eps);
    const int cpr = C >> 3;                                   // chunks per row; divides 256 (host-checked)
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int col = (t % cpr) * 8;
    const int row0 = t / cpr;
    const int row_step = (gridDim.x * 256) / cpr;
    const bf16x8 g = load_bf16x8(gamma + col);
    const bf16x8 b = load_bf16x8(beta + col);
    float gf[8], bfv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gf[e] = (float)g[e]; bfv[e] = (float)b[e]; }
    const int64_t base = (int64_t)frame * rows_per_frame;
    auto apply = [&](const bf16x8& v) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = ((float)v[e] - mean) * rstd * gf[e] + bfv[e];
            if (SWISH) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.44269504088896340736f));  // y * sigmoid(y)
            o[e] = f32_to_bf16(y);
        }
        return o;
    };
    int row = row0;
    for (; row + 3 * row_step < rows_per_frame; row += 4 * row_step) {
        bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load_bf16x8(x + (base + row + u * row_step) * ld + col);
#pragma unroll
        for (int u = 0; u < 4; ++u) store_bf16x8(out + (base + row + u * row_step) * ldo + col, apply(v[u]));
    }
    for (; row < rows_per_frame; row += row_step) store_bf16x8(out + (base + row) * ldo + col, apply(load_bf16x8(x + (base + row) * ld + col)));
}

// ---------------------------------------------------------------------------------------------------------------
// Haar (patch_size 4 = two 2x2x2 levels). One thread = one (t', y', x', rgb channel): a 4x4x4 block of pixels <-> 64
// coefficients. Each level is +-sum of 8 values times 1/8 (three 1/sqrt2 stages and the reference's 1/(2 sqrt2) rescale).
// Coefficient channel = band2*24 + band1*3 + c with band = 4*t_bit + 2*h_bit + w_bit (bit 1 = high-pass).
// ---------------------------------------------------------------------------------------------------------------
G3_DEVICE void haar8(const float (&v)[8], float (&o)[8], float scale) {  // index = 4*t + 2*h + w  ->  band = 4*lt + 2*lh + lw
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i] + v[i + 4]; a[i + 4] = v[i] - v[i + 4]; }  // t
#pragma unroll
    for (int i = 0; i < 8; i += 4) { b[i] = a[i] + a[i + 2]; b[i + 1] = a[i + 1] + a[i + 3]; b[i + 2] = a[i] - a[i + 2]; b[i + 3] = a[i + 1] - a[i + 3]; }  // h
#pragma unroll
    for (int i = 0; i < 8; i += 2) { o[i] = (b[i] + b[i + 1]) * scale; o[i + 1] = (b[i] - b[i + 1]) * scale; }  // w
}

// One thread = one output position (t', y', x') and all three colour channels: its 192 coefficients are one contiguous 384-byte row of the
// channels-last output. Reads: per (colour, frame, line) the thread's 4 pixels are one 8-byte load, neighbouring threads are neighbours in
// memory (512 contiguous bytes per wave instruction). Writes: the block's 128 rows are one contiguous 48 KiB span - staged through LDS
// (rows padded to 400 bytes: 16-byte ds accesses of 16 consecutive lanes then cover all 64 banks) and stored 16 bytes per lane, fully coalesced.
// (The first form - one thread per colour channel, 2-byte loads, 2-byte stores 6 bytes apart - ran at a tenth of the streaming rate.)
constexpr int HAAR_BLOCK = 128;
constexpr int HAAR_ROW_BYTES = 400;
__global__ __launch_bounds__(HAAR_BLOCK) void haar_patch_kernel(const bf16_t* __restrict__ video, bf16_t* __restrict__ out, int T, int H,
                                                                int W, int Tp, int Hp, int Wp) {
    __shared__ __attribute__((aligned(16))) char stage[HAAR_BLOCK * HAAR_ROW_BYTES];
    const int64_t total = (int64_t)Tp * Hp * Wp;
    const int64_t idx0 = (int64_t)blockIdx.x * HAAR_BLOCK;
    const int64_t idx = idx0 + threadIdx.x;
    if (idx < total) {
        const int xo = (int)(idx % Wp);
        const int yo = (int)((idx / Wp) % Hp);
        const int to = (int)(idx / ((int64_t)Wp * Hp));
        bf16_t* srow = reinterpret_cast<bf16_t*>(stage + threadIdx.x * HAAR_ROW_BYTES);
        for (int c = 0; c < 3; ++c) {
            const bf16_t* vc = video + (int64_t)c * T * H * W;
            float px[4][4][4];  // [t][y][x] of the 4x4x4 pixel block
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                int t = 4 * to + dt - 3;  // the first frame is repeated patch_size times in front (patching.py:162-163)
                t = t < 0 ? 0 : t;
#pragma unroll
                for (int dy = 0; dy < 4; ++dy) {
                    const bf16x4 v = *reinterpret_cast<const bf16x4*>(vc + ((int64_t)t * H + (4 * yo + dy)) * W + 4 * xo);
#pragma unroll
                    for (int dx = 0; dx < 4; ++dx) px[dt][dy][dx] = (float)v[dx];
                }
            }
            float l1[8][8];  // [level-1 band][level-1 block (bt,by,bx)]
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) {
                const int bt = blk >> 2, by = (blk >> 1) & 1, bx = blk & 1;
                float v[8], o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = px[2 * bt + (i >> 2)][2 * by + ((i >> 1) & 1)][2 * bx + (i & 1)];
                haar8(v, o, 0.125f);
#pragma unroll
                for (int b1 = 0; b1 < 8; ++b1) l1[b1][blk] = o[b1];
            }
#pragma unroll
            for (int b1 = 0; b1 < 8; ++b1) {
                float o[8];
                haar8(l1[b1], o, 0.125f);
#pragma unroll
                for (int b2 = 0; b2 < 8; ++b2) srow[b2 * 24 + b1 * 3 + c] = f32_to_bf16(o[b2]);
            }
        }
    }
    __syncthreads();
    const int64_t rows = total - idx0 < HAAR_BLOCK ? total - idx0 : HAAR_BLOCK;
    bf16_t* obase = out + idx0 * 192;
    for (int i = threadIdx.x; i < (int)rows * 24; i += HAAR_BLOCK) {  // 24 16-byte chunks per row
        const int r = i / 24, ch = i - r * 24;
        store_bf16x8(obase + (int64_t)i * 8, *reinterpret_cast<const bf16x8*>(stage + r * HAAR_ROW_BYTES + ch * 16));
    }
}

// inverse: even = (lo + hi), odd = (lo - hi) per axis, net scale per level 1 (1/sqrt2^3 * 2 sqrt2)
G3_DEVICE void ihaar8(const float (&band)[8], float (&v)[8]) {  // band = 4*lt+2*lh+lw -> index = 4*t+2*h+w
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i += 2) { a[i] = band[i] + band[i + 1]; a[i + 1] = band[i] - band[i + 1]; }  // w
#pragma unroll
    for (int i = 0; i < 8; i += 4) { b[i] = a[i] + a[i + 2]; b[i + 1] = a[i + 1] + a[i + 3]; b[i + 2] = a[i] - a[i + 2]; b[i + 3] = a[i + 1] - a[i + 3]; }  // h
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = b[i] + b[i + 4]; v[i + 4] = b[i] - b[i + 4]; }  // t
}

// One thread = one coefficient row (t', y', x'), all three colour channels: 24 16-byte loads of its contiguous 384 bytes (ld = 192) or
// scalar loads (any other ld), 8-byte stores of 4 pixels per (colour, frame, line) with neighbouring threads neighbours in memory.
__global__ __launch_bounds__(HAAR_BLOCK) void haar_unpatch_kernel(const bf16_t* __restrict__ coef, int64_t ld, bf16_t* __restrict__ video,
                                                                  int Tp, int Hp, int Wp, int Tout) {
    const int64_t idx = (int64_t)blockIdx.x * HAAR_BLOCK + threadIdx.x;
    const int64_t total = (int64_t)Tp * Hp * Wp;
    if (idx >= total) return;
    const int xo = (int)(idx % Wp);
    const int yo = (int)((idx / Wp) % Hp);
    const int to = (int)(idx / ((int64_t)Wp * Hp));
    const bf16_t* crow = coef + idx * ld;
    const int H = 4 * Hp, W = 4 * Wp;
    bf16_t cf[192];
    if (!(ld & 7) && !((uintptr_t)coef & 15)) {
#pragma unroll
        for (int j = 0; j < 24; ++j) *reinterpret_cast<bf16x8*>(cf + 8 * j) = load_bf16x8(crow + 8 * j);
    } else {
#pragma unroll
        for (int j = 0; j < 192; ++j) cf[j] = crow[j];
    }
    for (int c = 0; c < 3; ++c) {
        float l1[8][8];  // [band1][block]
#pragma unroll
        for (int b1 = 0; b1 < 8; ++b1) {
            float band[8], v[8];
#pragma unroll
            for (int b2 = 0; b2 < 8; ++b2) band[b2] = (float)cf[b2 * 24 + b1 * 3 + c];
            ihaar8(band, v);
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) l1[b1][blk] = v[blk];
        }
        bf16_t* vc = video + (int64_t)c * Tout * H * W;
#pragma unroll
        for (int bp = 0; bp < 4; ++bp) {  // the two level-1 blocks (bx = 0, 1) of one (bt, by): 2 frames x 2 lines x 4 pixels
            const int bt = bp >> 1, by = bp & 1;
            float v0[8], v1[8];
            {
                float band[8];
#pragma unroll
                for (int b1 = 0; b1 < 8; ++b1) band[b1] = l1[b1][4 * bt + 2 * by];
                ihaar8(band, v0);
#pragma unroll
                for (int b1 = 0; b1 < 8; ++b1) band[b1] = l1[b1][4 * bt + 2 * by + 1];
                ihaar8(band, v1);
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int t = 4 * to + 2 * bt + dt - 3;  // drop the first patch_size-1 frames (patching.py:298)
                if (t < 0 || t >= Tout) continue;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    bf16x4 o;  // index inside a block = 4 t + 2 h + w
                    o[0] = f32_to_bf16(v0[4 * dt + 2 * dy]);
                    o[1] = f32_to_bf16(v0[4 * dt + 2 * dy + 1]);
                    o[2] = f32_to_bf16(v1[4 * dt + 2 * dy]);
                    o[3] = f32_to_bf16(v1[4 * dt + 2 * dy + 1]);
                    *reinterpret_cast<bf16x4*>(vc + ((int64_t)t * H + (4 * yo + 2 * by + dy)) * W + 4 * xo) = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// resampling of channels-last tensors (16-byte chunks of channels per lane)
//   mode 0: avg-pool (1,2,2) stride 2 on the right/bottom zero-padded input   (layers3d.py:217-220)
//   mode 1: avg-pool (2,1,1) stride 2 on the front-replicated input            (layers3d.py:224-227)
//   mode 2: repeat_interleave(2) in time, drop the first frame                 (layers3d.py:170-173)
//   mode 3: repeat_interleave(2) in height and width                           (layers3d.py:177-178)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int Ti, int Hi, int Wi,
                                                       int To, int Ho, int Wo, int C, int mode) {
    const int cpr = C >> 3;
    const int64_t total = (int64_t)To * Ho * Wo * cpr;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int col = (int)(idx % cpr) * 8;
        const int64_t pos = idx / cpr;
        const int xo = (int)(pos % Wo);
        const int yo = (int)((pos / Wo) % Ho);
        const int to = (int)(pos / ((int64_t)Wo * Ho));
        auto src = [&](int t, int y, int x) { return in + (((int64_t)t * Hi + y) * Wi + x) * C + col; };
        bf16x8 o;
        if (mode == 0) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int y = 2 * yo + dy, x = 2 * xo + dx;
                    if (y < Hi && x < Wi) {
                        const bf16x8 v = load_bf16x8(src(to, y, x));
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
                    }
                }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(acc[e] * 0.25f);
        } else if (mode == 1) {
            const int t0 = max(2 * to - 1, 0), t1 = max(2 * to, 0);  // padded index i -> frame max(i-1, 0)
            const bf16x8 a = load_bf16x8(src(t0, yo, xo));
            const bf16x8 b = load_bf16x8(src(min(t1, Ti - 1), yo, xo));
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(((float)a[e] + (float)b[e]) * 0.5f);
        } else if (mode == 2) {
            o = load_bf16x8(src(Ti > 1 ? (to + 1) >> 1 : to, yo, xo));
        } else {
            o = load_bf16x8(src(to, yo >> 1, xo >> 1));
        }
        store_bf16x8(out + pos * C + col, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// row softmax (in place), scores * scale, one workgroup per row
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ x, int64_t ld, int n, float scale) {
    __shared__ float red[4];
    bf16_t* row = x + (int64_t)blockIdx.x * ld;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, (float)row[i] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += __expf((float)row[i] * scale - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int i = threadIdx.x; i < n; i += 256) row[i] = f32_to_bf16(__expf((float)row[i] * scale - mx) * inv);
}

// Same, for rows that fit the workgroup's registers (n <= 16384, n % 8 == 0, 16-byte aligned rows): one 16-byte read and one
// 16-byte write per 8 scores instead of three scalar passes. Same arithmetic and reduction order per thread as the kernel above
// would have with 8-wide strides (results agree to fp32 summation order).
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(bf16_t* __restrict__ x, int64_t ld, int n, float scale) {
    __shared__ float red[4];
    bf16_t* row = x + (int64_t)blockIdx.x * ld;
    const int nv = n >> 3;  // 8-element vectors in the row
    float v[8][8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int vi = threadIdx.x + 256 * j;
        if (vi < nv) {
            const bf16x8 t = load_bf16x8(row + 8 * vi);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[j][e] = (float)t[e] * scale;
                mx = fmaxf(mx, v[j][e]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (threadIdx.x + 256 * j < nv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[j][e] = __expf(v[j][e] - mx);
                s += v[j][e];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int vi = threadIdx.x + 256 * j;
        if (vi < nv) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(v[j][e] * inv);
            store_bf16x8(row + 8 * vi, o);
        }
    }
}

// out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void transpose2d_kernel(const bf16_t* __restrict__ in, int64_t ld_in, bf16_t* __restrict__ out,
                                                          int64_t ld_out, int R, int C) {
    __shared__ bf16_t tile[64][64 + 2];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(int64_t)(r0 + r) * ld_in + c0 + c] : (bf16_t)0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < C && r0 + r < R) out[(int64_t)(c0 + c) * ld_out + r0 + r] = tile[r][c];
    }
}

// causal attention over time for every pixel: q,k,v,o [T][HW][C]; one wave per (pixel, query frame)
__global__ __launch_bounds__(256) void temporal_attn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                            const bf16_t* __restrict__ v, bf16_t* __restrict__ o, int T, int HW,
                                                            int C, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (int64_t)HW * T) return;
    const int pix = (int)(wid % HW);
    const int ti = (int)(wid / HW);
    constexpr int MAXT = 64;
    float sc[MAXT];
    float mx = -INFINITY;
    for (int tj = 0; tj <= ti; ++tj) {
        float part = 0.f;
        for (int c = lane * 8; c < C; c += 512) {
            const bf16x8 a = load_bf16x8(q + ((int64_t)ti * HW + pix) * C + c);
            const bf16x8 b = load_bf16x8(k + ((int64_t)tj * HW + pix) * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) part += (float)a[e] * (float)b[e];
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) part += __shfl_xor(part, s, 64);
        part = (float)f32_to_bf16(part) * scale;  // bmm output is bf16 in the reference
        sc[tj] = part;
        mx = fmaxf(mx, part);
    }
    float den = 0.f;
    for (int tj = 0; tj <= ti; ++tj) { sc[tj] = __expf(sc[tj] - mx); den += sc[tj]; }
    const float inv = 1.0f / den;
    for (int c = lane * 8; c < C; c += 512) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int tj = 0; tj <= ti; ++tj) {
            const float p = (float)f32_to_bf16(sc[tj] * inv);
            const bf16x8 b = load_bf16x8(v + ((int64_t)tj * HW + pix) * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += p * (float)b[e];
        }
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = f32_to_bf16(acc[e]);
        store_bf16x8(o + ((int64_t)ti * HW + pix) * C + c, r);
    }
}

// Same attention, one wave per PIXEL: the pixel's T <= 16 key and value rows are read once into registers (a lane holds 8 of the C <= 512
// channels of every frame) and serve all T query frames; the first form reads them once per query frame (8.5 x the bytes at T = 16, through
// L2). Same operation order per output as temporal_attn_kernel => bitwise equal results (tested).
template <int TMAX>
__global__ __launch_bounds__(256) void temporal_attn_px_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                               const bf16_t* __restrict__ v, bf16_t* __restrict__ o, int T, int HW,
                                                               int C, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= HW) return;
    const int c = lane * 8;
    const bool act = c < C;
    bf16x8 rk[TMAX], rv[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const bool ok = act && t < T;
        rk[t] = ok ? load_bf16x8(k + ((int64_t)t * HW + pix) * C + c) : zero_bf16x8();
        rv[t] = ok ? load_bf16x8(v + ((int64_t)t * HW + pix) * C + c) : zero_bf16x8();
    }
    bf16x8 rq = act ? load_bf16x8(q + pix * C + c) : zero_bf16x8();
    static_for<0, TMAX>([&](auto tic) {
        constexpr int ti = decltype(tic)::value;
        if (ti >= T) return;  // wave-uniform
        const bf16x8 qn = (act && ti + 1 < T) ? load_bf16x8(q + ((int64_t)(ti + 1) * HW + pix) * C + c) : zero_bf16x8();  // next query row in flight
        float sc[ti + 1];
        float mx = -INFINITY;
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) part += (float)rq[e] * (float)rk[tj][e];
#pragma unroll
            for (int s2 = 32; s2 > 0; s2 >>= 1) part += __shfl_xor(part, s2, 64);
            part = (float)f32_to_bf16(part) * scale;  // bmm output is bf16 in the reference
            sc[tj] = part;
            mx = fmaxf(mx, part);
        }
        float den = 0.f;
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) { sc[tj] = __expf(sc[tj] - mx); den += sc[tj]; }
        const float inv = 1.0f / den;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
            const float p = (float)f32_to_bf16(sc[tj] * inv);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += p * (float)rv[tj][e];
        }
        if (act) {
            bf16x8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = f32_to_bf16(acc[e]);
            store_bf16x8(o + ((int64_t)ti * HW + pix) * C + c, r);
        }
        rq = qn;
    });
}

// any C (chunks per row not dividing the block: the column changes from chunk to chunk)
__global__ __launch_bounds__(256) void gn_apply_generic_kernel(const bf16_t* __restrict__ x, int64_t ld, const bf16_t* __restrict__ gamma,
                                                               const bf16_t* __restrict__ beta, const double* __restrict__ stats,
                                                               bf16_t* __restrict__ out, int64_t ldo, int rows_per_frame, int C, float eps, int swish) {
    const int frame = blockIdx.y;
    const double n = (double)rows_per_frame * C;
    const double mean_d = stats[frame * 2] / n;
    const double var_d = stats[frame * 2 + 1] / n - mean_d * mean_d;
    const float mean = (float)mean_d;
    const float rstd = rsqrtf((float)(var_d > 0 ? var_d : 0) + eps);
    const int cpr = C >> 3;
    const int64_t nchunks = (int64_t)rows_per_frame * cpr;
    const int64_t base = (int64_t)frame * rows_per_frame;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * 256) {
        const int64_t row = c / cpr;
        const int col = (int)(c - row * cpr) * 8;
        const bf16x8 v = load_bf16x8(x + (base + row) * ld + col);
        const bf16x8 g = load_bf16x8(gamma + col);
        const bf16x8 b = load_bf16x8(beta + col);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = ((float)v[e] - mean) * rstd * (float)g[e] + (float)b[e];
            if (swish) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.44269504088896340736f));
            o[e] = f32_to_bf16(y);
        }
        store_bf16x8(out + (base + row) * ldo + col, o);
    }
}

void launch_gn_apply(int gx, int frames, hipStream_t s, const void* x, int64_t ld, const void* gamma, const void* beta, const void* stats_f64, void* out,
                     int64_t ldo, int rows_per_frame, int C, float eps, int swish) {
    const int cpr = C >> 3;
    if (cpr <= 256 && (256 % cpr) == 0) {
        if (swish)
            hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(gx, frames), dim3(256), 0, s, (const bf16_t*)x, ld, (const bf16_t*)gamma, (const bf16_t*)beta,
                               (const double*)stats_f64, (bf16_t*)out, ldo, rows_per_frame, C, eps);
        else
            hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(gx, frames), dim3(256), 0, s, (const bf16_t*)x, ld, (const bf16_t*)gamma, (const bf16_t*)beta,
                               (const double*)stats_f64, (bf16_t*)out, ldo, rows_per_frame, C, eps);
    } else {
        hipLaunchKernelGGL(gn_apply_generic_kernel, dim3(gx, frames), dim3(256), 0, s, (const bf16_t*)x, ld, (const bf16_t*)gamma, (const bf16_t*)beta,
                           (const double*)stats_f64, (bf16_t*)out, ldo, rows_per_frame, C, eps, swish);
    }
}

int grid1d(int64_t work) {
    int64_t g = (work + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int g3_groupnorm_swish_cl_bf16(const void* x, int64_t ld, const void* gamma, const void* beta, void* stats_f64,
                                          void* out, int64_t ldo, int frames, int rows_per_frame, int C, float eps, int swish,
                                          void* stream) {
    if (!x || !gamma || !beta || !stats_f64 || !out) return g3_set_error(G3_ERR_ARG, "g3_groupnorm_swish_cl_bf16: null operand");
    if (frames <= 0 || rows_per_frame <= 0 || C <= 0 || (C & 7) || (ld & 7) || (ldo & 7))
        return g3_set_error(G3_ERR_ARG, "g3_groupnorm_swish_cl_bf16: C, ld, ldo must be multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(stats_f64, 0, sizeof(double) * 2 * frames, s);
    if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_groupnorm_swish_cl_bf16: memset: %s", hipGetErrorString(e));
    const int64_t chunks = (int64_t)rows_per_frame * (C >> 3);
    int gx = (int)((chunks + 256 * 8 - 1) / (256 * 8));
    gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(gx, frames), dim3(256), 0, s, (const bf16_t*)x, ld, rows_per_frame, C, (double*)stats_f64);
    launch_gn_apply(gx, frames, s, x, ld, gamma, beta, stats_f64, out, ldo, rows_per_frame, C, eps, swish);
    return g3_check_launch("g3_groupnorm_swish_cl_bf16");
}

// The two halves of g3_groupnorm_swish_cl_bf16 as separate calls: statistics ADDED to stats_f64 (caller zeroes; g3_conv3d_cl_gnstats_bf16
// produces the same numbers in the producing convolution's epilogue), and the normalisation given finished statistics.
extern "C" int g3_groupnorm_stats_cl_bf16(const void* x, int64_t ld, void* stats_f64, int frames, int rows_per_frame, int C, void* stream) {
    if (!x || !stats_f64) return g3_set_error(G3_ERR_ARG, "g3_groupnorm_stats_cl_bf16: null operand");
    if (frames <= 0 || rows_per_frame <= 0 || C <= 0 || (C & 7) || (ld & 7)) return g3_set_error(G3_ERR_ARG, "g3_groupnorm_stats_cl_bf16: C, ld must be multiples of 8");
    const int64_t chunks = (int64_t)rows_per_frame * (C >> 3);
    int gx = (int)((chunks + 256 * 8 - 1) / (256 * 8));
    gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(gx, frames), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, rows_per_frame, C, (double*)stats_f64);
    return g3_check_launch("g3_groupnorm_stats_cl_bf16");
}

extern "C" int g3_groupnorm_apply_cl_bf16(const void* x, int64_t ld, const void* gamma, const void* beta, const void* stats_f64, void* out,
                                          int64_t ldo, int frames, int rows_per_frame, int C, float eps, int swish, void* stream) {
    if (!x || !gamma || !beta || !stats_f64 || !out) return g3_set_error(G3_ERR_ARG, "g3_groupnorm_apply_cl_bf16: null operand");
    if (frames <= 0 || rows_per_frame <= 0 || C <= 0 || (C & 7) || (ld & 7) || (ldo & 7))
        return g3_set_error(G3_ERR_ARG, "g3_groupnorm_apply_cl_bf16: C, ld, ldo must be multiples of 8");
    const int64_t chunks = (int64_t)rows_per_frame * (C >> 3);
    int gx = (int)((chunks + 256 * 8 - 1) / (256 * 8));
    gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
    launch_gn_apply(gx, frames, (hipStream_t)stream, x, ld, gamma, beta, stats_f64, out, ldo, rows_per_frame, C, eps, swish);
    return g3_check_launch("g3_groupnorm_apply_cl_bf16");
}

extern "C" int g3_haar3d_patch_bf16(const void* video, void* out, int T, int H, int W, void* stream) {
    if (!video || !out) return g3_set_error(G3_ERR_ARG, "g3_haar3d_patch_bf16: null operand");
    if (T < 1 || ((T + 3) & 3) || (H & 3) || (W & 3)) return g3_set_error(G3_ERR_ARG, "g3_haar3d_patch_bf16: need (T+3), H, W multiples of 4");
    const int Tp = (T + 3) / 4, Hp = H / 4, Wp = W / 4;
    if (((uintptr_t)video & 7) || ((uintptr_t)out & 15)) return g3_set_error(G3_ERR_ARG, "g3_haar3d_patch_bf16: video must be 8-byte, out 16-byte aligned");
    const int64_t total = (int64_t)Tp * Hp * Wp;
    hipLaunchKernelGGL(haar_patch_kernel, dim3((unsigned)((total + HAAR_BLOCK - 1) / HAAR_BLOCK)), dim3(HAAR_BLOCK), 0, (hipStream_t)stream,
                       (const bf16_t*)video, (bf16_t*)out, T, H, W, Tp, Hp, Wp);
    return g3_check_launch("g3_haar3d_patch_bf16");
}

extern "C" int g3_haar3d_unpatch_bf16(const void* coef, int64_t ld, void* video, int Tp, int Hp, int Wp, void* stream) {
    if (!coef || !video) return g3_set_error(G3_ERR_ARG, "g3_haar3d_unpatch_bf16: null operand");
    if (Tp < 1 || Hp < 1 || Wp < 1 || ld < 192) return g3_set_error(G3_ERR_ARG, "g3_haar3d_unpatch_bf16: bad shape");
    if ((uintptr_t)video & 7) return g3_set_error(G3_ERR_ARG, "g3_haar3d_unpatch_bf16: video must be 8-byte aligned");
    const int64_t total = (int64_t)Tp * Hp * Wp;
    hipLaunchKernelGGL(haar_unpatch_kernel, dim3((unsigned)((total + HAAR_BLOCK - 1) / HAAR_BLOCK)), dim3(HAAR_BLOCK), 0, (hipStream_t)stream,
                       (const bf16_t*)coef, ld, (bf16_t*)video, Tp, Hp, Wp, 4 * Tp - 3);
    return g3_check_launch("g3_haar3d_unpatch_bf16");
}

extern "C" int g3_resample_cl_bf16(const void* in, void* out, int Ti, int Hi, int Wi, int C, int mode, void* stream) {
    if (!in || !out) return g3_set_error(G3_ERR_ARG, "g3_resample_cl_bf16: null operand");
    if (Ti <= 0 || Hi <= 0 || Wi <= 0 || C <= 0 || (C & 7)) return g3_set_error(G3_ERR_ARG, "g3_resample_cl_bf16: bad shape");
    int To = Ti, Ho = Hi, Wo = Wi;
    if (mode == 0) { Ho = (Hi + 1) / 2; Wo = (Wi + 1) / 2; }
    else if (mode == 1) { To = (Ti + 1) / 2; }
    else if (mode == 2) { To = Ti > 1 ? 2 * Ti - 1 : 1; }
    else if (mode == 3) { Ho = 2 * Hi; Wo = 2 * Wi; }
    else return g3_set_error(G3_ERR_ARG, "g3_resample_cl_bf16: unknown mode %d", mode);
    hipLaunchKernelGGL(resample_kernel, dim3(grid1d((int64_t)To * Ho * Wo * (C >> 3))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                       (bf16_t*)out, Ti, Hi, Wi, To, Ho, Wo, C, mode);
    return g3_check_launch("g3_resample_cl_bf16");
}

extern "C" int g3_softmax_rows_bf16(void* x, int64_t ld, int rows, int n, float scale, void* stream) {
    if (!x || rows <= 0 || n <= 0) return g3_set_error(G3_ERR_ARG, "g3_softmax_rows_bf16: bad argument");
    if (n <= 16384 && !(n & 7) && !(ld & 7) && !((uintptr_t)x & 15))
        hipLaunchKernelGGL(softmax_rows_reg_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ld, n, scale);
    else
        hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ld, n, scale);
    return g3_check_launch("g3_softmax_rows_bf16");
}

extern "C" int g3_transpose2d_bf16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int R, int C, void* stream) {
    if (!in || !out || R <= 0 || C <= 0) return g3_set_error(G3_ERR_ARG, "g3_transpose2d_bf16: bad argument");
    hipLaunchKernelGGL(transpose2d_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ld_in,
                       (bf16_t*)out, ld_out, R, C);
    return g3_check_launch("g3_transpose2d_bf16");
}

extern "C" int g3_temporal_attn_cl_bf16(const void* q, const void* k, const void* v, void* o, int T, int HW, int C, float scale,
                                        void* stream) {
    if (!q || !k || !v || !o) return g3_set_error(G3_ERR_ARG, "g3_temporal_attn_cl_bf16: null operand");
    if (T <= 0 || T > 64 || HW <= 0 || C <= 0 || (C & 7)) return g3_set_error(G3_ERR_ARG, "g3_temporal_attn_cl_bf16: need T <= 64 and C %% 8 == 0");
    if (g3_opt_tok_tattn_px && C <= 512 && T <= 16 && !(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15)) {
        hipLaunchKernelGGL(temporal_attn_px_kernel<16>, dim3((unsigned)((HW + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, T, HW, C, scale);
        return g3_check_launch("g3_temporal_attn_cl_bf16");
    }
    const int64_t waves = (int64_t)HW * T;
    hipLaunchKernelGGL(temporal_attn_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, T, HW, C, scale);
    return g3_check_launch("g3_temporal_attn_cl_bf16");
}
