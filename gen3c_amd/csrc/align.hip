// Depth alignment for autoregressive 3D-cache updates (one image, fp32).
//
// Replaces cosmos_predict1/diffusion/inference/camera_utils.py:225-345 (_align_inv_depth_to_depth, align_depth), which
// the reference runs as torch.quantile + torch.linalg.lstsq followed by 100 autograd/Adam iterations through
// unproject_points (~40 small kernels and a host round trip per iteration).
//
// Here the whole op is one C call that enqueues, on the caller's stream and without any host synchronisation:
//   rigid part
//     prep        : inverse depths, validity, valid counts, sortable keys (positive floats order like their bit patterns)
//     select x4   : MSB-first 8-bit radix SELECT of the 8 order statistics torch.quantile(q = 0.1, 0.9, 'linear') needs
//                   (floor/ceil rank for both quantiles of both maps) - a 256-bin histogram pass per digit instead of a sort
//     fit         : fp64 sums over the inlier set in a fixed two-stage order -> closed-form affine fit -> aligned depth
//   non-rigid part (optional), per iteration
//     arap_sign   : e = sign(box3(sc)/9 - sc)
//     adam_step   : pixel-local gradient of the reference's loss written out (see oracle/align_oracle.py) + the Adam update
//                   in torch.optim.Adam's operation order
// Everything is HBM-streaming over H*W floats (3.6 MB at 704x1280); the 100 iterations cost 200 launches of ~2 us.
//
// The optimisation follows sign() of fp32 residuals: results agree with the reference's autograd trajectory to O(lr),
// not bitwise (tests/test_align_gpu.py states the tolerance); the rigid part agrees to fp32 rounding.
#include "common.hpp"
#include <math.h>
#include <stddef.h>
#include <stdint.h>

namespace {

constexpr int NSEL = 8;        // order statistics: [array 0/1][quantile 0/1][floor/ceil]
constexpr int RED_BLOCKS = 256;  // partial-sum blocks of the fit reduction

struct AlignState {           // lives in the caller's workspace
    unsigned count[2];         // valid entries of source_inv / target_inv
    unsigned rank[NSEL];       // remaining rank inside the current prefix bucket
    unsigned prefix[NSEL];     // key bits decided so far
    float weight[4];           // lerp weights [array][quantile]
    float qlo[2], qhi[2];      // quantile values per array
    float scale, bias;
    unsigned mask_count;       // pixels in target_mask (data-loss normaliser)
    unsigned hist[NSEL][256];
    double partial[RED_BLOCKS][5];
};

__global__ void align_zero_state(AlignState* st) {
    unsigned* p = reinterpret_cast<unsigned*>(st);
    const int n = (int)((offsetof(AlignState, partial)) / 4);
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}

// keys: valid value bits, or 0xFFFFFFFF. tvalid = target_mask & target_depth > 0 ; svalid = source_inv > 0.
__global__ __launch_bounds__(256) void align_prep_kernel(const float* __restrict__ src_depth, const float* __restrict__ tgt_depth,
                                                         const uint8_t* __restrict__ tmask, float* __restrict__ src_inv,
                                                         float* __restrict__ tgt_inv, unsigned* __restrict__ key_s,
                                                         unsigned* __restrict__ key_t, AlignState* st, int n) {
    unsigned cs = 0, ct = 0, cm = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float si = 1.0f / src_depth[i];
        const float td = tgt_depth[i];
        const float ti = 1.0f / td;
        src_inv[i] = si;
        tgt_inv[i] = ti;
        const bool sv = si > 0.f;
        const bool m = tmask ? tmask[i] != 0 : true;
        const bool tv = m && td > 0.f;
        key_s[i] = sv ? __float_as_uint(si) : 0xFFFFFFFFu;
        key_t[i] = tv ? __float_as_uint(ti) : 0xFFFFFFFFu;
        cs += sv;
        ct += tv;
        cm += m;
    }
    __shared__ unsigned sh[3];
    if (threadIdx.x < 3) sh[threadIdx.x] = 0;
    __syncthreads();
    atomicAdd(&sh[0], cs);
    atomicAdd(&sh[1], ct);
    atomicAdd(&sh[2], cm);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&st->count[0], sh[0]);
        atomicAdd(&st->count[1], sh[1]);
        atomicAdd(&st->mask_count, sh[2]);
    }
}

// torch.quantile rank arithmetic (ATen/native/Sorting.cpp quantile_compute): rank = q * (n - 1) in fp32, floor / ceil, weight.
__global__ void align_rank_kernel(AlignState* st) {
    const int a = threadIdx.x >> 1, qi = threadIdx.x & 1;  // 4 threads
    if (threadIdx.x >= 4) return;
    const unsigned n = st->count[a];
    const float q = qi ? 0.9f : 0.1f;
    const float r = n ? q * (float)(n - 1) : 0.f;
    const float lo = floorf(r), hi = ceilf(r);
    st->rank[(a * 2 + qi) * 2 + 0] = (unsigned)lo;
    st->rank[(a * 2 + qi) * 2 + 1] = (unsigned)hi;
    st->weight[a * 2 + qi] = r - lo;
}

// one radix digit: histogram of digit `shift` over keys matching each selector's prefix above that digit
__global__ __launch_bounds__(256) void align_hist_kernel(const unsigned* __restrict__ key_s, const unsigned* __restrict__ key_t,
                                                         AlignState* st, int n, int shift) {
    __shared__ unsigned h[NSEL][256];
    for (int i = threadIdx.x; i < NSEL * 256; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    unsigned pre[NSEL];
#pragma unroll
    for (int s = 0; s < NSEL; ++s) pre[s] = st->prefix[s];
    const unsigned himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const unsigned k = a ? key_t[i] : key_s[i];
            if (k == 0xFFFFFFFFu) continue;
#pragma unroll
            for (int s = a * 4; s < a * 4 + 4; ++s)
                if ((k & himask) == pre[s]) atomicAdd(&h[s][(k >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NSEL * 256; i += 256) {
        const unsigned v = (&h[0][0])[i];
        if (v) atomicAdd(&(&st->hist[0][0])[i], v);
    }
}

// pick the bucket that holds each selector's rank, descend into it, clear the histogram for the next digit
__global__ void align_pick_kernel(AlignState* st, int shift) {
    const int s = threadIdx.x;
    if (s < NSEL) {
        unsigned r = st->rank[s], acc = 0;
        int b = 0;
        for (; b < 255; ++b) {
            const unsigned c = st->hist[s][b];
            if (r < acc + c) break;
            acc += c;
        }
        st->rank[s] = r - acc;
        st->prefix[s] |= (unsigned)b << shift;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NSEL * 256; i += blockDim.x) (&st->hist[0][0])[i] = 0;
}

__global__ void align_quantile_kernel(AlignState* st) {
    const int a = threadIdx.x >> 1, qi = threadIdx.x & 1;
    if (threadIdx.x >= 4) return;
    const float lo = __uint_as_float(st->prefix[(a * 2 + qi) * 2 + 0]);
    const float hi = __uint_as_float(st->prefix[(a * 2 + qi) * 2 + 1]);
    const float w = st->weight[a * 2 + qi];
    const float v = w < 0.5f ? lo + w * (hi - lo) : hi - (hi - lo) * (1.0f - w);  // at::lerp
    if (qi) st->qhi[a] = v; else st->qlo[a] = v;
}

// inlier sums (fp64, fixed order: contiguous chunk per thread -> tree per block -> sequential over blocks)
__global__ __launch_bounds__(256) void align_fit_partial_kernel(const float* __restrict__ src_inv, const float* __restrict__ tgt_inv,
                                                                AlignState* st, int n) {
    const float slo = st->qlo[0], shi = st->qhi[0], tlo = st->qlo[1], thi = st->qhi[1];
    double s[5] = {0, 0, 0, 0, 0};
    const int per = (n + RED_BLOCKS * 256 - 1) / (RED_BLOCKS * 256);
    const int start = (blockIdx.x * 256 + threadIdx.x) * per;
    for (int i = start; i < min(start + per, n); ++i) {
        const float x = src_inv[i], y = tgt_inv[i];
        if (x > slo && x < shi && y > tlo && y < thi) {
            s[0] += 1.0; s[1] += x; s[2] += y; s[3] += (double)x * x; s[4] += (double)x * y;
        }
    }
    __shared__ double sh[256][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) sh[threadIdx.x][j] = s[j];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
#pragma unroll
            for (int j = 0; j < 5; ++j) sh[threadIdx.x][j] += sh[threadIdx.x + off][j];
        __syncthreads();
    }
    if (threadIdx.x < 5) st->partial[blockIdx.x][threadIdx.x] = sh[0][threadIdx.x];
}
__global__ void align_fit_final_kernel(AlignState* st) {
    if (threadIdx.x != 0) return;
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < RED_BLOCKS; ++b)
        for (int j = 0; j < 5; ++j) s[j] += st->partial[b][j];
    const double det = s[0] * s[3] - s[1] * s[1];
    st->scale = (float)((s[0] * s[4] - s[1] * s[2]) / det);
    st->bias = (float)((s[3] * s[2] - s[1] * s[4]) / det);
}
__global__ __launch_bounds__(256) void align_apply_kernel(const float* __restrict__ src_inv, const AlignState* st, float* __restrict__ out,
                                                          float* __restrict__ sc, float* __restrict__ m1, float* __restrict__ v2, int n) {
    const float scale = st->scale, bias = st->bias;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        out[i] = 1.0f / (src_inv[i] * scale + bias);
        if (sc) { sc[i] = 1.0f; m1[i] = 0.f; v2[i] = 0.f; }
    }
}

// ---- non-rigid refinement
G3_DEVICE float sgn(float x) { return (float)((x > 0.f) - (x < 0.f)); }
G3_DEVICE float box3(const float* __restrict__ a, int y, int x, int h, int w) {  // zero padding, row-major accumulation order
    float acc = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            acc = acc + ((yy >= 0 && yy < h && xx >= 0 && xx < w) ? a[yy * w + xx] : 0.f);
        }
    return acc;
}
__global__ __launch_bounds__(256) void align_arap_sign_kernel(const float* __restrict__ sc, float* __restrict__ e, int h, int w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    e[i] = sgn(box3(sc, y, x, h, w) * (1.0f / 9.0f) - sc[i]);
}

struct AdamArgs {
    float T[12];     // rows of inv(c2w)[:3,:4]
    float Kinv[9];
    float lambda_arap, step, bc2_sqrt, inv_hw;
};
__global__ __launch_bounds__(256) void align_adam_kernel(const float* __restrict__ ds, const float* __restrict__ dt,
                                                         const uint8_t* __restrict__ tmask, const float* __restrict__ e,
                                                         float* __restrict__ sc, float* __restrict__ m1, float* __restrict__ v2,
                                                         const AlignState* st, AdamArgs a, int h, int w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    float g = 0.f;
    const bool m = tmask ? tmask[i] != 0 : true;
    const float s = sc[i];
    if (m) {
        const float fx = (float)x, fy = (float)y;
        float un[3], cs[3], ct[3];
        const float d_s = ds[i] * s, d_t = dt[i];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            un[r] = (a.Kinv[r * 3 + 0] * fx + a.Kinv[r * 3 + 1] * fy) + a.Kinv[r * 3 + 2];
            cs[r] = d_s * un[r];
            ct[r] = d_t * un[r];
        }
        float sg[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float ps = ((a.T[r * 4 + 0] * cs[0] + a.T[r * 4 + 1] * cs[1]) + a.T[r * 4 + 2] * cs[2]) + a.T[r * 4 + 3];
            const float pt = ((a.T[r * 4 + 0] * ct[0] + a.T[r * 4 + 1] * ct[1]) + a.T[r * 4 + 2] * ct[2]) + a.T[r * 4 + 3];
            sg[r] = sgn(ps - pt);
        }
        float gdv = 0.f;
        float gc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) gc[k] = (a.T[0 * 4 + k] * sg[0] + a.T[1 * 4 + k] * sg[1]) + a.T[2 * 4 + k] * sg[2];
        gdv = (gc[0] * un[0] + gc[1] * un[1]) + gc[2] * un[2];
        const unsigned nm = st->mask_count;
        const float inv_data = nm ? (float)(1.0 / (3.0 * (double)nm)) : 0.f;
        g = (gdv * ds[i]) * inv_data;
    }
    const float ei = e[i];
    g = g + a.lambda_arap * ((box3(e, y, x, h, w) * (1.0f / 9.0f) - ei) * a.inv_hw);
    // torch.optim.Adam (betas 0.9 / 0.999, eps 1e-8): lerp_, mul_/addcmul_, sqrt/div/add, addcdiv_
    const float mm = m1[i] + (g - m1[i]) * (float)(1.0 - 0.9);  // the Python-double hyper-parameters, rounded to fp32 once
    const float vv = v2[i] * 0.999f + (float)(1.0 - 0.999) * g * g;
    m1[i] = mm;
    v2[i] = vv;
    const float denom = sqrtf(vv) / a.bc2_sqrt + 1e-8f;
    sc[i] = s - a.step * (mm / denom);
}
__global__ __launch_bounds__(256) void align_scale_kernel(float* __restrict__ out, const float* __restrict__ sc, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = out[i] * sc[i];
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t g3_align_depth_workspace_bytes(int H, int W) {
    const size_t n = (size_t)H * W;
    return align_up(sizeof(AlignState)) + 8 * align_up(n * 4);
}

extern "C" int g3_align_depth_f32(const float* source_depth, const float* target_depth, const uint8_t* target_mask, const float* Kinv_host,
                                  const float* T_host, int non_rigid, int num_iters, float lambda_arap, float lr, float* out_depth,
                                  void* workspace, size_t workspace_bytes, int H, int W, void* stream) {
    if (!source_depth || !target_depth || !out_depth || !workspace) return g3_set_error(G3_ERR_ARG, "g3_align_depth_f32: null operand");
    if (H <= 0 || W <= 0 || (int64_t)H * W > (1 << 30)) return g3_set_error(G3_ERR_ARG, "g3_align_depth_f32: bad shape %dx%d", H, W);
    if (workspace_bytes < g3_align_depth_workspace_bytes(H, W))
        return g3_set_error(G3_ERR_ARG, "g3_align_depth_f32: workspace too small (%zu < %zu)", workspace_bytes, g3_align_depth_workspace_bytes(H, W));
    if (non_rigid && (!Kinv_host || !T_host || num_iters < 0))
        return g3_set_error(G3_ERR_ARG, "g3_align_depth_f32: non-rigid alignment needs Kinv, T = inv(c2w) and num_iters >= 0");
    if ((uintptr_t)workspace & 255) return g3_set_error(G3_ERR_ARG, "g3_align_depth_f32: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int n = H * W;
    char* ws = (char*)workspace;
    AlignState* st = (AlignState*)ws;
    ws += align_up(sizeof(AlignState));
    const size_t plane = align_up((size_t)n * 4);
    float* src_inv = (float*)(ws + 0 * plane);
    float* tgt_inv = (float*)(ws + 1 * plane);
    unsigned* key_s = (unsigned*)(ws + 2 * plane);
    unsigned* key_t = (unsigned*)(ws + 3 * plane);
    float* sc = (float*)(ws + 4 * plane);
    float* m1 = (float*)(ws + 5 * plane);
    float* v2 = (float*)(ws + 6 * plane);
    float* e = (float*)(ws + 7 * plane);
    const int grid = min((n + 255) / 256, 1024);

    hipLaunchKernelGGL(align_zero_state, dim3(1), dim3(256), 0, s, st);
    hipLaunchKernelGGL(align_prep_kernel, dim3(grid), dim3(256), 0, s, source_depth, target_depth, target_mask, src_inv, tgt_inv, key_s, key_t, st, n);
    hipLaunchKernelGGL(align_rank_kernel, dim3(1), dim3(64), 0, s, st);
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(align_hist_kernel, dim3(grid), dim3(256), 0, s, key_s, key_t, st, n, shift);
        hipLaunchKernelGGL(align_pick_kernel, dim3(1), dim3(256), 0, s, st, shift);
    }
    hipLaunchKernelGGL(align_quantile_kernel, dim3(1), dim3(64), 0, s, st);
    hipLaunchKernelGGL(align_fit_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, s, src_inv, tgt_inv, st, n);
    hipLaunchKernelGGL(align_fit_final_kernel, dim3(1), dim3(64), 0, s, st);
    hipLaunchKernelGGL(align_apply_kernel, dim3(grid), dim3(256), 0, s, src_inv, st, out_depth, non_rigid ? sc : (float*)nullptr, m1, v2, n);
    if (non_rigid) {
        AdamArgs a;
        for (int i = 0; i < 12; ++i) a.T[i] = T_host[i];
        for (int i = 0; i < 9; ++i) a.Kinv[i] = Kinv_host[i];
        a.lambda_arap = lambda_arap;
        a.inv_hw = (float)(1.0 / ((double)H * W));
        double b1t = 1.0, b2t = 1.0;
        const int blocks = (n + 255) / 256;
        for (int t = 1; t <= num_iters; ++t) {
            b1t *= 0.9;
            b2t *= 0.999;
            a.step = (float)((double)lr / (1.0 - b1t));
            a.bc2_sqrt = (float)sqrt(1.0 - b2t);
            hipLaunchKernelGGL(align_arap_sign_kernel, dim3(blocks), dim3(256), 0, s, sc, e, H, W);
            hipLaunchKernelGGL(align_adam_kernel, dim3(blocks), dim3(256), 0, s, out_depth, target_depth, target_mask, e, sc, m1, v2, st, a, H, W);
        }
        hipLaunchKernelGGL(align_scale_kernel, dim3(grid), dim3(256), 0, s, out_depth, sc, n);
    }
    return g3_check_launch("g3_align_depth_f32");
}
