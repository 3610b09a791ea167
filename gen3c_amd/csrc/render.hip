// 3D-cache renderer: unproject -> project/warp -> bilinear splat -> (optional) mesh-occlusion masking.
//
// Replaces, for the GEN3C cache path (cache_3d.py:183-223 -> forward_warp(depth1=None, world_points1=...)):
//   project_points / forward_warp / bilinear_splatting   forward_warp_utils_pytorch.py:171-336, 462-486, 576-695
//   points_to_mesh + get_camera_rays                      forward_warp_utils_pytorch.py:49-132, 151-168
//   ray_triangle_intersection (NVIDIA Warp kernel)        ray_triangle_intersection_warp.py:23-105
//   unproject_points, reliable_depth_mask_range_batch     forward_warp_utils_pytorch.py:338-353, 410-460
//
// The reference runs ~40 elementwise torch kernels + 8 index_put_(accumulate) scatters per splat and a brute-force
// rays x triangles loop. Here an item (one target frame of one cache buffer) costs three streaming passes:
//   (1) project: world point -> camera z, flow, validity, per-GROUP max of log1p(z) (the reference takes that max over
//       every item of one forward_warp call = warp_chunk_size items, so groups reproduce its pairing);
//   (2) splat:   4 corner weights, 5 fp32 atomics per corner (r, g, b, z, weight) into an [h+2][w+2][5] accumulator;
//   (3) resolve: normalise, fill, clamp, crop.
// Mesh occlusion is a conservative rasteriser: each boundary patch (2 triangles of the 4x-downsampled mesh) is
// projected to its pixel bounding box and ONLY those rays are tested with the same Moller-Trumbore arithmetic the
// reference applies to every (ray, triangle) pair; min-t is an atomicMin on the float bit pattern, so the result is
// independent of evaluation order and identical to the brute-force min.
//
// Arithmetic: fp32 with the explicit operation order of oracle/warp_oracle.py (the library is built with
// -ffp-contract=off): pixel indices and masks are bit-exact w.r.t. the oracle; accumulated floats depend on atomic
// order (as they do in the reference) and on the log / exp of the depth weight (hardware log2 / exp2 in the window splat, ~1e-5 relative).
// One deliberate difference on garbage input: a NaN camera-space z makes the reference's `log_depth1.max()` NaN, which poisons the weights
// of EVERY pixel of the call (nan_to_num then marks all of them valid with NaN colours); here fmaxf ignores the NaN, the pixel itself is
// masked by `z > 0` and the rest renders normally. GEN3C never produces NaN points (unproject zeroes points whose depth is <= 0).
#include "common.hpp"
#include <algorithm>
#include <mutex>

namespace {

constexpr int ACC_C = 5;  // r, g, b, z, weight

struct Mat { float m[16]; };

// ---------------------------------------------------------------------------------------------------------------
// (1) project
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_project_kernel(const float* __restrict__ points, const float* __restrict__ w2c,
                                                           const float* __restrict__ Kmat, const float* __restrict__ mask1,
                                                           float* __restrict__ zbuf, float* __restrict__ flow,
                                                           float* __restrict__ cam_out, float* __restrict__ maskz,
                                                           unsigned* __restrict__ group_max, int n, int h, int w,
                                                           int group_size, const int* __restrict__ src = nullptr) {
    const int item = blockIdx.y;
    const int hw = h * w;
    const int sitem = src ? src[item] : item;  // cache entry (source view) this item renders: points / mask1 are per SOURCE when src is given
    const float* W = w2c + item * 16;
    const float* K = Kmat + item * 9;
    // This pass is VALU-bound before it is HBM-bound (~150 instructions per pixel against 32 bytes): the maximum of log1p(z) is taken as
    // log1p of the maximum z (log1p is monotone; one libm call per thread instead of one per pixel) and the pixel's row / column are stepped
    // instead of divided out of the linear index. The two IEEE divisions stay: the flow is bit-exact against the reference's goldens.
    float zmax = 0.f;  // max(z, 0) over this thread's pixels; NaN z is ignored by fmaxf
    const int stride = gridDim.x * 256;
    const int step_y = stride / w, step_x = stride - step_y * w;
    int pix = blockIdx.x * 256 + threadIdx.x;
    int py = pix / w, px = pix - py * w;
    for (; pix < hw; pix += stride) {
        const int64_t o = (int64_t)item * hw + pix;
        const int64_t so = (int64_t)sitem * hw + pix;
        const float x = points[so * 3 + 0], y = points[so * 3 + 1], zz = points[so * 3 + 2];
        float cam[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) cam[i] = ((W[i * 4 + 0] * x + W[i * 4 + 1] * y) + W[i * 4 + 2] * zz) + W[i * 4 + 3] * 1.0f;
        float pr[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pr[i] = (K[i * 3 + 0] * cam[0] + K[i * 3 + 1] * cam[1]) + K[i * 3 + 2] * cam[2];
        const float z = pr[2];
        const float u = pr[0] / (z + 1e-7f);
        const float v = pr[1] / (z + 1e-7f);
        flow[((int64_t)item * 2 + 0) * hw + pix] = u - (float)px;
        flow[((int64_t)item * 2 + 1) * hw + pix] = v - (float)py;
        zbuf[o] = z;
        const float m = (mask1 ? mask1[so] : 1.0f) * ((z > 0.f) ? 1.0f : 0.0f);
        maskz[o] = m;
        if (cam_out) { cam_out[o * 3 + 0] = cam[0]; cam_out[o * 3 + 1] = cam[1]; cam_out[o * 3 + 2] = cam[2]; }
        zmax = fmaxf(zmax, fmaxf(z, 0.f));
        px += step_x; py += step_y;
        if (px >= w) { px -= w; ++py; }
    }
    float local_max = log1pf(zmax);  // >= 0
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, o, 64));
    // one atomic per WORKGROUP: thousands of same-address atomics per item serialise in the L2 (they were most of this kernel's time)
    __shared__ float wave_max[4];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = local_max;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        atomicMax(group_max + item / group_size, __float_as_uint(m));  // non-negative floats order like uints
    }
}

// z-only pre-pass of the FUSED form (round 5): the splat recomputes a pixel's projection itself (warp_splat_windows_kernel<true>), but its depth weight
// exp(50 log1p(z) / max) needs the maximum of log1p(z) over the whole GROUP of items before any pixel is weighted. This pass reads the points (12 bytes per
// pixel), evaluates z with the operation order of warp_project_kernel (the same float, bit for bit) and leaves the group maximum; nothing else is written.
__global__ __launch_bounds__(256) void warp_zmax_kernel(const float* __restrict__ points, const float* __restrict__ w2c, const float* __restrict__ Kmat,
                                                        unsigned* __restrict__ group_max, int n, int h, int w, int group_size, const int* __restrict__ src) {
    const int item = blockIdx.y;
    const int hw = h * w;
    const int sitem = src ? src[item] : item;
    const float* W = w2c + item * 16;
    const float* K = Kmat + item * 9;
    float zmax = 0.f;
    const int stride = gridDim.x * 256;
    // four pixels per trip, their loads first: with one load per trip the pass waited out a memory latency per pixel (49 us per 16 items for 173 MB)
    for (int pix0 = blockIdx.x * 256 + threadIdx.x; pix0 < hw; pix0 += 4 * stride) {
        float x[4], y[4], zz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pix = min(pix0 + u * stride, hw - 1);  // (a repeated pixel does not change a maximum)
            const int64_t so = (int64_t)sitem * hw + pix;
            x[u] = points[so * 3 + 0]; y[u] = points[so * 3 + 1]; zz[u] = points[so * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float cam[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) cam[i] = ((W[i * 4 + 0] * x[u] + W[i * 4 + 1] * y[u]) + W[i * 4 + 2] * zz[u]) + W[i * 4 + 3] * 1.0f;
            const float z = (K[6] * cam[0] + K[7] * cam[1]) + K[8] * cam[2];
            zmax = fmaxf(zmax, fmaxf(z, 0.f));
        }
    }
    float local_max = log1pf(zmax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, o, 64));
    __shared__ float wave_max[4];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = local_max;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        // one L2 atomic per workgroup on a handful of addresses serialises (~10 ns each, 14 080 workgroups per 16 items): most workgroups are below
        // the running maximum and find that out with a load (a stale smaller value only costs the atomic it would have taken anyway)
        unsigned* gm = group_max + item / group_size;
        if (__float_as_uint(m) > __hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(gm, __float_as_uint(m));
    }
}

// small vector helpers of the mesh-occlusion test (also used by the resolve pass when it applies the occlusion itself)
struct V3 { float x, y, z; };
G3_DEVICE V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
G3_DEVICE V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
G3_DEVICE float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

G3_DEVICE V3 pixel_ray(const float* Ki, int px, int py) {  // get_camera_rays: K^-1 [x,y,1], normalised
    const float xs = (float)px, ys = (float)py;
    V3 d;
    d.x = (Ki[0] * xs + Ki[1] * ys) + Ki[2] * 1.0f;
    d.y = (Ki[3] * xs + Ki[4] * ys) + Ki[5] * 1.0f;
    d.z = (Ki[6] * xs + Ki[7] * ys) + Ki[8] * 1.0f;
    float nrm = sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z);
    if (nrm == 0.f) nrm = 1.f;
    return {d.x / nrm, d.y / nrm, d.z / nrm};
}


// ---------------------------------------------------------------------------------------------------------------
// (2) splat
// ---------------------------------------------------------------------------------------------------------------
G3_DEVICE int clampi(long long v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }

struct SplatGeom { int fx, cx, fy, cy; float nw, sw, ne, se; };
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

G3_DEVICE SplatGeom splat_geom(float flow_x, float flow_y, int px, int py, int h, int w) {
    const float tx = flow_x + (float)px, ty = flow_y + (float)py;  // trans_pos = flow12 + grid
    const float ox = tx + 1.0f, oy = ty + 1.0f;
    SplatGeom g;
    // floor/ceil -> .long() -> clamp; NaN/inf follow the CPU reference (NaN -> INT64_MIN -> clamps to 0). The clamp is taken on the float
    // (integer-valued, bounds exactly representable: the same result as converting to int64 first, for every input; fmaxf(NaN, 0) = 0) - the
    // int64 form cost ~60 VALU instructions per pixel in a kernel that is VALU-bound (PMC: profiles/r3_pmc_render.txt)
    const float flx = floorf(ox), clx = ceilf(ox), fly = floorf(oy), cly = ceilf(oy);
    auto clampf = [](float f, int hi) -> int { return (int)fminf(fmaxf(f, 0.f), (float)hi); };
    g.fx = clampf(flx, w + 1);
    g.cx = clampf(clx, w + 1);
    g.fy = clampf(fly, h + 1);
    g.cy = clampf(cly, h + 1);
    const float oxc = fminf(fmaxf(ox, 0.f), (float)(w + 1));
    const float oyc = fminf(fmaxf(oy, 0.f), (float)(h + 1));
    const float wy_f = 1.0f - (oyc - (float)g.fy);
    const float wy_c = 1.0f - ((float)g.cy - oyc);
    const float wx_f = 1.0f - (oxc - (float)g.fx);
    const float wx_c = 1.0f - ((float)g.cx - oxc);
    g.nw = wy_f * wx_f; g.sw = wy_c * wx_f; g.ne = wy_f * wx_c; g.se = wy_c * wx_c;
    return g;
}

__global__ __launch_bounds__(256) void warp_splat_kernel(const float* __restrict__ image, const float* __restrict__ zbuf,
                                                         const float* __restrict__ flow, const float* __restrict__ maskz,
                                                         const unsigned* __restrict__ group_max, float* __restrict__ accum,
                                                         int n, int h, int w, int group_size) {
    const int item = blockIdx.y;
    const int hw = h * w;
    const float lmax = __uint_as_float(group_max[item / group_size]);
    const int aw = w + 2;
    float* acc_item = accum + (int64_t)item * (h + 2) * aw * ACC_C;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < hw; pix += gridDim.x * 256) {
        const int64_t o = (int64_t)item * hw + pix;
        const float m = maskz[o];
        if (m == 0.f) continue;  // zero weight: contributes nothing (the reference adds exact zeros)
        const int py = pix / w, px = pix - py * w;
        const float z = zbuf[o];
        const SplatGeom g = splat_geom(flow[((int64_t)item * 2 + 0) * hw + pix], flow[((int64_t)item * 2 + 1) * hw + pix], px, py, h, w);
        const float logd = log1pf(fmaxf(z, 0.f));
        const float expo = logd / (lmax + 1e-7f) * 50.0f;
        const float dw = expf(fminf(expo, 80.0f)) + 1e-7f;
        const float r = image[((int64_t)item * 3 + 0) * hw + pix];
        const float gc = image[((int64_t)item * 3 + 1) * hw + pix];
        const float b = image[((int64_t)item * 3 + 2) * hw + pix];
        const float wts[4] = {g.nw * m * 1.0f / dw, g.sw * m * 1.0f / dw, g.ne * m * 1.0f / dw, g.se * m * 1.0f / dw};
        const int ys[4] = {g.fy, g.cy, g.fy, g.cy};
        const int xs[4] = {g.fx, g.fx, g.cx, g.cx};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float* a = acc_item + ((int64_t)ys[c] * aw + xs[c]) * ACC_C;
            const float wt = wts[c];
            unsafeAtomicAdd(a + 0, r * wt);
            unsafeAtomicAdd(a + 1, gc * wt);
            unsafeAtomicAdd(a + 2, b * wt);
            unsafeAtomicAdd(a + 3, z * wt);
            unsafeAtomicAdd(a + 4, wt);
        }
    }
}

// Tiled splat (default): the accumulator atomics of the kernel above are bound by the L2's read-modify-write rate (20 per
// source pixel, ~4 pixels hitting every texel channel). Here a workgroup owns a 32 x 32 tile of SOURCE pixels, whose
// destinations are (for a smooth flow) a compact patch: it accumulates them with LDS atomics into a WIN x WIN x 5 window
// anchored at the tile's smallest destination corner, then adds the window into the global accumulator with ONE atomic per
// touched texel channel. Corners outside the window (depth discontinuities inside the tile, extreme zoom) go straight to
// global atomics, so any flow stays correct. Same contributions as the direct kernel; only the (already order-dependent)
// summation order differs.
constexpr int TS = 32;    // source tile edge
constexpr int ORG_N = 4;  // per source tile: window origin (x, y) and used extent (columns, rows); origin x = INT_MAX: nothing in the tile
constexpr int WIN = 40;   // destination window edge (32 KiB of LDS: 4-5 workgroups per CU; 48: 0.105 vs 0.095 ms per item on the bench scene)
__global__ __launch_bounds__(256) void warp_splat_tiled_kernel(const float* __restrict__ image, const float* __restrict__ zbuf,
                                                               const float* __restrict__ flow, const float* __restrict__ maskz,
                                                               const unsigned* __restrict__ group_max, float* __restrict__ accum,
                                                               int n, int h, int w, int group_size, int tiles_x) {
    __shared__ float win[WIN * WIN * ACC_C];
    __shared__ int org[2];
    const int item = blockIdx.y;
    const int hw = h * w;
    const float lmax = __uint_as_float(group_max[item / group_size]);
    const int aw = w + 2;
    float* acc_item = accum + (int64_t)item * (h + 2) * aw * ACC_C;
    const int ty0 = (blockIdx.x / tiles_x) * TS, tx0 = (blockIdx.x % tiles_x) * TS;
    if (threadIdx.x == 0) { org[0] = 0x7fffffff; org[1] = 0x7fffffff; }
    for (int i = threadIdx.x; i < WIN * WIN * ACC_C; i += 256) win[i] = 0.f;
    __syncthreads();

    // 4 pixels per thread: rows ty0 + (threadIdx.x >> 5) + 8 k, column tx0 + (threadIdx.x & 31)
    SplatGeom g[4];
    float wscale[4], col[4][4];  // m / dw ; r, g, b, z
    bool on[4];
    int mnx = 0x7fffffff, mny = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int py = ty0 + (threadIdx.x >> 5) + 8 * k, px = tx0 + (threadIdx.x & 31);
        on[k] = false;
        if (py >= h || px >= w) continue;
        const int pix = py * w + px;
        const int64_t o = (int64_t)item * hw + pix;
        const float m = maskz[o];
        if (m == 0.f) continue;
        on[k] = true;
        const float z = zbuf[o];
        g[k] = splat_geom(flow[((int64_t)item * 2 + 0) * hw + pix], flow[((int64_t)item * 2 + 1) * hw + pix], px, py, h, w);
        const float logd = log1pf(fmaxf(z, 0.f));
        const float expo = logd / (lmax + 1e-7f) * 50.0f;
        const float dw = expf(fminf(expo, 80.0f)) + 1e-7f;
        wscale[k] = dw;
        col[k][0] = image[((int64_t)item * 3 + 0) * hw + pix];
        col[k][1] = image[((int64_t)item * 3 + 1) * hw + pix];
        col[k][2] = image[((int64_t)item * 3 + 2) * hw + pix];
        col[k][3] = z;
        mnx = min(mnx, g[k].fx);
        mny = min(mny, g[k].fy);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, o, 64));
        mny = min(mny, __shfl_xor(mny, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&org[0], mnx); atomicMin(&org[1], mny); }
    __syncthreads();
    const int ox = org[0], oy = org[1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!on[k]) continue;
        const int py = ty0 + (threadIdx.x >> 5) + 8 * k, px = tx0 + (threadIdx.x & 31);
        const float m = maskz[(int64_t)item * hw + py * w + px];
        const float dw = wscale[k];
        const float wts[4] = {g[k].nw * m * 1.0f / dw, g[k].sw * m * 1.0f / dw, g[k].ne * m * 1.0f / dw, g[k].se * m * 1.0f / dw};
        const int ys[4] = {g[k].fy, g[k].cy, g[k].fy, g[k].cy};
        const int xs[4] = {g[k].fx, g[k].fx, g[k].cx, g[k].cx};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float wt = wts[c];
            const int lx = xs[c] - ox, ly = ys[c] - oy;
            const float v[ACC_C] = {col[k][0] * wt, col[k][1] * wt, col[k][2] * wt, col[k][3] * wt, wt};
            if ((unsigned)lx < (unsigned)WIN && (unsigned)ly < (unsigned)WIN) {
                float* a = win + (ly * WIN + lx) * ACC_C;
#pragma unroll
                for (int e = 0; e < ACC_C; ++e) atomicAdd(a + e, v[e]);
            } else {
                float* a = acc_item + ((int64_t)ys[c] * aw + xs[c]) * ACC_C;
#pragma unroll
                for (int e = 0; e < ACC_C; ++e) unsafeAtomicAdd(a + e, v[e]);
            }
        }
    }
    __syncthreads();
    if (ox == 0x7fffffff) return;  // nothing valid in this tile
    for (int i = threadIdx.x; i < WIN * WIN; i += 256) {
        const float* a = win + i * ACC_C;
        // a texel is touched iff some contribution was added; contributions may be exact zeros (zero bilinear weight), which
        // the direct kernel also adds - skipping them changes nothing
        if (a[0] == 0.f && a[1] == 0.f && a[2] == 0.f && a[3] == 0.f && a[4] == 0.f) continue;
        const int ly = i / WIN, lx = i - ly * WIN;
        const int gy = oy + ly, gx = ox + lx;
        if (gy > h + 1 || gx > w + 1) continue;  // cannot happen (corners are clamped), defensive
        float* d = acc_item + ((int64_t)gy * aw + gx) * ACC_C;
#pragma unroll
        for (int e = 0; e < ACC_C; ++e)
            if (a[e] != 0.f) unsafeAtomicAdd(d + e, a[e]);
    }
}

// Window splat (default; gen3c_amd/renderer.py: _WINDOW_SPLAT): the tiled kernel above pays one global atomic per touched texel channel when
// it flushes its window (~5 M per 704 x 1280 item). Here the flush is a plain, fully coalesced STORE of the window (WIN x WIN x 5 floats = 32 KiB at WIN = 40) plus its origin
// into a per-source-tile workspace, and a second kernel owns the DESTINATION: a workgroup per 32 x 32 output tile lists the source tiles
// whose windows overlap it (ascending tile order), sums their texels - plus the global accumulator, which only the rare out-of-window corners
// still reach - and resolves the pixel in the same pass: no global atomics on the common path and no separate resolve pass over the
// accumulator (3.85 + 0.21 -> 3.27 + 0.47 ms per 32 items). What bounds both forms is the LDS accumulation itself: rocprofv3 PMC
// (tools/gpu_pmc_render.sh) shows ~100 LDS-busy cycles per ds_add_f32 wave instruction (SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS, address conflicts
// flagged on 80 % of them) and 76 % of the wave cycles waiting on LDS issue - 20 float atomics per source pixel in the plain form. The
// window kernel therefore merges the contributions of neighbouring pixels in registers first (see its accumulation loop).
// cross-lane moves for the contribution merge below (gfx9 DPP wavefront shifts, gfx950 v_permlane32_swap): no LDS traffic
G3_DEVICE int lane_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1: lane i <- lane i - 1 */, 0xf, 0xf, false); }
G3_DEVICE int lane_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1: lane i <- lane i + 1 */, 0xf, 0xf, false); }
G3_DEVICE float lane_prev(float v) { return __int_as_float(lane_prev(__float_as_int(v))); }
G3_DEVICE int from_lane_minus32(int v) {  // lanes 32..63 receive the value of lane - 32 (lanes 0..31 keep their own)
    int a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a;
}
G3_DEVICE int from_lane_plus32(int v) {   // lanes 0..31 receive the value of lane + 32
    int a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return b;
}
G3_DEVICE float from_lane_minus32(float v) { return __int_as_float(from_lane_minus32(__float_as_int(v))); }

// One scan over up to NCH x 256 tile rectangles of an item (`origins` entries: origin x, y, extent x, y; x = INT_MAX: empty tile): every thread's loads
// go out first, then ONE pair of barriers. hit(t, og) selects; the selected rectangles are handed to emit(position, t, og) in ascending tile order at
// positions at, at + 1, ... - when all of them fit below `cap` (otherwise none is emitted). Returns the number selected.
template <int NCH, class Hit, class Emit>
G3_DEVICE int scan_rects(const int* __restrict__ org_item, int ntiles, int base, int* cnt_lds /* NCH x 4 ints of LDS */, int at, int cap, Hit hit, Emit emit) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int4 og[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int t = base + 256 * c + (int)threadIdx.x;
        og[c] = make_int4(0x7fffffff, 0x7fffffff, 0, 0);
        if (t < ntiles) og[c] = *reinterpret_cast<const int4*>(org_item + ORG_N * t);
    }
    bool h[NCH];
    unsigned long long bal[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        h[c] = og[c].x != 0x7fffffff && hit(base + 256 * c + (int)threadIdx.x, og[c]);
        bal[c] = __ballot(h[c]);
        if (lane == 0) cnt_lds[c * 4 + wv] = __popcll(bal[c]);
    }
    __syncthreads();
    int total = 0, off[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i == wv) off[c] = total;
            total += cnt_lds[c * 4 + i];
        }
    }
    if (at + total <= cap) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (h[c]) emit(at + off[c] + __popcll(bal[c] & ((1ull << lane) - 1ull)), base + 256 * c + (int)threadIdx.x, og[c]);
    }
    __syncthreads();
    return total;
}

// warp_project_kernel's arithmetic for one pixel, operation for operation (the library is built with -ffp-contract=off): world point -> camera ->
// pixel, flow = u - px (splat_geom adds px back: the rounding of the reference's flow12 + grid), validity = mask * (z > 0). Shared by the fused
// splat and the extent pre-pass so that both see the same floats.
G3_DEVICE void project_pixel(const float* __restrict__ W, const float* __restrict__ K, int px, int py, float& x_flx, float& y_fly, float& z_zin, float& mk) {
    const float x = x_flx, y = y_fly, zz = z_zin;
    float cam[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) cam[i] = ((W[i * 4 + 0] * x + W[i * 4 + 1] * y) + W[i * 4 + 2] * zz) + W[i * 4 + 3] * 1.0f;
    float pr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pr[i] = (K[i * 3 + 0] * cam[0] + K[i * 3 + 1] * cam[1]) + K[i * 3 + 2] * cam[2];
    const float z = pr[2];
    const float u = pr[0] / (z + 1e-7f);
    const float v = pr[1] / (z + 1e-7f);
    x_flx = u - (float)px;
    y_fly = v - (float)py;
    z_zin = z;
    mk = mk * ((z > 0.f) ? 1.0f : 0.0f);
}

// The resolve of one texel from its five sums (forward_warp_utils_pytorch.py:660-695: normalise, nan_to_num, fill -1, clamp), with the mesh
// occlusion's `keep` applied the way mesh_apply_kernel does. One function for the gather pass and for the splat's single-writer texels.
struct ResolvedTexel { float c[3], m, d; };
G3_DEVICE ResolvedTexel resolve_sums(const float (&sum)[ACC_C], bool occlusion, float keep) {
    ResolvedTexel r;
    float wt = sum[4];
    if (wt != wt) wt = 1000.0f;  // nan_to_num(nan=1000)
    const bool ok = wt > 0.f;
    const float dval = ok ? sum[3] / wt : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = ok ? sum[c] / wt : -1.0f;
        v = fminf(fmaxf(v, -1.0f), 1.0f);
        if (occlusion) v = (v + 1.0f) * keep - 1.0f;
        r.c[c] = v;
    }
    r.m = occlusion ? (ok ? 1.0f : 0.0f) * keep : (ok ? 1.0f : 0.0f);
    r.d = occlusion ? dval * keep : dval;
    return r;
}

// Extent pre-pass of the single-writer form (round 5, g3_render_items_f32 with render_exclusive = 1; replaces warp_zmax_kernel there): one
// workgroup per SOURCE tile, the thread -> pixel map of the splat. It evaluates the projection and the corner geometry only and publishes, before
// any tile splats, (a) the group maximum of log1p(z) the depth weights need and (b) every tile's destination rectangle (origin + used extent,
// clamped to the window) in `origins`. With all rectangles known, a destination texel that lies in exactly ONE rectangle has a single writer: the
// splat resolves it from LDS straight into frame / mask / depth and only texels shared between tiles make the round trip through the window
// workspace. A rectangle wider than the window (a tile across a depth discontinuity) is published at its full size: its corners beyond the window go
// to the dense accumulator, and the gather pass reads the accumulator for exactly those texels instead of for every pixel of a dirty item.
__global__ __launch_bounds__(256) void warp_extent_kernel(const float* __restrict__ points, const float* __restrict__ w2c, const float* __restrict__ Kmat,
                                                          const float* __restrict__ mask1, unsigned* __restrict__ group_max, int* __restrict__ origins,
                                                          int n, int h, int w, int group_size, int tiles_x, const int* __restrict__ src_idx) {
    __shared__ int red[4][4];
    __shared__ float wave_max[4];
    const int item = blockIdx.y;
    const int hw = h * w;
    const int sitem = src_idx ? src_idx[item] : item;
    const int ty0 = (blockIdx.x / tiles_x) * TS, tx0 = (blockIdx.x % tiles_x) * TS;
    const int64_t slot = (int64_t)item * gridDim.x + blockIdx.x;
    const float* W = w2c + item * 16;
    const float* K = Kmat + item * 9;
    float x[4], y[4], zz[4], mk[4];
    bool inb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int py = ty0 + 4 * (threadIdx.x >> 5) + k, px = tx0 + (threadIdx.x & 31);
        inb[k] = py < h && px < w;
        const int pix = inb[k] ? py * w + px : 0;
        const int64_t so = (int64_t)sitem * hw + pix;
        x[k] = points[so * 3 + 0];
        y[k] = points[so * 3 + 1];
        zz[k] = points[so * 3 + 2];
        mk[k] = mask1 ? mask1[so] : 1.0f;
    }
    int mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -1, mxy = -1;
    float zmax = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int py = ty0 + 4 * (threadIdx.x >> 5) + k, px = tx0 + (threadIdx.x & 31);
        project_pixel(W, K, px, py, x[k], y[k], zz[k], mk[k]);
        if (inb[k]) zmax = fmaxf(zmax, fmaxf(zz[k], 0.f));  // over every pixel of the item, masked or not (warp_zmax_kernel / warp_project_kernel)
        if (inb[k] && mk[k] != 0.f) {
            const SplatGeom gk = splat_geom(x[k], y[k], px, py, h, w);
            mnx = min(mnx, gk.fx);
            mny = min(mny, gk.fy);
            mxx = max(mxx, gk.cx);
            mxy = max(mxy, gk.cy);
        }
    }
    float local_max = log1pf(zmax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, o, 64));
        mny = min(mny, __shfl_xor(mny, o, 64));
        mxx = max(mxx, __shfl_xor(mxx, o, 64));
        mxy = max(mxy, __shfl_xor(mxy, o, 64));
        local_max = fmaxf(local_max, __shfl_xor(local_max, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        int* r = red[threadIdx.x >> 6];
        r[0] = mnx; r[1] = mny; r[2] = mxx; r[3] = mxy;
        wave_max[threadIdx.x >> 6] = local_max;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        // one L2 atomic per workgroup on a handful of addresses serialises (~10 ns each, 14 080 workgroups per 16 items): most workgroups are below
        // the running maximum and find that out with a load (a stale smaller value only costs the atomic it would have taken anyway)
        unsigned* gm = group_max + item / group_size;
        if (__float_as_uint(m) > __hip_atomic_load(gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(gm, __float_as_uint(m));
        const int ox = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
        const int oy = min(min(red[0][1], red[1][1]), min(red[2][1], red[3][1]));
        int* og = origins + ORG_N * slot;
        if (ox == 0x7fffffff) {  // nothing valid in this tile: the splat returns and the gather skips it by its origin
            og[0] = ox; og[1] = ox; og[2] = 0; og[3] = 0;
        } else {
            const int fx = max(max(red[0][2], red[1][2]), max(red[2][2], red[3][2])) - ox + 1;
            const int fy = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3])) - oy + 1;
            // UNCLAMPED extent: the tile's window is its first min(WIN, extent) columns / rows; corners beyond it go to the dense accumulator, and
            // the full rectangle is what tells the other tiles (and the gather pass) which texels this tile may reach
            og[0] = ox; og[1] = oy; og[2] = fx; og[3] = fy;
        }
    }
}

// FUSED (round 5, g3_render_items_f32): the projection of warp_project_kernel is evaluated HERE from the points (same operation order: z, flow and validity are
// the same floats) - the z / flow / validity planes are neither written nor read back (32 bytes per pixel and item less on the memory side).
// EXCL (round 5, with FUSED): the tile's destination rectangle comes from the extent pre-pass (warp_extent_kernel) instead of being reduced here, and
// after the accumulation every window texel no OTHER tile's rectangle covers is resolved from LDS straight into frame / mask / depth (the arithmetic
// of the gather pass on a single contribution: 0 + value); only shared texels are written to the window workspace.
template <bool FUSED, bool EXCL = false>
__global__ __launch_bounds__(256) void warp_splat_windows_kernel(const float* __restrict__ image, const float* __restrict__ zbuf,
                                                                 const float* __restrict__ flow, const float* __restrict__ maskz,
                                                                 const unsigned* __restrict__ group_max, float* __restrict__ accum,
                                                                 float* __restrict__ windows, int* __restrict__ origins, int n, int h, int w,
                                                                 int group_size, int tiles_x, const int* __restrict__ src_idx = nullptr,
                                                                 unsigned* __restrict__ dirty = nullptr, unsigned epoch = 0,
                                                                 const float* __restrict__ points = nullptr, const float* __restrict__ w2c = nullptr,
                                                                 const float* __restrict__ Kmat = nullptr, const float* __restrict__ mask1 = nullptr,
                                                                 float* __restrict__ frame = nullptr, float* __restrict__ mask_out = nullptr,
                                                                 float* __restrict__ depth_out = nullptr, int occlusion = 0, int full_extent = 0) {
    __shared__ __attribute__((aligned(16))) float win[WIN * WIN * ACC_C];
    __shared__ int org_w[4][4];  // per-wave minima of the north-west / maxima of the south-east destination corners: window origin and extent
    __shared__ int owner[WIN * WIN];  // which pixel of the tile stores (instead of atomically adding) into a window texel: see the phases below
    const int item = blockIdx.y;
    const int hw = h * w;
    const int sitem = src_idx ? src_idx[item] : item;  // the image is per SOURCE view when src_idx is given
    const float lmax = __uint_as_float(group_max[item / group_size]);
    const int aw = w + 2;
    float* acc_item = accum + (int64_t)item * (h + 2) * aw * ACC_C;
    const int ty0 = (blockIdx.x / tiles_x) * TS, tx0 = (blockIdx.x % tiles_x) * TS;
    const int64_t slot = (int64_t)item * gridDim.x + blockIdx.x;
    // All 28 loads of the thread's four pixels go out first, unconditionally (an out-of-image pixel reads pixel 0 and is switched off below): with
    // the loads behind `if (mask != 0)` every pixel paid two dependent HBM round trips and a workgroup - LDS holds only four per CU - sat idle
    // through eight of them. The zero fill of the window runs underneath them.
    float mk[4], zin[4], flx[4], fly[4], rgb[4][3];
    bool inb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int py = ty0 + 4 * (threadIdx.x >> 5) + k, px = tx0 + (threadIdx.x & 31);  // a thread owns 4 consecutive rows of one column
        inb[k] = py < h && px < w;
        const int pix = inb[k] ? py * w + px : 0;
        const int64_t o = (int64_t)item * hw + pix;
        if constexpr (FUSED) {  // (raw operands now, the projection below - behind the window's zero fill)
            const int64_t so = (int64_t)sitem * hw + pix;
            flx[k] = points[so * 3 + 0];
            fly[k] = points[so * 3 + 1];
            zin[k] = points[so * 3 + 2];
            mk[k] = mask1 ? mask1[so] : 1.0f;
        } else {
            mk[k] = maskz[o];
            zin[k] = zbuf[o];
            flx[k] = flow[((int64_t)item * 2 + 0) * hw + pix];
            fly[k] = flow[((int64_t)item * 2 + 1) * hw + pix];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[k][c] = image[((int64_t)sitem * 3 + c) * hw + pix];
    }
    {
        f32x4* wz = reinterpret_cast<f32x4*>(win);
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        for (int i = threadIdx.x; i < WIN * WIN * ACC_C / 4; i += 256) wz[i] = zero4;
        for (int i = threadIdx.x; i < WIN * WIN; i += 256) owner[i] = -1;
    }
    if constexpr (FUSED) {
        // warp_project_kernel's arithmetic (project_pixel above)
        const float* W = w2c + item * 16;
        const float* K = Kmat + item * 9;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int py = ty0 + 4 * (threadIdx.x >> 5) + k, px = tx0 + (threadIdx.x & 31);
            project_pixel(W, K, px, py, flx[k], fly[k], zin[k], mk[k]);
        }
    }
    // Accumulate into the window. A thread owns 4 consecutive rows of one column; a wave 8 rows x 32 columns (lanes 0..31: rows 8 v .. 8 v + 3,
    // lanes 32..63: rows 8 v + 4 .. 8 v + 7). LDS float atomics are what bounds this kernel (see above: ~6 LDS cycles per LANE), so contributions
    // that meet in one destination texel are summed in registers first and every texel then gets, as far as possible, ONE plain store.
    // A pixel has a west and an east pair of corners; each corner is (texel id = y << 16 | x, 5 values) and id -1 means "nothing left to write".
    //   (1) inside the pixel: corners that coincide (floor == ceil on an integer coordinate) are summed;
    //   (2) across lanes: the east corners of lane i - 1 (DPP wave shift) are added to ANY west corner of lane i with the same texel in rows
    //       k - 1 .. k + 1. Under the reference's default trajectories (camera moving along x) flow_y is 0 up to rounding, so a row's y lands on,
    //       just below or just above an integer pixel by pixel and "same row, same corner" (the round-2 rule) matched only ~54 % of the
    //       neighbours; matching by texel id finds a home for ~all of them (tools/render_merge_model.py: 1.7 -> 0.3 atomic corners per pixel);
    //   (3) down the thread's column, and from the last row of lane i to the first rows of lane i + 32 (v_permlane32_swap): a west corner is
    //       added to a west corner of the next row with the same texel.
    // Merging only ever joins equal texel ids, so whatever is left of a chain still carries the id of everything summed into it.
    // Every lane runs the cross-lane moves (they would read garbage from lanes masked off by a branch).
    const int lane = threadIdx.x & 63;
    const bool upper = lane >= 32;
    float Wv[4][2][ACC_C];  // west corners [row][0 north / 1 south][r g b z weight]: they receive sums, so all five values are kept
    float Ew[4][2], colk[4][4];  // east corners stay products of the pixel's (r g b z) and a weight until somebody needs the values
    int tw[4][2], te[4][2];
    int mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -1, mxy = -1;
    const float inv_lmax = 1.0f / (lmax + 1e-7f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        tw[k][0] = tw[k][1] = te[k][0] = te[k][1] = -1;
        if (inb[k] && mk[k] != 0.f) {
            const int py = ty0 + 4 * (threadIdx.x >> 5) + k, px = tx0 + (threadIdx.x & 31);
            const float z = zin[k];
            const SplatGeom gk = splat_geom(flx[k], fly[k], px, py, h, w);
            // depth weight exp(log1p(z) / max * 50): hardware log2 / exp2 and reciprocals instead of libm and IEEE divisions (this kernel is
            // VALU-bound). ~1e-5 relative on a weight that appears in numerator and denominator of the resolve; indices and masks do not depend on it
            const float logd = __logf(1.0f + fmaxf(z, 0.f));
            const float expo = logd * inv_lmax * 50.0f;
            const float dw = __expf(fminf(expo, 80.0f)) + 1e-7f;
            const float m = mk[k];
            colk[k][0] = rgb[k][0]; colk[k][1] = rgb[k][1]; colk[k][2] = rgb[k][2]; colk[k][3] = z;
            mnx = min(mnx, gk.fx);
            mny = min(mny, gk.fy);
            mxx = max(mxx, gk.cx);
            mxy = max(mxy, gk.cy);
            const float m_dw = m * __builtin_amdgcn_rcpf(dw);
            const float wts[4] = {gk.nw * m_dw, gk.sw * m_dw, gk.ne * m_dw, gk.se * m_dw};
#pragma unroll
            for (int e = 0; e < 4; ++e) { Wv[k][0][e] = colk[k][e] * wts[0]; Wv[k][1][e] = colk[k][e] * wts[1]; }
            Wv[k][0][4] = wts[0]; Wv[k][1][4] = wts[1]; Ew[k][0] = wts[2]; Ew[k][1] = wts[3];
            tw[k][0] = (gk.fy << 16) | gk.fx; tw[k][1] = (gk.cy << 16) | gk.fx;
            te[k][0] = (gk.fy << 16) | gk.cx; te[k][1] = (gk.cy << 16) | gk.cx;
            // (1) coinciding corners of this pixel
            if (tw[k][1] == tw[k][0]) {
#pragma unroll
                for (int e = 0; e < ACC_C; ++e) Wv[k][0][e] += Wv[k][1][e];
                tw[k][1] = -1;
            }
            if (te[k][1] == te[k][0]) {
                Ew[k][0] += Ew[k][1];
                te[k][1] = -1;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (te[k][c] < 0) continue;
                const int cc = te[k][c] == tw[k][0] ? 0 : (te[k][c] == tw[k][1] ? 1 : -1);
                if (cc < 0) continue;
#pragma unroll
                for (int e = 0; e < ACC_C; ++e) {
                    const float v = e < 4 ? colk[k][e < 4 ? e : 0] * Ew[k][c] : Ew[k][c];
                    if (cc == 0) Wv[k][0][e] += v; else Wv[k][1][e] += v;
                }
                te[k][c] = -1;
            }
        } else {
#pragma unroll
            for (int e = 0; e < ACC_C; ++e) Wv[k][0][e] = Wv[k][1][e] = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) colk[k][e] = 0.f;
            Ew[k][0] = Ew[k][1] = 0.f;
        }
    }
    // (2) east corners of the left neighbour -> my west corners
    {
        unsigned mytake = 0;  // bit 2 k + c: I took east corner (k, c) of lane - 1
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pcol[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) pcol[e] = lane_prev(colk[k][e]);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int p_t = lane_prev(te[k][c]);
                const float p_w = lane_prev(Ew[k][c]);
                float pv[ACC_C];
#pragma unroll
                for (int e = 0; e < 4; ++e) pv[e] = pcol[e] * p_w;
                pv[4] = p_w;
                const bool can = (lane & 31) != 0 && p_t >= 0;  // lanes 0 / 32 have no left neighbour in their rows
                bool done = false;
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) {
                    const int j = dj == 0 ? k : (dj == 1 ? k - 1 : k + 1);
                    if (j < 0 || j > 3) continue;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const bool hit = can && !done && tw[j][cc] == p_t;
                        if (hit) {
#pragma unroll
                            for (int e = 0; e < ACC_C; ++e) Wv[j][cc][e] += pv[e];
                        }
                        done = done || hit;
                    }
                }
                if (done) mytake |= 1u << (2 * k + c);
            }
        }
        const unsigned n_take = (unsigned)lane_next((int)mytake);
        const unsigned given = (lane & 31) != 31 ? n_take : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if ((given >> (2 * k + c)) & 1u) te[k][c] = -1;
    }
    // (3) down the column inside the thread
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bool done = false;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const bool hit = !done && tw[k][c] >= 0 && tw[k + 1][cc] == tw[k][c];
                if (hit) {
#pragma unroll
                    for (int e = 0; e < ACC_C; ++e) Wv[k + 1][cc][e] += Wv[k][c][e];
                }
                done = done || hit;
            }
            if (done) tw[k][c] = -1;
        }
    }
    // ... and across the wave's halves: the west corners of the last row of lane i -> the first two rows of lane i + 32
    {
        unsigned mytake = 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int l_t = from_lane_minus32(tw[3][c]);
            float lv[ACC_C];
#pragma unroll
            for (int e = 0; e < ACC_C; ++e) lv[e] = from_lane_minus32(Wv[3][c][e]);
            bool done = false;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const bool hit = upper && !done && l_t >= 0 && tw[j][cc] == l_t;
                    if (hit) {
#pragma unroll
                        for (int e = 0; e < ACC_C; ++e) Wv[j][cc][e] += lv[e];
                    }
                    done = done || hit;
                }
            }
            if (done) mytake |= 1u << c;
        }
        const unsigned u_take = (unsigned)from_lane_plus32((int)mytake);  // evaluated by EVERY lane: inside `!upper && ...` the swap would run with half the wave masked off
        if (!upper) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if ((u_take >> c) & 1u) tw[3][c] = -1;
        }
    }
    // window origin = minimum north-west corner of the tile (per-wave minima, no initialisation pass and no LDS atomics); the barrier also
    // closes the zero fill
    int ox, oy, ex, ey;
    if constexpr (EXCL) {  // the extent pre-pass published the same four numbers (same geometry code, same floats)
        const int4 og = *reinterpret_cast<const int4*>(origins + ORG_N * slot);
        ox = og.x; oy = og.y; ex = min(WIN, og.z); ey = min(WIN, og.w);  // (published unclamped)
        __syncthreads();
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, o, 64));
            mny = min(mny, __shfl_xor(mny, o, 64));
            mxx = max(mxx, __shfl_xor(mxx, o, 64));
            mxy = max(mxy, __shfl_xor(mxy, o, 64));
        }
        if (lane == 0) { int* ow = org_w[threadIdx.x >> 6]; ow[0] = mnx; ow[1] = mny; ow[2] = mxx; ow[3] = mxy; }
        __syncthreads();
        ox = min(min(org_w[0][0], org_w[1][0]), min(org_w[2][0], org_w[3][0]));
        oy = min(min(org_w[0][1], org_w[1][1]), min(org_w[2][1], org_w[3][1]));
        // used part of the window: only its rows are written out below and only its texels are read by the gather (a smooth flow fills ~34 x 34
        // of the 40 x 40 texels)
        const int fx = max(max(org_w[0][2], org_w[1][2]), max(org_w[2][2], org_w[3][2])) - ox + 1;
        const int fy = max(max(org_w[0][3], org_w[1][3]), max(org_w[2][3], org_w[3][3])) - oy + 1;
        ex = min(WIN, fx);
        ey = min(WIN, fy);
        // full_extent (g3_render_items_f32): the UNCLAMPED rectangle is published - what lies beyond the window went to the dense accumulator, and the
        // gather pass reads the accumulator for exactly those texels instead of for every pixel of an item that has such a tile somewhere
        if (threadIdx.x == 0) {
            int* og = origins + ORG_N * slot;
            og[0] = ox; og[1] = oy; og[2] = full_extent && ox != 0x7fffffff ? fx : ex; og[3] = full_extent && ox != 0x7fffffff ? fy : ey;
        }
    }
    if (ox == 0x7fffffff) return;  // nothing valid in this tile: the gather skips it by its origin
    // After the merge almost every destination texel receives exactly one value. LDS float atomics cost ~6 LDS cycles per LANE (PMC), plain
    // stores 2 cycles per wave instruction, so the texels are first given an owner: every west corner still alive writes its id to
    // owner[its texel] (last writer wins), and after a barrier the corner that reads its own id back STORES its five values into the
    // zero-initialised window; everything else - corners that lost a texel, east corners nobody took, out-of-window corners - follows
    // after a second barrier as atomics.
    int wl[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            wl[k][c] = -1;
            if (tw[k][c] < 0) continue;
            const int lx = (tw[k][c] & 0xffff) - ox, ly = (tw[k][c] >> 16) - oy;
            if ((unsigned)lx < (unsigned)WIN && (unsigned)ly < (unsigned)WIN) {
                wl[k][c] = ly * WIN + lx;
                owner[wl[k][c]] = (2 * k + c) * 256 + (int)threadIdx.x;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (wl[k][c] >= 0 && owner[wl[k][c]] == (2 * k + c) * 256 + (int)threadIdx.x) {
                float* a = win + wl[k][c] * ACC_C;
#pragma unroll
                for (int e = 0; e < ACC_C; ++e) a[e] = Wv[k][c][e];
                tw[k][c] = -1;
            }
        }
    }
    __syncthreads();
    auto add_texel = [&](int tex, const float (&v)[ACC_C]) {
        const int x = tex & 0xffff, y = tex >> 16;
        const int lx = x - ox, ly = y - oy;
        if ((unsigned)lx < (unsigned)WIN && (unsigned)ly < (unsigned)WIN) {
            float* a = win + (ly * WIN + lx) * ACC_C;
#pragma unroll
            for (int e = 0; e < ACC_C; ++e) atomicAdd(a + e, v[e]);
        } else {
            float* a = acc_item + ((int64_t)y * aw + x) * ACC_C;
#pragma unroll
            for (int e = 0; e < ACC_C; ++e) unsafeAtomicAdd(a + e, v[e]);
            if (dirty) dirty[item] = epoch;  // this launch put something into the item's dense accumulator: the gather must read (and clear) it
        }
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (tw[k][c] >= 0) add_texel(tw[k][c], Wv[k][c]);
            if (te[k][c] >= 0) {
                const float ev[ACC_C] = {colk[k][0] * Ew[k][c], colk[k][1] * Ew[k][c], colk[k][2] * Ew[k][c], colk[k][3] * Ew[k][c], Ew[k][c]};
                add_texel(te[k][c], ev);
            }
        }
    }
    __syncthreads();
    if constexpr (EXCL) {
        {
            // which of my window's texels lie in another tile's (full) rectangle: scan the item's rectangles, keep those that meet mine in LDS (the owner map
            // is no longer needed) and test this thread's texels (i = thread + 256 j) against them
            constexpr int NTX = (WIN * WIN + 255) / 256;
            bool shared_tx[NTX];
#pragma unroll
            for (int j = 0; j < NTX; ++j) shared_tx[j] = false;
            unsigned* lst = reinterpret_cast<unsigned*>(owner);  // one packed word per rectangle (WIN x WIN = 1 600 of them fit; a scan selects <= 1 023)
            const int ntiles = (int)gridDim.x;
            const int* org_item = origins + (int64_t)item * ntiles * ORG_N;
            for (int base = 0; base < ntiles; base += 1024) {
                const int total = scan_rects<4>(
                    org_item, ntiles, base, &org_w[0][0], 0, WIN * WIN,
                    [&](int t, const int4& og) { return t != (int)blockIdx.x && og.x < ox + ex && og.x + og.z > ox && og.y < oy + ey && og.y + og.w > oy; },
                    [&](int pos, int, const int4& og) {
                        // the part of the other rectangle inside my window, in window coordinates (not empty: that is the hit test):
                        // x0 | y0 << 6 | (columns - 1) << 12 | (rows - 1) << 18
                        const int x0 = max(og.x - ox, 0), y0 = max(og.y - oy, 0);
                        const int x1 = min(og.x - ox + og.z, WIN), y1 = min(og.y - oy + og.w, WIN);
                        lst[pos] = (unsigned)x0 | ((unsigned)y0 << 6) | ((unsigned)(x1 - x0 - 1) << 12) | ((unsigned)(y1 - y0 - 1) << 18);
                    });
                for (int i = 0; i < total; ++i) {
                    // a texel (lx | ly << 16) is inside iff the packed 16-bit differences to (x0 | y0 << 16) stay below the sizes: three VALU operations per
                    // test (this kernel is VALU-bound)
                    const unsigned e = lst[i];
                    const u16x2 r0 = __builtin_bit_cast(u16x2, (e & 63u) | (((e >> 6) & 63u) << 16));
                    const u16x2 rs = __builtin_bit_cast(u16x2, ((e >> 12) & 63u) | (((e >> 18) & 63u) << 16));
#pragma unroll
                    for (int j = 0; j < NTX; ++j) {
                        const int tix = (int)threadIdx.x + 256 * j;
                        const int ly = tix / WIN, lx = tix - ly * WIN;
                        const u16x2 d = __builtin_bit_cast(u16x2, (unsigned)lx | ((unsigned)ly << 16)) - r0;
                        const u16x2 m = __builtin_elementwise_min(d, rs);
                        shared_tx[j] = shared_tx[j] || (__builtin_bit_cast(unsigned, m) == __builtin_bit_cast(unsigned, d));
                    }
                }
                if (base + 1024 < ntiles) __syncthreads();  // the next scan overwrites the list
            }
            float* wdst = windows + slot * (WIN * WIN * ACC_C);
#pragma unroll
            for (int j = 0; j < NTX; ++j) {
                const int tix = (int)threadIdx.x + 256 * j;
                const int ly = tix / WIN, lx = tix - ly * WIN;
                if (ly >= ey || lx >= ex) continue;
                const float* a = win + tix * ACC_C;
                if (shared_tx[j]) {
#pragma unroll
                    for (int e = 0; e < ACC_C; ++e) wdst[tix * ACC_C + e] = a[e];
                    continue;
                }
                const int gy = oy + ly, gx = ox + lx;  // accumulator texel (gy, gx) = output pixel (gy - 1, gx - 1); the border ring is cropped
                if (gy < 1 || gy > h || gx < 1 || gx > w) continue;
                float sum[ACC_C];
#pragma unroll
                for (int e = 0; e < ACC_C; ++e) {
                    sum[e] = 0.f;
                    sum[e] += a[e];  // the gather pass's 0 + value (a -0 becomes +0 there too)
                }
                const ResolvedTexel r = resolve_sums(sum, occlusion != 0, 1.0f);
                const int pix = (gy - 1) * w + (gx - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) frame[((int64_t)item * 3 + c) * hw + pix] = r.c[c];
                mask_out[(int64_t)item * hw + pix] = r.m;
                if (depth_out) depth_out[(int64_t)item * hw + pix] = r.d;
            }
        }
    } else {
        // texel-major (r g b z weight of a texel adjacent): a planar layout was measured worse on both sides - the gather pass's row segments and the
        // single-writer form's sparse shared texels touch five lines instead of one (+42 % fetched bytes, splat +10 %)
        f32x4* dst = reinterpret_cast<f32x4*>(windows + slot * (WIN * WIN * ACC_C));
        const f32x4* src = reinterpret_cast<const f32x4*>(win);
        for (int i = threadIdx.x; i < ey * (WIN * ACC_C / 4); i += 256) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(256) void warp_gather_resolve_kernel(const float* __restrict__ windows, const int* __restrict__ origins,
                                                                  float* __restrict__ accum, float* __restrict__ frame,
                                                                  float* __restrict__ mask, float* __restrict__ depth, int n, int h, int w,
                                                                  int ntiles, int tiles_x, const unsigned* __restrict__ dirty = nullptr,
                                                                  unsigned epoch = 0, unsigned* __restrict__ tmin = nullptr,
                                                                  const float* __restrict__ Kinv = nullptr, const unsigned* __restrict__ occ_stamps = nullptr,
                                                                  int exclusive = 0) {
    __shared__ int lst[256 * 5];  // overlapping source tiles of one scan chunk: tile, ox, oy, ex, ey
    __shared__ int wave_cnt[16];
    __shared__ int any_big;  // some rectangle of the list is wider than a window (only then can a texel lie beyond a window)
    if (threadIdx.x == 0) any_big = 0;  // (ordered before the emits by the scan's first barrier)
    const int item = blockIdx.y;
    const int hw = h * w;
    const int aw = w + 2;
    const int dy0 = (blockIdx.x / tiles_x) * TS, dx0 = (blockIdx.x % tiles_x) * TS;  // output pixel (py, px) = accumulator texel (py + 1, px + 1)
    float* acc_item = accum + (int64_t)item * (h + 2) * aw * ACC_C;
    // dirty == nullptr: the caller zeroed the accumulator before the splat and it is read unconditionally (g3_warp_splat_resolve_f32).
    // Otherwise (g3_render_items_f32) the accumulator is all zero except for items the splat kernel stamped with this launch's epoch: only those
    // read it (20 bytes per pixel saved on the common path) and put the zeros back, so the buffer never needs a clearing pass.
    // exclusive: 0 = rectangles clamped to the window, accumulator read per ITEM (g3_warp_splat_resolve_f32); 2 = the rectangles are the tiles' full
    // extents and the accumulator is read per TEXEL (below); 1 = 2 + single-writer texels (resolved by the splat, skipped here).
    const bool read_acc = exclusive == 0 && (dirty == nullptr || dirty[item] == epoch);
    const int* org_item = origins + (int64_t)item * ntiles * ORG_N;
    const float* win_item = windows + (int64_t)item * ntiles * (WIN * WIN * ACC_C);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    float sum[4][ACC_C];
    int gy[4], gx[4];
    bool inb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // (a wave owning a compact 16 x 16 quadrant or an 8-row band, so that it can skip the rectangles that miss it - on lane tests or on scalar
        // compares of the list entries - was measured equal or slower, before and after the loads below were batched)
        const int py = dy0 + (threadIdx.x >> 5) + 8 * k, px = dx0 + (threadIdx.x & 31);
        inb[k] = py < h && px < w;
        gy[k] = py + 1;
        gx[k] = px + 1;
        float* a = acc_item + ((int64_t)gy[k] * aw + gx[k]) * ACC_C;
        bool any = false;
#pragma unroll
        for (int e = 0; e < ACC_C; ++e) {
            sum[k][e] = (inb[k] && read_acc) ? a[e] : 0.f;
            any = any || sum[k][e] != 0.f;
        }
        if (dirty != nullptr && any) {
#pragma unroll
            for (int e = 0; e < ACC_C; ++e) a[e] = 0.f;
        }
    }
    // the source tiles of one scan chunk (256 rectangles) whose rectangle meets this tile, compacted in ascending tile order into lst from entry
    // `at` on (when they fit: the return value is the number of hits either way). Plain form: the rectangle is the used part of the window.
    // Single-writer form: the tile's full extent, of which the first WIN columns / rows are its window
    auto scan_chunk = [&](int base, int at) -> int {
        const int t = base + threadIdx.x;
        int ox = 0x7fffffff, oy = 0x7fffffff, ex = 0, ey = 0;
        if (t < ntiles) {
            const int4 og = *reinterpret_cast<const int4*>(org_item + ORG_N * t);
            ox = og.x; oy = og.y; ex = og.z; ey = og.w;
        }
        // texels [oy, oy + ey) x [ox, ox + ex) against this tile's texels [dy0 + 1, dy0 + TS] x [dx0 + 1, dx0 + TS]
        const bool hit = ox != 0x7fffffff && ox <= dx0 + TS && ox + ex > dx0 + 1 && oy <= dy0 + TS && oy + ey > dy0 + 1;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < wv) off += wave_cnt[i];
            total += wave_cnt[i];
        }
        if (hit && at + total <= 256) {
            const int pos = at + off + __popcll(bal & ((1ull << lane) - 1ull));  // ascending tile order: the summation order below is fixed
            int* l = lst + 5 * pos;
            l[0] = t; l[1] = ox; l[2] = oy; l[3] = ex; l[4] = ey;
        }
        __syncthreads();
        return total;
    };
    // Single-writer form: a texel inside exactly ONE tile's rectangle - and inside that tile's window - was resolved and written by that tile's splat
    // workgroup (it is not in the window workspace); the same count decides it here. A texel in the part of a rectangle BEYOND the window received that
    // tile's contributions through the dense accumulator: it is read (and the zero put back) for exactly those texels.
    bool need[4];
    int cnt[4] = {0, 0, 0, 0};
    bool beyond[4] = {false, false, false, false};
#pragma unroll
    for (int k = 0; k < 4; ++k) need[k] = inb[k];
    auto count_entries = [&](int total) {
        for (int i = 0; i < total; ++i) {
            const int tox = lst[5 * i + 1], toy = lst[5 * i + 2], tfx = lst[5 * i + 3], tfy = lst[5 * i + 4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lx = gx[k] - tox, ly = gy[k] - toy;
                const bool in_rect = (unsigned)lx < (unsigned)tfx && (unsigned)ly < (unsigned)tfy;
                cnt[k] += in_rect ? 1 : 0;
                beyond[k] = beyond[k] || (in_rect && (lx >= WIN || ly >= WIN));
            }
        }
    };
    auto decide = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            need[k] = inb[k] && (exclusive != 1 || cnt[k] != 1 || beyond[k]);
            if (inb[k] && beyond[k]) {
                float* a = acc_item + ((int64_t)gy[k] * aw + gx[k]) * ACC_C;
                bool any = false;
#pragma unroll
                for (int e = 0; e < ACC_C; ++e) {
                    sum[k][e] = a[e];
                    any = any || sum[k][e] != 0.f;
                }
                if (any) {
#pragma unroll
                    for (int e = 0; e < ACC_C; ++e) a[e] = 0.f;
                }
            }
        }
    };
    // The sums, one memory round trip per rectangle: the four texels' loads go out together and unconditionally (a texel outside the rectangle reads the
    // window's first texel - valid workspace memory - and its value is dropped by a select), then one wait, then the adds in the fixed order. With the
    // loads behind `if (inside)` every (rectangle, texel) pair was a dependent round trip of its own and the waves spent 78 % of their cycles waiting
    // (profiles/r3_pmc_render.txt). A rectangle none of the wave's texels lies in is skipped.
    constexpr int SUM_U = 2;  // rectangles per trip (4: 125 registers, more dropped loads - measured 20 % slower; 1: 8 % slower)
    auto sum_entries = [&](int total) {
        for (int i0 = 0; i0 < total; i0 += SUM_U) {
            bool hit[SUM_U][4], any = false;
            const float* a[SUM_U][4];
#pragma unroll
            for (int u = 0; u < SUM_U; ++u) {
                const int i = min(i0 + u, total - 1);
                const bool live = i0 + u < total;
                const int tt = lst[5 * i], tox = lst[5 * i + 1], toy = lst[5 * i + 2];
                const int tex = exclusive != 0 ? min(WIN, lst[5 * i + 3]) : lst[5 * i + 3], tey = exclusive != 0 ? min(WIN, lst[5 * i + 4]) : lst[5 * i + 4];
                const float* wb = win_item + (int64_t)tt * (WIN * WIN * ACC_C);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int lx = gx[k] - tox, ly = gy[k] - toy;
                    hit[u][k] = live && need[k] && (unsigned)lx < (unsigned)tex && (unsigned)ly < (unsigned)tey;
                    a[u][k] = wb + (hit[u][k] ? (ly * WIN + lx) * ACC_C : 0);
                    any = any || hit[u][k];
                }
            }
            if (__ballot(any) == 0ull) continue;
            float v[SUM_U][4][ACC_C];
#pragma unroll
            for (int u = 0; u < SUM_U; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4u q = *reinterpret_cast<const f32x4u*>(a[u][k]);  // (20-byte texels: 4-byte aligned 16-byte loads)
                    v[u][k][0] = q.x; v[u][k][1] = q.y; v[u][k][2] = q.z; v[u][k][3] = q.w;
                    v[u][k][4] = a[u][k][4];
                }
            }
#pragma unroll
            for (int u = 0; u < SUM_U; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int e = 0; e < ACC_C; ++e) sum[k][e] = hit[u][k] ? sum[k][e] + v[u][k][e] : sum[k][e];
                }
            }
        }
    };
    bool done = false;
    if (exclusive != 0) {
        // one scan when every rectangle that meets this tile fits the list (a few dozen even along a depth edge): count, decide, sum from LDS
        int nlist = 0;
        bool fits = true;
        for (int base = 0; base < ntiles && fits; base += 1024) {
            const int total = scan_rects<4>(
                org_item, ntiles, base, wave_cnt, nlist, 256,
                [&](int, const int4& og) { return og.x <= dx0 + TS && og.x + og.z > dx0 + 1 && og.y <= dy0 + TS && og.y + og.w > dy0 + 1; },
                [&](int pos, int t, const int4& og) {
                    int* l = lst + 5 * pos;
                    l[0] = t; l[1] = og.x; l[2] = og.y; l[3] = og.z; l[4] = og.w;
                    if (og.z > WIN || og.w > WIN) any_big = 1;
                });
            fits = nlist + total <= 256;
            if (fits) nlist += total;
        }
        if (fits) {
            if (exclusive == 1 || any_big) count_entries(nlist);  // (form 2 needs the count only to find texels beyond a window)
            decide();
            sum_entries(nlist);
            done = true;
        } else {  // otherwise chunk by chunk: once to count, once (below) to sum
            for (int base = 0; base < ntiles; base += 256) {
                const int total = scan_chunk(base, 0);
                count_entries(total);
                __syncthreads();
            }
            decide();
        }
    }
    if (!done) {
        for (int base = 0; base < ntiles; base += 256) {
            const int total = scan_chunk(base, 0);
            sum_entries(total);
            __syncthreads();
        }
    }
    const bool occluded_tile = tmin && occ_stamps[(int64_t)item * ntiles + blockIdx.x] == epoch;  // the rasteriser hit this tile
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!inb[k]) continue;
        const int pix = (gy[k] - 1) * w + (gx[k] - 1);
        // mesh occlusion applied here when the rasteriser ran before this pass (g3_render_items_f32): the arithmetic of mesh_apply_kernel on the
        // values this thread is about to write, instead of a separate read-modify-write pass over frame / mask / depth
        float mesh_z = 0.f;
        if (occluded_tile) {  // read the pixel's tmin and leave +inf behind
            const unsigned bits = tmin[(int64_t)item * hw + pix];
            if (bits != 0x7f800000u) tmin[(int64_t)item * hw + pix] = 0x7f800000u;
            const float t = (bits == 0x7f800000u) ? 0.f : __uint_as_float(bits);
            const V3 d = pixel_ray(Kinv + item * 9, gx[k] - 1, gy[k] - 1);
            mesh_z = t * d.z;
        }
        if (!need[k]) {
            // resolved by the splat with keep = 1. An occluding triangle in front (keep = 0) turns the pixel into (v + 1) * 0 - 1 = -1, mask 0 and
            // depth * 0: the depth the splat stored is the dval of the test (foreground masking always has a depth output)
            if (occluded_tile) {
                const float dval = depth[(int64_t)item * hw + pix];
                if (((mesh_z + 0.02f) < dval) && (mesh_z > 0.f)) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) frame[((int64_t)item * 3 + c) * hw + pix] = -1.0f;
                    mask[(int64_t)item * hw + pix] = 0.0f;
                    depth[(int64_t)item * hw + pix] = dval * 0.0f;
                }
            }
            continue;
        }
        float keep = 1.0f;
        if (occluded_tile) {
            float wt = sum[k][4];
            if (wt != wt) wt = 1000.0f;
            const float dval = wt > 0.f ? sum[k][3] / wt : 0.0f;  // (resolve_sums computes the same dval)
            keep = (((mesh_z + 0.02f) < dval) && (mesh_z > 0.f)) ? 0.f : 1.f;
        }
        const ResolvedTexel r = resolve_sums(sum[k], tmin != nullptr, keep);
#pragma unroll
        for (int c = 0; c < 3; ++c) frame[((int64_t)item * 3 + c) * hw + pix] = r.c[c];
        mask[(int64_t)item * hw + pix] = r.m;
        if (depth) depth[(int64_t)item * hw + pix] = r.d;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// (3) resolve
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_resolve_kernel(const float* __restrict__ accum, float* __restrict__ frame,
                                                           float* __restrict__ mask, float* __restrict__ depth, int n,
                                                           int h, int w) {
    const int item = blockIdx.y;
    const int hw = h * w;
    const int aw = w + 2;
    const float* acc_item = accum + (int64_t)item * (h + 2) * aw * ACC_C;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < hw; pix += gridDim.x * 256) {
        const int py = pix / w, px = pix - py * w;
        const float* a = acc_item + ((int64_t)(py + 1) * aw + (px + 1)) * ACC_C;
        float wt = a[4];
        if (wt != wt) wt = 1000.0f;  // nan_to_num(nan=1000)
        const bool ok = wt > 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = ok ? a[c] / wt : -1.0f;
            v = fminf(fmaxf(v, -1.0f), 1.0f);
            frame[((int64_t)item * 3 + c) * hw + pix] = v;
        }
        mask[(int64_t)item * hw + pix] = ok ? 1.0f : 0.0f;
        if (depth) depth[(int64_t)item * hw + pix] = ok ? a[3] / wt : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// mesh occlusion
// ---------------------------------------------------------------------------------------------------------------
G3_DEVICE void bilinear_src(int i, float scale, int n_in, int& i0, int& i1, float& l0, float& l1) {
    float c = ((float)i + 0.5f) * scale - 0.5f;
    c = fmaxf(c, 0.f);
    i0 = (int)floorf(c);
    i1 = min(i0 + 1, n_in - 1);
    l1 = c - (float)i0;
    l0 = 1.0f - l1;
}

// cam == nullptr (g3_render_items_f32): the camera-space points are recomputed from the cached world points with warp_project_kernel's
// expression (same operation order, so the same bits) - only ~1/4 of the pixels are sampled, and the projection then has no 12-byte-per-pixel
// cam buffer to write.
__global__ __launch_bounds__(256) void mesh_downsample_kernel(const float* __restrict__ cam, const uint8_t* __restrict__ bmask,
                                                              float* __restrict__ pts, uint8_t* __restrict__ m, int n, int h,
                                                              int w, int nh, int nw, const int* __restrict__ src = nullptr,
                                                              const float* __restrict__ points = nullptr, const float* __restrict__ w2c = nullptr) {
    const int item = blockIdx.y;
    const int sitem = src ? src[item] : item;
    const float* W = w2c ? w2c + item * 16 : nullptr;
    const float* psrc = points ? points + (int64_t)sitem * h * w * 3 : nullptr;
    auto cam_at = [&](int y, int x, int k) -> float {
        if (cam) return cam[((int64_t)item * h * w + (int64_t)y * w + x) * 3 + k];
        const float* p = psrc + ((int64_t)y * w + x) * 3;
        return ((W[k * 4 + 0] * p[0] + W[k * 4 + 1] * p[1]) + W[k * 4 + 2] * p[2]) + W[k * 4 + 3] * 1.0f;
    };
    const float sy = (float)h / (float)nh, sx = (float)w / (float)nw;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < nh * nw; idx += gridDim.x * 256) {
        const int i = idx / nw, j = idx - i * nw;
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        bilinear_src(i, sy, h, y0, y1, wy0, wy1);
        bilinear_src(j, sx, w, x0, x1, wx0, wx1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float top = cam_at(y0, x0, k) * wx0 + cam_at(y0, x1, k) * wx1;
            const float bot = cam_at(y1, x0, k) * wx0 + cam_at(y1, x1, k) * wx1;
            pts[((int64_t)item * nh * nw + idx) * 3 + k] = top * wy0 + bot * wy1;
        }
        const int my = (int)floorf((float)i * sy), mx = (int)floorf((float)j * sx);
        m[(int64_t)item * nh * nw + idx] = bmask[(int64_t)sitem * h * w + (int64_t)my * w + mx] ? 1 : 0;
    }
}

// Moller-Trumbore for the ray through pixel (px, py) and one triangle given as (e1, e2, s = -v0, q = s x e1, e2q = e2 . q): the branch order
// and epsilons of ray_triangle_intersection_warp.py:23-105; min-t is an atomicMin on the float bit pattern (t > 0: uint order == float order).
// per-item counters of the mesh pass live CNT_STRIDE ints apart: counters of different items in ONE cache line serialise in the L2 (32 items x
// ~300 wave atomics on one line: 95 us for a pass that otherwise takes 5)
constexpr int CNT_STRIDE = 32;
// g3_render_items_f32 keeps tmin at +inf between renders: the rasteriser stamps the 32 x 32 output tiles it hits with the render's epoch, and the
// resolve pass reads tmin only in stamped tiles and puts the +inf back - no per-render fill of tmin and no tmin read where no triangle landed
// (8 bytes per pixel of traffic for the ~95 % of the tiles without a hit).
struct OccTiles { unsigned* stamps; int tiles_x; unsigned epoch; };  // stamps of ONE item; nullptr: tmin is filled / read everywhere
G3_DEVICE unsigned* occ_stamp(const OccTiles& O, int px, int py) { return O.stamps ? O.stamps + (py / TS) * O.tiles_x + px / TS : nullptr; }
struct TriSetup { V3 e1, e2, s, q; float e2q; };
G3_DEVICE TriSetup tri_setup(V3 v0, V3 v1, V3 v2) {
    TriSetup t;
    t.e1 = sub(v1, v0); t.e2 = sub(v2, v0);
    t.s = {0.f - v0.x, 0.f - v0.y, 0.f - v0.z};  // ray origin (0) - v0
    t.q = cross(t.s, t.e1);
    t.e2q = dot(t.e2, t.q);
    return t;
}
G3_DEVICE void ray_tri_min(const TriSetup& T, V3 d, float eps, unsigned* __restrict__ out_px, unsigned* __restrict__ stamp = nullptr, unsigned epoch = 0) {
    const V3 hh = cross(d, T.e2);
    const float a = dot(T.e1, hh);
    if (fabsf(a) < eps) return;
    const float f = 1.0f / a;
    const float u = f * dot(T.s, hh);
    if (u < 0.f || u > 1.f) return;
    const float v = f * dot(d, T.q);
    if (v < 0.f || (u + v) > 1.f) return;
    const float t = f * T.e2q;
    // values only ever decrease, so a (possibly stale) read that is already <= t makes the atomic redundant; overlapping skirt triangles
    // cover most pixels several times
    if (t > eps && __float_as_uint(t) < *(volatile unsigned*)out_px) {
        atomicMin(out_px, __float_as_uint(t));
        if (stamp) *stamp = epoch;  // this 32 x 32 output tile has a hit in this render: the resolve pass reads (and resets) its tmin, see OccTiles
    }
}
// Where the vertices of the 4x-downsampled mesh come from: a precomputed [nh][nw][3] array, or computed from the cached world points of the
// item's source view and the item's world-to-camera matrix with mesh_downsample_kernel's expressions (same operation order, same bits).
// g3_render_items_f32 computes them in mesh_mark_kernel for the boundary patches only (~2 % of the mesh; downsampling every vertex of every
// item is 43 us per 32 items of mostly unused work) into the same [nh][nw][3] layout, which the rasteriser then reads.
struct MeshVerts {
    const float* P;
    const float* psrc;
    const float* W;
    int h, w, nh, nw;
};
G3_DEVICE V3 mesh_vertex(const MeshVerts& M, int i, int j) {
    if (M.P) {
        const float* p = M.P + ((int64_t)i * M.nw + j) * 3;
        return {p[0], p[1], p[2]};
    }
    const float sy = (float)M.h / (float)M.nh, sx = (float)M.w / (float)M.nw;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bilinear_src(i, sy, M.h, y0, y1, wy0, wy1);
    bilinear_src(j, sx, M.w, x0, x1, wx0, wx1);
    float v[3];
    const float* W = M.W;
    auto cam_at = [&](int y, int x, int k) -> float {
        const float* p = M.psrc + ((int64_t)y * M.w + x) * 3;
        return ((W[k * 4 + 0] * p[0] + W[k * 4 + 1] * p[1]) + W[k * 4 + 2] * p[2]) + W[k * 4 + 3] * 1.0f;
    };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float top = cam_at(y0, x0, k) * wx0 + cam_at(y0, x1, k) * wx1;
        const float bot = cam_at(y1, x0, k) * wx0 + cam_at(y1, x1, k) * wx1;
        v[k] = top * wy0 + bot * wy1;
    }
    return {v[0], v[1], v[2]};
}
G3_DEVICE void patch_triangle(const MeshVerts& M, int patch, int tri, V3& v0, V3& v1, V3& v2) {
    const int pi = patch / (M.nw - 1), pj = patch - pi * (M.nw - 1);
    // points_to_mesh: (tl, tr, bl) and (tr, br, bl)
    v0 = tri == 0 ? mesh_vertex(M, pi, pj) : mesh_vertex(M, pi, pj + 1);
    v1 = tri == 0 ? mesh_vertex(M, pi, pj + 1) : mesh_vertex(M, pi + 1, pj + 1);
    v2 = mesh_vertex(M, pi + 1, pj);
}

// The lanes of a wave sweep the conservative pixel bounding box of one triangle (tri = 0 / 1 of a mesh patch). A triangle that touches the
// camera plane has no finite projection: its box is the whole image. `heavy` == nullptr: swept here all the same (one wave, h * w / 64 rounds);
// otherwise its three vertices are appended to the item's heavy list and mesh_raster_heavy_kernel spreads its pixels over the chip (a full
// list: swept here after all).
G3_DEVICE void mesh_raster_tri(int item, int patch, int tri, int lane, const MeshVerts& M, const float* __restrict__ Kmat,
                               const float* __restrict__ Kinv, unsigned* __restrict__ tmin, float eps,
                               int* __restrict__ heavy_cnt, float* __restrict__ heavy, int heavy_cap, const OccTiles& O) {
    const int h = M.h, w = M.w;
    const float* K = Kmat + item * 9;
    const float* Ki = Kinv + item * 9;
    unsigned* out = tmin + (int64_t)item * h * w;
    V3 v0, v1, v2;
    patch_triangle(M, patch, tri, v0, v1, v2);
    int x_lo = 0, x_hi = w - 1, y_lo = 0, y_hi = h - 1;
    if (v0.z > 1e-4f && v1.z > 1e-4f && v2.z > 1e-4f) {
        float minx = 3e38f, maxx = -3e38f, miny = 3e38f, maxy = -3e38f;
        const V3 vs[3] = {v0, v1, v2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float X = (K[0] * vs[k].x + K[1] * vs[k].y + K[2] * vs[k].z) / vs[k].z;
            const float Y = (K[3] * vs[k].x + K[4] * vs[k].y + K[5] * vs[k].z) / vs[k].z;
            minx = fminf(minx, X); maxx = fmaxf(maxx, X); miny = fminf(miny, Y); maxy = fmaxf(maxy, Y);
        }
        if (!(maxx >= -2.f && minx <= (float)w + 1.f && maxy >= -2.f && miny <= (float)h + 1.f)) return;  // off-screen
        x_lo = max(0, (int)floorf(fmaxf(minx, -2.f)) - 1);
        y_lo = max(0, (int)floorf(fmaxf(miny, -2.f)) - 1);
        x_hi = min(w - 1, (int)ceilf(fminf(maxx, (float)w + 1.f)) + 1);
        y_hi = min(h - 1, (int)ceilf(fminf(maxy, (float)h + 1.f)) + 1);
        if (x_hi < x_lo || y_hi < y_lo) return;
    } else if (heavy) {
        int at = 0;
        if (lane == 0) at = atomicAdd(heavy_cnt + item * CNT_STRIDE, 1);
        at = __shfl(at, 0, 64);
        if (at < heavy_cap) {
            if (lane == 0) {
                float* e = heavy + ((int64_t)item * heavy_cap + at) * 9;
                e[0] = v0.x; e[1] = v0.y; e[2] = v0.z; e[3] = v1.x; e[4] = v1.y; e[5] = v1.z; e[6] = v2.x; e[7] = v2.y; e[8] = v2.z;
            }
            return;
        }
    }
    const TriSetup T = tri_setup(v0, v1, v2);
    const int bw = x_hi - x_lo + 1;
    const int npix = bw * (y_hi - y_lo + 1);
    for (int k = lane; k < npix; k += 64) {
        const int py = y_lo + k / bw, px = x_lo + k % bw;
        ray_tri_min(T, pixel_ray(Ki, px, py), eps, out + (int64_t)py * w + px, occ_stamp(O, px, py), O.epoch);
    }
}

// one wave per mesh patch (g3_mesh_occlusion_f32: no workspace for a heavy list)
__global__ __launch_bounds__(256) void mesh_raster_wave_kernel(const float* __restrict__ pts, const uint8_t* __restrict__ m,
                                                               const float* __restrict__ Kmat, const float* __restrict__ Kinv,
                                                               unsigned* __restrict__ tmin, int n, int h, int w, int nh, int nw,
                                                               float eps) {
    const int item = blockIdx.y;
    const int patch = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (patch >= (nh - 1) * (nw - 1)) return;
    const int pi = patch / (nw - 1), pj = patch - pi * (nw - 1);
    const uint8_t* mi = m + (int64_t)item * nh * nw;
    if (!(mi[pi * nw + pj] | mi[pi * nw + pj + 1] | mi[(pi + 1) * nw + pj] | mi[(pi + 1) * nw + pj + 1])) return;
    const MeshVerts M = {pts + (int64_t)item * nh * nw * 3, nullptr, nullptr, h, w, nh, nw};
    for (int tri = 0; tri < 2; ++tri) mesh_raster_tri(item, patch, tri, threadIdx.x & 63, M, Kmat, Kinv, tmin, eps, nullptr, nullptr, 0, OccTiles{nullptr, 0, 0u});
}

// g3_render_items_f32: a THREAD per mesh patch tests the boundary mask and the patches that pass are appended to the item's list (one atomic
// per wave); mesh_raster_list_kernel then gives every listed TRIANGLE to a wave, grid-stride. One wave per patch of the mesh is 1.8 M waves per
// 32 items of 704 x 1280 of which ~2 % have a boundary patch, and those sit next to each other (the disc outlines), i.e. in few workgroups.
__global__ __launch_bounds__(256) void mesh_mark_kernel(const uint8_t* __restrict__ bmask, const int* __restrict__ src, int* __restrict__ list_cnt,
                                                        int* __restrict__ list, int list_cap, int n, int h, int w, int nh, int nw,
                                                        const float* __restrict__ points, const float* __restrict__ w2c, float* __restrict__ pts_ds) {
    const int item = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int npatch = (nh - 1) * (nw - 1);
    const int mine = blockIdx.x * 256 + threadIdx.x;
    bool boundary = false;
    if (mine < npatch) {
        const int pi = mine / (nw - 1), pj = mine - pi * (nw - 1);
        const uint8_t* bm = bmask + (int64_t)(src ? src[item] : item) * h * w;
        const float sy = (float)h / (float)nh, sx = (float)w / (float)nw;
        // the vertex mask of mesh_downsample_kernel (nearest sample of the boundary mask), read at the patch's four vertices
        auto vm = [&](int i, int j) -> int { return bm[(int64_t)((int)floorf((float)i * sy)) * w + (int)floorf((float)j * sx)]; };
        boundary = (vm(pi, pj) | vm(pi, pj + 1) | vm(pi + 1, pj) | vm(pi + 1, pj + 1)) != 0;
    }
    const unsigned long long bal = __ballot(boundary);
    if (bal == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(list_cnt + item * CNT_STRIDE, __popcll(bal));
    base = __shfl(base, 0, 64);
    if (boundary) {
        const int at = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (at < list_cap) list[(int64_t)item * list_cap + at] = mine;
        // the patch's four vertices, computed where they are needed only (~2 % of the mesh; neighbouring patches write the same values twice)
        const MeshVerts M = {nullptr, points + (int64_t)(src ? src[item] : item) * h * w * 3, w2c + item * 16, h, w, nh, nw};
        const int pi = mine / (nw - 1), pj = mine - pi * (nw - 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = pi + (c >> 1), j = pj + (c & 1);
            const V3 v = mesh_vertex(M, i, j);
            float* o = pts_ds + ((int64_t)item * nh * nw + (int64_t)i * nw + j) * 3;
            o[0] = v.x; o[1] = v.y; o[2] = v.z;
        }
    }
}

__global__ __launch_bounds__(256) void mesh_raster_list_kernel(const float* __restrict__ pts_ds, const float* __restrict__ Kmat,
                                                               const float* __restrict__ Kinv, unsigned* __restrict__ tmin,
                                                               const int* __restrict__ list_cnt, const int* __restrict__ list, int list_cap, int n,
                                                               int h, int w, int nh, int nw, float eps, int* __restrict__ heavy_cnt,
                                                               float* __restrict__ heavy, int heavy_cap, unsigned* __restrict__ stamps, int ntiles,
                                                               int tiles_x, unsigned epoch) {
    const int item = blockIdx.y;
    const OccTiles O = {stamps + (int64_t)item * ntiles, tiles_x, epoch};
    const int ntri = 2 * min(list_cnt[item * CNT_STRIDE], list_cap);
    const MeshVerts M = {pts_ds + (int64_t)item * nh * nw * 3, nullptr, nullptr, h, w, nh, nw};
    for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < ntri; e += gridDim.x * 4)
        mesh_raster_tri(item, list[(int64_t)item * list_cap + (e >> 1)], e & 1, threadIdx.x & 63, M, Kmat, Kinv, tmin, eps, heavy_cnt, heavy, heavy_cap, O);
}

// Triangles on the heavy list against every pixel: a thread owns HEAVY_PX pixels (ray computed once per pixel), the triangles are the inner,
// wave-uniform loop. Nothing listed (the normal case): the workgroups leave at once.
constexpr int HEAVY_PX = 8;
__global__ __launch_bounds__(256) void mesh_raster_heavy_kernel(const float* __restrict__ Kinv, unsigned* __restrict__ tmin,
                                                                const int* __restrict__ heavy_cnt, const float* __restrict__ heavy, int heavy_cap,
                                                                int n, int h, int w, float eps, unsigned* __restrict__ stamps, int ntiles, int tiles_x,
                                                                unsigned epoch) {
    const int item = blockIdx.y;
    const int cnt = min(heavy_cnt[item * CNT_STRIDE], heavy_cap);
    if (cnt == 0) return;
    const OccTiles O = {stamps + (int64_t)item * ntiles, tiles_x, epoch};
    const float* Ki = Kinv + item * 9;
    unsigned* out = tmin + (int64_t)item * h * w;
    const int hw = h * w;
    for (int r = 0; r < HEAVY_PX; ++r) {
        const int pix = (blockIdx.x * HEAVY_PX + r) * 256 + threadIdx.x;
        if (pix >= hw) break;
        const int py = pix / w, px = pix - py * w;
        const V3 d = pixel_ray(Ki, px, py);
        for (int i = 0; i < cnt; ++i) {
            const float* e = heavy + ((int64_t)item * heavy_cap + i) * 9;
            ray_tri_min(tri_setup({e[0], e[1], e[2]}, {e[3], e[4], e[5]}, {e[6], e[7], e[8]}), d, eps, out + pix, occ_stamp(O, px, py), O.epoch);
        }
    }
}

__global__ __launch_bounds__(256) void mesh_apply_kernel(const unsigned* __restrict__ tmin, const float* __restrict__ Kinv,
                                                         float* __restrict__ frame, float* __restrict__ mask,
                                                         float* __restrict__ depth, int n, int h, int w) {
    const int item = blockIdx.y;
    const int hw = h * w;
    const float* Ki = Kinv + item * 9;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < hw; pix += gridDim.x * 256) {
        const unsigned bits = tmin[(int64_t)item * hw + pix];
        const float t = (bits == 0x7f800000u) ? 0.f : __uint_as_float(bits);
        const int py = pix / w, px = pix - py * w;
        const V3 d = pixel_ray(Ki, px, py);
        const float mesh_z = t * d.z;
        const int64_t o = (int64_t)item * hw + pix;
        const bool closer = ((mesh_z + 0.02f) < depth[o]) && (mesh_z > 0.f);
        const float keep = closer ? 0.f : 1.f;
        mask[o] = mask[o] * keep;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t oc = ((int64_t)item * 3 + c) * hw + pix;
            frame[oc] = (frame[oc] + 1.0f) * keep - 1.0f;
        }
        depth[o] = depth[o] * keep;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// cache construction helpers
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unproject_kernel(const float* __restrict__ depth, const float* __restrict__ c2w,
                                                        const float* __restrict__ Kinv, float* __restrict__ points, int n,
                                                        int h, int w) {
    const int item = blockIdx.y;
    const int hw = h * w;
    const float* Ki = Kinv + item * 9;
    const float* C = c2w + item * 16;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < hw; pix += gridDim.x * 256) {
        const int py = pix / w, px = pix - py * w;
        const float d = depth[(int64_t)item * hw + pix];
        const float xs = (float)px, ys = (float)py;
        float un[3], camp[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            un[r] = (Ki[r * 3 + 0] * xs + Ki[r * 3 + 1] * ys) + Ki[r * 3 + 2] * 1.0f;
            camp[r] = d * un[r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float val = ((C[r * 4 + 0] * camp[0] + C[r * 4 + 1] * camp[1]) + C[r * 4 + 2] * camp[2]) + C[r * 4 + 3] * 1.0f;
            points[((int64_t)item * hw + pix) * 3 + r] = (d > 0.f) ? val : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void reliable_mask_kernel(const float* __restrict__ depth, uint8_t* __restrict__ out, int n,
                                                            int h, int w, int window, float thr, float eps) {
    const int item = blockIdx.y;
    const int hw = h * w;
    const int r = window / 2;
    const float* d = depth + (int64_t)item * hw;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < hw; pix += gridDim.x * 256) {
        const int py = pix / w, px = pix - py * w;
        float mx = -INFINITY, mn = INFINITY, sm = 0.f;
        for (int dy = -r; dy <= r; ++dy)
            for (int dx = -r; dx <= r; ++dx) {  // same accumulation order as the oracle: row-major over the window
                const int y = py + dy, x = px + dx;
                const bool in = (y >= 0 && y < h && x >= 0 && x < w);
                const float v = in ? d[y * w + x] : 0.f;
                if (in) { mx = fmaxf(mx, v); mn = fminf(mn, v); }
                sm = sm + v;
            }
        const float mean = sm / (float)(window * window);
        const float ratio = (mx - mn) / (mean + eps);
        out[(int64_t)item * hw + pix] = ((ratio < thr) && (d[pix] > 0.f)) ? 1 : 0;
    }
}

int grid_x(int work) {
    int g = (work + 255) / 256;
    return g > 1024 ? 1024 : (g < 1 ? 1 : g);
}

}  // namespace

extern "C" int g3_warp_project_f32(const float* points, const float* w2c, const float* K, const float* mask1, float* z,
                                   float* flow, float* cam_points, float* maskz, void* group_max, int n, int h, int w,
                                   int group_size, void* stream) {
    if (!points || !w2c || !K || !z || !flow || !maskz || !group_max) return g3_set_error(G3_ERR_ARG, "g3_warp_project_f32: null operand");
    if (n <= 0 || h <= 0 || w <= 0 || group_size <= 0) return g3_set_error(G3_ERR_ARG, "g3_warp_project_f32: bad shape");
    hipLaunchKernelGGL(warp_project_kernel, dim3(min(grid_x(h * w), 256), n), dim3(256), 0, (hipStream_t)stream, points, w2c, K, mask1, z,
                       flow, cam_points, maskz, (unsigned*)group_max, n, h, w, group_size);
    return g3_check_launch("g3_warp_project_f32");
}

extern "C" int g3_warp_splat_f32(const float* image, const float* z, const float* flow, const float* maskz,
                                 const void* group_max, float* accum, int n, int h, int w, int group_size, void* stream) {
    if (!image || !z || !flow || !maskz || !group_max || !accum) return g3_set_error(G3_ERR_ARG, "g3_warp_splat_f32: null operand");
    if (n <= 0 || h <= 0 || w <= 0 || group_size <= 0) return g3_set_error(G3_ERR_ARG, "g3_warp_splat_f32: bad shape");
    if (g3_opt_splat_tiled) {
        const int tiles_x = (w + TS - 1) / TS, tiles_y = (h + TS - 1) / TS;
        hipLaunchKernelGGL(warp_splat_tiled_kernel, dim3(tiles_x * tiles_y, n), dim3(256), 0, (hipStream_t)stream, image, z, flow, maskz,
                           (const unsigned*)group_max, accum, n, h, w, group_size, tiles_x);
    } else {
        hipLaunchKernelGGL(warp_splat_kernel, dim3(grid_x(h * w), n), dim3(256), 0, (hipStream_t)stream, image, z, flow, maskz,
                           (const unsigned*)group_max, accum, n, h, w, group_size);
    }
    return g3_check_launch("g3_warp_splat_f32");
}

extern "C" int g3_warp_resolve_f32(const float* accum, float* frame, float* mask, float* depth, int n, int h, int w,
                                   void* stream) {
    if (!accum || !frame || !mask) return g3_set_error(G3_ERR_ARG, "g3_warp_resolve_f32: null operand");
    if (n <= 0 || h <= 0 || w <= 0) return g3_set_error(G3_ERR_ARG, "g3_warp_resolve_f32: bad shape");
    hipLaunchKernelGGL(warp_resolve_kernel, dim3(grid_x(h * w), n), dim3(256), 0, (hipStream_t)stream, accum, frame, mask, depth, n, h, w);
    return g3_check_launch("g3_warp_resolve_f32");
}

extern "C" size_t g3_warp_windows_workspace_bytes(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    const size_t ntiles = (size_t)((w + TS - 1) / TS) * ((h + TS - 1) / TS);
    return (size_t)n * ntiles * ((size_t)WIN * WIN * ACC_C * sizeof(float) + ORG_N * sizeof(int));
}

// splat + resolve without global atomics on the common path (warp_splat_windows_kernel / warp_gather_resolve_kernel above). `accum` as for
// g3_warp_splat_f32 (zeroed by the caller; receives only out-of-window corners); `workspace` >= g3_warp_windows_workspace_bytes, 16-byte aligned.
extern "C" int g3_warp_splat_resolve_f32(const float* image, const float* z, const float* flow, const float* maskz, const void* group_max,
                                         float* accum, void* workspace, float* frame, float* mask, float* depth, int n, int h, int w,
                                         int group_size, void* stream) {
    if (!image || !z || !flow || !maskz || !group_max || !accum || !workspace || !frame || !mask)
        return g3_set_error(G3_ERR_ARG, "g3_warp_splat_resolve_f32: null operand");
    if (n <= 0 || h <= 0 || w <= 0 || group_size <= 0) return g3_set_error(G3_ERR_ARG, "g3_warp_splat_resolve_f32: bad shape");
    if ((uintptr_t)workspace & 15) return g3_set_error(G3_ERR_ARG, "g3_warp_splat_resolve_f32: workspace must be 16-byte aligned");
    if (h + 2 > 32767 || w + 2 > 65535) return g3_set_error(G3_ERR_ARG, "g3_warp_splat_resolve_f32: image too large for the packed texel ids (h < 32766, w < 65534)");
    const int tiles_x = (w + TS - 1) / TS, tiles_y = (h + TS - 1) / TS, ntiles = tiles_x * tiles_y;
    float* windows = (float*)workspace;
    int* origins = (int*)((char*)workspace + (size_t)n * ntiles * WIN * WIN * ACC_C * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(warp_splat_windows_kernel<false>, dim3(ntiles, n), dim3(256), 0, s, image, z, flow, maskz, (const unsigned*)group_max, accum,
                       windows, origins, n, h, w, group_size, tiles_x, (const int*)nullptr, (unsigned*)nullptr, 0u, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr);
    hipLaunchKernelGGL(warp_gather_resolve_kernel, dim3(ntiles, n), dim3(256), 0, s, (const float*)windows, (const int*)origins, accum,
                       frame, mask, depth, n, h, w, ntiles, tiles_x);
    return g3_check_launch("g3_warp_splat_resolve_f32");
}

extern "C" int g3_mesh_occlusion_f32(const float* cam_points, const uint8_t* boundary_mask, const float* K, const float* Kinv,
                                     float* pts_ds, uint8_t* mask_ds, void* tmin, float* frame, float* mask, float* depth,
                                     int n, int h, int w, int factor, void* stream) {
    if (!cam_points || !boundary_mask || !K || !Kinv || !pts_ds || !mask_ds || !tmin || !frame || !mask || !depth)
        return g3_set_error(G3_ERR_ARG, "g3_mesh_occlusion_f32: null operand");
    if (n <= 0 || factor <= 0 || h / factor < 2 || w / factor < 2) return g3_set_error(G3_ERR_ARG, "g3_mesh_occlusion_f32: bad shape");
    const int nh = h / factor, nw = w / factor;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)tmin, 0x7f800000, (size_t)n * h * w, s);  // +inf bit pattern
    if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_mesh_occlusion_f32: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(mesh_downsample_kernel, dim3(grid_x(nh * nw), n), dim3(256), 0, s, cam_points, boundary_mask, pts_ds, mask_ds, n, h, w, nh, nw);
    const int npatch = (nh - 1) * (nw - 1);
    hipLaunchKernelGGL(mesh_raster_wave_kernel, dim3((npatch + 3) / 4, n), dim3(256), 0, s, pts_ds, mask_ds, K, Kinv, (unsigned*)tmin, n, h, w, nh, nw, 1e-8f);
    hipLaunchKernelGGL(mesh_apply_kernel, dim3(grid_x(h * w), n), dim3(256), 0, s, (const unsigned*)tmin, Kinv, frame, mask, depth, n, h, w);
    return g3_check_launch("g3_mesh_occlusion_f32");
}

extern "C" int g3_unproject_points_f32(const float* depth, const float* c2w, const float* Kinv, float* points, int n, int h,
                                       int w, void* stream) {
    if (!depth || !c2w || !Kinv || !points) return g3_set_error(G3_ERR_ARG, "g3_unproject_points_f32: null operand");
    if (n <= 0 || h <= 0 || w <= 0) return g3_set_error(G3_ERR_ARG, "g3_unproject_points_f32: bad shape");
    hipLaunchKernelGGL(unproject_kernel, dim3(grid_x(h * w), n), dim3(256), 0, (hipStream_t)stream, depth, c2w, Kinv, points, n, h, w);
    return g3_check_launch("g3_unproject_points_f32");
}

extern "C" int g3_reliable_depth_mask_f32(const float* depth, uint8_t* out, int n, int h, int w, int window, float ratio_thresh,
                                          float eps, void* stream) {
    if (!depth || !out) return g3_set_error(G3_ERR_ARG, "g3_reliable_depth_mask_f32: null operand");
    if (n <= 0 || h <= 0 || w <= 0 || window <= 0 || (window & 1) == 0) return g3_set_error(G3_ERR_ARG, "g3_reliable_depth_mask_f32: window must be odd");
    hipLaunchKernelGGL(reliable_mask_kernel, dim3(grid_x(h * w), n), dim3(256), 0, (hipStream_t)stream, depth, out, n, h, w, window, ratio_thresh, eps);
    return g3_check_launch("g3_reliable_depth_mask_f32");
}

/* ---- one call per batch of render items ---------------------------------------------------------------------------------------------------
 * Cache3D_Base.render_cache (cache_3d.py:151-236) renders n = (target frame, cache buffer) items from n_src cached source views; the reference
 * expands the sources to one copy per item and loops forward_warp over pairs. Here an item names its source (src_index), so nothing is
 * replicated, and project -> window splat -> gather / resolve -> (mesh occlusion) run back to back on one workspace. */
namespace {
struct RenderWs {
    size_t z, flow, maskz, gmax, accum, windows, origins, dirty, occ, pts_ds, tmin, heavy_cnt, heavy, list, total;
};
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
size_t render_heavy_cap(size_t nh, size_t nw) { return (nh > 1 && nw > 1) ? std::min((nh - 1) * (nw - 1) * 2, (size_t)16384) : 0; }
RenderWs render_ws_layout(int n, int h, int w, int group_size, int factor) {
    RenderWs L;
    const size_t hw = (size_t)h * w;
    const size_t ntiles = (size_t)((w + TS - 1) / TS) * ((h + TS - 1) / TS);
    const size_t nh = h / factor, nw = w / factor;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = align256(o + bytes); return at; };
    L.accum = take((size_t)n * (h + 2) * (w + 2) * ACC_C * sizeof(float));  // first: the part g3_render_workspace_init has to zero
    L.dirty = take((size_t)n * sizeof(unsigned));
    L.occ = take((size_t)n * ntiles * sizeof(unsigned));  // epoch stamps of the output tiles the occlusion rasteriser hit
    L.z = take(n * hw * sizeof(float));
    L.flow = take(n * hw * 2 * sizeof(float));
    L.maskz = take(n * hw * sizeof(float));
    L.gmax = take(((size_t)(n + group_size - 1) / group_size) * sizeof(unsigned));
    L.windows = take((size_t)n * ntiles * WIN * WIN * ACC_C * sizeof(float));
    L.origins = take((size_t)n * ntiles * ORG_N * sizeof(int));
    L.pts_ds = take((size_t)n * nh * nw * 3 * sizeof(float));
    L.tmin = take(n * hw * sizeof(unsigned));
    L.heavy_cnt = take((size_t)n * CNT_STRIDE * sizeof(int));  // per item, a 128-byte line of its own: [0] heavy triangles, [1] listed boundary patches
    L.heavy = take((size_t)n * render_heavy_cap(nh, nw) * 9 * sizeof(float));  // vertices of the triangles that touch the camera plane
    L.list = take((nh > 1 && nw > 1) ? (size_t)n * (nh - 1) * (nw - 1) * sizeof(int) : 0);
    L.total = o;
    return L;
}
unsigned g_render_epoch = 0;  // stamps the items whose dense accumulator a launch touched; any value that differs from earlier launches' works
std::mutex g_render_epoch_mu;
// Side stream of the occlusion pass (one per device, created on first use): marking / rasterising the boundary mesh reads the cached world points
// and cameras only, nothing the projection or the splat produce, and it is a latency-bound pass that leaves most of the chip idle - so it runs
// next to project + splat (VALU-bound) and joins before the resolve pass.
hipStream_t g_render_side[16] = {};
hipStream_t render_side_stream() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lock(g_render_epoch_mu);
    if (!g_render_side[dev] && hipStreamCreateWithFlags(&g_render_side[dev], hipStreamNonBlocking) != hipSuccess) g_render_side[dev] = nullptr;
    return g_render_side[dev];
}
}  // namespace

extern "C" size_t g3_render_workspace_bytes(int n, int h, int w, int group_size) {
    if (n <= 0 || h <= 0 || w <= 0 || group_size <= 0) return 0;
    return render_ws_layout(n, h, w, group_size, 4).total;
}

extern "C" int g3_render_workspace_init(void* workspace, int n, int h, int w, int group_size, void* stream) {
    if (!workspace || ((uintptr_t)workspace & 255)) return g3_set_error(G3_ERR_ARG, "g3_render_workspace_init: workspace must be 256-byte aligned");
    if (n <= 0 || h <= 0 || w <= 0 || group_size <= 0) return g3_set_error(G3_ERR_ARG, "g3_render_workspace_init: bad shape");
    const RenderWs L = render_ws_layout(n, h, w, group_size, 4);
    hipError_t e = hipMemsetAsync((char*)workspace + L.accum, 0, L.z - L.accum, (hipStream_t)stream);  // accumulators + dirty / occlusion stamps
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)((char*)workspace + L.tmin), 0x7f800000, (size_t)n * h * w, (hipStream_t)stream);  // +inf; the renders keep it so
    if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_render_workspace_init: memset: %s", hipGetErrorString(e));
    return G3_OK;
}

extern "C" int g3_render_items_f32(const float* points_src, const float* image_src, const float* mask_src, const uint8_t* boundary_src,
                                   const int* src_index, const float* w2c, const float* K, const float* Kinv, void* workspace, float* frame,
                                   float* mask, float* depth, float* flow_out, int n, int n_src, int h, int w, int group_size, void* stream) {
    if (!points_src || !image_src || !src_index || !w2c || !K || !workspace || !frame || !mask)
        return g3_set_error(G3_ERR_ARG, "g3_render_items_f32: null operand");
    if (n <= 0 || n_src <= 0 || h <= 0 || w <= 0 || group_size <= 0) return g3_set_error(G3_ERR_ARG, "g3_render_items_f32: bad shape");
    if ((uintptr_t)workspace & 255) return g3_set_error(G3_ERR_ARG, "g3_render_items_f32: workspace must be 256-byte aligned");
    if (boundary_src && (!Kinv || !depth)) return g3_set_error(G3_ERR_ARG, "g3_render_items_f32: foreground masking needs Kinv and a depth output");
    const int factor = 4;  // mesh_downsample_factor (forward_warp_utils_pytorch.py:290)
    if (boundary_src && (h / factor < 2 || w / factor < 2)) return g3_set_error(G3_ERR_ARG, "g3_render_items_f32: frame too small for the occlusion mesh");
    const RenderWs L = render_ws_layout(n, h, w, group_size, factor);
    char* ws = (char*)workspace;
    float* z = (float*)(ws + L.z);
    float* flow = flow_out ? flow_out : (float*)(ws + L.flow);
    float* maskz = (float*)(ws + L.maskz);
    unsigned* gmax = (unsigned*)(ws + L.gmax);
    float* accum = (float*)(ws + L.accum);
    float* windows = (float*)(ws + L.windows);
    int* origins = (int*)(ws + L.origins);
    unsigned* dirty = (unsigned*)(ws + L.dirty);
    unsigned* occ = (unsigned*)(ws + L.occ);
    hipStream_t s = (hipStream_t)stream;
    unsigned epoch;
    {
        std::lock_guard<std::mutex> lock(g_render_epoch_mu);
        epoch = ++g_render_epoch;
        if (epoch == 0) epoch = ++g_render_epoch;  // 0 is what g3_render_workspace_init leaves in the stamps
    }
    if (h + 2 > 32767 || w + 2 > 65535) return g3_set_error(G3_ERR_ARG, "g3_render_items_f32: image too large for the packed texel ids (h < 32766, w < 65534)");
    hipError_t e = hipMemsetAsync(gmax, 0, ((size_t)(n + group_size - 1) / group_size) * sizeof(unsigned), s);
    if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_render_items_f32: memset: %s", hipGetErrorString(e));
    // mesh occlusion: downsampled mesh of the boundary patches -> per-pixel nearest hit (tmin); the resolve pass below applies it
    unsigned* tmin = nullptr;
    hipEvent_t ev_join = nullptr;
    if (boundary_src) {
        // fork: everything already queued on `s` (the previous render's resolve pass puts tmin / the stamps back) precedes the side stream's work
        hipStream_t ms = s;
        hipStream_t side = g3_opt_render_overlap ? render_side_stream() : nullptr;
        hipEvent_t ev_fork = nullptr;
        if (side && hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) == hipSuccess &&
            hipEventRecord(ev_fork, s) == hipSuccess && hipStreamWaitEvent(side, ev_fork, 0) == hipSuccess)
            ms = side;
        if (ev_fork) (void)hipEventDestroy(ev_fork);  // released once the wait it feeds has been satisfied
        const int nh = h / factor, nw = w / factor;
        int* heavy_cnt = (int*)(ws + L.heavy_cnt);
        int* list_cnt = heavy_cnt + 1;
        float* heavy = (float*)(ws + L.heavy);
        int* list = (int*)(ws + L.list);
        tmin = (unsigned*)(ws + L.tmin);
        const int npatch = (nh - 1) * (nw - 1);
        const int heavy_cap = (int)render_heavy_cap(nh, nw);
        const int mesh_tiles_x = (w + TS - 1) / TS, mesh_ntiles = mesh_tiles_x * ((h + TS - 1) / TS);
        e = hipMemsetAsync(heavy_cnt, 0, (size_t)n * CNT_STRIDE * sizeof(int), ms);
        if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_render_items_f32: memset: %s", hipGetErrorString(e));
        float* pts_ds = (float*)(ws + L.pts_ds);
        hipLaunchKernelGGL(mesh_mark_kernel, dim3((npatch + 255) / 256, n), dim3(256), 0, ms, boundary_src, src_index, list_cnt, list, npatch, n, h, w, nh, nw,
                           points_src, w2c, pts_ds);
        hipLaunchKernelGGL(mesh_raster_list_kernel, dim3(256, n), dim3(256), 0, ms, (const float*)pts_ds, K, Kinv, tmin, (const int*)list_cnt,
                           (const int*)list, npatch, n, h, w, nh, nw, 1e-8f, heavy_cnt, heavy, heavy_cap, occ, mesh_ntiles, mesh_tiles_x, epoch);
        hipLaunchKernelGGL(mesh_raster_heavy_kernel, dim3((h * w + 256 * HEAVY_PX - 1) / (256 * HEAVY_PX), n), dim3(256), 0, ms, Kinv, tmin,
                           (const int*)heavy_cnt, (const float*)heavy, heavy_cap, n, h, w, 1e-8f, occ, mesh_ntiles, mesh_tiles_x, epoch);
        if (ms != s) {
            if (hipEventRecord(ev_join, ms) != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_render_items_f32: event record failed");
        } else if (ev_join) {
            (void)hipEventDestroy(ev_join);
            ev_join = nullptr;
        }
    }
    const int tiles_x = (w + TS - 1) / TS, tiles_y = (h + TS - 1) / TS, ntiles = tiles_x * tiles_y;
    // fused form (default; g3_set_option("render_fused", 0) and callers that want the flow plane take the three-plane form): a z-only pre-pass for the
    // group maxima, the projection itself inside the splat - z / flow / validity never touch memory
    const bool exclusive = g3_opt_render_fused && !flow_out && g3_opt_render_exclusive;
    const int full_extent = (exclusive || g3_opt_render_full_extent) ? 1 : 0;
    if (exclusive) {
        // single-writer form (opt-in: g3_set_option("render_exclusive", 1)): the pre-pass publishes every tile's destination rectangle (and the group maxima); texels only one tile
        // reaches are resolved by the splat itself, the window workspace carries the shared ones only (warp_extent_kernel)
        hipLaunchKernelGGL(warp_extent_kernel, dim3(ntiles, n), dim3(256), 0, s, points_src, w2c, K, mask_src, gmax, origins, n, h, w, group_size, tiles_x,
                           src_index);
        hipLaunchKernelGGL((warp_splat_windows_kernel<true, true>), dim3(ntiles, n), dim3(256), 0, s, image_src, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const unsigned*)gmax, accum, windows, origins, n, h, w, group_size, tiles_x, src_index, dirty, epoch,
                           points_src, w2c, K, mask_src, frame, mask, depth, boundary_src ? 1 : 0);
    } else if (g3_opt_render_fused && !flow_out) {
        hipLaunchKernelGGL(warp_zmax_kernel, dim3(min(grid_x(h * w), 256), n), dim3(256), 0, s, points_src, w2c, K, gmax, n, h, w, group_size, src_index);
        hipLaunchKernelGGL(warp_splat_windows_kernel<true>, dim3(ntiles, n), dim3(256), 0, s, image_src, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const unsigned*)gmax, accum, windows, origins, n, h, w, group_size, tiles_x, src_index, dirty, epoch,
                           points_src, w2c, K, mask_src, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, full_extent);
    } else {
        hipLaunchKernelGGL(warp_project_kernel, dim3(min(grid_x(h * w), 256), n), dim3(256), 0, s, points_src, w2c, K, mask_src, z, flow, (float*)nullptr, maskz,
                           gmax, n, h, w, group_size, src_index);
        hipLaunchKernelGGL(warp_splat_windows_kernel<false>, dim3(ntiles, n), dim3(256), 0, s, image_src, (const float*)z, (const float*)flow, (const float*)maskz,
                           (const unsigned*)gmax, accum, windows, origins, n, h, w, group_size, tiles_x, src_index, dirty, epoch, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, full_extent);
    }
    if (ev_join) {  // join: the resolve pass applies the occlusion
        if (hipStreamWaitEvent(s, ev_join, 0) != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_render_items_f32: stream wait failed");
        (void)hipEventDestroy(ev_join);
    }
    hipLaunchKernelGGL(warp_gather_resolve_kernel, dim3(ntiles, n), dim3(256), 0, s, (const float*)windows, (const int*)origins, accum, frame, mask,
                       depth, n, h, w, ntiles, tiles_x, (const unsigned*)dirty, epoch, tmin, Kinv, (const unsigned*)occ, exclusive ? 1 : (full_extent ? 2 : 0));
    return g3_check_launch("g3_render_items_f32");
}
