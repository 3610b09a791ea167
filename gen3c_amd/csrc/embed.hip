// DiT input / output plumbing kernels (bf16, HBM-streaming, a few MB each):
//   patchify   : channel concat of up to 4 sources + 1x2x2 patch gather into the token-major matrix the embedding GEMM reads
//                (general_dit_video_conditioned.py:77-101 torch.cat's + blocks.py:154-159 Rearrange, THWBD row order)
//   unpatchify : final-layer output rows -> [B, C, T, H, W] (general_dit.py:348-357 "(p1 p2 t C)" column order)
//   timestep   : sinusoidal embedding (blocks.py:38-57, flip_sin_to_cos -> [cos | sin]) rounded to bf16, and its affine
//                RMSNorm (general_dit.py:173-177: x * rsqrt(mean(x^2) + 1e-6) * weight in fp32 -> bf16)
#include "common.hpp"
#include <stdint.h>

namespace {

struct PatchSrc {
    const bf16_t* p[4];
    int chans[4];
    int64_t t_stride[4];  // elements between frames (0: the source has no T axis and is broadcast, e.g. the padding mask)
    int64_t c_stride[4];  // elements between channels
    int64_t b_stride[4];  // elements between batch items
    int n;
};

__global__ __launch_bounds__(256) void dit_patchify_kernel(PatchSrc s, bf16_t* __restrict__ out, int B, int T, int H, int W, int pt,
                                                          int ps, int ctot) {
    const int Tp = T / pt, Hp = H / ps, Wp = W / ps;
    const int pd = ctot * pt * ps * ps;
    const int64_t total = (int64_t)Tp * Hp * Wp * B * pd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % pd);
        int64_t row = i / pd;
        const int b = (int)(row % B);
        row /= B;
        const int wp = (int)(row % Wp);
        row /= Wp;
        const int hp = (int)(row % Hp);
        const int tp = (int)(row / Hp);
        // col = ((c*pt + r)*ps + m)*ps + n
        const int nn = col % ps;
        const int m = (col / ps) % ps;
        const int r = (col / (ps * ps)) % pt;
        int c = col / (ps * ps * pt);
        int k = 0;
        while (k < s.n - 1 && c >= s.chans[k]) {
            c -= s.chans[k];
            ++k;
        }
        const int t = tp * pt + r, y = hp * ps + m, x = wp * ps + nn;
        out[i] = s.p[k][(int64_t)b * s.b_stride[k] + (int64_t)c * s.c_stride[k] + (int64_t)t * s.t_stride[k] + (int64_t)y * W + x];
    }
}

__global__ __launch_bounds__(256) void dit_unpatchify_kernel(const bf16_t* __restrict__ y, int64_t ldy, bf16_t* __restrict__ out, int B,
                                                            int Co, int T, int H, int W, int pt, int ps) {
    const int Tp = T / pt, Hp = H / ps, Wp = W / ps;
    const int64_t total = (int64_t)B * Co * T * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int x = (int)(r % W); r /= W;
        const int yy = (int)(r % H); r /= H;
        const int t = (int)(r % T); r /= T;
        const int c = (int)(r % Co);
        const int b = (int)(r / Co);
        const int tp = t / pt, tt = t % pt, hp = yy / ps, p1 = yy % ps, wp = x / ps, p2 = x % ps;
        const int64_t row = (((int64_t)tp * Hp + hp) * Wp + wp) * B + b;
        const int col = ((p1 * ps + p2) * pt + tt) * Co + c;
        out[i] = y[row * ldy + col];
    }
}

// one workgroup per batch item; D = 2 * half
__global__ __launch_bounds__(256) void timestep_embed_kernel(const float* __restrict__ t, const bf16_t* __restrict__ norm_w,
                                                            bf16_t* __restrict__ t_sin, bf16_t* __restrict__ emb, int D) {
    const int b = blockIdx.x;
    const int half = D / 2;
    const float tv = t[b];
    float ss = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) {
        const int j = i < half ? i : i - half;
        const float expo = (-9.210340371976184f * (float)j) / (float)half;  // -ln(10000) * arange / (half - 0)
        const float ang = tv * expf(expo);
        const float v = (float)f32_to_bf16(i < half ? cosf(ang) : sinf(ang));  // flip_sin_to_cos: [cos | sin], then .to(bf16)
        t_sin[(int64_t)b * D + i] = f32_to_bf16(v);
        ss += v * v;
    }
    __shared__ float red[256];
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float r = rsqrtf(red[0] / (float)D + 1e-6f);
    for (int i = threadIdx.x; i < D; i += 256) {
        const float v = (float)t_sin[(int64_t)b * D + i];
        emb[(int64_t)b * D + i] = f32_to_bf16((v * r) * (float)norm_w[i]);
    }
}

int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int g3_dit_patchify_bf16(const void* const* srcs, const int* chans, const int* has_t, int n_src, void* out, int B, int T,
                                    int H, int W, int patch_t, int patch_s, void* stream) {
    if (!srcs || !chans || !has_t || !out || n_src < 1 || n_src > 4) return g3_set_error(G3_ERR_ARG, "g3_dit_patchify_bf16: 1..4 sources");
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || patch_t <= 0 || patch_s <= 0 || T % patch_t || H % patch_s || W % patch_s)
        return g3_set_error(G3_ERR_ARG, "g3_dit_patchify_bf16: T/H/W must be multiples of the patch size (T=%d H=%d W=%d)", T, H, W);
    PatchSrc s;
    int ctot = 0;
    for (int k = 0; k < 4; ++k) {
        s.p[k] = nullptr; s.chans[k] = 0; s.t_stride[k] = s.c_stride[k] = s.b_stride[k] = 0;
    }
    for (int k = 0; k < n_src; ++k) {
        if (!srcs[k] || chans[k] <= 0) return g3_set_error(G3_ERR_ARG, "g3_dit_patchify_bf16: bad source %d", k);
        s.p[k] = (const bf16_t*)srcs[k];
        s.chans[k] = chans[k];
        s.t_stride[k] = has_t[k] ? (int64_t)H * W : 0;
        s.c_stride[k] = has_t[k] ? (int64_t)T * H * W : (int64_t)H * W;
        s.b_stride[k] = s.c_stride[k] * chans[k];
        ctot += chans[k];
    }
    s.n = n_src;
    const int64_t total = (int64_t)(T / patch_t) * (H / patch_s) * (W / patch_s) * B * ctot * patch_t * patch_s * patch_s;
    hipLaunchKernelGGL(dit_patchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)out, B, T, H, W, patch_t, patch_s, ctot);
    return g3_check_launch("g3_dit_patchify_bf16");
}

extern "C" int g3_dit_unpatchify_bf16(const void* y, int64_t ldy, void* out, int B, int C_out, int T, int H, int W, int patch_t, int patch_s,
                                      void* stream) {
    if (!y || !out) return g3_set_error(G3_ERR_ARG, "g3_dit_unpatchify_bf16: null operand");
    if (B <= 0 || C_out <= 0 || T % patch_t || H % patch_s || W % patch_s || ldy < (int64_t)patch_s * patch_s * patch_t * C_out)
        return g3_set_error(G3_ERR_ARG, "g3_dit_unpatchify_bf16: bad shape");
    const int64_t total = (int64_t)B * C_out * T * H * W;
    hipLaunchKernelGGL(dit_unpatchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, ldy, (bf16_t*)out, B, C_out, T, H,
                       W, patch_t, patch_s);
    return g3_check_launch("g3_dit_unpatchify_bf16");
}

extern "C" int g3_timestep_embedding_bf16(const float* timesteps, const void* norm_weight, void* t_sin, void* emb, int B, int D, void* stream) {
    if (!timesteps || !norm_weight || !t_sin || !emb) return g3_set_error(G3_ERR_ARG, "g3_timestep_embedding_bf16: null operand");
    if (B <= 0 || D <= 0 || (D & 1)) return g3_set_error(G3_ERR_ARG, "g3_timestep_embedding_bf16: D must be even");
    hipLaunchKernelGGL(timestep_embed_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, timesteps, (const bf16_t*)norm_weight, (bf16_t*)t_sin, (bf16_t*)emb, D);
    return g3_check_launch("g3_timestep_embedding_bf16");
}
