// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the GEN3C denoising path.
// Wave = 64 lanes everywhere. No portability shims: this code only targets gfx950.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define G3_DEVICE __device__ __forceinline__

// status codes of the C ABI (include/gen3c_hip.h)
#define G3_OK 0
#define G3_ERR_ARG 1
#define G3_ERR_LAUNCH 2
#define G3_ERR_RESOURCE 3

int g3_set_error(int code, const char* fmt, ...);
int g3_check_launch(const char* what);

// runtime switches for A/B measurements (g3_set_option); defaults come from the environment on first use
extern int g3_opt_gemm_regstage;  // 1: register-staged GEMM even when the direct-to-LDS path applies
extern int g3_opt_gemm_rowmajor_tiles;  // 1: plain row-major tile order inside an XCD run (A/B); 0: 4-token-tile super-rows
extern int g3_opt_gemm_unpinned;        // 1: compiler-scheduled GEMM main loop (no sched_group_barrier pinning) (A/B)
extern int g3_opt_tok_tattn_px;         // tokenizer temporal attention: one wave per pixel (default 1) instead of one per (pixel, query frame)
extern int g3_opt_gemm_persistent;      // one-wave-per-SIMD GEMM: persistent tile loop (next tile's first stage requested under the epilogue)
extern int g3_opt_gemm_deferred;        // 1 (default): persistent one-wave-per-SIMD GEMM with the epilogue deferred into the next tile's K loop (gemm_w4e.hpp)
extern int g3_opt_gemm_deferred_grid;    // tests: workgroups of the tile loop of the deferred-epilogue GEMM (0 = one per CU)
extern int g3_opt_gemm_tokens_first;    // gemm_w4e.hpp: LDS-DMA piece order (0 weight rows get the two-tile lead, 1 token rows, 2 = default: token rows where N <= 4096)
extern int g3_opt_conv_w4;              // tokenizer convolutions on the one-wave-per-SIMD kernel where it applies (default 1)
extern int g3_opt_gemm_pingpong;        // plain K%64==0 GEMMs: 2 (default) / 1 = phase-staggered ping-pong kernel with 2 / 4 phases per K tile, 0 = classic
extern int g3_opt_gemm_wide_store;      // 1 (default): LDS-transposed full-line epilogue when the operands allow 16-byte rows
extern int g3_opt_render_exclusive;     // 1 (needs render_fused; default 0): every tile's destination rectangle is published by a pre-pass and texels a single tile reaches are resolved inside the splat
extern int g3_opt_render_full_extent;   // 1 (default): g3_render_items_f32's tiles publish their unclamped destination rectangle and the gather pass reads the dense accumulator per texel, not per item
extern int g3_opt_render_fused;         // 1 (default): g3_render_items_f32 projects inside the splat (z-only pre-pass for the group maxima) instead of writing z / flow / validity planes
extern int g3_opt_render_overlap;       // 1 (default): g3_render_items_f32 runs the occlusion pass on a side stream next to project + splat
extern int g3_opt_splat_tiled;          // 1 (default): LDS-windowed splat; 0: direct global atomics (A/B)
extern int g3_opt_attn_xcd_heads;  // 1 (default): w4b attention launches a 1-D grid and gives every XCD its own (batch, head) pairs
extern int g3_opt_ln_wave_rows;        // 1: LayerNorm + AdaLN at D = 4096 with one wave per row (no LDS round trip / barrier); 0: one workgroup per row
extern int g3_opt_norm_octets;         // 1 (default): per-head RMSNorm + RoPE in the octet form (one 8-lane group keeps a row's cos / sin for 8 heads) where H % 8 == 0
extern int g3_opt_attn_variant;   // 1 non-pipelined, 2 software-pipelined, 3 LDS-DMA + pinned interleave, 4 (default) = 3 with the softmax scale folded into Q and the running max into the MFMA's C operand

G3_DEVICE float bf16_to_f32(bf16_t v) { return (float)v; }
G3_DEVICE bf16_t f32_to_bf16(float v) { return (bf16_t)v; }  // RNE (v_cvt_pk_bf16_f32 on gfx950)

G3_DEVICE bf16x8 load_bf16x8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
G3_DEVICE void store_bf16x8(bf16_t* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
G3_DEVICE bf16x8 zero_bf16x8() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return __builtin_bit_cast(bf16x8, z);
}

G3_DEVICE float wave_xor_f32(float v, int mask) { return __shfl_xor(v, mask, 64); }

G3_DEVICE float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU(x) = x * Phi(x) with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below one bf16 ulp of the output),
// folded so that no sign handling is left:  with z = |x|/sqrt2, t = 1/(1 + p z), q = poly(t) t exp(-z^2) = 1 - erf(z):
//   x >= 0: 0.5 x (2 - q) = x - 0.5 |x| q ;   x < 0: 0.5 x q = -0.5 |x| q      =>   GELU(x) = max(x, 0) - 0.5 |x| q.
// 1 rcp + 1 exp2 + 12 VALU (the 0.5 lives in the coefficients) instead of libm's ~40-instruction erff, in a GEMM epilogue
// that evaluates it 128 times per lane.
G3_DEVICE float gelu_erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float poly = __builtin_fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    poly = __builtin_fmaf(poly, t, 0.5f * 1.421413741f);
    poly = __builtin_fmaf(poly, t, 0.5f * -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.5f * 0.254829592f);
    const float ex = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.44269504088896340736f));  // exp(-z^2), z^2 = x^2 / 2
    const float a = ((poly * t) * ex) * ax;
    return fmaxf(x, 0.0f) - a;
}
// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, class F> G3_DEVICE void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Race screen: -DG3_AB_JITTER=<n> (tools/ab_flags.py) makes pseudo-randomly chosen waves sleep n*64 cycles at tile / phase
// boundaries. A kernel whose LDS hand-offs are correctly fenced gives bit-identical results under any such timing.
#ifdef G3_AB_JITTER
#define G3_JITTER(a, b) do { if (((((int)(a)) * 7 + ((int)(b)) * 3) & 15) == 5) __builtin_amdgcn_s_sleep(G3_AB_JITTER); } while (0)
#else
#define G3_JITTER(a, b) ((void)0)
#endif

// Barrier that PUBLISHES LDS-DMA data (global_load_lds): the data is ordered for another wave's ds_read only by the issuing
// wave's vmcnt wait followed by a barrier. hipcc usually emits that wait at a __syncthreads() that follows LDS-DMA, but it is
// not obliged to (measured: it dropped it at the attention prologue) - so the wait is explicit wherever DMA data is handed over.
G3_DEVICE void lds_dma_publish_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
G3_DEVICE float silu(float x) { return x / (1.0f + __expf(-x)); }
