// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the GEN3C denoising path.
// Wave = 64 lanes everywhere. No portability shims: this code only targets gfx950.
#pragma once
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

int g3_set_error(int code, const char* fmt, ...);
int g3_check_launch(const char* what);

// runtime switches for A/B measurements (g3_set_option); defaults come from the environment on first use
extern int g3_opt_gemm_regstage;  // 1: register-staged GEMM even when the direct-to-LDS path applies
extern int g3_opt_gemm_rowmajor_tiles;  // 1: plain row-major tile order inside an XCD run (A/B); 0: 4-token-tile super-rows
extern int g3_opt_gemm_unpinned;        // 1: compiler-scheduled GEMM main loop (no sched_group_barrier pinning) (A/B)
extern int g3_opt_gemm_pingpong;        // 1 (default): phase-staggered ping-pong kernel for plain K%64==0 GEMMs
extern int g3_opt_attn_variant;   // 1: non-pipelined attention kernel, 2: software-pipelined (default)

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
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below one bf16 ulp of the GELU output): 1 rcp + 1 exp + 6 fma
// instead of libm's ~40-instruction erff in a GEMM epilogue that evaluates it 128 times per lane.
G3_DEVICE float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float e = 1.0f - poly * t * __expf(-z * z);  // erf(|x|/sqrt2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}
G3_DEVICE float silu(float x) { return x / (1.0f + __expf(-x)); }
