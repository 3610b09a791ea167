// Error plumbing + version for the C ABI declared in include/gen3c_hip.h.
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {
thread_local char g_err[512] = "";
}

int g3_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int g3_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return G3_OK;
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
int g3_opt_gemm_regstage = env_int("G3_GEMM_REGSTAGE", 0);
int g3_opt_attn_xcd_heads = env_int("G3_ATTN_XCD_HEADS", 1);  // w4b: every XCD works on its own heads (1-D grid remap)
int g3_opt_attn_variant = env_int("G3_ATTN_VARIANT", 0);  // 0 = automatic (attention.hip: flash_attn_launch)
int g3_opt_gemm_rowmajor_tiles = env_int("G3_GEMM_ROWMAJOR_TILES", 0);
int g3_opt_gemm_wide_store = env_int("G3_GEMM_WIDE_STORE", 1);
int g3_opt_splat_tiled = env_int("G3_SPLAT_TILED", 1);
int g3_opt_render_overlap = env_int("G3_RENDER_OVERLAP", 1);
int g3_opt_render_exclusive = env_int("G3_RENDER_EXCLUSIVE", 0);  // with render_fused: extent pre-pass + single-writer texels resolved inside the splat (less traffic, more instructions: slower)
int g3_opt_render_full_extent = env_int("G3_RENDER_FULL_EXTENT", 1);  // g3_render_items_f32: tiles publish their unclamped rectangle, the gather pass reads the dense accumulator per texel
int g3_opt_render_fused = env_int("G3_RENDER_FUSED", 1);  // g3_render_items_f32: projection inside the splat + z-only pre-pass (no z / flow / validity planes)
int g3_opt_gemm_pingpong = env_int("G3_GEMM_PINGPONG", 3);  // 3: one wave per SIMD (gemm_w4.hpp) where it applies, else the 2-phase ping-pong kernel
int g3_opt_tok_tattn_px = env_int("G3_TOK_TATTN_PX", 1);
int g3_opt_gemm_persistent = env_int("G3_GEMM_PERSISTENT", 0);  // measured slower or equal (profiles/r3_gemm_persistent_ab.txt): off
int g3_opt_gemm_deferred = env_int("G3_GEMM_DEFERRED", 1);  // block GEMMs with the epilogue deferred into the next tile's K loop (gemm_w4e.hpp) where it applies
int g3_opt_gemm_deferred_grid = env_int("G3_GEMM_DEFERRED_GRID", 0);  // tests: workgroup count of the deferred-epilogue GEMM (0 = one per CU)
int g3_opt_gemm_tokens_first = env_int("G3_GEMM_TOKENS_FIRST", 2);  // gemm_w4e.hpp piece order: 0 weights first, 1 tokens first, 2 (default) tokens first where N <= 4096
int g3_opt_conv_w4 = env_int("G3_CONV_W4", 1);  // tokenizer convolutions on the one-wave-per-SIMD kernel (gemm_w4_conv.hpp) where it applies
int g3_opt_ln_wave_rows = env_int("G3_LN_WAVE_ROWS", 0);  // norm_rope.hip: ln_modulate_wave_kernel (A/B: profiles/r6_ln_wave_ab.txt)
int g3_opt_norm_octets = env_int("G3_NORM_OCTETS", 1);  // norm_rope.hip: octet form of the per-head RMSNorm + RoPE pass (0: one group per (row, head), A/B)
int g3_opt_gemm_unpinned = env_int("G3_GEMM_UNPINNED", 1);  // measured: pinning the LDS prefetch does not help this kernel (profiles/r1_v4_gemm_pin_ab.txt)

extern "C" int g3_set_option(const char* name, int value) {
    if (!name) return g3_set_error(G3_ERR_ARG, "g3_set_option: null name");
    if (!strcmp(name, "gemm_regstage")) { g3_opt_gemm_regstage = value; return G3_OK; }
    if (!strcmp(name, "gemm_wide_store")) { g3_opt_gemm_wide_store = value; return G3_OK; }
    if (!strcmp(name, "splat_tiled")) { g3_opt_splat_tiled = value; return G3_OK; }
    if (!strcmp(name, "render_overlap")) { g3_opt_render_overlap = value; return G3_OK; }
    if (!strcmp(name, "render_fused")) { g3_opt_render_fused = value; return G3_OK; }
    if (!strcmp(name, "render_exclusive")) { g3_opt_render_exclusive = value; return G3_OK; }
    if (!strcmp(name, "render_full_extent")) { g3_opt_render_full_extent = value; return G3_OK; }
    if (!strcmp(name, "gemm_pingpong")) { g3_opt_gemm_pingpong = value; return G3_OK; }
    if (!strcmp(name, "gemm_unpinned")) { g3_opt_gemm_unpinned = value; return G3_OK; }
    if (!strcmp(name, "conv_w4")) { g3_opt_conv_w4 = value; return G3_OK; }
    if (!strcmp(name, "gemm_persistent")) { g3_opt_gemm_persistent = value; return G3_OK; }
    if (!strcmp(name, "gemm_deferred")) { g3_opt_gemm_deferred = value; return G3_OK; }
    if (!strcmp(name, "gemm_tokens_first")) { g3_opt_gemm_tokens_first = value; return G3_OK; }
    if (!strcmp(name, "gemm_deferred_grid")) { g3_opt_gemm_deferred_grid = value; return G3_OK; }
    if (!strcmp(name, "tok_tattn_px")) { g3_opt_tok_tattn_px = value; return G3_OK; }
    if (!strcmp(name, "gemm_rowmajor_tiles")) { g3_opt_gemm_rowmajor_tiles = value; return G3_OK; }
    if (!strcmp(name, "ln_wave_rows")) { g3_opt_ln_wave_rows = value; return G3_OK; }
    if (!strcmp(name, "norm_octets")) { g3_opt_norm_octets = value; return G3_OK; }
    if (!strcmp(name, "attn_variant")) { g3_opt_attn_variant = value; return G3_OK; }
    if (!strcmp(name, "attn_xcd_heads")) { g3_opt_attn_xcd_heads = value; return G3_OK; }
    return g3_set_error(G3_ERR_ARG, "g3_set_option: unknown option %s", name);
}

extern "C" const char* g3_last_error(void) { return g_err; }
extern "C" int g3_abi_version(void) { return 1; }

// Device facts the host side uses to size grids (CU count) and to refuse running on anything but gfx950.
extern "C" int g3_device_info(int device, int* cu_count, int* is_gfx950, char* arch_name, int arch_name_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_device_info: %s", hipGetErrorString(e));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (is_gfx950) *is_gfx950 = (strncmp(prop.gcnArchName, "gfx950", 6) == 0) ? 1 : 0;
    if (arch_name && arch_name_len > 0) snprintf(arch_name, arch_name_len, "%s", prop.gcnArchName);
    return G3_OK;
}

// hipEvent helpers so bench.py can time kernels on the exact stream they were launched on without torch types.
extern "C" int g3_event_create(void** ev) {
    hipEvent_t e;
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_event_create: %s", hipGetErrorString(r));
    *ev = (void*)e;
    return G3_OK;
}
extern "C" int g3_event_record(void* ev, void* stream) {
    hipError_t r = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    if (r != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_event_record: %s", hipGetErrorString(r));
    return G3_OK;
}
extern "C" int g3_event_elapsed_ms(void* start, void* stop, float* ms) {
    hipError_t r = hipEventSynchronize((hipEvent_t)stop);
    if (r == hipSuccess) r = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    if (r != hipSuccess) return g3_set_error(G3_ERR_LAUNCH, "g3_event_elapsed_ms: %s", hipGetErrorString(r));
    return G3_OK;
}
extern "C" int g3_event_destroy(void* ev) {
    (void)hipEventDestroy((hipEvent_t)ev);
    return G3_OK;
}
