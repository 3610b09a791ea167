"""ctypes binding of libgen3c_hip.so (C ABI: include/gen3c_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, the product path raises. The CPU
oracle under oracle/ is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libgen3c_hip.so"
_lib = None

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float

# status codes of include/gen3c_hip.h
G3_OK, G3_ERR_ARG, G3_ERR_LAUNCH, G3_ERR_RESOURCE = 0, 1, 2, 3

# name -> argtypes (restype is int unless listed in _RESTYPES). Mirrors include/gen3c_hip.h one-to-one; the
# CPU test tests/test_abi.py checks that every symbol declared in the header is listed here and exported.
SIGNATURES = {
    "g3_last_error": [],
    "g3_abi_version": [],
    "g3_set_option": [C.c_char_p, i32],
    "g3_device_info": [i32, C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32],
    "g3_event_create": [C.POINTER(vp)],
    "g3_event_record": [vp, vp],
    "g3_event_elapsed_ms": [vp, vp, C.POINTER(f32)],
    "g3_event_destroy": [vp],
    "g3_gemm_bf16_nt": [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp, i32, i64, vp, i64, vp],
    "g3_flash_attn_fwd_kvseg_bf16": [vp, i64, i64, i64, vp, i64, i64, i64, vp, i64, i64, i64, i32, i64, vp, i64, i64, i64, i32, i32, i32,
                                     i32, i32, f32, vp],
    "g3_gemm_kernel_name": [i32, i32, i32, i32],
    "g3_gemv_bf16": [vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp],
    "g3_flash_attn_kernel_name": [i32, i32, i32, i32],
    "g3_flash_attn_fwd_bf16": [vp, i64, i64, i64, vp, i64, i64, i64, vp, i64, i64, i64, vp, i64, i64, i64, i32, i32, i32,
                               i32, i32, f32, vp],
    "g3_cross_attn_fwd_bf16": [vp, i64, i64, i64, vp, f32, vp, i64, i64, i64, vp, i64, i64, i64, vp, i64, i64, i64, i32, i32, i32, i32, i32,
                               i32, f32, vp],
    "g3_flash_attn_fwd_ex_bf16": [vp, i64, i64, i64, vp, i64, i64, i64, vp, i64, i64, i64, i32, i64, vp, vp, vp, i64, i64, i64, i32, i32, i32,
                                  i32, i32, f32, i32, vp],
    "g3_flash_attn_kernel_name_ex": [i32, i32, i32, i32, i32],
    "g3_attn_merge_partials_bf16": [vp, vp, i32, i64, i64, i64, vp, i64, i64, i64, i32, i32, i32, i32, vp],
    "g3_transpose_v_bf16": [vp, i64, vp, i64, i32, i32, i32, i32, vp],
    "g3_layernorm_modulate_bf16": [vp, i64, vp, vp, i64, i32, vp, i64, i32, i32, f32, vp],
    "g3_posemb_layernorm_modulate_bf16": [vp, i64, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, i64, i32, vp, i64, i32, f32, vp],
    "g3_qk_rmsnorm_rope_bf16": [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, vp],
    "g3_qk_rmsnorm_rope_pair_bf16": [vp, i64, vp, i32, vp, i32, vp, vp, vp, i64, i32, i32, i32, f32, vp],
    "g3_gemm_qk_norm_rope_bf16": [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, f32, vp, i64, vp],
    "g3_add_inplace_bf16": [vp, vp, i64, vp],
    "g3_warp_project_f32": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "g3_warp_splat_f32": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "g3_warp_resolve_f32": [vp, vp, vp, vp, i32, i32, i32, vp],
    "g3_warp_windows_workspace_bytes": [i32, i32, i32],
    "g3_warp_splat_resolve_f32": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "g3_render_workspace_bytes": [i32, i32, i32, i32],
    "g3_render_workspace_init": [vp, i32, i32, i32, i32, vp],
    "g3_render_items_f32": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "g3_mesh_occlusion_f32": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "g3_unproject_points_f32": [vp, vp, vp, vp, i32, i32, i32, vp],
    "g3_dit_patchify_bf16": [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp],
    "g3_dit_unpatchify_bf16": [vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "g3_timestep_embedding_bf16": [vp, vp, vp, vp, i32, i32, vp],
    "g3_reliable_depth_mask_f32": [vp, vp, i32, i32, i32, i32, f32, f32, vp],
    "g3_align_depth_workspace_bytes": [i32, i32],
    "g3_align_depth_f32": [vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, C.c_size_t, i32, i32, vp],
    "g3_conv3d_cl_bf16": [vp, i64, vp, i64, vp, vp, i64, vp, i64] + [i32] * 17 + [vp],
    "g3_conv3d_cl_gnstats_bf16": [vp, i64, vp, i64, vp, vp, i64, vp, i64] + [i32] * 17 + [vp, i32, vp],
    "g3_groupnorm_stats_cl_bf16": [vp, i64, vp, i32, i32, i32, vp],
    "g3_groupnorm_apply_cl_bf16": [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, f32, i32, vp],
    "g3_groupnorm_swish_cl_bf16": [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, f32, i32, vp],
    "g3_haar3d_patch_bf16": [vp, vp, i32, i32, i32, vp],
    "g3_haar3d_unpatch_bf16": [vp, i64, vp, i32, i32, i32, vp],
    "g3_resample_cl_bf16": [vp, vp, i32, i32, i32, i32, i32, vp],
    "g3_softmax_rows_bf16": [vp, i64, i32, i32, f32, vp],
    "g3_transpose2d_bf16": [vp, i64, vp, i64, i32, i32, vp],
    "g3_temporal_attn_cl_bf16": [vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "g3_spatial_attn_d512_bf16": [vp, vp, vp, i64, i64, vp, i32, i32, f32, vp],
    "g3_edm_prepare_input_bf16": [vp, vp, vp, vp, vp, vp, i64, i32, i32, f32, f32, f32, f32, vp],
    "g3_edm_cfg_euler_step_bf16": [vp, vp, vp, vp, vp, vp, i64, i32, i32, f32, f32, f32, f32, f32, f32, f32, vp],
}
_RESTYPES = {"g3_last_error": C.c_char_p, "g3_flash_attn_kernel_name": C.c_char_p, "g3_gemm_kernel_name": C.c_char_p, "g3_flash_attn_kernel_name_ex": C.c_char_p, "g3_align_depth_workspace_bytes": C.c_size_t,
             "g3_warp_windows_workspace_bytes": C.c_size_t, "g3_render_workspace_bytes": C.c_size_t}


class Gen3cHipError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load the shared library (once). Raises Gen3cHipError with build instructions if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; it must be the HIP runtime of the process BEFORE our library (linked against the
    # same SONAME) is mapped, otherwise two runtimes coexist and launches fail with "no ROCm-capable device".
    import torch  # noqa: F401
    if not _LIB_PATH.exists():
        raise Gen3cHipError(
            f"{_LIB_PATH} is missing: build it with `python -m gen3c_amd.build` (or __graft_entry__.build()). "
            "gen3c_amd has no CPU/PyTorch fallback for its HIP kernels."
        )
    lib = C.CDLL(str(_LIB_PATH))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/binding drift; fail loudly
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().g3_last_error()
    return msg.decode() if msg else "?"


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().g3_last_error()
        raise Gen3cHipError(f"{what or 'gen3c_hip call'} failed (rc={rc}): {msg.decode() if msg else '?'}")
