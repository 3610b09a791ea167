/*
 * gen3c_hip.h - C ABI of libgen3c_hip.so: the MI355X (gfx950) kernels of the GEN3C-Cosmos-7B denoising path.
 *
 * The reference (nv-tlabs/GEN3C, a Cosmos-Predict1 fork) has no FFI boundary for this path: the hot operators sit
 * behind Python module seams and third-party CUDA libraries (TransformerEngine, cuBLAS via nn.Linear, ATen
 * index_put_, NVIDIA Warp).  Each entry point below names the reference call site(s) it replaces; INTEGRATION.md
 * shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless marked "host"; nothing is allocated or freed inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on that stream;
 *   - tensors are bf16 (uint16 storage) unless the name says f32; leading dims / strides are in ELEMENTS;
 *   - return value: 0 = G3_OK, non-zero = error (g3_last_error() returns a thread-local message);
 *   - thread-compatible: no mutable global state besides lazily-set kernel attributes and the g3_set_option A/B switches (process-wide,
 *     meant for tools; per-call choices go through the *_ex entry points).
 */
#ifndef GEN3C_HIP_H
#define GEN3C_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G3_OK 0
#define G3_ERR_ARG 1
#define G3_ERR_LAUNCH 2
#define G3_ERR_RESOURCE 3 /* the device refused a static resource request (dynamic LDS size): the caller may take another path */

/* epilogues of g3_gemm_bf16_nt */
#define G3_EPI_NONE 0           /* C = A.W^T                                   */
#define G3_EPI_GELU 1           /* C = gelu_erf(A.W^T)      attention.py:86,94-99 (GPT2FeedForward)            */
#define G3_EPI_GATED_RESIDUAL 2 /* C = R + gate[m%rows] * (A.W^T)   blocks.py:455-471 (x + gate * block(...))   */
#define G3_EPI_BIAS 3           /* C = A.W^T + bias[m%rows]                                                     */
#define G3_EPI_BIAS_RESIDUAL 4  /* C = A.W^T + bias[m%rows] + R                                                 */

const char* g3_last_error(void);
int g3_abi_version(void);
/* runtime switches for A/B measurements: "gemm_regstage", "gemm_rowmajor_tiles", "gemm_unpinned" (0/1), "gemm_pingpong" (3 = one wave per SIMD, the
 * default), "gemm_deferred" (1 = the persistent block GEMM with the deferred epilogue where it applies, the default; 0 = epilogue behind every K loop),
 * "gemm_persistent", "conv_w4" (2 = without the in-gap tap change), "attn_variant" (0 = automatic), "attn_xcd_heads", "splat_tiled", "render_overlap", "render_fused", "render_full_extent"
 * (1, default: the renderer's tiles publish unclamped destination rectangles and the gather pass reads the dense accumulator per texel),
 * "render_exclusive" (1: extent pre-pass + single-writer texels resolved inside the splat - a quarter less memory traffic, 18 % slower; default 0),
 * "tok_tattn_px". Outputs do not depend on them. */
int g3_set_option(const char* name /*host*/, int value);
int g3_device_info(int device, int* cu_count, int* is_gfx950, char* arch_name /*host*/, int arch_name_len);

/* hipEvent helpers (opaque handles) so a host can time a stream without linking HIP itself. */
int g3_event_create(void** ev);
int g3_event_record(void* ev, void* stream);
int g3_event_elapsed_ms(void* start, void* stop, float* ms /*host*/);
int g3_event_destroy(void* ev);

/* ---- DiT linears ---------------------------------------------------------------------------------------------
 * C[M,N] = epi(A[M,K] . W[N,K]^T): replaces nn.Linear(bias=False) in Attention.to_q/to_k/to_v/to_out
 * (cosmos_predict1/diffusion/module/attention.py:207-223), GPT2FeedForward.layer1/layer2 (attention.py:61-62),
 * PatchEmbed.proj (blocks.py:160-162) and FinalLayer.linear (blocks.py:205-206).
 * Needs K, lda, ldw multiples of 8; N, ldc (ldg, ldr) multiples of 4. gate/bias is [gate_rows][ldg]. */
int g3_gemm_bf16_nt(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                    int epilogue, const void* gate, int gate_rows, int64_t ldg, const void* residual, int64_t ldr,
                    void* stream);

/* Name of the kernel the call above launches for this shape under the options in force (aligned operands assumed): profilers / bench.py only. */
const char* g3_gemm_kernel_name(int M, int N, int K, int epilogue);

/* out[M<=8][N] = (act_in(a) . w^T) (+ add): TimestepEmbedding (blocks.py:60-80) and adaLN_modulation
 * (blocks.py:411-415, 442-447; FinalLayer blocks.py:212-216, 230). act_in: 0 none, 1 SiLU. */
int g3_gemv_bf16(const void* a, int64_t lda, const void* w, int64_t ldw, const void* add, int64_t ldadd, void* out,
                 int64_t ldo, int M, int N, int K, int act_in, void* stream);

/* ---- attention -------------------------------------------------------------------------------------------------
 * O = softmax(Q K^T * softmax_scale) V, no mask, head_dim 128: replaces TransformerEngine DotProductAttention
 * (attention.py:228-238, 288). Element (s, b, h, d) of Q is at q + s*q_row + b*q_batch + h*q_head + d (same for K, O);
 * V is passed TRANSPOSED: element (b, h, d, kv) at vt + b*vt_batch + h*vt_head + d*vt_row + kv, with
 * vt_row >= S_kv rounded up to 8 and the tail [S_kv, vt_row) zero (g3_transpose_v_bf16 writes exactly this). */
int g3_flash_attn_fwd_bf16(const void* q, int64_t q_row, int64_t q_batch, int64_t q_head, const void* k, int64_t k_row,
                           int64_t k_batch, int64_t k_head, const void* vt, int64_t vt_row, int64_t vt_batch,
                           int64_t vt_head, void* o, int64_t o_row, int64_t o_batch, int64_t o_head, int Sq, int Skv,
                           int B, int H, int head_dim, float softmax_scale, void* stream);

/* Name of the kernel g3_flash_attn_fwd_bf16 launches for this problem under the current "attn_variant" option (0 = automatic choice
 * between the 8-wave and the one-wave-per-SIMD kernels): for profilers and bench.py's roofline line, never needed to run the op. */
const char* g3_flash_attn_kernel_name(int Sq, int Skv, int B, int H);

/* The CROSS-attention form of the same call (Attention.forward with a context: cal_qkv's to_q[1] norm + DotProductAttention, attention.py:247-297), two options:
 *  - q_norm_weight != NULL: q holds the RAW to_q[0] projection; the per-head te.pytorch.RMSNorm(128, eps) with this weight (attention.py:130-131, 262-273) is
 *    applied inside the kernel's Q load, so the cross-attention's Q never makes the separate read + write pass of g3_qk_rmsnorm_rope_bf16 (no RoPE here:
 *    cross-attention applies none). NULL: q is used as it is.
 *  - kv_dense > 0: keys with an ALL-ZERO TAIL - the caller guarantees that the K rows AND the V^T columns of keys [kv_dense, S_kv) are exactly zero: a T5
 *    context that text_encoder zero-pads to 512 tokens (to_k / to_v carry no bias and RMSNorm(0) = 0, so the padding stays zero through cal_qkv). Those keys
 *    score exactly 0 and add nothing to the output but they DO stay in the softmax denominator (the reference attends over all 512 context tokens unmasked,
 *    general_dit.py:407-410): only the first ceil64(kv_dense) keys go through the kernel's tile loop, the tail enters in closed form (m' = max(m, 0),
 *    l' = l 2^(m-m') + n_tail 2^(-m')). Same function of the same inputs. 0: every key goes through the loop. */
int g3_cross_attn_fwd_bf16(const void* q, int64_t q_row, int64_t q_batch, int64_t q_head, const void* q_norm_weight, float q_norm_eps,
                           const void* k, int64_t k_row, int64_t k_batch, int64_t k_head, const void* vt, int64_t vt_row, int64_t vt_batch,
                           int64_t vt_head, void* o, int64_t o_row, int64_t o_batch, int64_t o_head, int Sq, int Skv, int kv_dense, int B, int H,
                           int head_dim, float softmax_scale, void* stream);

/* Same, with V^T stored in KEY SEGMENTS: keys [s*vt_seg_len, (s+1)*vt_seg_len) live in the block at vt + s*vt_seg_stride (each block
 * laid out as above with its own vt_row >= vt_seg_len). vt_seg_len must be a multiple of 64 and divide S_kv. This is what a rank-major
 * all-gather of per-rank V^T shards produces under context parallelism (module/parallel.py:110-163 gathers; TE's CP attention is the
 * reference counterpart), so no rank has to re-transpose the gathered V. */
int g3_flash_attn_fwd_kvseg_bf16(const void* q, int64_t q_row, int64_t q_batch, int64_t q_head, const void* k, int64_t k_row,
                                 int64_t k_batch, int64_t k_head, const void* vt, int64_t vt_row, int64_t vt_batch, int64_t vt_head,
                                 int vt_seg_len, int64_t vt_seg_stride, void* o, int64_t o_row, int64_t o_batch, int64_t o_head, int Sq,
                                 int Skv, int B, int H, int head_dim, float softmax_scale, void* stream);

/* Extended form of the two calls above (one entry point: vt_seg_len == 0 = plain V^T) for hosts that shard the KEYS of a row over several
 * launches - the context-parallel schedule that starts on the local K / V shard before any remote shard has arrived, which is what
 * TransformerEngine's ring does behind attn_op.set_context_parallel_group (general_dit.py:536-541, module/attention.py:228-238):
 *   variant   per-call kernel choice, 0 = the process-wide "attn_variant" option (whose 0 = automatic). Nothing global is touched, so
 *             concurrent launches on several streams can use different kernels (4 = 8-wave kernel, 11 = one-wave-per-SIMD kernel).
 *   o         bf16 result as above, or NULL when o_partial / lse are given:
 *   o_partial fp32 [Sq][B][H][128] addressed with the SAME element strides o_row / o_batch / o_head: the softmax-NORMALISED result over
 *             this launch's keys only;  lse fp32 [B][H][Sq] contiguous: log2(sum_k exp2(s_qk * scale * log2 e)) of those keys.
 * g3_attn_merge_partials_bf16 combines n_parts (1..8) such parts of the same rows into the bf16 result over the union of their keys:
 *   out = sum_i w_i * o_part_i,  w_i = 2^(lse_i - max lse) / sum_j 2^(lse_j - max lse).  o_parts / lse_parts are HOST arrays of device pointers. */
int g3_flash_attn_fwd_ex_bf16(const void* q, int64_t q_row, int64_t q_batch, int64_t q_head, const void* k, int64_t k_row, int64_t k_batch,
                              int64_t k_head, const void* vt, int64_t vt_row, int64_t vt_batch, int64_t vt_head, int vt_seg_len,
                              int64_t vt_seg_stride, void* o, float* o_partial, float* lse, int64_t o_row, int64_t o_batch, int64_t o_head,
                              int Sq, int Skv, int B, int H, int head_dim, float softmax_scale, int variant, void* stream);
const char* g3_flash_attn_kernel_name_ex(int Sq, int Skv, int B, int H, int variant);
int g3_attn_merge_partials_bf16(const float* const* o_parts /*host*/, const float* const* lse_parts /*host*/, int n_parts, int64_t p_row,
                                int64_t p_batch, int64_t p_head, void* out, int64_t o_row, int64_t o_batch, int64_t o_head, int Sq, int B,
                                int H, int head_dim, void* stream);

/* V [S][B][H][128] (row stride ld_in) -> V^T [B][H][128][ldvt], zero-filling kv in [S, ldvt). */
int g3_transpose_v_bf16(const void* v, int64_t ld_in, void* vt, int64_t ldvt, int S, int B, int H, int head_dim,
                        void* stream);

/* ---- norms -----------------------------------------------------------------------------------------------------
 * out = LayerNorm(x; no affine, eps) * (1 + scale[row % mod_rows]) + shift[row % mod_rows]
 * replaces DITBuildingBlock.norm_state + adaln_norm_state (blocks.py:339-341, 408) and FinalLayer (blocks.py:204, 239). */
int g3_layernorm_modulate_bf16(const void* x, int64_t ldx, const void* shift, const void* scale, int64_t ldmod,
                               int mod_rows, void* out, int64_t ldo, int rows, int D, float eps, void* stream);

/* per-head RMSNorm(weight[128], eps) then (optional) non-interleaved RoPE with f32 cos/sin tables [S][128]:
 * replaces te.pytorch.RMSNorm + apply_rotary_pos_emb(fused=True) in Attention.cal_qkv (attention.py:262-280).
 * in/out rows are (s, b) pairs, b fastest; cos_table == sin_table == NULL skips RoPE (cross-attention, k of context). */
int g3_qk_rmsnorm_rope_bf16(const void* in, int64_t ld_in, const void* weight, const float* cos_table,
                            const float* sin_table, void* out, int64_t ld_out, int S, int B, int H, int head_dim,
                            float eps, void* stream);

/* The same pass over TWO adjacent head ranges with their own norm weights in one launch: heads [0, H_q) of a row with weight_q, heads
 * [H_q, H_q + H_k) with weight_k - q and k of the fused QKV buffer as Attention.cal_qkv normalises and rotates them (attention.py:262-280:
 * to_q[1] / to_k[1] then apply_rotary_pos_emb on both). in / out: [S*B][(H_q + H_k) * 128] views (out may alias in). */
int g3_qk_rmsnorm_rope_pair_bf16(const void* in, int64_t ld_in, const void* weight_q, int H_q, const void* weight_k, int H_k,
                                 const float* cos_table, const float* sin_table, void* out, int64_t ld_out, int S, int B, int head_dim,
                                 float eps, void* stream);

/* Q / K(/V) projection with that per-head RMSNorm (+ RoPE) in the GEMM's epilogue (Attention.cal_qkv, attention.py:247-280, as one call):
 *   C[:, 0:n_q] = rope(rmsnorm(A W^T, norm_q)),  C[:, n_q:n_q+n_k] = rope(rmsnorm(A W^T, norm_k)),  C[:, n_q+n_k:N] = A W^T (e.g. v).
 * Row m is token (s = m / B, b = m % B); n_q, n_k, N multiples of 128 (whole heads); cos / sin fp32 [M/B][128] or both NULL. Same
 * rounding points as g3_gemm_bf16_nt followed by g3_qk_rmsnorm_rope_bf16 in place (which is what runs where the fused kernel does not apply).
 * vt != NULL: the remaining (v) heads are written TRANSPOSED, vt[b][h][d][s] with row length vt_ld >= M/B (g3_transpose_v_bf16's layout:
 * what g3_flash_attn_fwd_bf16 takes; positions >= M/B written as zeros up to the end of the last 32-row block, the rest of the tail is the
 * caller's: allocate it zeroed once), and their columns of C are NOT written. */
int g3_gemm_qk_norm_rope_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                              int n_q, int n_k, const void* norm_q, const void* norm_k, const float* cos_table,
                              const float* sin_table, int B, float eps, void* vt, int64_t vt_ld, void* stream);

/* Top of a DiT block in ONE pass over x:  x += extra_per_block_pos_emb  (in place; blocks.py:547-548), then
 * out = LayerNorm(x) * (1 + scale) + shift  as g3_layernorm_modulate_bf16. The [S*B, D] embedding is not read from memory: it is
 * rebuilt per row from LearnablePosEmbAxis' three tables pe_t [T,D], pe_h [Hp,D], pe_w [Wp,D] with the reference's bf16 rounding
 * points (position_embedding.py:218-233; normalize, attention.py:108-124):  bf16(bf16(bf16(pe_t+pe_h)+pe_w) / pos_norm[s]),
 * pos_norm [T*Hp*Wp] bf16 = 1e-6 + ||.||_2 / sqrt(D) per token. x: [T*Hp*Wp*B, D] rows (s, b), b fastest. With context
 * parallelism pe_t / pos_norm point at this rank's frames.  * pe_h == pe_w == pos_norm == NULL: pe_t is the finished (summed, normalised, rounded) embedding [T*Hp*Wp][D] and every row
 * reads one table row instead of three (what gen3c_amd/dit.py passes: the table is input-independent and built once per shape). */
int g3_posemb_layernorm_modulate_bf16(void* x, int64_t ldx, const void* pe_t, const void* pe_h, const void* pe_w,
                                      const void* pos_norm, int T, int Hp, int Wp, int B, const void* shift, const void* scale,
                                      int64_t ldmod, int mod_rows, void* out, int64_t ldo, int D, float eps, void* stream);

/* x += y (n % 8 == 0): "x = x + extra_per_block_pos_emb" (blocks.py:547-548). */
int g3_add_inplace_bf16(void* x, const void* y, int64_t n, void* stream);

/* DiT input/output plumbing. patchify: channel-concatenates up to 4 bf16 sources ([B,C_i,T,H,W] contiguous, or [B,C_i,H,W]
 * broadcast over T when has_t[i] == 0 - the padding mask) and gathers patch_t x patch_s x patch_s patches into the token-major
 * matrix out[(t h w) b][(c r m n)] the embedding GEMM reads (general_dit_video_conditioned.py:77-101, blocks.py:154-159).
 * srcs / chans / has_t are HOST arrays of n_src entries. unpatchify: final-layer rows y[(t h w) b][(p1 p2 t C)] (ld = ldy) ->
 * out [B,C_out,T,H,W] (general_dit.py:348-357). timestep_embedding: timesteps f32 [B] (device) -> t_sin bf16 [B,D] = [cos | sin]
 * sinusoid (blocks.py:38-57) and emb bf16 [B,D] = its affine RMSNorm with norm_weight [D], eps 1e-6 (general_dit.py:173-177). */
int g3_dit_patchify_bf16(const void* const* srcs, const int* chans, const int* has_t, int n_src, void* out, int B, int T, int H,
                         int W, int patch_t, int patch_s, void* stream);
int g3_dit_unpatchify_bf16(const void* y, int64_t ldy, void* out, int B, int C_out, int T, int H, int W, int patch_t, int patch_s,
                           void* stream);
int g3_timestep_embedding_bf16(const float* timesteps, const void* norm_weight, void* t_sin, void* emb, int B, int D, void* stream);

/* ---- EDM-Euler sampler step (model_v2w.py:130-149, 201-259; diffusers 0.32.2 EDMEulerScheduler) ------------------
 * Two fused elementwise passes around the two network calls of one denoise step, over a [B,C,T,H,W] latent of n
 * elements (hw = H*W, indicator = f32 [T], 1 on conditioning frames). Scalar coefficients are evaluated by the host
 * in the dtypes the reference uses (see gen3c_amd/sampler.py) and passed as floats.
 *   prepare: new_xt = bf16(ind*((gt+noise*aug)*c_in_aug/c_in_bf16) + (1-ind)*xt); new_xt_scaled = bf16(new_xt*c_in_step)
 *   step   : net = cond + g*(cond-uncond); replace conditioning frames by the (un-preconditioned) latent; Euler update. */
int g3_edm_prepare_input_bf16(const void* xt, const void* gt_latent, const float* noise, const float* indicator,
                              void* new_xt, void* new_xt_scaled, int64_t n, int T, int hw, float augment_sigma,
                              float c_in_aug, float c_in_bf16, float c_in_step, void* stream);
int g3_edm_cfg_euler_step_bf16(const void* out_cond, const void* out_uncond, const void* new_xt, const void* gt_latent,
                               const float* indicator, void* xt_next, int64_t n, int T, int hw, float guidance,
                               float c_skip_bf16, float c_out_bf16, float c_skip, float c_out, float sigma,
                               float sigma_next, void* stream);

/* ---- 3D-cache renderer (all f32; n "items" = (target frame, cache buffer) pairs, each h x w) -----------------------
 * Replaces forward_warp(depth1=None, world_points1=...) + bilinear_splatting + the mesh-occlusion branch
 * (cosmos_predict1/diffusion/inference/forward_warp_utils_pytorch.py:171-336, 462-486, 576-695, 49-132) and the NVIDIA
 * Warp ray/triangle kernel (ray_triangle_intersection_warp.py:23-105).
 * The reference takes max(log1p(z)) over every item of one forward_warp call (warp_chunk_size = 2 items,
 * cache_3d.py:183); `group_size` reproduces that grouping: item i belongs to group i / group_size.
 *   project : points [n][h][w][3], w2c [n][16], K [n][9], mask1 [n][h][w] or NULL ->
 *             z [n][h][w], flow [n][2][h][w] (= forward_warp's flow12), cam_points [n][h][w][3] or NULL,
 *             maskz = mask1*(z>0) [n][h][w], group_max [ceil(n/group_size)] u32 (float bits; ZERO it before the call)
 *   splat   : accumulates (r,g,b,z,weight) into accum [n][h+2][w+2][5] (ZERO it before the call)
 *   resolve : frame [n][3][h][w] (fill -1, clamp [-1,1]), mask [n][h][w], depth [n][h][w] or NULL (fill 0)
 *   mesh_occlusion: boundary_mask [n][h][w] u8, Kinv [n][9]; scratch pts_ds [n][h/f][w/f][3], mask_ds [n][h/f][w/f] u8,
 *             tmin [n][h][w] u32; zeroes mask/depth and sets frame to -1 where the boundary mesh is closer by > 0.02. */
int g3_warp_project_f32(const float* points, const float* w2c, const float* K, const float* mask1, float* z, float* flow,
                        float* cam_points, float* maskz, void* group_max, int n, int h, int w, int group_size, void* stream);
int g3_warp_splat_f32(const float* image, const float* z, const float* flow, const float* maskz, const void* group_max,
                      float* accum, int n, int h, int w, int group_size, void* stream);
int g3_warp_resolve_f32(const float* accum, float* frame, float* mask, float* depth, int n, int h, int w, void* stream);

/* g3_warp_splat_f32 + g3_warp_resolve_f32 in one call without global atomics on the common path: every 32x32 source tile stores its 40x40
 * destination window (32 KiB; size the workspace with g3_warp_windows_workspace_bytes, never from this text) into `workspace` and a destination-owning pass sums the overlapping windows in a fixed order and resolves the pixel
 * (bilinear_splatting + the normalisation / clamp of forward_warp, forward_warp_utils_pytorch.py:576-695,300-334). Same contributions as
 * the two-call form (whose accumulator still receives the rare out-of-window corners here: zero it first). workspace: g3_warp_windows_workspace_bytes(n, h, w) bytes, 16-byte aligned. depth may be NULL. */
size_t g3_warp_windows_workspace_bytes(int n, int h, int w);
int g3_warp_splat_resolve_f32(const float* image, const float* z, const float* flow, const float* maskz, const void* group_max,
                              float* accum, void* workspace, float* frame, float* mask, float* depth, int n, int h, int w,
                              int group_size, void* stream);
/* One call per batch of render items (Cache3D_Base.render_cache, cache_3d.py:151-236): n items = (target camera, cached source view) pairs rendered
 * from n_src source views WITHOUT replicating the sources - item i reads source src_index[i] (device int32 [n]): points_src [n_src][h][w][3],
 * image_src [n_src][3][h][w], mask_src [n_src][h][w] or NULL, boundary_src u8 [n_src][h][w] or NULL (NULL = no foreground masking; else Kinv
 * [n][9] and depth are required). w2c [n][16], K [n][9] are the TARGET cameras. Runs project -> window splat -> gather / resolve -> (mesh
 * occlusion) as g3_warp_project_f32 + g3_warp_splat_resolve_f32 + g3_mesh_occlusion_f32 would on expanded inputs (same arithmetic, same
 * group_size pairing). Outputs frame [n][3][h][w], mask [n][h][w], depth [n][h][w] or NULL, flow_out [n][2][h][w] or NULL.
 * workspace: g3_render_workspace_bytes(n, h, w, group_size) bytes, 256-byte aligned, prepared ONCE by g3_render_workspace_init for exactly this
 * (n, h, w, group_size) and then reused call after call: the dense out-of-window accumulator inside it is kept all-zero by the kernels
 * themselves (items whose accumulator a launch touched are stamped, and only those are read back and cleared), so no per-call clearing pass;
 * likewise the nearest-hit buffer of the occlusion pass is left at +inf by the resolve pass (per-tile stamps). With foreground masking the
 * occlusion is applied inside the resolve pass (the arithmetic of g3_mesh_occlusion_f32's apply step on the values being written), and the
 * marking / rasterising pass runs on a stream the library owns, forked from and joined back into `stream` with events inside this call (all
 * inputs must be ready on `stream` at call time, as for every entry point; option render_overlap = 0 keeps everything on `stream`). Limits:
 * h < 32766, w < 65534 (packed texel ids); calls that share a workspace must be ordered (same stream). */
size_t g3_render_workspace_bytes(int n, int h, int w, int group_size);
int g3_render_workspace_init(void* workspace, int n, int h, int w, int group_size, void* stream);
int g3_render_items_f32(const float* points_src, const float* image_src, const float* mask_src, const uint8_t* boundary_src,
                        const int* src_index, const float* w2c, const float* K, const float* Kinv, void* workspace, float* frame, float* mask,
                        float* depth, float* flow_out, int n, int n_src, int h, int w, int group_size, void* stream);
int g3_mesh_occlusion_f32(const float* cam_points, const uint8_t* boundary_mask, const float* K, const float* Kinv,
                          float* pts_ds, uint8_t* mask_ds, void* tmin, float* frame, float* mask, float* depth, int n, int h,
                          int w, int factor, void* stream);
/* cache construction: unproject_points(is_depth=True) (forward_warp_utils_pytorch.py:410-460; c2w = inverse(w2c), Kinv =
 * inverse(K), both [n][..] f32 computed by the host) and reliable_depth_mask_range_batch (:338-353) -> u8 mask. */
int g3_unproject_points_f32(const float* depth, const float* c2w, const float* Kinv, float* points, int n, int h, int w,
                            void* stream);
int g3_reliable_depth_mask_f32(const float* depth, uint8_t* out, int n, int h, int w, int window, float ratio_thresh,
                               float eps, void* stream);
/* autoregressive cache update: camera_utils.align_depth (camera_utils.py:225-345) for one [H][W] fp32 depth map, enqueued on
 * `stream` without host synchronisation. Rigid part: 10 %/90 % quantile outlier rejection (torch.quantile 'linear' rank
 * arithmetic) + affine fit in inverse depth. non_rigid != 0 adds `num_iters` Adam steps (lr, betas 0.9/0.999, eps 1e-8) on a
 * per-pixel scale map with the reference's data + lambda_arap * 3x3-ARAP loss; T_host = first three rows of inv(c2w) (12
 * floats - unproject_points inverts what align_depth passes in its w2c slot) and Kinv_host = inverse(K) (9 floats) are HOST
 * pointers, read at call time. target_mask: u8 [H][W] or NULL (= all pixels). out_depth may not alias the inputs.
 * workspace: device memory, 256-byte aligned, at least g3_align_depth_workspace_bytes(H, W) bytes. */
size_t g3_align_depth_workspace_bytes(int H, int W);
int g3_align_depth_f32(const float* source_depth, const float* target_depth, const uint8_t* target_mask, const float* Kinv_host,
                       const float* T_host, int non_rigid, int num_iters, float lambda_arap, float lr, float* out_depth,
                       void* workspace, size_t workspace_bytes, int H, int W, void* stream);

/* ---- causal video tokenizer (Cosmos-Tokenize1-CV8x8x8; tokenizer/modules/{layers3d,patching,utils}.py) ---------------
 * Activations are channels-last bf16 [T][H][W][C] for one batch item.
 * conv3d_cl: CausalConv3d (layers3d.py:50-97) as an implicit GEMM on the MFMA kernel. Output position (to,yo,xo), tap
 *   (dt,dy,dx) reads ti = max(to*st+ot+dt, 0) (first frame replicated in front), yi = yo*sh+oh+dy, xi = xo*sw+ow+dx
 *   (outside the frame = zero padding). w is tap-major [kt*kh*kw][N][ldw]; bias [N] or NULL; residual (needs bias) is
 *   added after the bias: out = conv(in) + bias + residual  (res-block skip, hybrid up/down-sample "+ x").
 * groupnorm_swish_cl: CausalNormalize = GroupNorm(1 group, eps) per frame (+ optional x*sigmoid(x)); stats_f64 is a
 *   [frames][2] double scratch. haar3d_(un)patch: Patcher3D/UnPatcher3D, patch_size 4; video is planar [3][T][H][W].
 * resample_cl modes: 0 avg-pool(1,2,2) on right/bottom zero pad, 1 avg-pool(2,1,1) on front-replicated input,
 *   2 repeat_interleave(2) in time minus the first frame, 3 repeat_interleave(2) in H and W (layers3d.py:135-234).
 * softmax_rows / transpose2d / temporal_attn_cl: CausalAttnBlock and CausalTemporalAttnBlock (layers3d.py:345-427). */
int g3_conv3d_cl_bf16(const void* in, int64_t ld_in, const void* w, int64_t ldw, const void* bias, const void* residual,
                      int64_t ldr, void* out, int64_t ld_out, int K, int N, int Ti, int Hi, int Wi, int To, int Ho, int Wo,
                      int kt, int kh, int kw, int st, int sh, int sw, int ot, int oh, int ow, void* stream);
/* g3_conv3d_cl_gnstats_bf16: the same convolution, which additionally ADDS per output frame (gn_rows_per_frame = Ho*Wo consecutive output rows)
 * the sum and sum of squares of its stored bf16 outputs to gn_stats_f64[frame][0 / 1] (doubles, zeroed by the caller) - in the kernel's epilogue
 * where the one-wave-per-SIMD convolution kernel runs, by a statistics pass otherwise. g3_groupnorm_stats_cl_bf16 / g3_groupnorm_apply_cl_bf16
 * are the two halves of g3_groupnorm_swish_cl_bf16: statistics (added to a zeroed buffer), and the normalisation given finished statistics
 * - CausalNormalize of a tensor whose producing convolution already delivered them reads the tensor once instead of twice. */
int g3_conv3d_cl_gnstats_bf16(const void* in, int64_t ld_in, const void* w, int64_t ldw, const void* bias, const void* residual,
                              int64_t ldr, void* out, int64_t ld_out, int K, int N, int Ti, int Hi, int Wi, int To, int Ho, int Wo,
                              int kt, int kh, int kw, int st, int sh, int sw, int ot, int oh, int ow, void* gn_stats_f64,
                              int gn_rows_per_frame, void* stream);
int g3_groupnorm_stats_cl_bf16(const void* x, int64_t ld, void* stats_f64, int frames, int rows_per_frame, int C, void* stream);
int g3_groupnorm_apply_cl_bf16(const void* x, int64_t ld, const void* gamma, const void* beta, const void* stats_f64, void* out,
                               int64_t ldo, int frames, int rows_per_frame, int C, float eps, int swish, void* stream);
int g3_groupnorm_swish_cl_bf16(const void* x, int64_t ld, const void* gamma, const void* beta, void* stats_f64, void* out,
                               int64_t ldo, int frames, int rows_per_frame, int C, float eps, int swish, void* stream);
int g3_haar3d_patch_bf16(const void* video, void* out, int T, int H, int W, void* stream);
int g3_haar3d_unpatch_bf16(const void* coef, int64_t ld, void* video, int Tp, int Hp, int Wp, void* stream);
int g3_resample_cl_bf16(const void* in, void* out, int Ti, int Hi, int Wi, int C, int mode, void* stream);
int g3_softmax_rows_bf16(void* x, int64_t ld, int rows, int n, float scale, void* stream);
int g3_transpose2d_bf16(const void* in, int64_t ld_in, void* out, int64_t ld_out, int R, int C, void* stream);
int g3_temporal_attn_cl_bf16(const void* q, const void* k, const void* v, void* o, int T, int HW, int C, float scale,
                             void* stream);
/* CausalAttnBlock's attention core (tokenizer/modules/layers3d.py:362-377: per frame w = softmax(q^T k * C^-0.5), h = v w^T) for C = 512 in ONE
 * flash-style pass: q, k, o [frames][hw][512] bf16 (channels-last pixels), vt = V^T [frames][512][ld_vt] (g3_transpose2d_bf16 per frame;
 * vt_frame_stride in elements), hw % 64 == 0. fp32 softmax statistics and accumulation, P rounded to bf16 before P.V like the bf16 reference.
 * Replaces the scores-GEMM + g3_softmax_rows_bf16 + P.V-GEMM sequence of earlier rounds (csrc/attention_d512.hip). */
int g3_spatial_attn_d512_bf16(const void* q, const void* k, const void* vt, int64_t ld_vt, int64_t vt_frame_stride, void* o, int frames,
                              int hw, float softmax_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEN3C_HIP_H */
