// Fused elementwise halves of one EDM-Euler denoise step of the GEN3C sampler
// (cosmos_predict1/diffusion/model/model_v2w.py:130-149, 201-259; diffusers 0.32.2 EDMEulerScheduler).
//
// The reference runs ~25 separate bf16/fp32 elementwise torch kernels per step over the [B,16,T,88,160] latent.
// Here each side of the two network calls is ONE pass: (1) build the network input, (2) CFG + conditioning-frame
// replacement + Euler update. bf16 roundings are placed exactly where the reference's dtype promotion puts them
// (comments name the torch expression each rounding belongs to); scalar coefficients are computed by the host in
// the dtypes the reference uses and arrive as floats.
#include "common.hpp"

namespace {

G3_DEVICE float rbf(float v) { return (float)f32_to_bf16(v); }  // one bf16 rounding

struct PrepArgs {
    const bf16_t* xt;        // current sample (bf16)
    const bf16_t* gt_latent; // condition.gt_latent (bf16)
    const float* noise;      // arch_invariant_rand(seed) fp32
    const float* indicator;  // [T] 0/1 per latent frame (already zeroed by the host when augment_sigma >= sigma)
    bf16_t* new_xt;          // model_v2w.py:138
    bf16_t* new_xt_scaled;   // model_v2w.py:139
    int64_t n; int T; int hw;
    float augment_sigma;     // condition_augment_sigma
    float c_in_aug;          // 1/sqrt(augment_sigma^2 + sigma_data^2)            (python floats, fp32 tensor math)
    float inv_c_in_bf16;     // divisor c_in of _reverse_precondition_input, evaluated in bf16 (sigma is a bf16 tensor)
    float c_in_step;         // scheduler.scale_model_input: 1/sqrt(sigma^2 + sigma_data^2) in fp32
};

__global__ __launch_bounds__(256) void edm_prepare_kernel(PrepArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const int t = (int)((i / a.hw) % a.T);
        const float ind = a.indicator[t];
        const float x = (float)a.xt[i];
        float v = x;
        if (ind != 0.f) {
            // augment_latent = latent + noise * augment_sigma                      (fp32)
            float al = (float)a.gt_latent[i] + a.noise[i] * a.augment_sigma;
            al = al * a.c_in_aug;            // scheduler.precondition_inputs       (fp32)
            al = al / a.inv_c_in_bf16;       // xt / c_in, c_in a bf16 0-dim tensor (fp32 result)
            v = ind * al + (1.f - ind) * x;  // model_v2w.py:246                    (fp32)
        }
        const bf16_t nx = f32_to_bf16(v);    // new_xt.to(bf16)
        a.new_xt[i] = nx;
        a.new_xt_scaled[i] = f32_to_bf16((float)nx * a.c_in_step);  // bf16 tensor * fp32 0-dim -> bf16
    }
}

struct StepArgs {
    const bf16_t* out_cond; const bf16_t* out_uncond;
    const bf16_t* new_xt; const bf16_t* gt_latent; const float* indicator;
    bf16_t* xt_next;
    int64_t n; int T; int hw;
    float guidance;
    float c_skip_bf16, c_out_bf16;  // _reverse_precondition_output coefficients, evaluated in bf16 (sigma bf16 tensor)
    float c_skip, c_out;            // scheduler.precondition_outputs coefficients (fp32 sigma)
    float sigma, sigma_next;        // fp32
};

__global__ __launch_bounds__(256) void edm_step_kernel(StepArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const int t = (int)((i / a.hw) % a.T);
        const float ind = a.indicator[t];
        const float oc = (float)a.out_cond[i], ou = (float)a.out_uncond[i];
        // net_output = cond + guidance * (cond - uncond)          three bf16 tensor ops (model_v2w.py:144)
        float no = rbf(oc + rbf(a.guidance * rbf(oc - ou)));
        const float x = (float)a.new_xt[i];
        if (ind != 0.f) {
            // latent_unscaled = (latent - c_skip * xt) / c_out     bf16 tensor ops (model_v2w.py:254-259)
            const float lu = rbf(rbf((float)a.gt_latent[i] - rbf(a.c_skip_bf16 * x)) / a.c_out_bf16);
            // new_output = indicator * latent_unscaled + (1 - indicator) * net_output   (bf16, :147)
            no = rbf(rbf(ind * lu) + rbf(rbf(1.f - ind) * no));
        }
        // EDMEulerScheduler.step: sample upcast to fp32; c_out * model_output is a bf16 product
        const float x0 = a.c_skip * x + rbf(a.c_out * no);
        const float deriv = (x - x0) / a.sigma;
        a.xt_next[i] = f32_to_bf16(x + deriv * (a.sigma_next - a.sigma));  // prev_sample.to(model_output.dtype)
    }
}

int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b > 256 * 8 ? 256 * 8 : b);
}

}  // namespace

extern "C" int g3_edm_prepare_input_bf16(const void* xt, const void* gt_latent, const float* noise,
                                         const float* indicator, void* new_xt, void* new_xt_scaled, int64_t n, int T,
                                         int hw, float augment_sigma, float c_in_aug, float c_in_bf16, float c_in_step,
                                         void* stream) {
    if (!xt || !gt_latent || !noise || !indicator || !new_xt || !new_xt_scaled)
        return g3_set_error(G3_ERR_ARG, "g3_edm_prepare_input_bf16: null operand");
    if (n <= 0 || T <= 0 || hw <= 0) return g3_set_error(G3_ERR_ARG, "g3_edm_prepare_input_bf16: bad shape");
    PrepArgs a{(const bf16_t*)xt, (const bf16_t*)gt_latent, noise, indicator, (bf16_t*)new_xt, (bf16_t*)new_xt_scaled,
               n, T, hw, augment_sigma, c_in_aug, c_in_bf16, c_in_step};
    hipLaunchKernelGGL(edm_prepare_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a);
    return g3_check_launch("g3_edm_prepare_input_bf16");
}

extern "C" int g3_edm_cfg_euler_step_bf16(const void* out_cond, const void* out_uncond, const void* new_xt,
                                          const void* gt_latent, const float* indicator, void* xt_next, int64_t n,
                                          int T, int hw, float guidance, float c_skip_bf16, float c_out_bf16,
                                          float c_skip, float c_out, float sigma, float sigma_next, void* stream) {
    if (!out_cond || !out_uncond || !new_xt || !gt_latent || !indicator || !xt_next)
        return g3_set_error(G3_ERR_ARG, "g3_edm_cfg_euler_step_bf16: null operand");
    if (n <= 0 || T <= 0 || hw <= 0 || sigma <= 0.f) return g3_set_error(G3_ERR_ARG, "g3_edm_cfg_euler_step_bf16: bad argument");
    StepArgs a{(const bf16_t*)out_cond, (const bf16_t*)out_uncond, (const bf16_t*)new_xt, (const bf16_t*)gt_latent,
               indicator, (bf16_t*)xt_next, n, T, hw, guidance, c_skip_bf16, c_out_bf16, c_skip, c_out, sigma, sigma_next};
    hipLaunchKernelGGL(edm_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a);
    return g3_check_launch("g3_edm_cfg_euler_step_bf16");
}
