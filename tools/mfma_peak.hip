// Sustained MFMA rate of the GPU under a pure v_mfma_f32_32x32x16_bf16 stream (no memory traffic in the loop), with
// zero and with random operands: the matrix pipe's clock under load is power-managed and depends on operand toggling,
// so this - not the 2.5 PFLOP/s datasheet figure - is what a perfect bf16 kernel could sustain on this board.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__global__ __launch_bounds__(512, 2) void mfma_stream(const bf16x8* __restrict__ ops, float* out, int iters) {
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = ops[(threadIdx.x + 64 * i) & 1023];
    for (int i = 0; i < 2; ++i) b[i] = ops[(threadIdx.x * 3 + 17 * i) & 1023];
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const int n = 1024 * 8;
    std::vector<unsigned short> h(n);
    bf16x8* d;
    float* o;
    hipMalloc(&d, n * 2);
    hipMalloc(&o, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;
    for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < n; ++i) {
            // 0: zeros; 1: random N(0,1)-like bf16 (sign + exponent around 1.0 + random mantissa); 2: random small-magnitude
            unsigned short v = 0;
            if (mode == 1) v = (unsigned short)(((rand() & 1) << 15) | ((0x7e + (rand() & 1)) << 7) | (rand() & 0x7f));
            if (mode == 2) v = (unsigned short)(((rand() & 1) << 15) | ((0x70 + (rand() & 7)) << 7) | (rand() & 0x7f));
            h[i] = v;
        }
        hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_stream, dim3(blocks), dim3(512), 0, 0, d, o, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 8 * iters * 32 * 2.0 * 32 * 32 * 16;
            printf("mode %d (%s) rep %d: %.3f ms  %.0f TFLOP/s\n", mode, mode == 0 ? "zeros" : mode == 1 ? "random ~1.0" : "random small", rep, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
