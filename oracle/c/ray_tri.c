/* CPU ORACLE (test infrastructure - never linked or loaded by gen3c_amd/): brute-force ray x triangle depth.
 *
 * Plain-C restatement of the reference's NVIDIA-Warp kernel `ray_triangle_intersection_kernel`
 * (cosmos_predict1/diffusion/inference/ray_triangle_intersection_warp.py:23-105): Moller-Trumbore for EVERY (ray, triangle)
 * pair, ray origin 0, eps 1e-8, the depth of a ray = min t over all triangles with t > eps, 0 where nothing is hit.
 * Same fp32 operation order as oracle/warp_oracle.py:ray_triangle_depth (products summed left to right, no FMA: build with
 * -ffp-contract=off), which tests/test_warp_oracle_golden.py replays bit for bit against this file. It exists because the
 * numpy version needs minutes and gigabytes at 704x1280 x thousands of triangles; this one needs seconds.
 *
 *   gcc -O3 -ffp-contract=off -fopenmp -shared -fPIC -o oracle/_build/libray_tri.so oracle/c/ray_tri.c
 */
#include <math.h>
#include <stdint.h>

/* rays: [n_rays][3] unit directions; tris: [n_tris][3 vertices][3]; out: [n_rays] */
void g3o_ray_triangle_depth(const float* rays, int64_t n_rays, const float* tris, int64_t n_tris, float eps, float* out) {
#pragma omp parallel for schedule(static, 4096)
    for (int64_t r = 0; r < n_rays; ++r) {
        const float dx = rays[3 * r], dy = rays[3 * r + 1], dz = rays[3 * r + 2];
        float best = 1e10f;
        for (int64_t k = 0; k < n_tris; ++k) {
            const float* T = tris + 9 * k;
            const float e1x = T[3] - T[0], e1y = T[4] - T[1], e1z = T[5] - T[2];
            const float e2x = T[6] - T[0], e2y = T[7] - T[1], e2z = T[8] - T[2];
            /* h = d x e2 */
            const float hx = dy * e2z - dz * e2y, hy = dz * e2x - dx * e2z, hz = dx * e2y - dy * e2x;
            const float a = (e1x * hx + e1y * hy) + e1z * hz;
            if (fabsf(a) < eps) continue;
            const float f = 1.0f / a;
            const float sx = 0.0f - T[0], sy = 0.0f - T[1], sz = 0.0f - T[2];
            const float u = f * ((sx * hx + sy * hy) + sz * hz);
            if (u < 0.0f || u > 1.0f) continue;
            /* q = s x e1 */
            const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
            const float v = f * ((dx * qx + dy * qy) + dz * qz);
            if (v < 0.0f || (u + v) > 1.0f) continue;
            const float t = f * ((e2x * qx + e2y * qy) + e2z * qz);
            if (t > eps && t < best) best = t;
        }
        out[r] = best < 1e10f ? best : 0.0f;
    }
}
