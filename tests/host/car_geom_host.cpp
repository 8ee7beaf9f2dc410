// Host shim for CPU-side unit tests of cross_attention_renderer_amd/csrc/car_geom.h (test infrastructure).
// Compiled by g++ (tests/test_geom_host.py) with -ffp-contract=off; it loops the very same inline
// functions the HIP kernels call, so the geometry can be checked against the oracle without a GPU.
#include "car_geom.h"

extern "C" {

void host_pose_setup(const float* c2w_ctx, const float* c2w_q, const float* K_ctx, const float* K_q,
                     int b, int V, int H, CarPose* out) {
    for (int i = 0; i < b; ++i)
        car_pose_setup(c2w_ctx + 16 * V * i, c2w_q + 16 * i, K_ctx + 16 * V * i, K_q + 16 * i, V, H, out + V * i);
}

void host_ray_setup(const CarPose* poses, const float* uv, int b, int V, int R, CarRay* out) {
    for (int n = 0; n < b * V; ++n)
        for (int r = 0; r < R; ++r) {
            const float* p = uv + 2 * ((n / V) * R + r);
            car_ray_setup(poses[n], p[0], p[1], out + (size_t)n * R + r);
        }
}

void host_sample_setup(const CarPose* poses, const CarRay* rays, const float* interval, int b, int V, int R,
                       int P, int H, int W, CarSample* out) {
    for (int n = 0; n < b * V; ++n)
        for (int r = 0; r < R; ++r) {
            const CarRay& ray = rays[(size_t)n * R + r];
            for (int p = 0; p < P; ++p) {
                CarSample* S = out + ((size_t)n * R + r) * P + p;
                for (int i = 0; i < 2; ++i) S->grid[i] = ray.start[i] + (ray.end[i] - ray.start[i]) * interval[p];
                car_sample_setup(poses[n], poses + (n / V) * V, ray, V, H, W, S);
            }
        }
}

void host_bilinear_taps(const float* grid, int n, int W, int H, int mode, int* idx, float* w) {
    for (int i = 0; i < n; ++i) car_bilinear_taps(grid[2 * i], grid[2 * i + 1], W, H, mode, idx + 4 * i, w + 4 * i);
}

int host_sizeof_pose() { return (int)sizeof(CarPose); }
int host_sizeof_ray() { return (int)sizeof(CarRay); }
int host_sizeof_sample() { return (int)sizeof(CarSample); }
}
