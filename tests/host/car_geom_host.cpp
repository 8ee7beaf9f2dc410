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

// the merged lattice as car_project_maps' merge kernel builds it (csrc/car_render.hip merge_kernel): levels [n][h][w][C], r = how
// many times coarser than the widest level; lat [2 modes][lh][lw][C]
void host_lattice_build(const float* const* g, const int* h, const int* w, const int* r, int n_levels, int C, int lh, int lw, int pad, float* lat) {
    for (int mode = 0; mode < 2; ++mode)
        for (int jy = 0; jy < lh; ++jy)
            for (int jx = 0; jx < lw; ++jx) {
                float* o = lat + (((size_t)mode * lh + jy) * lw + jx) * C;
                for (int c = 0; c < C; ++c) o[c] = 0.0f;
                for (int l = n_levels - 1; l >= 0; --l) {
                    const float r2 = (float)(2 * r[l]);
                    int t4[4];
                    float w4[4];
                    car_bilinear_taps_px((float)(jx - pad + 1 - r[l]) / r2, (float)(jy - pad + 1 - r[l]) / r2, w[l], h[l], mode, t4, w4);
                    for (int t = 0; t < 4; ++t)
                        for (int c = 0; c < C; ++c) o[c] = fmaf(w4[t], g[l][(size_t)t4[t] * C + c], o[c]);
                }
            }
}
void host_lattice_taps(const float* grid, int n, int lw, int lh, int pad, float sx, float sy, int* node, int* flags, float* w) {
    for (int i = 0; i < n; ++i) car_lattice_taps(grid[2 * i], grid[2 * i + 1], lw, lh, pad, sx, sy, node + i, flags + i, w + 4 * i);
}

int host_sizeof_pose() { return (int)sizeof(CarPose); }
int host_sizeof_ray() { return (int)sizeof(CarRay); }
int host_sizeof_sample() { return (int)sizeof(CarSample); }
}
