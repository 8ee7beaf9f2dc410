// car_geometry.hip — stage kernels for the per-ray and per-sample geometry.
// One thread per ray / per sample; all arithmetic lives in car_geom.h.  Compiled with -ffp-contract=off.
#include "car_common.h"
#include "car_geom.h"

namespace {

__global__ void pose_kernel(const float* c2w_ctx, const float* c2w_q, const float* K_ctx, const float* K_q,
                            int b, int V, int H, CarPose* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b) return;
    car_pose_setup(c2w_ctx + 16 * V * i, c2w_q + 16 * i, K_ctx + 16 * V * i, K_q + 16 * i, V, H, out + V * i);
}

__global__ void ray_kernel(const CarPose* __restrict__ poses, const float* __restrict__ uv, int b, int V, int R,
                           int H, int W, int P, int no_sample, const float* __restrict__ depth_steps,
                           CarRay* __restrict__ rays, float* __restrict__ coords9, float* __restrict__ phi_x,
                           int ld_phi) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)b * V * R) return;
    const int n = (int)(i / R), r = (int)(i % R);
    const int sc = n / V, v = n % V;
    const CarPose& Ps = poses[n];
    const float u = uv[2 * ((long)sc * R + r)], w = uv[2 * ((long)sc * R + r) + 1];
    CarRay ray;
    if (!no_sample) {
        car_ray_setup(Ps, u, w, &ray);
    } else {
        // geometry.get_epipolar_lines_volumetric: uniform depths on the query ray; valid = some sample strictly inside
        car_pixel_ray(Ps.q_rel, Ps.kq, u, w, ray.d, ray.m);
        const float o[3] = {Ps.q_rel[3], Ps.q_rel[7], Ps.q_rel[11]};
        bool any_in = false;
        float first[2] = {0, 0}, last[2] = {0, 0};
        for (int p = 0; p < P; ++p) {
            const float s = depth_steps[p];
            const float q[3] = {o[0] + s * ray.d[0], o[1] + s * ray.d[1], o[2] + s * ray.d[2]};
            float gg[2];
            car_project_grid(Ps.kc, q, H, W, gg);
            any_in = any_in || (gg[0] < 1.0f && gg[0] > -1.0f && gg[1] < 1.0f && gg[1] > -1.0f);
            if (p == 0) { first[0] = gg[0]; first[1] = gg[1]; }
            last[0] = gg[0]; last[1] = gg[1];
        }
        ray.start[0] = first[0]; ray.start[1] = first[1];
        ray.end[0] = last[0]; ray.end[1] = last[1];
        ray.overlaps = any_in ? 1.0f : 0.0f;
        ray.pad = 0.0f;
    }
    rays[i] = ray;
    const float o3[3] = {Ps.q_rel[3], Ps.q_rel[7], Ps.q_rel[11]};
    if (coords9) {
        float* c = coords9 + 9 * i;
        for (int k = 0; k < 3; ++k) { c[k] = ray.d[k]; c[3 + k] = ray.m[k]; c[6 + k] = o3[k]; }
    }
    if (phi_x) {
        float* c = phi_x + ((long)sc * R + r) * ld_phi + 9 * v;
        for (int k = 0; k < 3; ++k) { c[k] = ray.d[k]; c[3 + k] = ray.m[k]; c[6 + k] = o3[k]; }
    }
}

__global__ void sample_kernel(const CarPose* __restrict__ poses, const CarRay* __restrict__ rays,
                              const float* __restrict__ steps, int b, int V, int R, int P, int H, int W,
                              int no_sample, float* __restrict__ pixel_val, float* __restrict__ pt,
                              float* __restrict__ g, float* __restrict__ grid_in, float* __restrict__ xenc,
                              int ld_xenc, int col_xenc, float* __restrict__ pt_in) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)b * V * R * P) return;
    const int p = (int)(i % P);
    const long nr = i / P;
    const int n = (int)(nr / R);
    const CarPose& Ps = poses[n];
    const CarRay ray = rays[nr];
    CarSample S;
    if (!no_sample) {
        for (int k = 0; k < 2; ++k) S.grid[k] = ray.start[k] + (ray.end[k] - ray.start[k]) * steps[p];
    } else {
        const float s = steps[p];
        const float q[3] = {Ps.q_rel[3] + s * ray.d[0], Ps.q_rel[7] + s * ray.d[1], Ps.q_rel[11] + s * ray.d[2]};
        car_project_grid(Ps.kc, q, H, W, S.grid);
    }
    car_sample_setup(Ps, poses + (n / V) * V, ray, V, H, W, &S);
    if (pixel_val) { pixel_val[2 * i] = S.grid[0]; pixel_val[2 * i + 1] = S.grid[1]; }
    if (pt) for (int k = 0; k < 3; ++k) pt[3 * i + k] = S.pt[k];
    if (g) for (int k = 0; k < CAR_G_DIM; ++k) g[CAR_G_DIM * i + k] = S.g[k];
    if (grid_in)
        for (int s = 0; s < V; ++s) {
            grid_in[(i * V + s) * 2 + 0] = S.grid_in[s][0];
            grid_in[(i * V + s) * 2 + 1] = S.grid_in[s][1];
        }
    if (pt_in)
        for (int s = 0; s < V; ++s)
            for (int k = 0; k < 3; ++k) pt_in[(i * V + s) * 3 + k] = S.pt_in[s][k];
    if (xenc) {
        if (V == 1) {          // models.py:482-483: pt[isnan] = 0 (already scrubbed); tanh(pt/5), tanh(pt/100)
            float* x = xenc + i * (long)ld_xenc + col_xenc;
            for (int k = 0; k < 3; ++k) { x[k] = tanhf(S.pt[k] / 5.0f); x[3 + k] = tanhf(S.pt[k] / 100.0f); }
        } else {
            for (int s = 0; s < V; ++s) {
                float* x = xenc + (i * V + s) * (long)ld_xenc + col_xenc;
                for (int k = 0; k < 3; ++k) x[k] = tanhf(S.pt_in[s][k] / 5.0f);
            }
        }
    }
}

__global__ void finalize_kernel(const CarRay* __restrict__ rays, const float* __restrict__ rgb_in, int ld_in, int b,
                                int V, int R, float* __restrict__ rgb, float* __restrict__ valid) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)b * R) return;
    const int sc = (int)(i / R), r = (int)(i % R);
    bool any = false;
    for (int v = 0; v < V; ++v) any = any || (rays[((long)(sc * V + v)) * R + r].overlaps != 0.0f);
    const float m = any ? 1.0f : 0.0f;
    for (int k = 0; k < 3; ++k) rgb[3 * i + k] = rgb_in[i * ld_in + k] * m + 1.0f * (1.0f - m);
    valid[i] = m;
}

}  // namespace

extern "C" int car_pose_setup(const float* c2w_ctx, const float* c2w_q, const float* K_ctx, const float* K_q,
                              int b, int V, int H, float* poses, void* stream) {
    CAR_REQUIRE(c2w_ctx && c2w_q && K_ctx && K_q && poses, "car_pose_setup: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && V <= CAR_MAX_VIEWS && H > 0, "car_pose_setup: bad sizes b=%d V=%d H=%d", b, V, H);
    (void)hipGetLastError();
    hipLaunchKernelGGL(pose_kernel, dim3(car_div_up(b, 64)), dim3(64), 0, (hipStream_t)stream, c2w_ctx, c2w_q, K_ctx,
                       K_q, b, V, H, (CarPose*)poses);
    CAR_CHECK_LAUNCH("car_pose_setup");
    return CAR_OK;
}

extern "C" int car_ray_setup(const float* poses, const float* uv, int b, int V, int R, int H, int W, int P,
                             int no_sample, const float* depth_steps, float* rays, float* coords9, float* phi_x,
                             int ld_phi, void* stream) {
    CAR_REQUIRE(poses && uv && rays, "car_ray_setup: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && V <= CAR_MAX_VIEWS && R > 0 && H > 1 && W > 1 && P > 0, "car_ray_setup: bad sizes");
    CAR_REQUIRE(!no_sample || depth_steps, "car_ray_setup: no_sample needs depth_steps");
    CAR_REQUIRE(!phi_x || ld_phi >= 9 * V, "car_ray_setup: ld_phi (%d) < 9*V", ld_phi);
    const long n = (long)b * V * R;
    (void)hipGetLastError();
    hipLaunchKernelGGL(ray_kernel, dim3(car_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, (const CarPose*)poses,
                       uv, b, V, R, H, W, P, no_sample, depth_steps, (CarRay*)rays, coords9, phi_x, ld_phi);
    CAR_CHECK_LAUNCH("car_ray_setup");
    return CAR_OK;
}

extern "C" int car_sample_setup(const float* poses, const float* rays, const float* steps, int b, int V, int R,
                                int P, int H, int W, int no_sample, float* pixel_val, float* pt, float* g,
                                float* grid_in, float* xenc, int ld_xenc, int col_xenc, float* pt_in, void* stream) {
    CAR_REQUIRE(poses && rays && steps, "car_sample_setup: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && V <= CAR_MAX_VIEWS && R > 0 && P > 0 && H > 1 && W > 1, "car_sample_setup: bad sizes");
    CAR_REQUIRE(!xenc || (ld_xenc >= col_xenc + (V == 1 ? 6 : 3) && col_xenc >= 0), "car_sample_setup: xenc window out of row");
    const long n = (long)b * V * R * P;
    (void)hipGetLastError();
    hipLaunchKernelGGL(sample_kernel, dim3(car_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const CarPose*)poses, (const CarRay*)rays, steps, b, V, R, P, H, W, no_sample, pixel_val, pt, g,
                       grid_in, xenc, ld_xenc, col_xenc, pt_in);
    CAR_CHECK_LAUNCH("car_sample_setup");
    return CAR_OK;
}

namespace {
__global__ void project_points_kernel(const CarPose* __restrict__ poses, const float* __restrict__ pts, long npts, int V,
                                      int view, int H, int W, long total, float* __restrict__ grid) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int sc = (int)(i / npts);
    const float q[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    car_project_grid(poses[sc * V + view].kc, q, H, W, grid + 2 * i);
}
}  // namespace

extern "C" int car_project_points(const float* poses, const float* pts, int n_scenes, long npts, int V, int view, int H,
                                  int W, float* grid, void* stream) {
    CAR_REQUIRE(poses && pts && grid, "car_project_points: null pointer");
    CAR_REQUIRE(n_scenes > 0 && npts > 0 && V > 0 && view >= 0 && view < V && H > 1 && W > 1, "car_project_points: bad sizes");
    const long total = (long)n_scenes * npts;
    (void)hipGetLastError();
    hipLaunchKernelGGL(project_points_kernel, dim3(car_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const CarPose*)poses, pts, npts, V, view, H, W, total, grid);
    CAR_CHECK_LAUNCH("car_project_points");
    return CAR_OK;
}

namespace {
// The row lists of the cross-view exchange for V > 2 context views (models.py:345-475 as engine._encode_three_views restates it): sample j of
// context c carries V rows — component 0: its own features (map c, border padding, its own grid point, its point in frame c); component k >= 1,
// for the other views o in ascending order: view o's features (zeros padding) where CONTEXT o's sample j — moved into frame c, projected with
// view o's intrinsics — lands, with that point's encoding.  One thread per (scene, context, sample, component).
__global__ void exchange_rows_kernel(const CarPose* __restrict__ poses, const float* __restrict__ pixel_val, const float* __restrict__ pt_in,
                                     const float* __restrict__ ptenc, int V, long pts, int H, int W, long total, int* __restrict__ row_src,
                                     float* __restrict__ row_grid, float* __restrict__ row_pe) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % V);
    const long sj = idx / V;                    // (scene, context, sample)
    const long j = sj % pts;
    const int n = (int)(sj / pts), c = n % V, sc = n / V;
    const int o = k == 0 ? c : (k - 1 < c ? k - 1 : k);
    const long so = ((long)(sc * V + o) * pts + j);                    // context o's sample j
    float g2[2];
    if (k == 0) { g2[0] = pixel_val[2 * so]; g2[1] = pixel_val[2 * so + 1]; }
    else {
        const float q[3] = {pt_in[(so * V + c) * 3], pt_in[(so * V + c) * 3 + 1], pt_in[(so * V + c) * 3 + 2]};
        car_project_grid(poses[sc * V + o].kc, q, H, W, g2);
    }
    row_src[idx] = (sc * V + o) | (k == 0 ? 0 : (1 << 30));
    row_grid[2 * idx] = g2[0]; row_grid[2 * idx + 1] = g2[1];
    const float* pe = ptenc + (so * V + c) * 4;
    row_pe[4 * idx] = pe[0]; row_pe[4 * idx + 1] = pe[1]; row_pe[4 * idx + 2] = pe[2]; row_pe[4 * idx + 3] = 0.0f;
}
}  // namespace

extern "C" int car_exchange_rows(const float* poses, const float* pixel_val, const float* pt_in, const float* ptenc, int n_scenes, int V, long pts,
                                 int H, int W, int* row_src, float* row_grid, float* row_pe, void* stream) {
    CAR_REQUIRE(poses && pixel_val && pt_in && ptenc && row_src && row_grid && row_pe, "car_exchange_rows: null pointer");
    CAR_REQUIRE(n_scenes > 0 && V > 1 && V <= CAR_MAX_VIEWS && pts > 0 && H > 1 && W > 1, "car_exchange_rows: bad sizes");
    const long total = (long)n_scenes * V * pts * V;
    (void)hipGetLastError();
    hipLaunchKernelGGL(exchange_rows_kernel, dim3(car_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, (const CarPose*)poses, pixel_val, pt_in,
                       ptenc, V, pts, H, W, total, row_src, row_grid, row_pe);
    CAR_CHECK_LAUNCH("car_exchange_rows");
    return CAR_OK;
}

extern "C" int car_finalize(const float* rays, const float* rgb_in, int ld_in, int b, int V, int R, float* rgb,
                            float* valid, void* stream) {
    CAR_REQUIRE(rays && rgb_in && rgb && valid, "car_finalize: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && R > 0 && ld_in >= 3, "car_finalize: bad sizes");
    const long n = (long)b * R;
    (void)hipGetLastError();
    hipLaunchKernelGGL(finalize_kernel, dim3(car_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, (const CarRay*)rays,
                       rgb_in, ld_in, b, V, R, rgb, valid);
    CAR_CHECK_LAUNCH("car_finalize");
    return CAR_OK;
}
