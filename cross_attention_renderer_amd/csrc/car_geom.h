// car_geom.h — per-ray / per-sample epipolar geometry, shared by every HIP kernel on the render path.
//
// All functions are `__host__ __device__` inline so that (a) the stand-alone stage kernels and the fused
// kernels execute literally the same arithmetic and (b) the same code can be compiled by g++ into a tiny
// host shim for CPU-side unit tests (tests/host/).  The translation units that include this header are
// compiled with -ffp-contract=off: every fused multiply-add below is written explicitly (fmaf / fma) at
// exactly the places where the reference's CPU kernels fuse (small MKL bmm chains, torch.cross,
// vector_norm), so that the fp32 quantities feeding the fp64 Pluecker intersection are reproduced to the
// last bit wherever IEEE arithmetic allows it.
//
// Reference behaviour restated here (yilundu/cross_attention_renderer):
//   query rays           geometry.py:236-245, 353-371, 409-433   (plucker_embedding / lift / get_ray_directions)
//   segment clipping     epipolar.py:23-43, 74-162, 175-253       (project_rays and helpers)
//   sample positions     models.py:240-275
//   closest point        geometry.py:98-162                        (get_3d_point_epipolar / get_intersection, fp64)
//   cross-view exchange  models.py:30-39, 285-331; geometry.py:374-393; utils/util.py:16-19
//   geometric query      models.py:494-528; geometry.py:313-324
//   bilinear taps        torch grid_sample(bilinear, align_corners=False, border|zeros) as called at models.py:278, 317
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CAR_HD __host__ __device__ __forceinline__
#else
#define CAR_HD inline
#endif

#define CAR_MAX_VIEWS 3

// One record per (scene b, context view v).  All matrices are the top three rows of a 4x4, row-major.
struct CarPose {
    float q_rel[12];               // query camera -> context frame v        inv(c2w_ctx[v]) @ c2w_qry   (models.py:208)
    float c_rel[12];               // context v -> its own frame (~identity) inv(c2w_ctx[v]) @ c2w_ctx[v] (models.py:207)
    float T[CAR_MAX_VIEWS][12];    // frame v -> frame s                     inv(c2w_ctx[s]) @ c2w_ctx[v] (models.py:285-286)
    float kc[4];                   // fx, fy, cx, cy of context view v (pixel units)
    float k01[9];                  // K_ctx[v][:3,:3] with rows 0,1 divided by H (models.py:226-228)
    float kq[4];                   // fx, fy, cx, cy of the query camera
    float inv_q[12];               // world -> query camera                  inv(c2w_qry)                (geometry.py:404)
    float pad[7];
};
static_assert(sizeof(CarPose) == 96 * sizeof(float), "CarPose must be 96 floats");

// One record per (scene-view n, ray r).
struct CarRay {
    float d[3];        // unit direction of the query ray in context frame v
    float m[3];        // moment o x d
    float start[2];    // epipolar segment start in grid_sample coordinates [-1,1], NaN/Inf scrubbed to 0
    float end[2];      // segment end
    float overlaps;    // 1.0 if the segment overlaps the image (epipolar.py:251) / some sample inside (geometry.py:185)
    float pad;
};
static_assert(sizeof(CarRay) == 12 * sizeof(float), "CarRay must be 12 floats");

#define CAR_G_DIM 16   // channels of the geometric query `local_coords` (models.py:528)

// ----------------------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------------------
CAR_HD bool car_finite(float x) { return (x - x) == 0.0f; }
CAR_HD bool car_finite_d(double x) { return (x - x) == 0.0; }
CAR_HD float car_scrub(float x, float v) { return car_finite(x) ? x : v; }

// torch.cross component pattern on the reference's CPU build: fma(a_j, b_k, -round(a_k * b_j))
CAR_HD void car_cross_f(const float* a, const float* b, float* c) {
    c[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    c[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    c[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
CAR_HD void car_cross_d(const double* a, const double* b, double* c) {
    c[0] = fma(a[1], b[2], -(a[2] * b[1]));
    c[1] = fma(a[2], b[0], -(a[0] * b[2]));
    c[2] = fma(a[0], b[1], -(a[1] * b[0]));
}
// vector_norm pattern: sqrt(fma(z,z,fma(y,y,x*x)))
CAR_HD float car_norm3_f(const float* v) { return sqrtf(fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0]))); }
CAR_HD double car_norm3_d(const double* v) { return sqrt(fma(v[2], v[2], fma(v[1], v[1], v[0] * v[0]))); }
// F.normalize: v / max(||v||, 1e-12)
CAR_HD void car_normalize3(float* v) {
    float n = car_norm3_f(v);
    n = n > 1e-12f ? n : 1e-12f;
    v[0] = v[0] / n; v[1] = v[1] / n; v[2] = v[2] / n;
}
// [x,y,z,1] through a 3x4, as an FMA chain in k order (MKL bmm with K=4): used by the ray lift (geometry.py:417)
CAR_HD void car_affine_fma(const float* M, float x, float y, float z, float* out) {
    for (int i = 0; i < 3; ++i) {
        float acc = x * M[4 * i + 0];
        acc = fmaf(y, M[4 * i + 1], acc);
        acc = fmaf(z, M[4 * i + 2], acc);
        out[i] = fmaf(1.0f, M[4 * i + 3], acc);
    }
}
// the same product as an elementwise multiply + left-to-right sum (encode_relative_point, models.py:36)
CAR_HD void car_affine_seq(const float* M, const float* p, float* out) {
    for (int i = 0; i < 3; ++i)
        out[i] = ((p[0] * M[4 * i + 0] + p[1] * M[4 * i + 1]) + p[2] * M[4 * i + 2]) + M[4 * i + 3];
}

// ----------------------------------------------------------------------------------------------------
// pose algebra (a3).  The reference uses torch.inverse (LAPACK sgetrf/sgetri) + matmul in fp32; here the
// inverse is a partial-pivot Gauss-Jordan in fp64 rounded once to fp32, the products are fp32 FMA chains.
// ----------------------------------------------------------------------------------------------------
CAR_HD void car_inverse4(const float* A, float* Ainv) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = (double)A[4 * i + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        double best = fabs(a[c][c]);
        for (int r = c + 1; r < 4; ++r) { double v = fabs(a[r][c]); if (v > best) { best = v; piv = r; } }
        if (piv != c) for (int j = 0; j < 8; ++j) { double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
        double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            double f = a[r][c];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Ainv[4 * i + j] = (float)a[i][4 + j];
}
CAR_HD void car_matmul4_top3(const float* A, const float* B, float* C12) {   // rows 0..2 of A@B, FMA chain over k
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = A[4 * i + 0] * B[0 + j];
            acc = fmaf(A[4 * i + 1], B[4 + j], acc);
            acc = fmaf(A[4 * i + 2], B[8 + j], acc);
            C12[4 * i + j] = fmaf(A[4 * i + 3], B[12 + j], acc);
        }
}
// c2w_ctx: [V][16] of scene b; c2w_q, K_q: [16]; K_ctx: [V][16]; out: pose[v] for v < V
CAR_HD void car_pose_setup(const float* c2w_ctx, const float* c2w_q, const float* K_ctx, const float* K_q,
                           int V, int H, CarPose* out) {
    float inv_ctx[CAR_MAX_VIEWS][16];
    float inv_q[16];
    for (int v = 0; v < V; ++v) car_inverse4(c2w_ctx + 16 * v, inv_ctx[v]);
    car_inverse4(c2w_q, inv_q);
    for (int v = 0; v < V; ++v) {
        CarPose& P = out[v];
        car_matmul4_top3(inv_ctx[v], c2w_q, P.q_rel);
        car_matmul4_top3(inv_ctx[v], c2w_ctx + 16 * v, P.c_rel);
        for (int s = 0; s < CAR_MAX_VIEWS; ++s) {
            if (s < V) car_matmul4_top3(inv_ctx[s], c2w_ctx + 16 * v, P.T[s]);
            else for (int j = 0; j < 12; ++j) P.T[s][j] = 0.0f;
        }
        const float* K = K_ctx + 16 * v;
        P.kc[0] = K[0]; P.kc[1] = K[5]; P.kc[2] = K[2]; P.kc[3] = K[6];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) P.k01[3 * i + j] = (i < 2) ? K[4 * i + j] / (float)H : K[4 * i + j];
        P.kq[0] = K_q[0]; P.kq[1] = K_q[5]; P.kq[2] = K_q[2]; P.kq[3] = K_q[6];
        for (int j = 0; j < 12; ++j) P.inv_q[j] = inv_q[j];
        for (int j = 0; j < 7; ++j) P.pad[j] = 0.0f;
    }
}

// ----------------------------------------------------------------------------------------------------
// a4: a pixel's ray as a Pluecker line in the frame given by the 3x4 `M` (geometry.py:236-245)
// ----------------------------------------------------------------------------------------------------
CAR_HD void car_pixel_ray(const float* M, const float* k4, float u, float v, float* d, float* m) {
    const float x = (u - k4[2]) / k4[0];
    const float y = (v - k4[3]) / k4[1];
    float w[3];
    car_affine_fma(M, x, y, 1.0f, w);
    const float o[3] = {M[3], M[7], M[11]};
    d[0] = w[0] - o[0]; d[1] = w[1] - o[1]; d[2] = w[2] - o[2];
    car_normalize3(d);
    car_cross_f(o, d, m);
}

// ----------------------------------------------------------------------------------------------------
// a5: clip the projection of the ray (o, d) to the unit image square (epipolar.py:175-253)
// ----------------------------------------------------------------------------------------------------
CAR_HD bool car_in_bounds01(float x, float y) {
    const float e = 1e-6f;
    return (x >= -e) && (y >= -e) && (x <= 1.0f + e) && (y <= 1.0f + e);
}
struct CarHit { float t, x, y; bool valid; };

CAR_HD CarHit car_frame_hit(const float* K, const float* o, const float* d, int dim, float value) {
    const int od = 1 - dim;
    const float fs = K[3 * dim + dim], fo = K[3 * od + od], cs = K[3 * dim + 2], co = K[3 * od + 2];
    const float o_s = o[dim], o_o = o[od], o_z = o[2], d_s = d[dim], d_o = d[od], d_z = d[2];
    const float c = (value - cs) / fs;
    CarHit h;
    h.t = (c * o_z - o_s) / (d_s - c * d_z);
    const float num = fo * (o_o * (c * d_z - d_s) + d_o * (o_s - c * o_z));
    const float den = d_z * o_s - d_s * o_z;
    const float other = co + num / den;
    h.x = dim == 0 ? value : other;
    h.y = dim == 0 ? other : value;
    const float z = o_z + h.t * d_z;
    h.valid = car_in_bounds01(h.x, h.y) && (z > -1e-6f);
    return h;
}
CAR_HD void car_pinhole01(const float* K, const float* p, float* xy) {      // epipolar.project
    const float s = p[2] + 1e-8f;
    const float q0 = p[0] / s, q1 = p[1] / s, q2 = p[2] / s;
    xy[0] = fmaf(K[2], q2, fmaf(K[1], q1, K[0] * q0));
    xy[1] = fmaf(K[5], q2, fmaf(K[4], q1, K[3] * q0));
}
// returns overlaps_image; xy_min / xy_max in the [0,1] image
CAR_HD bool car_project_ray(const float* K01, const float* o, const float* d, float* xy_min, float* xy_max) {
    const float INF = INFINITY;
    CarHit hits[4] = {car_frame_hit(K01, o, d, 0, 0.0f), car_frame_hit(K01, o, d, 0, 1.0f),
                      car_frame_hit(K01, o, d, 1, 0.0f), car_frame_hit(K01, o, d, 1, 1.0f)};
    // min over t with invalid -> +inf, max with invalid -> -inf; first index wins ties (torch.min/max on CPU)
    int imin = 0, imax = 0;
    float tmin = hits[0].valid ? hits[0].t : INF, tmax = hits[0].valid ? hits[0].t : -INF;
    for (int i = 1; i < 4; ++i) {
        const float a = hits[i].valid ? hits[i].t : INF;
        const float b = hits[i].valid ? hits[i].t : -INF;
        if (a < tmin) { tmin = a; imin = i; }
        if (b > tmax) { tmax = b; imax = i; }
    }
    const bool depth_zero = o[2] < 1e-6f;
    const bool at_camera = car_norm3_f(o) < 1e-6f;
    const float* p0 = at_camera ? d : o;
    float xy0[2], xyi[2];
    car_pinhole01(K01, p0, xy0);
    bool ok0 = car_in_bounds01(xy0[0], xy0[1]) && (p0[2] > -1e-6f);
    if (depth_zero && !at_camera) ok0 = false;
    car_pinhole01(K01, d, xyi);
    const bool oki = car_in_bounds01(xyi[0], xyi[1]) && (d[2] > -1e-6f);
    xy_min[0] = ok0 ? xy0[0] : hits[imin].x;  xy_min[1] = ok0 ? xy0[1] : hits[imin].y;
    xy_max[0] = oki ? xyi[0] : hits[imax].x;  xy_max[1] = oki ? xyi[1] : hits[imax].y;
    return (ok0 ? true : hits[imin].valid) && (oki ? true : hits[imax].valid);
}

// full per-ray record for the default (epipolar-segment) sampling mode
CAR_HD void car_ray_setup(const CarPose& P, float u, float v, CarRay* out) {
    car_pixel_ray(P.q_rel, P.kq, u, v, out->d, out->m);
    const float o[3] = {P.q_rel[3], P.q_rel[7], P.q_rel[11]};
    float a[2], b[2];
    const bool ov = car_project_ray(P.k01, o, out->d, a, b);
    for (int i = 0; i < 2; ++i) {
        out->start[i] = car_scrub((a[i] - 0.5f) * 2.0f, 0.0f);
        out->end[i] = car_scrub((b[i] - 0.5f) * 2.0f, 0.0f);
    }
    out->overlaps = ov ? 1.0f : 0.0f;
    out->pad = 0.0f;
}

// geometry.project + normalize_for_grid_sample: camera-frame point -> grid coordinate of an HxW image
CAR_HD void car_project_grid(const float* k4, const float* p, int H, int W, float* g) {
    const float zz = p[2] + 1e-12f;
    float x = k4[0] * p[0] / zz + k4[2];
    float y = k4[1] * p[1] / zz + k4[3];
    x = car_scrub(x, 1e10f);
    y = car_scrub(y, 1e10f);
    g[0] = x / (float)(W - 1) * 2.0f - 1.0f;
    g[1] = y / (float)(H - 1) * 2.0f - 1.0f;
}

// ----------------------------------------------------------------------------------------------------
// a6/a8/a9/a13: everything geometric about one sample of one ray
// ----------------------------------------------------------------------------------------------------
struct CarSample {
    float grid[2];                       // pixel_val: where this sample reads its own view (grid coords)
    float pt[3];                         // closest point on the query ray, context frame v, fp32
    float g[CAR_G_DIM];                  // local_coords (models.py:528)
    float pt_in[CAR_MAX_VIEWS][3];       // nan_to_num(T[s] pt): the point in every context frame s
    float grid_in[CAR_MAX_VIEWS][2];     // where the point lands in view s (project + normalize_for_grid_sample)
};

// `grid` (pixel_val) is given; fills the rest.  n_view is the number of context views.
CAR_HD void car_sample_setup(const CarPose& P, const CarPose* poses_of_scene, const CarRay& ray, int n_view,
                             int H, int W, CarSample* S) {
    // pixel coordinate with the align_corners=True convention (geometry.py:100-101)
    const float px = (S->grid[0] + 1.0f) / 2.0f * (float)(W - 1);
    const float py = (S->grid[1] + 1.0f) / 2.0f * (float)(H - 1);
    // the sample's pixel ray in the context frame (plucker_embedding with c_rel ~ I)
    float l2f[3], m2f[3];
    car_pixel_ray(P.c_rel, P.kc, px, py, l2f, m2f);
    // closest point on the query line, fp64 (geometry.py:132-151)
    const double l1[3] = {ray.d[0], ray.d[1], ray.d[2]}, m1[3] = {ray.m[0], ray.m[1], ray.m[2]};
    const double l2[3] = {l2f[0], l2f[1], l2f[2]}, m2[3] = {m2f[0], m2f[1], m2f[2]};
    double n[3], l2xn[3], first[3];
    car_cross_d(l1, l2, n);
    car_cross_d(l2, n, l2xn);
    car_cross_d(m1, l2xn, first);
    const double dotm = (m2[0] * n[0] + m2[1] * n[1]) + m2[2] * n[2];
    const double nn = car_norm3_d(n);
    const double den = nn * nn + 1e-12;
    for (int i = 0; i < 3; ++i) {
        const double p1 = (-first[i] + dotm * l1[i]) / den;
        S->pt[i] = car_finite_d(p1) ? (float)p1 : 0.0f;
    }
    // the point in every context frame, and where it lands there
    for (int s = 0; s < n_view; ++s) {
        float q[3];
        car_affine_seq(P.T[s], S->pt, q);
        car_project_grid(poses_of_scene[s].kc, q, H, W, S->grid_in[s]);
        for (int i = 0; i < 3; ++i) {                                      // torch.nan_to_num(x, 0)
            float t = q[i];
            if (t != t) t = 0.0f;
            else if (t == INFINITY) t = 3.4028234663852886e38f;
            else if (t == -INFINITY) t = -3.4028234663852886e38f;
            S->pt_in[s][i] = t;
        }
    }
    // geometric query: [cam_ray(3), 0(3), ray_dir(3), tanh(depth/{1,10,100,1000})(4), o_q(3)]
    float cr[3] = {(px - P.kc[2]) / P.kc[0], (py - P.kc[3]) / P.kc[1], 1.0f};
    car_normalize3(cr);
    const float o[3] = {P.q_rel[3], P.q_rel[7], P.q_rel[11]};
    const float dv[3] = {S->pt[0] - o[0], S->pt[1] - o[1], S->pt[2] - o[2]};
    const float depth = car_scrub(car_norm3_f(dv), 1000000.0f);
    float* g = S->g;
    g[0] = cr[0]; g[1] = cr[1]; g[2] = cr[2];
    g[3] = 0.0f; g[4] = 0.0f; g[5] = 0.0f;
    g[6] = ray.d[0]; g[7] = ray.d[1]; g[8] = ray.d[2];
    g[9] = tanhf(depth); g[10] = tanhf(depth / 10.0f); g[11] = tanhf(depth / 100.0f); g[12] = tanhf(depth / 1000.0f);
    g[13] = o[0]; g[14] = o[1]; g[15] = o[2];
}

// ----------------------------------------------------------------------------------------------------
// a7/a10: bilinear taps of grid_sample(align_corners=False).  mode 0 = border, 1 = zeros.
// Returns the four (clamped, always addressable) texel indices y*W+x and their weights; a tap that falls
// outside the map has weight 0.  Coordinates may be ~1e10 (geometry.project scrubbing): the float->int
// conversion is guarded.
// ----------------------------------------------------------------------------------------------------
// ... from texel coordinates (ix, iy): texel i has its centre at i
CAR_HD void car_bilinear_taps_px(float ix, float iy, int W, int H, int mode, int* idx, float* w) {
    if (mode == 0) {
        ix = fminf(fmaxf(ix, 0.0f), (float)(W - 1));
        iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));
    }
    // NaN (cannot happen after scrubbing, but be safe) and huge values -> far outside -> all taps masked
    if (!(ix > -4.0f)) ix = -4.0f;
    if (!(iy > -4.0f)) iy = -4.0f;
    if (ix > (float)W + 4.0f) ix = (float)W + 4.0f;
    if (iy > (float)H + 4.0f) iy = (float)H + 4.0f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
    const float wx0 = x1f - ix, wx1 = ix - x0f, wy0 = y1f - iy, wy1 = iy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int cx0 = x0 < 0 ? 0 : (x0 >= W ? W - 1 : x0), cx1 = x1 < 0 ? 0 : (x1 >= W ? W - 1 : x1);
    const int cy0 = y0 < 0 ? 0 : (y0 >= H ? H - 1 : y0), cy1 = y1 < 0 ? 0 : (y1 >= H ? H - 1 : y1);
    idx[0] = cy0 * W + cx0; w[0] = (vx0 && vy0) ? wx0 * wy0 : 0.0f;    // nw
    idx[1] = cy0 * W + cx1; w[1] = (vx1 && vy0) ? wx1 * wy0 : 0.0f;    // ne
    idx[2] = cy1 * W + cx0; w[2] = (vx0 && vy1) ? wx0 * wy1 : 0.0f;    // sw
    idx[3] = cy1 * W + cx1; w[3] = (vx1 && vy1) ? wx1 * wy1 : 0.0f;    // se
}
CAR_HD void car_bilinear_taps(float gx, float gy, int W, int H, int mode, int* idx, float* w) {
    car_bilinear_taps_px(((gx + 1.0f) * (float)W - 1.0f) / 2.0f, ((gy + 1.0f) * (float)H - 1.0f) / 2.0f, W, H, mode, idx, w);
}

// ----------------------------------------------------------------------------------------------------
// The merged lattice (car_project_maps).  grid_sample of a level is a tensor product of hat functions on
// that level's texel centres; with u = (x + 1) W_max - 1 (twice the finest level's texel coordinate) the centres of a
// level r times coarser sit at u = 2 r i + r - 1, all integers, and so do the kinks the two padding modes add (zeros: one
// texel beyond the edge; border: the clamp at the outer centres).  The SUM over the levels is therefore bilinear inside
// every cell of the integer lattice u in [-pad, 2 W_max - 1 + pad], pad = r_max + 1, and four taps of one map — built once
// per stereo pair and padding mode — give what twelve taps of three levels give.  Outside the lattice both sums are constant
// (zero / the border value), so the lookup clamps.  Returns the node index y0 * lw + x0 of the north-west tap, flags and the weights
// (nw, ne, sw, se).  The north-west node is never in the last column / row: a point clamped onto the far edge takes the node before
// it with weights (0, 1), so the east taps are ALWAYS one node over and the south taps one row down (flag bits 1 and 2, always set;
// kept for the callers that still read them) — the fused kernel's tap addresses are "node, +1 node, +1 row, +both" without a decode.
// Flag 4: on / beyond the outer ring.
// ----------------------------------------------------------------------------------------------------
CAR_HD void car_lattice_taps(float gx, float gy, int lw, int lh, int pad, float sx, float sy, int* node, int* flags, float* w) {
    float ux = (gx + 1.0f) * sx - 1.0f, uy = (gy + 1.0f) * sy - 1.0f;
    const float lox = -(float)pad, hix = (float)(lw - 1 - pad), loy = -(float)pad, hiy = (float)(lh - 1 - pad);
    if (!(ux > lox)) ux = lox;                                         // also NaN
    if (!(uy > loy)) uy = loy;
    if (ux > hix) ux = hix;
    if (uy > hiy) uy = hiy;
    float x0f = floorf(ux), y0f = floorf(uy);
    if (x0f > hix - 1.0f) x0f = hix - 1.0f;                            // on the far edge: the cell before it, weights (0, 1)
    if (y0f > hiy - 1.0f) y0f = hiy - 1.0f;
    const float wx0 = (x0f + 1.0f) - ux, wx1 = ux - x0f, wy0 = (y0f + 1.0f) - uy, wy1 = uy - y0f;
    const int x0 = (int)x0f + pad, y0 = (int)y0f + pad;
    *node = y0 * lw + x0;
    // 4: the point sits on or beyond the lattice's outer ring: with zeros padding every level's function — and so every node the
    // four taps touch — is exactly zero there (with border padding the ring carries the edge values: the caller ignores the bit)
    *flags = 3 | ((ux <= lox || ux >= hix || uy <= loy || uy >= hiy) ? 4 : 0);
    w[0] = wx0 * wy0; w[1] = wx1 * wy0; w[2] = wx0 * wy1; w[3] = wx1 * wy1;
}
