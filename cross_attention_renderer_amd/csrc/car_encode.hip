// car_encode.hip — first layer of the per-sample point-feature MLP, evaluated as a gather over pre-projected feature
// maps (reference models.py:278, 317, 330-341).
//
// The reference gathers the C=576 pyramid features of a sample (4 bilinear taps per level), appends tanh(pt/5) and
// applies  h = relu(W1 [f ; tanh(pt/5)] + b1)  with W1 of shape (C, C+3): a K=579 GEMM per sample and per source view,
// 51 % of the path's FLOPs.  Both the 1x1 convolution and the bilinear blend are linear, so
//     W1[:, :C] . (sum_t w_t F[pix_t])  ==  sum_t w_t (W1[:, :C] F)[pix_t] ,
// i.e. the layer can be applied ONCE PER TEXEL of the feature pyramid (G_l = W1[:, ch_l] F_l, 21.7 GFLOP per stereo
// pair, done by car_linear when the pyramid or W1 changes) instead of once per sample (1.4 TFLOP per 8192-ray
// chunk).  What is left per sample is this kernel: 4 taps x n_levels of C-wide rows of G, the 3-term point
// projection, bias and ReLU.  The result differs from the reference's order of operations only by fp32 rounding
// (~1e-7 relative); it is checked against the oracle at 1e-4 like every other stage.
//
// Layout: G_l is [n_maps, Hl, Wl, C] channel-last, one texel = C contiguous floats.  A workgroup handles 16 output
// rows; tap indices/weights are computed once per (row, level) into LDS, then every thread owns float4 channel
// quads: all reads and the write are 16 B per lane and contiguous across the lanes of a row.  The kernel is bound
// by L2 / Infinity-Cache bandwidth (12 x C x 4 bytes read per row, C x 4 written).
#include "car_common.h"
#include "car_geom.h"

namespace {

constexpr int kRows = 16;

// lat_w > 0: `map[0]` is the merged lattice of car_merge_lattice ([n_maps][2 padding modes][lat_h][lat_w][C]): one four-tap lookup of the
// row's (map, padding mode) lattice replaces the four taps per level
struct EncodeLattice { int h, w, pad; float sx, sy; };
struct EncodeLevels {
    const float* map[CAR_MAX_LEVELS];
    int h[CAR_MAX_LEVELS], w[CAR_MAX_LEVELS];
    int n_levels;
};

__global__ void __launch_bounds__(256) encode_kernel(EncodeLevels L, int Cg, const float* __restrict__ pixel_val,
                                                     const float* __restrict__ grid_in, const float* __restrict__ ptenc,
                                                     const float* __restrict__ wpt, int V, long pts, long rows,
                                                     float* __restrict__ out, int ld_out, const int* __restrict__ row_src, EncodeLattice lat) {
    __shared__ int s_idx[kRows][CAR_MAX_LEVELS][4];
    __shared__ float s_w[kRows][CAR_MAX_LEVELS][4];
    __shared__ float s_pe[kRows][4];
    const int tid = threadIdx.x;
    const long row0 = (long)blockIdx.x * kRows;

    if (tid < kRows * L.n_levels) {
        const int rl = tid / L.n_levels, l = tid % L.n_levels;
        long row = row0 + rl;
        if (row >= rows) row = rows - 1;
        float gx, gy;
        int mode, m;
        if (row_src) {                           // explicit rows (car_gather_encode_rows): map | padding mode << 30, the row's own grid point
            const int src = row_src[row];
            m = src & 0x3fffffff; mode = (src >> 30) & 1;
            gx = pixel_val[2 * row]; gy = pixel_val[2 * row + 1];
        } else {
            const long i = row / V;              // sample (n, r, p)
            const int s = (int)(row % V);        // source view of this row
            const int n = (int)(i / pts);
            const int v = n % V, sc = n / V;
            if (s == v) { gx = pixel_val[2 * i]; gy = pixel_val[2 * i + 1]; mode = 0; m = n; }
            else { gx = grid_in[(i * V + s) * 2]; gy = grid_in[(i * V + s) * 2 + 1]; mode = 1; m = sc * V + s; }
        }
        int idx[4];
        float w[4];
        if (lat.w > 0) {
            int node, flags;
            car_lattice_taps(gx, gy, lat.w, lat.h, lat.pad, lat.sx, lat.sy, &node, &flags, w);
            const bool dead = mode == 1 && (flags & 4);                // zeros padding, on or beyond the outer ring: exactly zero
            const int base = (m * 2 + mode) * lat.h * lat.w + node;
            idx[0] = base; idx[1] = base + 1; idx[2] = base + lat.w; idx[3] = base + lat.w + 1;
            for (int t = 0; t < 4; ++t) { s_idx[rl][l][t] = dead ? 0 : idx[t]; s_w[rl][l][t] = dead ? 0.0f : w[t]; }
        } else {
        car_bilinear_taps(gx, gy, L.w[l], L.h[l], mode, idx, w);
        for (int t = 0; t < 4; ++t) { s_idx[rl][l][t] = m * L.h[l] * L.w[l] + idx[t]; s_w[rl][l][t] = w[t]; }
        }
        if (l == 0) for (int k = 0; k < 4; ++k) s_pe[rl][k] = ptenc[row * 4 + k];
    }
    __syncthreads();

    const int qpr = Cg / 4;
    for (int item = tid; item < kRows * qpr; item += 256) {
        const int rl = item / qpr, q = item % qpr;
        const long row = row0 + rl;
        if (row >= rows) break;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < L.n_levels; ++l) {
            const float* base = L.map[l] + 4 * q;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 g = *reinterpret_cast<const float4*>(base + (long)s_idx[rl][l][t] * Cg);
                const float w = s_w[rl][l][t];
                acc.x = fmaf(w, g.x, acc.x); acc.y = fmaf(w, g.y, acc.y);
                acc.z = fmaf(w, g.z, acc.z); acc.w = fmaf(w, g.w, acc.w);
            }
        }
        const float p0 = s_pe[rl][0], p1 = s_pe[rl][1], p2 = s_pe[rl][2];
        const float4* wp = reinterpret_cast<const float4*>(wpt + 16 * q);      // rows 4q..4q+3 of [C][4] = (w0,w1,w2,b)
        const float4 a = wp[0], b4 = wp[1], c4 = wp[2], d4 = wp[3];
        acc.x += fmaf(a.z, p2, fmaf(a.y, p1, a.x * p0)) + a.w;
        acc.y += fmaf(b4.z, p2, fmaf(b4.y, p1, b4.x * p0)) + b4.w;
        acc.z += fmaf(c4.z, p2, fmaf(c4.y, p1, c4.x * p0)) + c4.w;
        acc.w += fmaf(d4.z, p2, fmaf(d4.y, p1, d4.x * p0)) + d4.w;
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        *reinterpret_cast<float4*>(out + row * ld_out + 4 * q) = acc;
    }
}

}  // namespace

extern "C" int car_gather_encode(const float* const* gmaps, const int* level_h, const int* level_w, int n_levels, int Cg,
                                 const float* pixel_val, const float* grid_in, const float* ptenc, const float* wpt,
                                 int n_maps, int V, long pts, float* out, int ld_out, void* stream) {
    CAR_REQUIRE(gmaps && level_h && level_w && pixel_val && grid_in && ptenc && wpt && out, "car_gather_encode: null pointer");
    CAR_REQUIRE(n_levels > 0 && n_levels <= CAR_MAX_LEVELS && Cg > 0 && Cg % 4 == 0, "car_gather_encode: bad level/channel count");
    CAR_REQUIRE(V == 2 && n_maps > 0 && n_maps % V == 0 && pts > 0, "car_gather_encode: needs V == 2 (got %d) and whole scenes", V);
    CAR_REQUIRE(ld_out >= Cg && ld_out % 4 == 0, "car_gather_encode: ld_out (%d) must be a multiple of 4 and >= C (%d)", ld_out, Cg);
    EncodeLevels L;
    L.n_levels = n_levels;
    for (int l = 0; l < CAR_MAX_LEVELS; ++l) {
        L.map[l] = l < n_levels ? gmaps[l] : nullptr;
        L.h[l] = l < n_levels ? level_h[l] : 0;
        L.w[l] = l < n_levels ? level_w[l] : 0;
        if (l < n_levels) CAR_REQUIRE(L.map[l] && L.h[l] > 0 && L.w[l] > 0 && (long)n_maps * L.h[l] * L.w[l] < 2147483647L, "car_gather_encode: bad level %d", l);
    }
    const long rows = (long)n_maps * pts * V;
    (void)hipGetLastError();
    hipLaunchKernelGGL(encode_kernel, dim3(car_div_up(rows, kRows)), dim3(256), 0, (hipStream_t)stream, L, Cg, pixel_val,
                       grid_in, ptenc, wpt, V, pts, rows, out, ld_out, (const int*)nullptr, EncodeLattice{0, 0, 0, 0.f, 0.f});
    CAR_CHECK_LAUNCH("car_gather_encode");
    return CAR_OK;
}

// The same layer for an explicit list of rows: row i gathers map (row_src[i] & 0x3fffffff) at row_grid[i] with padding mode
// (row_src[i] >> 30) & 1 (0 border, 1 zeros) and adds the point term of row_pe[i].  This is what the three-view exchange needs
// (models.py:345-475: every sample of context c carries its own features and, per other view o, view o's features where CONTEXT o's points
// land — nine different (map, grid, point) combinations per scene, listed by the host).
extern "C" int car_gather_encode_rows(const float* const* gmaps, const int* level_h, const int* level_w, int n_levels, int Cg,
                                      const int* row_src, const float* row_grid, const float* row_pe, const float* wpt, int n_maps,
                                      long rows, float* out, int ld_out, void* stream) {
    CAR_REQUIRE(gmaps && level_h && level_w && row_src && row_grid && row_pe && wpt && out, "car_gather_encode_rows: null pointer");
    CAR_REQUIRE(n_levels > 0 && n_levels <= CAR_MAX_LEVELS && Cg > 0 && Cg % 4 == 0, "car_gather_encode_rows: bad level/channel count");
    CAR_REQUIRE(n_maps > 0 && rows > 0, "car_gather_encode_rows: bad sizes");
    CAR_REQUIRE(ld_out >= Cg && ld_out % 4 == 0, "car_gather_encode_rows: ld_out (%d) must be a multiple of 4 and >= C (%d)", ld_out, Cg);
    EncodeLevels L;
    L.n_levels = n_levels;
    for (int l = 0; l < CAR_MAX_LEVELS; ++l) {
        L.map[l] = l < n_levels ? gmaps[l] : nullptr;
        L.h[l] = l < n_levels ? level_h[l] : 0;
        L.w[l] = l < n_levels ? level_w[l] : 0;
        if (l < n_levels) CAR_REQUIRE(L.map[l] && L.h[l] > 0 && L.w[l] > 0 && (long)n_maps * L.h[l] * L.w[l] < 1073741823L, "car_gather_encode_rows: bad level %d", l);
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(encode_kernel, dim3(car_div_up(rows, kRows)), dim3(256), 0, (hipStream_t)stream, L, Cg, row_grid,
                       (const float*)nullptr, row_pe, wpt, 1, rows, rows, out, ld_out, row_src, EncodeLattice{0, 0, 0, 0.f, 0.f});
    CAR_CHECK_LAUNCH("car_gather_encode_rows");
    return CAR_OK;
}

// The same rows from the merged lattice of car_merge_lattice (every level summed on the common lattice, one lattice per map and padding
// mode: DESIGN.md 4.3): four taps per row instead of four per level.
extern "C" int car_lattice_encode_rows(const float* lattice, int lat_h, int lat_w, int lat_pad, int Cg, const int* row_src, const float* row_grid,
                                       const float* row_pe, const float* wpt, int n_maps, long rows, float* out, int ld_out, void* stream) {
    CAR_REQUIRE(lattice && row_src && row_grid && row_pe && wpt && out, "car_lattice_encode_rows: null pointer");
    CAR_REQUIRE(Cg > 0 && Cg % 4 == 0 && n_maps > 0 && rows > 0, "car_lattice_encode_rows: bad sizes");
    CAR_REQUIRE(lat_pad >= 2 && lat_h > 2 * lat_pad + 1 && lat_w > 2 * lat_pad + 1 && ((lat_h - 2 * lat_pad) & 1) && ((lat_w - 2 * lat_pad) & 1),
                "car_lattice_encode_rows: bad lattice %d x %d, pad %d (car_merge_lattice)", lat_h, lat_w, lat_pad);
    CAR_REQUIRE((long)n_maps * 2 * lat_h * lat_w < 2147483647L, "car_lattice_encode_rows: too many lattice nodes");
    CAR_REQUIRE(ld_out >= Cg && ld_out % 4 == 0, "car_lattice_encode_rows: ld_out (%d) must be a multiple of 4 and >= C (%d)", ld_out, Cg);
    EncodeLevels L{};
    L.n_levels = 1;
    L.map[0] = lattice; L.h[0] = lat_h; L.w[0] = lat_w;
    const EncodeLattice lat{lat_h, lat_w, lat_pad, (float)((lat_w - 2 * lat_pad + 1) / 2), (float)((lat_h - 2 * lat_pad + 1) / 2)};
    (void)hipGetLastError();
    hipLaunchKernelGGL(encode_kernel, dim3(car_div_up(rows, kRows)), dim3(256), 0, (hipStream_t)stream, L, Cg, row_grid,
                       (const float*)nullptr, row_pe, wpt, 1, rows, rows, out, ld_out, row_src, lat);
    CAR_CHECK_LAUNCH("car_lattice_encode_rows");
    return CAR_OK;
}
