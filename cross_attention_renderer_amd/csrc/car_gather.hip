// car_gather.hip — stage kernel for the bilinear gathers from the channel-last feature pyramid
// (reference models.py:278, 317: F.grid_sample(bilinear, align_corners=False)).
//
// Layout: each pyramid level is [n_maps, Hl, Wl, Cl] (NHWC), so one texel is Cl contiguous floats; a thread
// owns one float4 of channels of one output row, consecutive lanes own consecutive channel quads: the four
// tap reads and the write are 16 B per lane, coalesced across the lanes that share a row (a 256-channel
// texel is exactly 64 lanes x float4).  This stand-alone stage is HBM/L2-bound: per output row it writes
// sum(Cl)*4 bytes and reads 4 taps of the same size out of L2 / Infinity Cache.
#include "car_common.h"
#include "car_geom.h"

namespace {

struct GatherLevels {
    const float* map[CAR_MAX_LEVELS];
    int c[CAR_MAX_LEVELS], h[CAR_MAX_LEVELS], w[CAR_MAX_LEVELS];
    int q0[CAR_MAX_LEVELS + 1];   // first channel quad of each level; q0[n_levels] = quads per row
    int n_levels;
};

__global__ void __launch_bounds__(256) gather_kernel(GatherLevels L, int n_maps, const float* __restrict__ grid,
                                                     long pts, int mode, int place, int V, float* __restrict__ out,
                                                     int ld_out, int col_out) {
    const int qpr = L.q0[L.n_levels];
    const long total = (long)n_maps * pts * qpr;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % qpr);
        const long mp = idx / qpr;                       // m*pts + i
        const int m = (int)(mp / pts);
        const long i = mp % pts;
        int l = 0;
        while (l + 1 < L.n_levels && q >= L.q0[l + 1]) ++l;
        const int cq = q - L.q0[l];
        const float gx = grid[2 * mp], gy = grid[2 * mp + 1];
        int tidx[4];
        float tw[4];
        car_bilinear_taps(gx, gy, L.w[l], L.h[l], mode, tidx, tw);
        const float* base = L.map[l] + (long)m * L.h[l] * L.w[l] * L.c[l] + 4 * cq;
        const float4 a = *reinterpret_cast<const float4*>(base + (long)tidx[0] * L.c[l]);
        const float4 b4 = *reinterpret_cast<const float4*>(base + (long)tidx[1] * L.c[l]);
        const float4 c4 = *reinterpret_cast<const float4*>(base + (long)tidx[2] * L.c[l]);
        const float4 d4 = *reinterpret_cast<const float4*>(base + (long)tidx[3] * L.c[l]);
        float4 r;   // ((nw + ne) + sw) + se, products rounded individually (ATen's vectorised CPU kernel order)
        r.x = ((a.x * tw[0] + b4.x * tw[1]) + c4.x * tw[2]) + d4.x * tw[3];
        r.y = ((a.y * tw[0] + b4.y * tw[1]) + c4.y * tw[2]) + d4.y * tw[3];
        r.z = ((a.z * tw[0] + b4.z * tw[1]) + c4.z * tw[2]) + d4.z * tw[3];
        r.w = ((a.w * tw[0] + b4.w * tw[1]) + c4.w * tw[2]) + d4.w * tw[3];
        long row;
        if (place == CAR_PLACE_PLAIN) row = mp;
        else if (place == CAR_PLACE_OWN) row = mp * V + (m % V);
        else { const int sc = m / 2, s = m % 2; row = (((long)(sc * 2 + (1 - s))) * pts + i) * 2 + s; }
        *reinterpret_cast<float4*>(out + row * ld_out + col_out + 4 * q) = r;
    }
}

}  // namespace

extern "C" int car_gather_bilinear(const float* const* maps, const int* level_c, const int* level_h,
                                   const int* level_w, int n_levels, int n_maps, const float* grid, long pts, int mode,
                                   int place, int V, float* out, int ld_out, int col_out, void* stream) {
    CAR_REQUIRE(maps && level_c && level_h && level_w && grid && out, "car_gather_bilinear: null pointer");
    CAR_REQUIRE(n_levels > 0 && n_levels <= CAR_MAX_LEVELS && n_maps > 0 && pts > 0, "car_gather_bilinear: bad sizes");
    CAR_REQUIRE(mode == 0 || mode == 1, "car_gather_bilinear: mode must be 0 (border) or 1 (zeros)");
    CAR_REQUIRE(place == CAR_PLACE_PLAIN || place == CAR_PLACE_OWN || (place == CAR_PLACE_OTHER2 && V == 2 && n_maps % 2 == 0),
                "car_gather_bilinear: bad placement %d for V=%d", place, V);
    GatherLevels L;
    L.n_levels = n_levels;
    int q = 0;
    for (int l = 0; l < n_levels; ++l) {
        CAR_REQUIRE(maps[l] && level_c[l] > 0 && level_c[l] % 4 == 0 && level_h[l] > 0 && level_w[l] > 0,
                    "car_gather_bilinear: level %d needs a channel count that is a positive multiple of 4", l);
        L.map[l] = maps[l]; L.c[l] = level_c[l]; L.h[l] = level_h[l]; L.w[l] = level_w[l];
        L.q0[l] = q;
        q += level_c[l] / 4;
    }
    L.q0[n_levels] = q;
    for (int l = n_levels; l < CAR_MAX_LEVELS; ++l) { L.map[l] = nullptr; L.c[l] = L.h[l] = L.w[l] = 0; if (l > n_levels) L.q0[l] = q; }
    CAR_REQUIRE(ld_out % 4 == 0 && col_out % 4 == 0 && col_out >= 0 && col_out + 4 * q <= ld_out,
                "car_gather_bilinear: output window [%d,%d) must be float4-aligned inside a row of %d", col_out, col_out + 4 * q, ld_out);
    const long total = (long)n_maps * pts * q;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    (void)hipGetLastError();
    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L, n_maps, grid, pts, mode,
                       place, V, out, ld_out, col_out);
    CAR_CHECK_LAUNCH("car_gather_bilinear");
    return CAR_OK;
}
