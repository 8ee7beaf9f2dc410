// car_gather.hip — stage kernel for the bilinear gathers from the channel-last feature pyramid
// (reference models.py:278, 317: F.grid_sample(bilinear, align_corners=False)).
//
// Layout: each pyramid level is [n_maps, Hl, Wl, Cl] (NHWC), so one texel is Cl contiguous floats; a thread
// owns one float4 of channels of one output row, consecutive lanes own consecutive channel quads: the four
// tap reads and the write are 16 B per lane, coalesced across the lanes that share a row (a 256-channel
// texel is exactly 64 lanes x float4).  This stand-alone stage is HBM-bound: per output row it writes
// sum(Cl)*4 bytes; the 4 taps of the same size come out of L2 / Infinity Cache (the maps are 75 MB).
#include "car_common.h"
#include "car_geom.h"

namespace {

struct GatherLevels {
    const float* map[CAR_MAX_LEVELS];
    int c[CAR_MAX_LEVELS], h[CAR_MAX_LEVELS], w[CAR_MAX_LEVELS];
    int q0[CAR_MAX_LEVELS + 1];   // first channel quad of each level; q0[n_levels] = quads per row
    int n_levels;
};

constexpr int kGatherWave = 7;      // launch_gather: the wave-task kernel (falls back to (32, 3) for channel counts it does not cover)
constexpr int kGatherCfg = kGatherWave;    // the product's configuration (tools/bench_gather.py: wave-tasks 55-65 % of the HBM peak, (32 rows, 3 items) 52-54 %, (16, 3) 49-50 %)

// A workgroup handles kGRows sampled points at a time: the bilinear tap indices / weights are computed once per (point,
// level) into LDS, then every thread moves float4s: 4 tap reads (L1 / L2 resident maps), one 16-byte store, kItems items
// (4 kItems tap loads) in flight per thread.  The stage is bound by the texture path (five bytes through it per byte written).
template <int kGRows, int kItems>
__global__ void __launch_bounds__(256) gather_kernel(GatherLevels L, int n_maps, const float* __restrict__ grid,
                                                     long pts, int run, int mode, int place, int V, float* __restrict__ out,
                                                     int ld_out, int col_out) {
    __shared__ int s_idx[kGRows][CAR_MAX_LEVELS][4];
    __shared__ float s_w[kGRows][CAR_MAX_LEVELS][4];
    __shared__ long s_row[kGRows];
    const int qpr = L.q0[L.n_levels];
    const long total_rows = (long)n_maps * pts;
    const int tid = threadIdx.x;
    // A work group's 16 rows: 16 consecutive points (run == 1), or the same step of 16 neighbouring rays when the points of a map
    // are rays x `run` steps — neighbouring rays sample neighbouring texels, so their taps share cache lines.
    const long rays = pts / run, rblocks = (rays + kGRows - 1) / kGRows;
    const long groups = run > 1 ? (long)n_maps * rblocks * run : (total_rows + kGRows - 1) / kGRows;
    __shared__ bool s_live[kGRows];
    for (long gidx = blockIdx.x; gidx < groups; gidx += gridDim.x) {
        for (int job = tid; job < kGRows * L.n_levels; job += 256) {
            const int rl = job / L.n_levels, l = job % L.n_levels;
            long mp;
            bool live_row;
            if (run > 1) {
                const long p = gidx % run, rb = (gidx / run) % rblocks, mm = gidx / (run * rblocks);
                long ray = rb * kGRows + rl;
                live_row = ray < rays;
                if (!live_row) ray = rays - 1;
                mp = mm * pts + ray * run + p;
            } else {
                mp = gidx * kGRows + rl;
                live_row = mp < total_rows;
                if (!live_row) mp = total_rows - 1;
            }
            if (l == 0) s_live[rl] = live_row;
            const int m = (int)(mp / pts);
            int tidx[4];
            float tw[4];
            car_bilinear_taps(grid[2 * mp], grid[2 * mp + 1], L.w[l], L.h[l], mode, tidx, tw);
            for (int t = 0; t < 4; ++t) { s_idx[rl][l][t] = m * L.h[l] * L.w[l] + tidx[t]; s_w[rl][l][t] = tw[t]; }
            if (l == 0) {
                const long i = mp % pts;
                long row;
                if (place == CAR_PLACE_PLAIN) row = mp;
                else if (place == CAR_PLACE_OWN) row = mp * V + (m % V);
                else { const int sc = m / 2, s = m % 2; row = (((long)(sc * 2 + (1 - s))) * pts + i) * 2 + s; }
                s_row[rl] = row;
            }
        }
        __syncthreads();
        // kItems items (4 kItems tap loads) in flight per thread before the first store: the stage lives on memory-level parallelism
        const int n_items = kGRows * qpr;
        for (int it0 = tid; it0 < n_items; it0 += kItems * 256) {
            float4 tp[kItems][4];
            float wv[kItems][4];
            long orow[kItems];
            int oq[kItems];
            bool ok[kItems];
#pragma unroll
            for (int u = 0; u < kItems; ++u) {
                const int item = it0 + u * 256;
                const int ci = item < n_items ? item : tid;             // clamp: harmless duplicate read, store masked
                const int rl = ci / qpr, q = ci % qpr;
                ok[u] = item < n_items && s_live[rl];
                int l = 0;
                while (l + 1 < L.n_levels && q >= L.q0[l + 1]) ++l;
                const float* base = L.map[l] + 4 * (q - L.q0[l]);
                const long cl = L.c[l];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    tp[u][t] = *reinterpret_cast<const float4*>(base + (long)s_idx[rl][l][t] * cl);
                    wv[u][t] = s_w[rl][l][t];
                }
                orow[u] = s_row[rl];
                oq[u] = q;
            }
#pragma unroll
            for (int u = 0; u < kItems; ++u) {
                if (!ok[u]) continue;
                float4 r;   // ((nw + ne) + sw) + se, products rounded individually (ATen's vectorised CPU kernel order)
                r.x = ((tp[u][0].x * wv[u][0] + tp[u][1].x * wv[u][1]) + tp[u][2].x * wv[u][2]) + tp[u][3].x * wv[u][3];
                r.y = ((tp[u][0].y * wv[u][0] + tp[u][1].y * wv[u][1]) + tp[u][2].y * wv[u][2]) + tp[u][3].y * wv[u][3];
                r.z = ((tp[u][0].z * wv[u][0] + tp[u][1].z * wv[u][1]) + tp[u][2].z * wv[u][2]) + tp[u][3].z * wv[u][3];
                r.w = ((tp[u][0].w * wv[u][0] + tp[u][1].w * wv[u][1]) + tp[u][2].w * wv[u][2]) + tp[u][3].w * wv[u][3];
                // streaming store (one 16-byte instruction): the gathered rows are written once and must not evict the feature maps from L2
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 rv = {r.x, r.y, r.z, r.w};
                __builtin_nontemporal_store(rv, reinterpret_cast<f32x4*>(out + orow[u] * ld_out + col_out + 4 * oq[u]));
            }
        }
        __syncthreads();
    }
}

// ---- the same stage with wave-uniform bookkeeping: a wave-instruction covers whole rows of one level (lanes = rows x channel quads of
//      that level: 1 row of a 256-channel level, 4 rows of a 64-channel one), so the level, the tap tables' rows and the output
//      window are found once per wave-task with scalar arithmetic instead of a division and a level search per float4.  Needs
//      every level's quad count to be a power of two (<= 64 lanes per row, or a multiple of 64).  Same arithmetic, same results. ----
constexpr int kWRows = 32;          // rows per work group

struct WaveLevels { int lpr[CAR_MAX_LEVELS], rpt[CAR_MAX_LEVELS], lpr_shift[CAR_MAX_LEVELS], seg_shift[CAR_MAX_LEVELS], tasks[CAR_MAX_LEVELS + 1]; };   // lanes per row, rows per task (powers of two)

__global__ void __launch_bounds__(256) gather_wave_kernel(GatherLevels L, WaveLevels WL, int n_maps, const float* __restrict__ grid, long pts, int run,
                                                          int mode, int place, int V, float* __restrict__ out, int ld_out, int col_out) {
    __shared__ __attribute__((aligned(16))) int s_idx[kWRows][CAR_MAX_LEVELS][4];
    __shared__ __attribute__((aligned(16))) float s_w[kWRows][CAR_MAX_LEVELS][4];
    __shared__ long s_row[kWRows];                                    // output row, -1 for a row past the end
    const long total_rows = (long)n_maps * pts;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long rays = pts / run, rblocks = (rays + kWRows - 1) / kWRows;
    const long groups = run > 1 ? (long)n_maps * rblocks * run : (total_rows + kWRows - 1) / kWRows;
    const int n_tasks = WL.tasks[L.n_levels];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    for (long gidx = blockIdx.x; gidx < groups; gidx += gridDim.x) {
        for (int job = tid; job < kWRows * L.n_levels; job += 256) {
            const int rl = job / L.n_levels, l = job % L.n_levels;
            long mp;
            bool live_row;
            if (run > 1) {
                const long p = gidx % run, rb = (gidx / run) % rblocks, mm = gidx / (run * rblocks);
                long ray = rb * kWRows + rl;
                live_row = ray < rays;
                if (!live_row) ray = rays - 1;
                mp = mm * pts + ray * run + p;
            } else {
                mp = gidx * kWRows + rl;
                live_row = mp < total_rows;
                if (!live_row) mp = total_rows - 1;
            }
            const int m = (int)(mp / pts);
            int tidx[4];
            float tw[4];
            car_bilinear_taps(grid[2 * mp], grid[2 * mp + 1], L.w[l], L.h[l], mode, tidx, tw);
            for (int t = 0; t < 4; ++t) { s_idx[rl][l][t] = m * L.h[l] * L.w[l] + tidx[t]; s_w[rl][l][t] = tw[t]; }
            if (l == 0) {
                const long i = mp % pts;
                long row;
                if (place == CAR_PLACE_PLAIN) row = mp;
                else if (place == CAR_PLACE_OWN) row = mp * V + (m % V);
                else { const int sc = m / 2, s = m % 2; row = (((long)(sc * 2 + (1 - s))) * pts + i) * 2 + s; }
                s_row[rl] = live_row ? row : -1;
            }
        }
        __syncthreads();
        // wave-tasks of this group, three at a time (12 tap loads in flight per lane)
        for (int t0 = wave; t0 < n_tasks; t0 += 12) {
            f32x4 tp[3][4];
            float4 wv[3];
            float* dst[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int task = t0 + 4 * u < n_tasks ? t0 + 4 * u : t0;      // clamp: harmless duplicate read, store masked
                int l = 0;
                while (l + 1 < L.n_levels && task >= WL.tasks[l + 1]) ++l;    // scalar: the task index is wave-uniform
                const int tl = task - WL.tasks[l], seg = tl & ((1 << WL.seg_shift[l]) - 1), rb = tl >> WL.seg_shift[l];
                const int r = rb * WL.rpt[l] + (lane >> WL.lpr_shift[l]), q = seg * WL.lpr[l] + (lane & (WL.lpr[l] - 1));
                const int4 id = *reinterpret_cast<const int4*>(&s_idx[r][l][0]);
                wv[u] = *reinterpret_cast<const float4*>(&s_w[r][l][0]);
                const float* base = L.map[l] + 4 * q;
                const long cl = L.c[l];
                tp[u][0] = *reinterpret_cast<const f32x4*>(base + (long)id.x * cl);
                tp[u][1] = *reinterpret_cast<const f32x4*>(base + (long)id.y * cl);
                tp[u][2] = *reinterpret_cast<const f32x4*>(base + (long)id.z * cl);
                tp[u][3] = *reinterpret_cast<const f32x4*>(base + (long)id.w * cl);
                const long orow = s_row[r];
                dst[u] = (orow >= 0 && t0 + 4 * u < n_tasks) ? out + orow * ld_out + col_out + 4 * (L.q0[l] + q) : nullptr;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (!dst[u]) continue;
                f32x4 r;    // ((nw + ne) + sw) + se, products rounded individually (ATen's vectorised CPU kernel order)
                r = ((tp[u][0] * wv[u].x + tp[u][1] * wv[u].y) + tp[u][2] * wv[u].z) + tp[u][3] * wv[u].w;
                __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(dst[u]));
            }
        }
        __syncthreads();
    }
}

}  // namespace

int launch_gather(int cfg, const GatherLevels& L, int n_maps, const float* grid, long pts, int run, int mode, int place, int V, float* out,
                  int ld_out, int col_out, void* stream);

extern "C" int car_gather_bilinear(const float* const* maps, const int* level_c, const int* level_h,
                                   const int* level_w, int n_levels, int n_maps, const float* grid, long pts, int run, int mode,
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
    CAR_REQUIRE((long)n_maps * level_h[0] * level_w[0] < 2147483647L, "car_gather_bilinear: map too large for 32-bit texel indices");
    if (run < 1 || pts % run != 0) run = 1;
    return launch_gather(kGatherCfg, L, n_maps, grid, pts, run, mode, place, V, out, ld_out, col_out, stream);
}

#ifdef CAR_ABLATION
// development build only: the same kernel with another (rows per group, items in flight) pair — tools/bench_gather.py
extern "C" int car_gather_bilinear_cfg(int cfg, const float* const* maps, const int* level_c, const int* level_h, const int* level_w, int n_levels,
                                       int n_maps, const float* grid, long pts, int run, int mode, int place, int V, float* out, int ld_out,
                                       int col_out, void* stream) {
    GatherLevels L;
    L.n_levels = n_levels;
    int q = 0;
    for (int l = 0; l < n_levels; ++l) { L.map[l] = maps[l]; L.c[l] = level_c[l]; L.h[l] = level_h[l]; L.w[l] = level_w[l]; L.q0[l] = q; q += level_c[l] / 4; }
    L.q0[n_levels] = q;
    for (int l = n_levels; l < CAR_MAX_LEVELS; ++l) { L.map[l] = nullptr; L.c[l] = L.h[l] = L.w[l] = 0; if (l > n_levels) L.q0[l] = q; }
    if (run < 1 || pts % run != 0) run = 1;
    return launch_gather(cfg, L, n_maps, grid, pts, run, mode, place, V, out, ld_out, col_out, stream);
}
#endif

namespace {
template <int ROWS, int ITEMS>
int launch_cfg(const GatherLevels& L, int n_maps, const float* grid, long pts, int run, int mode, int place, int V, float* out, int ld_out,
               int col_out, void* stream) {
    const long groups = run > 1 ? (long)n_maps * ((pts / run + ROWS - 1) / ROWS) * run : ((long)n_maps * pts + ROWS - 1) / ROWS;
    const unsigned blocks = (unsigned)(groups < 65536 ? groups : 65536);
    (void)hipGetLastError();
    hipLaunchKernelGGL((gather_kernel<ROWS, ITEMS>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, L, n_maps, grid, pts, run, mode,
                       place, V, out, ld_out, col_out);
    CAR_CHECK_LAUNCH("car_gather_bilinear");
    return CAR_OK;
}
}  // namespace

int launch_gather(int cfg, const GatherLevels& L, int n_maps, const float* grid, long pts, int run, int mode, int place, int V, float* out,
                  int ld_out, int col_out, void* stream) {
    if (cfg == kGatherWave) {
        WaveLevels WL;
        bool ok = true;
        int tasks = 0;
        for (int l = 0; l < L.n_levels; ++l) {
            const int quads = L.c[l] / 4;
            ok = ok && quads > 0 && (quads & (quads - 1)) == 0 && (quads <= 64 || quads % 64 == 0);
            WL.lpr[l] = quads < 64 ? quads : 64;
            WL.rpt[l] = 64 / WL.lpr[l];
            ok = ok && WL.rpt[l] <= kWRows;                            // a 4-channel level (64 rows per task > the group's 32) takes the per-float4 kernel
            const int segs = quads / WL.lpr[l];
            WL.lpr_shift[l] = WL.seg_shift[l] = 0;
            while ((1 << WL.lpr_shift[l]) < WL.lpr[l]) ++WL.lpr_shift[l];
            while ((1 << WL.seg_shift[l]) < segs) ++WL.seg_shift[l];
            WL.tasks[l] = tasks;
            tasks += (kWRows / WL.rpt[l]) * segs;
        }
        for (int l = L.n_levels; l <= CAR_MAX_LEVELS; ++l) WL.tasks[l] = tasks;
        for (int l = L.n_levels; l < CAR_MAX_LEVELS; ++l) { WL.lpr[l] = WL.rpt[l] = 1; WL.lpr_shift[l] = WL.seg_shift[l] = 0; }
        if (ok) {
            const long groups = run > 1 ? (long)n_maps * ((pts / run + kWRows - 1) / kWRows) * run : ((long)n_maps * pts + kWRows - 1) / kWRows;
            const unsigned blocks = (unsigned)(groups < 65536 ? groups : 65536);
            (void)hipGetLastError();
            hipLaunchKernelGGL(gather_wave_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L, WL, n_maps, grid, pts, run, mode, place, V,
                               out, ld_out, col_out);
            CAR_CHECK_LAUNCH("car_gather_bilinear");
            return CAR_OK;
        }
        cfg = 1;                                                       // odd channel counts: the per-float4 kernel
    }
    switch (cfg) {
        case 1: return launch_cfg<32, 3>(L, n_maps, grid, pts, run, mode, place, V, out, ld_out, col_out, stream);
        case 2: return launch_cfg<64, 3>(L, n_maps, grid, pts, run, mode, place, V, out, ld_out, col_out, stream);
        case 3: return launch_cfg<16, 4>(L, n_maps, grid, pts, run, mode, place, V, out, ld_out, col_out, stream);
        case 4: return launch_cfg<32, 4>(L, n_maps, grid, pts, run, mode, place, V, out, ld_out, col_out, stream);
        case 5: return launch_cfg<64, 4>(L, n_maps, grid, pts, run, mode, place, V, out, ld_out, col_out, stream);
        case 6: return launch_cfg<64, 6>(L, n_maps, grid, pts, run, mode, place, V, out, ld_out, col_out, stream);
        default: return launch_cfg<16, 3>(L, n_maps, grid, pts, run, mode, place, V, out, ld_out, col_out, stream);
    }
}
