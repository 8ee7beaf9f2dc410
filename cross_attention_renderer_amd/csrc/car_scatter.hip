// car_scatter.hip — grid_sample backward with respect to the maps WITHOUT floating-point atomics (what torch autograd
// does for F.grid_sample in models.py:261-344 when the pyramid needs a gradient).
// car_gather_bilinear_backward (car_backward.hip) adds w_tap * dout[row][c] into the texel with one fp32 atomic per (point, tap, channel):
// 1.36 G atomics per training step of the reference's shape, 3.5 ms — the fabric's atomic rate, not bandwidth.  Here the taps are BINNED by
// texel first (a counting sort on 8-byte (row, weight) records: integer atomics on 2 M counters, 7 M records) and every texel is then written
// ONCE, by the lanes that own its channels, as the sum over its records of w * dout[row][channels of its level] — whole 256-byte / 1-KB
// reads of dout rows, no atomics on floats, no zero fill of the maps beforehand (a texel nobody touched is stored as zero).
//   pass 1 count   one thread per (gather, map, point, level): count[texel] += 1 per live tap
//   pass 2 scan    exclusive prefix over all texels of all maps (three small kernels)
//   pass 3 fill    the same walk as pass 1: slot = cursor[texel]++ ; record[slot] = (row, weight)
//   pass 4 reduce  a 64-lane group per texel of a 256-channel level (16 lanes per texel of a 64-channel one): float4 per lane
// The order of a texel's records — hence of its fp32 additions — is the order in which pass 3's integer atomics happened to land, as the
// order of the float atomics was before; the sum is over the same terms.
#include "car_common.h"
#include "car_geom.h"

namespace {

constexpr int kMaxGathers = 4;

struct BinLevels {
    float* map[CAR_MAX_LEVELS];
    int c[CAR_MAX_LEVELS], h[CAR_MAX_LEVELS], w[CAR_MAX_LEVELS];
    int c0[CAR_MAX_LEVELS];                 // first column of the level's channels inside a row of dout
    int t0[CAR_MAX_LEVELS + 1];             // first texel of the level inside a map's counters
    int n_levels;
};
struct BinGathers {
    const float* grid[kMaxGathers];         // [n_maps, pts, 2]
    int mode[kMaxGathers], place[kMaxGathers];
    int n;
};
struct Record { int row; float w; };

__device__ __forceinline__ long place_row(int place, int V, int m, long i, long pts) {
    const long mp = (long)m * pts + i;
    if (place == CAR_PLACE_PLAIN) return mp;
    if (place == CAR_PLACE_OWN) return mp * V + (m % V);
    const int sc = m / 2, s = m % 2;
    return (((long)(sc * 2 + (1 - s))) * pts + i) * 2 + s;
}

// passes 1 and 3: FILL = false counts, FILL = true claims a slot per live tap and writes its record
template <bool FILL>
__global__ void __launch_bounds__(256) bin_kernel(BinLevels L, BinGathers G, int n_maps, long pts, int V, unsigned* __restrict__ counter,
                                                  Record* __restrict__ rec) {
    const long per = (long)n_maps * pts * L.n_levels;
    const long total = per * G.n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int gi = (int)(idx / per);
        const long r = idx % per;
        const int l = (int)(r % L.n_levels);
        const long mp = r / L.n_levels;
        const int m = (int)(mp / pts);
        const long i = mp % pts;
        int tidx[4];
        float tw[4];
        const float* g = G.grid[gi] + 2 * mp;
        car_bilinear_taps(g[0], g[1], L.w[l], L.h[l], G.mode[gi], tidx, tw);
        const long base = (long)m * L.t0[L.n_levels] + L.t0[l];
        const int row = (int)place_row(G.place[gi], V, m, i, pts);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (tw[t] == 0.0f) continue;
            const unsigned slot = atomicAdd(counter + base + tidx[t], 1u);
            if (FILL) rec[slot] = Record{row, tw[t]};
        }
    }
}

// exclusive scan of n counters in three kernels: per 1024-element block sums; scan of the block sums (one workgroup); offsets applied.
// `start` receives the exclusive prefix (n + 1 entries: the last is the total), `cursor` a second copy for pass 3 to increment.
constexpr int kScanBlock = 1024;
__global__ void __launch_bounds__(256) scan_sums_kernel(const unsigned* __restrict__ count, long n, unsigned* __restrict__ sums) {
    __shared__ unsigned red[4];
    const long b0 = (long)blockIdx.x * kScanBlock;
    unsigned s = 0;
    for (int k = threadIdx.x; k < kScanBlock; k += 256) s += b0 + k < n ? count[b0 + k] : 0u;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(1024) scan_blocks_kernel(unsigned* __restrict__ sums, int n_blocks, unsigned* __restrict__ total) {
    // exclusive scan of the block sums in place, by one workgroup: each thread a contiguous run
    __shared__ unsigned part[1024];
    const int per = (n_blocks + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    unsigned s = 0;
    for (int k = 0; k < per; ++k) s += b0 + k < n_blocks ? sums[b0 + k] : 0u;
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
    for (int k = 0; k < per; ++k) {
        if (b0 + k >= n_blocks) break;
        const unsigned v = sums[b0 + k];
        sums[b0 + k] = run;
        run += v;
    }
    if (threadIdx.x == 1023) *total = part[1023];
}
__global__ void __launch_bounds__(256) scan_apply_kernel(unsigned* __restrict__ count, long n, const unsigned* __restrict__ sums,
                                                         unsigned* __restrict__ start) {
    // a wave scans 256 consecutive counters (four per lane); the block's four waves chain through LDS
    __shared__ unsigned wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i0 = (long)blockIdx.x * kScanBlock + wave * 256 + lane * 4;
    unsigned v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? count[i0 + k] : 0u;
    const unsigned mine = v[0] + v[1] + v[2] + v[3];
    unsigned inc = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned off = sums[blockIdx.x] + inc - mine;
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < n) { start[i0 + k] = off; count[i0 + k] = off; }          // count becomes pass 3's cursor
        off += v[k];
    }
}

// pass 4: LANES lanes own one texel (float4 of channels each; LANES * 4 = the level's channel count, or a divisor of it: then the lanes
// loop over the channel groups).  A texel's records are read four at a time (row indices and weights are wave-uniform per texel group only
// when LANES == 64: they are loaded per lane and the loads of dout go out together).
__global__ void __launch_bounds__(256) reduce_kernel(BinLevels L, int n_maps, const unsigned* __restrict__ start, const Record* __restrict__ rec,
                                                     const float* __restrict__ dout, int ld_out, int col_out, int level, int lanes) {
    const int C = L.c[level];
    const long texels = (long)L.h[level] * L.w[level];
    const long total = (long)n_maps * texels;
    const int per_block = 256 / lanes;
    const int sub = threadIdx.x % lanes;
    const long tex = (long)blockIdx.x * per_block + threadIdx.x / lanes;
    if (tex >= total) return;
    const int m = (int)(tex / texels);
    const long t = tex % texels;
    const long ci = (long)m * L.t0[L.n_levels] + L.t0[level] + t;
    const unsigned s0 = start[ci], s1 = start[ci + 1];
    float* out = L.map[level] + tex * C;
    const float* din = dout + col_out + L.c0[level];
    for (int c = 4 * sub; c < C; c += 4 * lanes) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned s = s0;
        for (; s + 4 <= s1; s += 4) {
            Record r[4];
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = rec[s + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(din + (long)r[k].row * ld_out + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc.x = fmaf(r[k].w, v[k].x, acc.x); acc.y = fmaf(r[k].w, v[k].y, acc.y);
                acc.z = fmaf(r[k].w, v[k].z, acc.z); acc.w = fmaf(r[k].w, v[k].w, acc.w);
            }
        }
        for (; s < s1; ++s) {
            const Record r = rec[s];
            const float4 v = *reinterpret_cast<const float4*>(din + (long)r.row * ld_out + c);
            acc.x = fmaf(r.w, v.x, acc.x); acc.y = fmaf(r.w, v.y, acc.y); acc.z = fmaf(r.w, v.z, acc.z); acc.w = fmaf(r.w, v.w, acc.w);
        }
        *reinterpret_cast<float4*>(out + c) = acc;
    }
}

unsigned blocks_for(long total, long cap) {
    const long blocks = (total + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

}  // namespace

// Bytes of workspace car_gather_bilinear_backward_binned needs for n_gathers gathers of n_maps x pts points into the given levels.
extern "C" size_t car_scatter_workspace_bytes(const int* level_h, const int* level_w, int n_levels, int n_maps, long pts, int n_gathers) {
    if (!level_h || !level_w || n_levels <= 0 || n_levels > CAR_MAX_LEVELS || n_maps <= 0 || pts <= 0 || n_gathers <= 0) return 0;
    long texels = 0;
    for (int l = 0; l < n_levels; ++l) texels += (long)level_h[l] * level_w[l];
    const long n = (long)n_maps * texels + 1;
    const long blocks = (n + kScanBlock - 1) / kScanBlock;
    const long recs = (long)n_gathers * n_maps * pts * n_levels * 4;
    // counters (-> cursor) | start | block sums (+ total) | records
    return (size_t)(((2 * n + blocks + 4) * sizeof(unsigned) + 15) / 16 * 16 + recs * sizeof(Record));
}

// dmaps[l] [n_maps, Hl, Wl, Cl] = sum over the n_gathers gathers of grid_sample's backward (NOT accumulated: every texel is written, the
// untouched ones as zero).  Gather j samples the maps at grids[j] [n_maps, pts, 2] with padding modes[j] (0 border, 1 zeros) and row
// placement places[j] (car_gather_bilinear's rule); all of them read the gradient of their gathered rows from dout (row stride ld_out,
// first column col_out, the levels' channels back to back as car_gather_bilinear wrote them).
extern "C" int car_gather_bilinear_backward_binned(float* const* dmaps, const int* level_c, const int* level_h, const int* level_w, int n_levels,
                                                   int n_maps, const float* const* grids, const int* modes, const int* places, int n_gathers,
                                                   long pts, int V, const float* dout, int ld_out, int col_out, void* workspace,
                                                   size_t workspace_bytes, void* stream) {
    CAR_REQUIRE(dmaps && level_c && level_h && level_w && grids && modes && places && dout && workspace, "car_gather_bilinear_backward_binned: null pointer");
    CAR_REQUIRE(n_levels > 0 && n_levels <= CAR_MAX_LEVELS && n_maps > 0 && pts > 0 && n_gathers > 0 && n_gathers <= kMaxGathers,
                "car_gather_bilinear_backward_binned: bad sizes");
    BinLevels L;
    BinGathers G;
    L.n_levels = n_levels;
    int c0 = 0;
    long t0 = 0;
    for (int l = 0; l < n_levels; ++l) {
        CAR_REQUIRE(dmaps[l] && level_c[l] > 0 && level_c[l] % 4 == 0 && level_h[l] > 0 && level_w[l] > 0,
                    "car_gather_bilinear_backward_binned: level %d needs a channel count that is a positive multiple of 4", l);
        L.map[l] = dmaps[l]; L.c[l] = level_c[l]; L.h[l] = level_h[l]; L.w[l] = level_w[l];
        L.c0[l] = c0; L.t0[l] = (int)t0;
        c0 += level_c[l];
        t0 += (long)level_h[l] * level_w[l];
    }
    L.t0[n_levels] = (int)t0;
    for (int l = n_levels; l < CAR_MAX_LEVELS; ++l) { L.map[l] = nullptr; L.c[l] = L.h[l] = L.w[l] = L.c0[l] = 0; if (l > n_levels) L.t0[l] = (int)t0; }
    CAR_REQUIRE(ld_out % 4 == 0 && col_out % 4 == 0 && col_out >= 0 && col_out + c0 <= ld_out && ((uintptr_t)dout & 15) == 0,
                "car_gather_bilinear_backward_binned: window [%d,%d) must be float4-aligned inside a row of %d", col_out, col_out + c0, ld_out);
    G.n = n_gathers;
    for (int j = 0; j < kMaxGathers; ++j) {
        G.grid[j] = j < n_gathers ? grids[j] : nullptr; G.mode[j] = j < n_gathers ? modes[j] : 0; G.place[j] = j < n_gathers ? places[j] : 0;
        if (j >= n_gathers) continue;
        CAR_REQUIRE(grids[j] && (modes[j] == 0 || modes[j] == 1), "car_gather_bilinear_backward_binned: gather %d: mode must be 0 (border) or 1 (zeros)", j);
        CAR_REQUIRE(places[j] == CAR_PLACE_PLAIN || places[j] == CAR_PLACE_OWN || (places[j] == CAR_PLACE_OTHER2 && V == 2 && n_maps % 2 == 0),
                    "car_gather_bilinear_backward_binned: bad placement %d for V=%d", places[j], V);
    }
    const long n = (long)n_maps * t0 + 1;                                  // one counter per texel and a closing one
    const long recs = (long)n_gathers * n_maps * pts * n_levels * 4;
    CAR_REQUIRE(n < (1l << 31) && recs < (1l << 32) && (long)n_maps * pts * (V > 0 ? V : 1) < (1l << 31),
                "car_gather_bilinear_backward_binned: too many texels / records for 32-bit counters");
    CAR_REQUIRE(workspace_bytes >= car_scatter_workspace_bytes(level_h, level_w, n_levels, n_maps, pts, n_gathers) && ((uintptr_t)workspace & 15) == 0,
                "car_gather_bilinear_backward_binned: workspace too small (car_scatter_workspace_bytes) or not 16-byte aligned");
    const long blocks = (n + kScanBlock - 1) / kScanBlock;
    unsigned* counter = reinterpret_cast<unsigned*>(workspace);
    unsigned* start = counter + n;
    unsigned* sums = start + n;
    Record* rec = reinterpret_cast<Record*>(reinterpret_cast<char*>(workspace) + ((2 * n + blocks + 4) * sizeof(unsigned) + 15) / 16 * 16);
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    if (hipMemsetAsync(counter, 0, n * sizeof(unsigned), st) != hipSuccess) { car_set_error("car_gather_bilinear_backward_binned: memset failed"); return CAR_E_LAUNCH; }
    const long work = (long)n_gathers * n_maps * pts * n_levels;
    hipLaunchKernelGGL(bin_kernel<false>, dim3(blocks_for(work, 65536)), dim3(256), 0, st, L, G, n_maps, pts, V, counter, rec);
    hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, st, counter, n, sums);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, st, sums, (int)blocks, sums + blocks);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, counter, n, sums, start);
    hipLaunchKernelGGL(bin_kernel<true>, dim3(blocks_for(work, 65536)), dim3(256), 0, st, L, G, n_maps, pts, V, counter, rec);
    for (int l = 0; l < n_levels; ++l) {
        // lanes per texel: the largest of 64 / 32 / 16 / 8 / 4 / 2 / 1 that the level's float4 count fills
        const int quads = level_c[l] / 4;
        int lanes = 64;
        while (lanes > quads) lanes >>= 1;
        const long texels = (long)n_maps * level_h[l] * level_w[l];
        const long per_block = 256 / lanes;
        hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((texels + per_block - 1) / per_block)), dim3(256), 0, st, L, n_maps, start, rec, dout, ld_out,
                           col_out, l, lanes);
    }
    CAR_CHECK_LAUNCH("car_gather_bilinear_backward_binned");
    return CAR_OK;
}
