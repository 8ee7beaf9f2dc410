// car_fused_tex.hip — the fused per-sample kernel with an LDS TEXEL CACHE in front of the 12-tap gather (same stage, same inputs,
// outputs, packed weights and arithmetic as car_fused.hip; results are bit-identical).
//
// Why.  car_fused.hip is bound by the texture-address / L1 path: every sample fetches 12 taps x 2 sources x 576 channels through it
// (55 KB per sample, 58 GB per 8192 rays), although the 16 neighbouring rays x 4 consecutive steps a "bundle" of four waves works
// on touch only ~14 / 22 / 43 DISTINCT texels of the three pyramid levels out of 256 tap requests each (measured on the bench
// pair; profiles/).  Here the distinct texels of a bundle are brought into LDS once per 32-channel chunk by LDS-DMA
// (global_load_lds: the bounding box of the bundle's taps per level and source, 128-byte slices) and the taps are read from LDS,
// which has four times the bandwidth of the L1 path and no request charge per lane: ~7x fewer bytes through the TA.
//
// Structure (differences from car_fused.hip):
//   * geometry phase: geometry wave b (64 samples = bundle b) also reduces the bounding box of its taps per (source, level) and
//     packs the boxes into the bundle's pool of kPool cache slices, coarsest level first.  The two coarse levels must fit (they
//     practically always do: ~40 of 152 slices); if the finest level's box does not fit as well, that level is gathered "direct"
//     (from global memory, as car_fused.hip does) — a wave-uniform choice per (bundle, source).  A workgroup whose coarse levels do
//     not fit (wildly diverging rays) gives its sample group back: it appends its index to the `redo` list and the host runs
//     car_fused.hip's kernel over that list afterwards;
//   * eight waves (two bundles, 128 samples) per workgroup at two waves per SIMD: with the taps coming from LDS there is little
//     latency left for a third wave to hide, and 256 registers per wave leave room for everything without spills;
//   * weight stream in HALF chunks (9 tiles = 18 KB, two buffers): 36 KB of LDS instead of 72 KB make room for two texel-cache
//     buffers of 2 x kPool slices (chunk c+1 is gathered from one while chunk c+2 lands in the other);
//   * no transposing stage tile: a lane owns (sample s, channels 8 q .. 8 q + 7) of the chunk — exactly its B-operand slot — and
//     reads its 32 bytes of every tap from the cache (pieces XOR-swizzled by slot so that different texels of one read land on
//     different banks);
//   * the key / query tail runs on chunks of one K step (8 tiles).
#include "car_common.h"
#include "car_geom.h"

namespace {

constexpr int kWaves = 8, kRows = 16, kGroup = kWaves * kRows;         // 128 samples per workgroup
constexpr int kStepsPerGroup = 4, kBundles = kWaves / kStepsPerGroup;  // 2 bundles of 16 rays x 4 steps
constexpr int kPieces = 3;                                             // LDS-DMA pieces per weight half chunk (18 KB over 8 waves)
constexpr int kThreads = kWaves * 64;

#include "car_fused_mma.h"

__device__ __forceinline__ int chunk_tile_offset(int) { return 0; }    // the stream helpers of car_fused_mma.h are not used here
__device__ __forceinline__ int chunk_tiles(int) { return 0; }

constexpr int kHalf = kTE / 2;                       // 9 tiles per weight half chunk
constexpr int kWBuf = kHalf * kTile;                 // floats per weight buffer (18 KB)
constexpr int kPool = 152;                           // cache slices (one texel x 32 channels = 128 B) per bundle and buffer
constexpr int kSlice = 32;                           // floats per slice
constexpr int kDmaPerBundle = kPool / 8;             // 19 DMA instructions of 8 slices fill a bundle's pool
constexpr int kDmaPerWave = (kDmaPerBundle + kStepsPerGroup - 1) / kStepsPerGroup;   // 5 of them per wave and chunk
constexpr int kCacheBuf = kBundles * kPool * kSlice; // floats per cache buffer (39 KB)

constexpr int kLdsWt = 0;                                      // [2][9][512]             weight half chunks               36 KB
constexpr int kLdsCache = kLdsWt + 2 * kWBuf;                  // [2][2][152][32]         texel cache                      76 KB
constexpr int kLdsTapI = kLdsCache + 2 * kCacheBuf;            // [128][2][3] uint        cached: slot | dx << 10 | dy-stride << 11 ; direct: byte offset | flags
constexpr int kLdsTapW = kLdsTapI + kGroup * 6;                // [128][2][3][4]          tap weights (nw, ne, sw, se)     12 KB
constexpr int kLdsPe = kLdsTapW + kGroup * 24;                 // [128][2][4]             tanh(pt_s/5)                      4 KB
constexpr int kLdsWpt = kLdsPe + kGroup * 8;                   // [576][4]                (W1[:,C:C+3], b1)                 9 KB
constexpr int kLdsBias = kLdsWpt + kC * 4;                     // [704]
constexpr int kLdsSrc = kLdsBias + kBiasFloats;                // [2][2][152] 64-bit      byte offset of every pool slice from gmap[0]
constexpr int kLdsDesc = kLdsSrc + kBundles * 2 * kPool * 2;   // [2][2][8] int           cached mask, -, -, slices used, ..., [7] give-up flag
constexpr int kLdsFloats = kLdsDesc + kBundles * 2 * 8;
constexpr size_t kLdsBytes = (size_t)kLdsFloats * sizeof(float);
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

struct TexArgs {
    const CarPose* poses;
    const CarRay* rays;
    const float* steps;
    const float* gmap[3];
    int gh[3], gw[3];
    const float* gmeta;
    const float* wpt;
    const float* blob;
    const float* bias;
    int b, V, R, P, H, W;
    long S;
    float* e;
    float* qry;
    float* g;
    float* logit;
    float* pt;
    float* pixel_val;
    int* redo;                 // [0] number of sample groups handed back, [1 + k] their indices
};

// weight chunks: W2 in halves (source 0: 36, source 1: 36) | K1 over e_1 (K steps 9..17) | K1 over e_0 (0..8) | K2 x4 | Q1 | Q2 x4
constexpr int kG_K1b = 4 * kKS, kG_K1a = kG_K1b + 9, kG_K2 = kG_K1a + 9, kG_Q1 = kG_K2 + 4, kG_Q2 = kG_Q1 + 1, kG_End = kG_Q2 + 4;
__device__ __forceinline__ int wchunk_tile_offset(int g) {
    if (g < kG_K1b) { const int h = g % (2 * kKS); return kOffW2 + (h >> 1) * kTE + (h & 1) * kHalf; }
    if (g < kG_K1a) return kOffK1 + (9 + g - kG_K1b) * kTD;
    if (g < kG_K2) return kOffK1 + (g - kG_K1a) * kTD;
    if (g < kG_Q1) return kOffK2 + (g - kG_K2) * kTD;
    if (g < kG_Q2) return kOffQ1;
    return kOffQ2 + (g - kG_Q2) * kTD;
}
__device__ __forceinline__ int wchunk_tiles(int g) { return g < kG_K1b ? kHalf : kTD; }

// LDS-DMA of 1 KB: lane l copies 16 bytes from gsrc (per lane) to lds_dst + 16 l (wave-uniform base); inline asm, see car_linear.hip
__device__ __forceinline__ void dma_1k(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const float* p) { return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)p); }

// piece p (0, 1) of weight chunk g into its buffer: wave w copies KB number 12 p + w (wrapped: re-copying identical bytes is harmless)
__device__ __forceinline__ void weights_issue(const float* __restrict__ blob, float* lds, int g, int p, int lane, int wave) {
    const int ge = g < kG_End ? g : kG_End - 1;
    const int nkb = 2 * wchunk_tiles(ge);
    int kb = kWaves * p + wave;
    kb = kb < nkb ? kb : kb - nkb;
    kb = kb < nkb ? kb : kb - nkb;
    const float* src = blob + (long)wchunk_tile_offset(ge) * kTile + kb * 256 + 4 * lane;
    dma_1k(src, lds_addr(lds + kLdsWt + (ge & 1) * kWBuf + kb * 256));
}
__device__ __forceinline__ void chunk_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ void mfma_single(f32x4& c0, const float* w0, const half8& bhi, const half8& blo) {
    const half8 ah = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0));
    const half8 al = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 256));
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bhi, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, blo, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bhi, c0, 0, 0, 0);
}

// a chained layer with 128 outputs over NSRC source tiles, one K step (two source tiles) per weight chunk
template <int NSRC, bool RELU>
__device__ __forceinline__ void chained_layer1(f32x4 (&acc)[kTD], const f32x4 (&src)[NSRC], float p, const float* __restrict__ blob,
                                               float* lds, int& g, int lane, int wave) {
#pragma unroll
    for (int m = 0; m < NSRC / 2; ++m) {
        const float* wl = lds + kLdsWt + (g & 1) * kWBuf + 4 * lane;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[e] = src[2 * m + (e >> 2)][e & 3];
            if (RELU) x[e] = fmaxf(x[e], 0.f);
        }
        half8 bhi, blo;
        split8(x, p, bhi, blo);
#pragma unroll
        for (int q = 0; q < kTD / 2; ++q) {
            const float* w0 = wl + (2 * q * 2) * 256;
            mfma_pair(acc[2 * q], acc[2 * q + 1], w0, w0 + 512, bhi, blo);
            if (q < kPieces) weights_issue(blob, lds, g + 1, q, lane, wave);
            __builtin_amdgcn_sched_barrier(0);
        }
        chunk_sync();
        ++g;
    }
}

// ABL > 0: variants of the development build.  Timing only (wrong results): 1 no tap reads / blends, 2 no cache DMA, 3 = 1 + 2,
// 4 no MFMAs in the e passes, 5 no barriers / weight DMA in the e passes (with 3).  6: the full kernel without the scheduling
// barriers between the slots of the e passes (correct results)
template <int ABL>
__global__ void __launch_bounds__(kThreads) fused_tex_kernel(const TexArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave index in an SGPR
    const int s = lane & 15, q4 = lane >> 4;
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    {   // workgroup b runs on XCD b % 8 (observed, speed only): a contiguous band of sample groups per XCD
        const int q8 = nblk / 8, r8 = nblk % 8, xcd = blk % 8, idx = blk / 8;
        blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int pgs = (a.P + kStepsPerGroup - 1) / kStepsPerGroup, bundles = (a.R + kBundles * kRows - 1) / (kBundles * kRows);
    const int pg = blk % pgs, bun = (blk / pgs) % bundles, nn = blk / (pgs * bundles);
    const int bundle = wave / kStepsPerGroup;
    const int ray_i = bun * (kBundles * kRows) + bundle * kRows + s, pp = pg * kStepsPerGroup + wave % kStepsPerGroup;
    const bool live = ray_i < a.R && pp < a.P;
    const long i = ((long)nn * a.R + (ray_i < a.R ? ray_i : a.R - 1)) * a.P + (pp < a.P ? pp : a.P - 1);

    for (int k = tid; k < kC; k += kThreads) *reinterpret_cast<float4*>(lds + kLdsWpt + 4 * k) = *reinterpret_cast<const float4*>(a.wpt + 4 * k);
    for (int k = tid; k < kBiasFloats; k += kThreads) lds[kLdsBias + k] = a.bias[k];
    int g = 0;
#pragma unroll
    for (int p_ = 0; p_ < kPieces; ++p_) weights_issue(a.blob, lds, 0, p_, lane, wave);

    // ---- geometry: wave b (of the first three) computes the 64 samples of bundle b, one per lane, both source views ------------
    const int P = a.P, V = a.V;
    if (wave < kBundles) {
        const int sg = wave * 64 + lane, gwv = sg >> 4, gs = sg & 15;     // sample sg = row gs of matrix wave gwv; bundle = wave
        const int g_ray = bun * (kBundles * kRows) + (gwv / kStepsPerGroup) * kRows + gs, g_pp = pg * kStepsPerGroup + gwv % kStepsPerGroup;
        const bool g_live = g_ray < a.R && g_pp < a.P;
        const long gi = ((long)nn * a.R + (g_ray < a.R ? g_ray : a.R - 1)) * a.P + (g_pp < a.P ? g_pp : a.P - 1);
        const int p = (int)(gi % P);
        const long nr = gi / P;
        const int n = (int)(nr / a.R);
        const int v = n % V, sc = n / V;
        const CarPose& Ps = a.poses[n];
        const CarRay ray = a.rays[nr];
        CarSample smp;
        for (int k = 0; k < 2; ++k) smp.grid[k] = ray.start[k] + (ray.end[k] - ray.start[k]) * a.steps[p];
        car_sample_setup(Ps, a.poses + sc * 2, ray, 2, a.H, a.W, &smp);
        int* desc = reinterpret_cast<int*>(lds + kLdsDesc) + wave * 16;
        unsigned long long* srct = reinterpret_cast<unsigned long long*>(lds + kLdsSrc) + wave * 2 * kPool;
        int giveup = 0;
#pragma unroll
        for (int sv = 0; sv < 2; ++sv) {
            float gx, gy;
            int mode, m;
            if (sv == v) { gx = smp.grid[0]; gy = smp.grid[1]; mode = 0; m = n; }
            else { gx = sv == 0 ? smp.grid_in[0][0] : smp.grid_in[1][0]; gy = sv == 0 ? smp.grid_in[0][1] : smp.grid_in[1][1]; mode = 1; m = sc * V + sv; }
            unsigned* tb = reinterpret_cast<unsigned*>(lds + kLdsTapI) + (sg * 2 + sv) * 3;
            float* tw = lds + kLdsTapW + (sg * 2 + sv) * 12;
            int used = 0, mask = 0, base[3] = {0, 0, 0};
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                int idx[4];
                float w[4];
                car_bilinear_taps(gx, gy, a.gw[l], a.gh[l], mode, idx, w);
                *reinterpret_cast<float4*>(tw + 4 * l) = make_float4(w[0], w[1], w[2], w[3]);
                const int gwl = a.gw[l];
                int x0 = idx[0] % gwl, y0 = idx[0] / gwl;
                const int dx = idx[1] != idx[0] ? 1 : 0, dy = idx[2] != idx[0] ? 1 : 0;
                // bounding box of the bundle's taps at this level (samples whose four weights are all zero read nothing that matters)
                const bool any = (w[0] != 0.f) || (w[1] != 0.f) || (w[2] != 0.f) || (w[3] != 0.f);
                int xlo = any ? x0 : 0x7fffffff, ylo = any ? y0 : 0x7fffffff, xhi = any ? x0 + dx : -1, yhi = any ? y0 + dy : -1;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    xlo = min(xlo, __shfl_xor(xlo, o, 64)); ylo = min(ylo, __shfl_xor(ylo, o, 64));
                    xhi = max(xhi, __shfl_xor(xhi, o, 64)); yhi = max(yhi, __shfl_xor(yhi, o, 64));
                }
                if (xhi < 0) { xlo = ylo = xhi = yhi = 0; }                 // nobody reads this level: a 1 x 1 box
                const int bw = xhi - xlo + 1, bh = yhi - ylo + 1, cnt = bw * bh;
                const bool fits = cnt <= kPool - used && bw < 64;            // wave-uniform
                if (!fits && l < 2) giveup = 1;                              // a coarse level must be cached
                if (fits) {
                    // this level is served from the cache: slot of the nw tap inside the bundle's pool, +1 for ne, +bw for the row below
                    if (!any) { x0 = xlo; y0 = ylo; }
                    const int slot = used + (y0 - ylo) * bw + (x0 - xlo);
                    tb[l] = (unsigned)slot | (any && dx ? 1u << 10 : 0u) | (any && dy ? (unsigned)bw << 11 : 0u);
                    // source of every slice of the box: byte offset of texel (ylo + t / bw, xlo + t % bw) of map m inside the level
                    for (int t = lane; t < cnt; t += 64) {
                        const int ty = (int)(((float)t + 0.5f) / (float)bw), tx = t - ty * bw;
                        srct[sv * kPool + used + t] = (unsigned long long)(reinterpret_cast<const char*>(a.gmap[l]) - reinterpret_cast<const char*>(a.gmap[0]))
                                                      + (unsigned long long)((m * a.gh[l] + ylo + ty) * gwl + xlo + tx) * (unsigned long long)(kC * 4);
                    }
                    base[l] = used;
                    used += cnt;
                    mask |= 1 << l;
                } else {
                    tb[l] = (unsigned)(m * a.gh[l] * gwl + idx[0]) * (unsigned)(kC * 4) | (dx ? 1u : 0u) | (dy ? 2u : 0u);
                    base[l] = used;
                }
            }
            if (lane == 0) { desc[sv * 8 + 0] = mask; desc[sv * 8 + 3] = used; }
            const float px = sv == 0 ? smp.pt_in[0][0] : smp.pt_in[1][0], py = sv == 0 ? smp.pt_in[0][1] : smp.pt_in[1][1],
                        pz = sv == 0 ? smp.pt_in[0][2] : smp.pt_in[1][2];
            *reinterpret_cast<float4*>(lds + kLdsPe + (sg * 2 + sv) * 4) = make_float4(tanhf(px / 5.0f), tanhf(py / 5.0f), tanhf(pz / 5.0f), 0.0f);
        }
        if (lane == 0) desc[7] = giveup;
        if (g_live) {
            a.pixel_val[2 * gi] = smp.grid[0]; a.pixel_val[2 * gi + 1] = smp.grid[1];
            a.pt[3 * gi + 0] = smp.pt[0]; a.pt[3 * gi + 1] = smp.pt[1]; a.pt[3 * gi + 2] = smp.pt[2];
#pragma unroll
            for (int k = 0; k < 16; k += 4) *reinterpret_cast<float4*>(a.g + 16 * gi + k) = make_float4(smp.g[k], smp.g[k + 1], smp.g[k + 2], smp.g[k + 3]);
        }
    }
    chunk_sync();                                                      // tables, tap records and weight half chunk 0 visible
    {
        const int* d = reinterpret_cast<const int*>(lds + kLdsDesc);
        if (d[7] | d[16 + 7]) {                                       // some bundle's coarse levels do not fit: car_fused.hip's kernel takes the group
            if (tid == 0) a.redo[1 + atomicAdd(a.redo, 1)] = blk;
            return;
        }
    }

    // ---- per-bundle state (wave-uniform, in SGPRs): slices used and whether the finest level is cached, per source -------------------
    const int* desc = reinterpret_cast<const int*>(lds + kLdsDesc) + bundle * 16;
    const unsigned long long* srct = reinterpret_cast<const unsigned long long*>(lds + kLdsSrc) + bundle * 2 * kPool;
    const int sub = wave % kStepsPerGroup;                             // this wave's DMA instructions of the bundle pool: sub, sub + 4, ...
    const int used_sv[2] = {__builtin_amdgcn_readfirstlane(desc[3]), __builtin_amdgcn_readfirstlane(desc[8 + 3])};
    const bool l2c_sv[2] = {(__builtin_amdgcn_readfirstlane(desc[0]) & 4) != 0, (__builtin_amdgcn_readfirstlane(desc[8]) & 4) != 0};
    const char* gbase = reinterpret_cast<const char*>(a.gmap[0]);

    // cache DMA instruction k (0..4) of this wave for (source sv, chunk c) into buffer buf: 8 slices of the bundle's pool
    auto cache_issue = [&](int sv, int c, int buf, int k) {
        const int ins = sub + 4 * k;
        const int used = used_sv[sv];
        if constexpr (ABL == 2 || ABL == 3 || ABL == 5) return;
        if (ins >= kDmaPerBundle || 8 * ins >= used) return;           // wave-uniform
        const int sl = 8 * ins + (lane >> 3);
        const int slc = sl < used ? sl : used - 1;                     // unused tail of the last instruction: a harmless duplicate
        const unsigned piece = (unsigned)(lane & 7) ^ (((unsigned)sl >> 1 & 3u) << 1);     // XOR swizzle, see fetch
        const char* src = gbase + srct[sv * kPool + slc] + (128u * (unsigned)c + 16u * piece);
        dma_1k(src, lds_addr(lds + kLdsCache + buf * kCacheBuf + (bundle * kPool + 8 * ins) * kSlice));
    };

    // ---- gather: lane (s, q4) blends channels 8 q4 .. 8 q4 + 7 of chunk c for its sample, tap by tap.  Where a tap lives and what it
    //      weighs does not depend on the chunk: the 12 LDS offsets (or, for a finest level that is gathered direct, global offsets)
    //      and weights of the sample are decoded once per source pass into registers. ------------------------------------------------
    float hacc[8];
    f32x4 tap[2];
    const int srow = wave * kRows + s;
    unsigned toff[12];         // tap k = 0..11: level 2 - k / 4 (finest first, as car_fused.hip adds them), corner k % 4 (nw, ne, sw, se)
    float tw[12];
    auto load_taps = [&](int sv) {
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const unsigned info = reinterpret_cast<const unsigned*>(lds + kLdsTapI)[(srow * 2 + sv) * 3 + l];
            const float4 w4 = *reinterpret_cast<const float4*>(lds + kLdsTapW + (srow * 2 + sv) * 12 + 4 * l);
            const float ww[4] = {w4.x, w4.y, w4.z, w4.w};
            const bool cached = l < 2 || l2c_sv[sv];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = 4 * (2 - l) + t;
                tw[k] = ww[t];
                if (cached) {                                          // byte offset inside a cache buffer: slice of the bundle pool, swizzled piece
                    const unsigned slot = (info & 1023u) + ((t & 1) ? (info >> 10 & 1u) : 0u) + ((t & 2) ? (info >> 11) : 0u);
                    toff[k] = ((unsigned)(bundle * kPool) + slot) * 128u + 16u * ((2u * q4) ^ ((slot >> 1 & 3u) << 1));
                } else {                                               // byte offset inside the level (chunk 0)
                    toff[k] = (info & ~3u) + ((t & 1) && (info & 1u) ? (unsigned)(kC * 4) : 0u) + ((t & 2) && (info & 2u) ? (unsigned)a.gw[l] * (kC * 4) : 0u)
                              + 32u * q4;
                }
            }
        }
    };
    auto affine = [&](int sv, int c) {
        const float4 pe = *reinterpret_cast<const float4*>(lds + kLdsPe + (srow * 2 + sv) * 4);
        const float4* wp = reinterpret_cast<const float4*>(lds + kLdsWpt + 4 * (32 * c + 8 * q4));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 w = wp[k];
            hacc[k] = fmaf(w.z, pe.z, fmaf(w.y, pe.y, w.x * pe.x)) + w.w;
        }
    };
    auto fetch = [&](int c, int buf, int k, bool l2_cached) {
        if constexpr (ABL == 1 || ABL == 3 || ABL == 5) return;
        if (k >= 4 || l2_cached) {                                     // wave-uniform (compile-time for the coarse levels): from the cache
            const char* p = reinterpret_cast<const char*>(lds + kLdsCache + buf * kCacheBuf) + toff[k];
            tap[0] = *reinterpret_cast<const f32x4*>(p);
            tap[1] = *reinterpret_cast<const f32x4*>(p + 16);
        } else {                                                       // the finest level from global memory
            const char* p = reinterpret_cast<const char*>(a.gmap[2] + 32 * c) + toff[k];
            tap[0] = *reinterpret_cast<const f32x4*>(p);
            tap[1] = *reinterpret_cast<const f32x4*>(p + 16);
        }
    };
    auto blend = [&](int k) {
        if constexpr (ABL == 1 || ABL == 3 || ABL == 5) return;
        const float w = tw[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) { hacc[j] = fmaf(w, tap[0][j], hacc[j]); hacc[4 + j] = fmaf(w, tap[1][j], hacc[4 + j]); }
    };

    float hp, e_up, e_down;
    const float* lsc = lds + kLdsBias + kBiasScale;
    auto uniform = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    {
        float hinv;
        pow2_scale(fmaxf(((a.gmeta[0] + a.gmeta[1]) + a.gmeta[2]) + lsc[5], 1e-30f), hp, hinv);
        const float dW2 = lsc[kLayerW2];
        e_up = uniform(hp / dW2); e_down = uniform(dW2 * hinv); hp = uniform(hp);
    }
    auto finish = [&](half8& bhi, half8& blo) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = fmaxf(hacc[k], 0.f);
        split8(x, hp, bhi, blo);
    };

    // prologue: cache of chunks 0 and 1 of source 0, then the gather of chunk 0 with nothing to hide it under
#pragma unroll
    for (int k = 0; k < kDmaPerWave; ++k) { cache_issue(0, 0, 0, k); cache_issue(0, 1, 1, k); }
    chunk_sync();
    half8 bhi, blo;
    {
        const bool l2c = l2c_sv[0];
        load_taps(0);
        affine(0, 0);
#pragma unroll
        for (int k = 0; k < 12; ++k) { fetch(0, 0, k, l2c); blend(k); }
        finish(bhi, blo);
    }
    __syncthreads();                                                   // every wave is done with cache buffer 0 before chunk 2 lands in it

    f32x4 acc[kTE];
    float m0 = 0.0f;
#pragma unroll 1
    for (int sv = 0; sv < 2; ++sv) {
        init_bias<kTE>(acc, lds + kLdsBias + kBiasE, q4, e_up);
#pragma unroll 1
        for (int c = 0; c < kKS; ++c) {
            // gathered during this chunk: chunk m+1 = (nsv, nc) from cache buffer (m+1) & 1; fetched into the cache: chunk m+2 =
            // (n2sv, n2c) into buffer m & 1 (free since the barrier that ended chunk m-1).  Past the end: harmless repeats.
            const int m = sv * kKS + c;
            const int nsv = (c + 1 < kKS) ? sv : 1, nc = (c + 1 < kKS) ? c + 1 : 0;
            const int n2sv = (c + 2 < kKS) ? sv : 1, n2c = (c + 2 < kKS) ? c + 2 : c + 2 - kKS;
            const int nbuf = (m + 1) & 1, n2buf = m & 1;
            const bool l2c = l2c_sv[nsv];
            if (c + 1 == kKS && sv == 0) load_taps(1);                 // the gather moves on to the other source's taps
            // 10 slots (5 per weight half chunk: four tile pairs and a single tile); the 12 gather taps are dealt over them (two in the
            // first slot of each half): a tap's 32 bytes are requested before the slot's MFMAs and blended after them.  The DMA rides
            // along: the three pieces of the next weight half chunk in slots 0-2, the next-but-one chunk's cache slices in slots 2-4.
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const float* wl = lds + kLdsWt + (g & 1) * kWBuf + 4 * lane;
#pragma unroll
                for (int qs = 0; qs < 5; ++qs) {
                    const int k = 6 * half + qs + (qs > 0 ? 1 : 0);    // tap of this slot (slot 0: taps 6 half, 6 half + 1)
                    if (half == 0 && qs == 0) affine(nsv, nc);
                    if (qs == 0) { fetch(nc, nbuf, k, l2c); blend(k); fetch(nc, nbuf, k + 1, l2c); }
                    else fetch(nc, nbuf, k, l2c);
                    if constexpr (ABL != 4) {
                        const float* w0 = wl + (2 * qs * 2) * 256;
                        if (qs < 4) mfma_pair(acc[kHalf * half + 2 * qs], acc[kHalf * half + 2 * qs + 1], w0, w0 + 512, bhi, blo);
                        else mfma_single(acc[kHalf * half + 8], w0, bhi, blo);
                    }
                    if (qs < kPieces) { if constexpr (ABL != 5) weights_issue(a.blob, lds, g + 1, qs, lane, wave); }
                    if (qs >= 2) cache_issue(n2sv, n2c, n2buf, 3 * half + qs - 2);
                    blend(qs == 0 ? k + 1 : k);
                    if constexpr (ABL != 6) __builtin_amdgcn_sched_barrier(0);
                }
                if (half == 1) finish(bhi, blo);                       // the next chunk's B operand
                if constexpr (ABL != 5) chunk_sync();
                ++g;
            }
        }
        scale_acc<kTE>(acc, e_down);
        if (sv == 0) {
            m0 = sample_max<kTE, false>(acc);
            if (live) store_rows<kTE>(acc, a.e + i * (2 * kE), q4);
        }
    }
    // ---- k1 = Wk1 [e_0 ; e_1] + bk1 (e_1 chained from the accumulators, then e_0 read back), key, qry, logit: as car_fused.hip ----
    float p, pinv;
    pow2_scale(fmaxf(fmaxf(m0, sample_max<kTE, false>(acc)), 1e-30f), p, pinv);
    f32x4 k1[kTD];
    init_bias<kTD>(k1, lds + kLdsBias + kBiasK1, q4, p / lsc[kLayerK1]);
    chained_layer1<kTE, false>(k1, acc, p, a.blob, lds, g, lane, wave);
    if (live) store_rows<kTE>(acc, a.e + i * (2 * kE) + kE, q4);
#pragma unroll
    for (int t = 0; t < kTE; ++t) acc[t] = *reinterpret_cast<const f32x4*>(a.e + i * (2 * kE) + 16 * t + 4 * q4);
    chained_layer1<kTE, false>(k1, acc, p, a.blob, lds, g, lane, wave);
    scale_acc<kTD>(k1, lsc[kLayerK1] * pinv);
    f32x4 key[kTD];
    pow2_scale(fmaxf(sample_max<kTD, true>(k1), 1e-30f), p, pinv);
    init_bias<kTD>(key, lds + kLdsBias + kBiasK2, q4, p / lsc[kLayerK2]);
    chained_layer1<kTD, true>(key, k1, p, a.blob, lds, g, lane, wave);
    scale_acc<kTD>(key, lsc[kLayerK2] * pinv);

    half8 ghi, glo;                                                    // B operand of the layer fed by g (k = 16: folded bias)
    {
        // g was written to the output tensor by the geometry waves of this workgroup many barriers ago; this lane's sample row
        const float* gl = a.g + 16 * i + 8 * (q4 & 1);
        float gx8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) gx8[k] = q4 < 2 ? gl[k] : (q4 == 2 && k == 0) ? 1.0f : 0.0f;
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(gx8[k]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));                           // >= 1: the bias column
        pow2_scale(m, p, pinv);
        split8(gx8, p, ghi, glo);
    }
    f32x4 t1[kTD], qv[kTD];
#pragma unroll
    for (int t = 0; t < kTD; ++t) t1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p_ = 0; p_ < kPieces; ++p_) weights_issue(a.blob, lds, g + 1, p_, lane, wave);
    small_layer(t1, ghi, glo, lds + kLdsWt + (g & 1) * kWBuf + 4 * lane);                        // q1
    chunk_sync();
    ++g;
    scale_acc<kTD>(t1, lsc[kLayerQ1] * pinv);
    pow2_scale(fmaxf(sample_max<kTD, true>(t1), 1e-30f), p, pinv);
    init_bias<kTD>(qv, lds + kLdsBias + kBiasQ2, q4, p / lsc[kLayerQ2]);
    chained_layer1<kTD, true>(qv, t1, p, a.blob, lds, g, lane, wave);
    scale_acc<kTD>(qv, lsc[kLayerQ2] * pinv);
    float dot = 0.0f;
#pragma unroll
    for (int t = 0; t < kTD; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dot = fmaf(key[t][r], qv[t][r], dot);
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if (live) {
        store_rows<kTD>(qv, a.qry + i * kD, q4);
        if (q4 == 0) a.logit[i] = dot / 16.0f;
    }
}

}  // namespace

// First half of car_fused_samples (car_fused.hip): every sample group whose coarse levels fit the texel cache; the others are
// appended to `redo` (zeroed by the caller: [0] count, [1 + k] group index) for car_fused.hip's kernel.
static int launch_tex(int abl, const float* poses, const float* rays, const float* steps, const float* const* gmaps,
                      const int* level_h, const int* level_w, int n_levels, int C, const float* gmeta,
                      const float* wpt, const float* blob, const float* bias, int b, int V, int R, int P, int H, int W,
                      float* e, float* qry, float* g, float* logit, float* pt, float* pixel_val, int* redo, void* stream) {
    CAR_REQUIRE(poses && rays && steps && gmaps && level_h && level_w && gmeta && wpt && blob && bias, "car_fused_samples_tex: null input");
    CAR_REQUIRE(e && qry && g && logit && pt && pixel_val && redo, "car_fused_samples_tex: null output");
    CAR_REQUIRE(n_levels == 3 && C == kC && V == 2, "car_fused_samples_tex: built for 3 pyramid levels, C = %d, V = 2 (got %d, %d, %d)", kC, n_levels, C, V);
    CAR_REQUIRE(b > 0 && R > 0 && P > 0 && H > 1 && W > 1, "car_fused_samples_tex: bad sizes");
    TexArgs a;
    a.poses = (const CarPose*)poses; a.rays = (const CarRay*)rays; a.steps = steps;
    for (int l = 0; l < 3; ++l) {
        a.gmap[l] = gmaps[l]; a.gh[l] = level_h[l]; a.gw[l] = level_w[l];
        CAR_REQUIRE(a.gmap[l] && a.gh[l] > 0 && a.gw[l] > 0 && (long)b * V * a.gh[l] * a.gw[l] * (kC * 4) < 4294967296L,
                    "car_fused_samples_tex: bad level %d (a level's projected map must stay below 4 GiB per call: render fewer scenes per call)", l);
    }
    a.gmeta = gmeta; a.wpt = wpt; a.blob = blob; a.bias = bias;
    a.b = b; a.V = V; a.R = R; a.P = P; a.H = H; a.W = W;
    a.S = (long)b * V * R * P;
    a.e = e; a.qry = qry; a.g = g; a.logit = logit; a.pt = pt; a.pixel_val = pixel_val; a.redo = redo;
    const long groups = (long)b * V * car_div_up(R, kBundles * kRows) * car_div_up(P, kStepsPerGroup);
    void (*kern)(const TexArgs) = fused_tex_kernel<0>;
#ifdef CAR_ABLATION
    switch (abl) {
        case 1: kern = fused_tex_kernel<1>; break;   case 2: kern = fused_tex_kernel<2>; break;   case 3: kern = fused_tex_kernel<3>; break;
        case 4: kern = fused_tex_kernel<4>; break;   case 5: kern = fused_tex_kernel<5>; break;   case 6: kern = fused_tex_kernel<6>; break;
        default: break;
    }
#else
    (void)abl;
#endif
    hipError_t e1 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e1 != hipSuccess) { car_set_error("car_fused_samples_tex: cannot reserve %zu bytes of LDS: %s", kLdsBytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(kThreads), kLdsBytes, (hipStream_t)stream, a);
    CAR_CHECK_LAUNCH("car_fused_samples_tex");
    return CAR_OK;
}

extern "C" int car_fused_samples_tex(const float* poses, const float* rays, const float* steps, const float* const* gmaps,
                                     const int* level_h, const int* level_w, int n_levels, int C, const float* gmeta,
                                     const float* wpt, const float* blob, const float* bias, int b, int V, int R, int P, int H, int W,
                                     float* e, float* qry, float* g, float* logit, float* pt, float* pixel_val, int* redo, void* stream) {
    return launch_tex(0, poses, rays, steps, gmaps, level_h, level_w, n_levels, C, gmeta, wpt, blob, bias, b, V, R, P, H, W, e, qry, g, logit, pt,
                      pixel_val, redo, stream);
}
#ifdef CAR_ABLATION
extern "C" int car_fused_samples_tex_ablate(int abl, const float* poses, const float* rays, const float* steps, const float* const* gmaps,
                                            const int* level_h, const int* level_w, int n_levels, int C, const float* gmeta,
                                            const float* wpt, const float* blob, const float* bias, int b, int V, int R, int P, int H, int W,
                                            float* e, float* qry, float* g, float* logit, float* pt, float* pixel_val, int* redo, void* stream) {
    return launch_tex(abl, poses, rays, steps, gmaps, level_h, level_w, n_levels, C, gmeta, wpt, blob, bias, b, V, R, P, H, W, e, qry, g, logit, pt,
                      pixel_val, redo, stream);
}
#endif
