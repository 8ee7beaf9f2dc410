// car_fused_w8.hip — development-only candidate for the fused per-sample kernel (never part of the product; built by
// tools/build_dev.py with CAR_DEV_UNIT=car_fused_w8.hip, compared bit for bit with the product by tools/bench_fused.py 200).
// Same arithmetic, same weight blob, same outputs as car_fused.hip; different occupancy:
//   * TWO waves per SIMD (8 waves, 256 registers each) instead of three, each wave owning 32 samples = two 16-row MFMA tiles;
//   * an A-operand pair read from LDS feeds both sample tiles: 4 ds_read_b128 per 12 MFMAs instead of per 6 (the product's
//     LDS read traffic halves);
//   * a workgroup is 256 samples (32 rays x 8 steps) instead of 192: the weight stream and the chunk barrier are shared by a
//     third more samples.
// The price: one wave fewer per SIMD to hide the gather and the LDS latency behind.
#include "car_common.h"
#include "car_geom.h"
#include <type_traits>
#include <utility>
#ifndef CAR_W8_VAR
#define CAR_W8_VAR 0
#endif

namespace {

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>), in order
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int kWaves = 8, kNT = 2, kRows = 16 * kNT, kGroup = kWaves * kRows;      // 256 samples per workgroup
constexpr int kWaveRays = 8, kWaveSteps = kRows / kWaveRays;          // a wave's 32 rows: 8 rays x 4 steps
constexpr int kTileSteps = 8, kStepWaves = kTileSteps / kWaveSteps, kRayWaves = kWaves / kStepWaves, kTileRays = kRayWaves * kWaveRays;
static_assert(kStepWaves * kRayWaves == kWaves && kTileSteps % kWaveSteps == 0, "tile shape");
__device__ __forceinline__ int tile_ray(int w, int s) { return (w / kStepWaves) * kWaveRays + (s & (kWaveRays - 1)); }
__device__ __forceinline__ int tile_step(int w, int s) { return (w % kStepWaves) * kWaveSteps + s / kWaveRays; }
constexpr int kThreads = 64 * kWaves;

constexpr int kPieces = 5;                         // LDS-DMA pieces per chunk: 8 waves x 1 KB each (36 KB: the last piece wraps)
constexpr unsigned kDeadTap = 0xc0000000u;
constexpr long kMaxMapBytes = 0x80000000L;

#include "car_fused_mma.h"

constexpr int kLdsStage = kLdsW + 2 * kChunkTiles * kTile;      // [8][32][36]            h tiles, wave private     36 KB
constexpr int kLdsBias = kLdsStage + kGroup * kStageLd;
constexpr int kLdsG = kLdsBias + kBiasFloats;                   // [256][16]
constexpr int kLdsTapB = kLdsG + kGroup * 16;                   // [256][2] uint
constexpr int kLdsTapW = kLdsTapB + kGroup * 2;                 // [256][2][4]
constexpr int kLdsPe = kLdsTapW + kGroup * 8;                   // [256][2][4]
constexpr int kLdsWpt = kLdsPe + kGroup * 8;                    // [144][4][4]
constexpr int kLdsE0 = kLdsTapB;                                // [8][1024]              e_0 rows of a K step (the other buffer: the wave's h tile)
constexpr int kLdsFloats = (kLdsWpt + kC * 4 > kLdsE0 + kWaves * 1024) ? kLdsWpt + kC * 4 : kLdsE0 + kWaves * 1024;
constexpr size_t kLdsBytes = (size_t)kLdsFloats * sizeof(float);
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

struct FusedArgs {
    const CarPose* poses;
    const CarRay* rays;
    const float* steps;
    const float* lattice;
    int lh, lw, pad;
    float sx, sy;
    unsigned map_bytes;
    const float* gmeta;
    const float* wpt;
    const float* blob;
    const float* bias;
    int b, V, R, P, H, W;
    int no_sample;
    long S;
    float* e;
    float* qry;
    float* g;
    float* logit;
    float* pt;
    float* pixel_val;
};

constexpr int kChK1 = 5;
constexpr int kG_W2b = kKS, kG_K1b = 2 * kKS, kG_K1a = kG_K1b + kChK1, kG_K2 = kG_K1a + kChK1, kG_Q1 = kG_K2 + 2, kG_Q2 = kG_Q1 + 1;
static_assert(kG_Q2 + 2 == kNumChunks, "chunk count");
__device__ __forceinline__ constexpr int chunk_tile_offset(int g) {
    if (g < kG_W2b) return kOffW2 + g * kTE;
    if (g < kG_K1b) return kOffW2 + (g - kG_W2b) * kTE;
    if (g < kG_K1a) return kOffK1 + 9 * kTD + (g - kG_K1b) * 2 * kTD;
    if (g < kG_K2) return kOffK1 + (g - kG_K1a) * 2 * kTD;
    if (g < kG_Q1) return kOffK2 + (g - kG_K2) * 2 * kTD;
    if (g < kG_Q2) return kOffQ1;
    return kOffQ2 + (g - kG_Q2) * 2 * kTD;
}
__device__ __forceinline__ constexpr int chunk_tiles(int g) {
    if (g < kG_K1b) return kTE;
    if (g == kG_K1a - 1 || g == kG_K2 - 1 || g == kG_Q1) return kTD;
    return 2 * kTD;
}
__device__ __forceinline__ NextChunk next_chunk_w2(const float* __restrict__ blob, float* lds, int gn) {
    const bool w2 = gn < kG_K1b;
    const int step = gn >= kG_W2b ? gn - kG_W2b : gn;
    NextChunk n;
    n.src = blob + (long)(w2 ? kOffW2 + step * kTE : chunk_tile_offset(kG_K1b)) * kTile;
    n.dst = lds + kLdsW + (gn & 1) * kChunkTiles * kTile;
    n.nkb = w2 ? 2 * kTE : 2 * chunk_tiles(kG_K1b);
    return n;
}

// an A-operand pair (two output tiles, hi and lo halves: four LDS reads) against the B operands of BOTH sample tiles: twelve MFMAs,
// each accumulator in the product's order (hi*hi, hi*lo, lo*hi), consecutive MFMAs on different accumulators
__device__ __forceinline__ void mfma_quad(f32x4& c00, f32x4& c01, f32x4& c10, f32x4& c11, const float* w0, const float* w1,
                                          const half8 (&bhi)[kNT], const half8 (&blo)[kNT]) {
    const half8 ah0 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0));
    const half8 ah1 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w1));
    const half8 al0 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 256));
    const half8 al1 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w1 + 256));
    c00 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bhi[0], c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bhi[0], c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bhi[1], c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bhi[1], c11, 0, 0, 0);
    c00 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, blo[0], c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, blo[0], c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, blo[1], c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, blo[1], c11, 0, 0, 0);
    c00 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bhi[0], c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bhi[0], c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bhi[1], c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bhi[1], c11, 0, 0, 0);
}

__device__ __forceinline__ void load_a(const float* w0, half8 (&a)[4]) {
    a[0] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0));
    a[1] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 512));
    a[2] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 256));
    a[3] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 768));
}
__device__ __forceinline__ void mma_quad(f32x4& c00, f32x4& c01, f32x4& c10, f32x4& c11, const half8 (&a)[4], const half8 (&bhi)[kNT], const half8 (&blo)[kNT]) {
    c00 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bhi[0], c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], bhi[0], c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bhi[1], c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], bhi[1], c11, 0, 0, 0);
    c00 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], blo[0], c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], blo[0], c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], blo[1], c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], blo[1], c11, 0, 0, 0);
    c00 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2], bhi[0], c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[3], bhi[0], c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2], bhi[1], c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[3], bhi[1], c11, 0, 0, 0);
}

// chained layer over both sample tiles (car_fused_mma.h chained_layer, one A read per pair of sample tiles)
template <int NSRC, bool RELU, int G0, int HOOK_OPS = 0, class Hook = NoHook>
__device__ __forceinline__ void chained_layer2(f32x4 (&acc)[kNT][kTD], const f32x4 (&src)[kNT][NSRC], const float (&p)[kNT],
                                               const float* __restrict__ blob, float* lds, int lane, int wave, Hook after = Hook()) {
    constexpr int kSteps = NSRC / 2;
#pragma unroll
    for (int m0 = 0; m0 < kSteps; m0 += 2) {
        const int nks = m0 + 1 < kSteps ? 2 : 1;
        const int g = G0 + m0 / 2;
        const float* wl = lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane;
        const NextChunk nx = next_chunk(blob, lds, g + 1);
#pragma unroll
        for (int kl = 0; kl < 2; ++kl) {
            if (kl < nks) {
                const int m = m0 + kl < kSteps ? m0 + kl : kSteps - 1;
                half8 bhi[kNT], blo[kNT];
#pragma unroll
                for (int st = 0; st < kNT; ++st) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        x[e] = src[st][2 * m + (e >> 2)][e & 3];
                        if (RELU) x[e] = fmaxf(x[e], 0.f);
                    }
                    split8(x, p[st], bhi[st], blo[st]);
                }
#pragma unroll
                for (int q = 0; q < kTD / 2; ++q) {
                    const float* w0 = wl + ((kl * kTD + 2 * q) * 2) * 256;
                    mfma_quad(acc[0][2 * q], acc[0][2 * q + 1], acc[1][2 * q], acc[1][2 * q + 1], w0, w0 + 512, bhi, blo);
                    if (kl == 0) {
                        if (q < 3) { if (q < kPieces) stream_issue_piece(nx, q, lane, wave); }
                        else {
#pragma unroll
                            for (int p_ = 3; p_ < kPieces; ++p_) stream_issue_piece(nx, p_, lane, wave);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                after(m);
            }
        }
        if (nks == 2) stream_sync<0, 2 * HOOK_OPS>(); else stream_sync<0, HOOK_OPS>();
    }
}

__global__ void __launch_bounds__(kThreads) fused_kernel_w8(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, q4 = lane >> 4;
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    {
        const int q8 = nblk / 8, r8 = nblk % 8, xcd = blk % 8, idx = blk / 8;
        blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int pgs = (a.P + kTileSteps - 1) / kTileSteps, bundles = (a.R + kTileRays - 1) / kTileRays;
    const int pg = blk % pgs, bun = (blk / pgs) % bundles, nn = blk / (pgs * bundles);
    // sample of row r of this wave's tile, relative to the workgroup's first ray at step 0 (i_base): 32-bit byte offsets from scalar bases
    const long i_base = ((long)nn * a.R + (long)bun * kTileRays) * a.P;
    auto row_rel = [&](int r) -> int {                                 // sample index - i_base of row r (clamped to a live sample)
        const int ray_r = bun * kTileRays + tile_ray(wave, r), pp_r = pg * kTileSteps + tile_step(wave, r);
        return ((ray_r < a.R ? ray_r : a.R - 1) - bun * kTileRays) * a.P + (pp_r < a.P ? pp_r : a.P - 1);
    };
    auto row_live = [&](int r) -> bool {
        return bun * kTileRays + tile_ray(wave, r) < a.R && pg * kTileSteps + tile_step(wave, r) < a.P;
    };

    float hp, hinv;
    pow2_scale(fmaxf(a.gmeta[0] + a.bias[kBiasScale + 5], 1e-30f), hp, hinv);
    for (int k = tid; k < kC; k += kThreads) {
        const float4 v = *reinterpret_cast<const float4*>(a.wpt + 4 * k);
        float* q = lds + kLdsWpt + 16 * (k >> 2) + (k & 3);
        q[0] = v.x; q[4] = v.y; q[8] = v.z; q[12] = v.w * hp;
    }
    for (int k = tid; k < kBiasFloats; k += kThreads) lds[kLdsBias + k] = a.bias[k];
    int g = 0;
    stream_issue_all(a.blob, lds, 0, lane, wave);

    // ---- geometry: one sample per lane of waves 0-3 ----
    const int P = a.P, V = a.V;
    if (wave < kGroup / 64) {
        const int sg = wave * 64 + lane, gwv = sg / kRows, gs = sg % kRows;
        const int g_ray = bun * kTileRays + tile_ray(gwv, gs), g_pp = pg * kTileSteps + tile_step(gwv, gs);
        const bool g_live = g_ray < a.R && g_pp < a.P;
        const long gi = ((long)nn * a.R + (g_ray < a.R ? g_ray : a.R - 1)) * a.P + (g_pp < a.P ? g_pp : a.P - 1);
        const int p = (int)(gi % P);
        const long nr = gi / P;
        const int n = (int)(nr / a.R);
        const int v = n % V, sc = n / V;
        const CarPose& Ps = a.poses[n];
        const CarRay ray = a.rays[nr];
        CarSample smp;
        if (!a.no_sample) {
            for (int k = 0; k < 2; ++k) smp.grid[k] = ray.start[k] + (ray.end[k] - ray.start[k]) * a.steps[p];
        } else {
            const float sd = a.steps[p];
            const float q[3] = {Ps.q_rel[3] + sd * ray.d[0], Ps.q_rel[7] + sd * ray.d[1], Ps.q_rel[11] + sd * ray.d[2]};
            car_project_grid(Ps.kc, q, a.H, a.W, smp.grid);
        }
        car_sample_setup(Ps, a.poses + sc * 2, ray, 2, a.H, a.W, &smp);
#pragma unroll
        for (int sv = 0; sv < 2; ++sv) {
            float gx, gy;
            int mode;
            if (sv == v) { gx = smp.grid[0]; gy = smp.grid[1]; mode = 0; }
            else { gx = sv == 0 ? smp.grid_in[0][0] : smp.grid_in[1][0]; gy = sv == 0 ? smp.grid_in[0][1] : smp.grid_in[1][1]; mode = 1; }
            int node, flags;
            float w[4];
            car_lattice_taps(gx, gy, a.lw, a.lh, a.pad, a.sx, a.sy, &node, &flags, w);
            const bool dead = mode == 1 && (flags & 4);
            const unsigned tap_off = (unsigned)node * (unsigned)(kC * 4);
            reinterpret_cast<unsigned*>(lds + kLdsTapB)[sg * 2 + sv] = dead ? kDeadTap : tap_off;
            *reinterpret_cast<float4*>(lds + kLdsTapW + (sg * 2 + sv) * 4) =
                dead ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(w[0] * hp, w[1] * hp, w[2] * hp, w[3] * hp);
            const float px = sv == 0 ? smp.pt_in[0][0] : smp.pt_in[1][0], py = sv == 0 ? smp.pt_in[0][1] : smp.pt_in[1][1],
                        pz = sv == 0 ? smp.pt_in[0][2] : smp.pt_in[1][2];
            *reinterpret_cast<float4*>(lds + kLdsPe + (sg * 2 + sv) * 4) = make_float4(tanhf(px / 5.0f) * hp, tanhf(py / 5.0f) * hp, tanhf(pz / 5.0f) * hp, 0.0f);
        }
        if (g_live) {
#if CAR_W8_VAR != 9
            a.pixel_val[2 * gi] = smp.grid[0]; a.pixel_val[2 * gi + 1] = smp.grid[1];
#endif
            a.pt[3 * gi + 0] = smp.pt[0]; a.pt[3 * gi + 1] = smp.pt[1]; a.pt[3 * gi + 2] = smp.pt[2];
        }
        float* gl = lds + kLdsG + sg * 16;
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            const float4 g4 = make_float4(smp.g[k], smp.g[k + 1], smp.g[k + 2], smp.g[k + 3]);
            *reinterpret_cast<float4*>(gl + k) = g4;
            if (g_live) *reinterpret_cast<float4*>(a.g + 16 * gi + k) = g4;
        }
    }
    __syncthreads();

    // ---- gather machinery: lane owns rows rr = (lane >> 3) + 8 it (it = 0..3: one step of the wave's 8 rays each) and channel quad
    //      qd = lane & 7.  Two tap batches in flight (tapA: row groups 0 and 2, tapB: 1 and 3), each issued four slots ahead ----
    const int qd = lane & 7, r0 = lane >> 3;
    float* stage = lds + kLdsStage + wave * kRows * kStageLd;
    f32x4 tapA[4], tapB[4];
    const unsigned qd16 = 16u * qd;
    const unsigned row_step = (unsigned)a.lw * (kC * 4);
    const int v_own = nn % a.V, sc_own = nn / a.V;
    const long map_floats = (long)a.lh * a.lw * kC;
    const __amdgpu_buffer_rsrc_t rsrc[2] = {
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lattice + ((long)(sc_own * a.V + 0) * 2 + (v_own == 0 ? 0 : 1)) * map_floats), 0,
                                          (int)a.map_bytes, 0x00027000),
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lattice + ((long)(sc_own * a.V + 1) * 2 + (v_own == 1 ? 0 : 1)) * map_floats), 0,
                                          (int)a.map_bytes, 0x00027000)};
    auto issue_row = [&](f32x4 (&tap)[4], int sv, int c, int it) {
        const int chunk_off = 128 * c;
        const unsigned tbv = reinterpret_cast<const unsigned*>(lds + kLdsTapB)[(wave * kRows + r0 + 8 * it) * 2 + sv];
        const unsigned o00 = tbv + qd16, o10 = o00 + row_step;
        auto ld = [&](unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc[sv], (int)off, chunk_off, 0)); };
        tap[0] = ld(o00);
        tap[1] = ld(o00 + (unsigned)(kC * 4));
        tap[2] = ld(o10);
        tap[3] = ld(o10 + (unsigned)(kC * 4));
    };
    // a row group of the chunk's h: point / bias term + the four taps, relu, into the wave's h tile
    auto gather_row = [&](const f32x4 (&tap)[4], int sv, int c, int it) {
        const int rr = r0 + 8 * it;
        const float4 pe = *reinterpret_cast<const float4*>(lds + kLdsPe + ((wave * kRows + rr) * 2 + sv) * 4);
        const float4* wp = reinterpret_cast<const float4*>(lds + kLdsWpt + 16 * (8 * c + qd));
        const float4 wx = wp[0], wy = wp[1], wz = wp[2], wb = wp[3];
        const float4 h0 = make_float4(fmaf(wx.x, pe.x, fmaf(wy.x, pe.y, fmaf(wz.x, pe.z, wb.x))), fmaf(wx.y, pe.x, fmaf(wy.y, pe.y, fmaf(wz.y, pe.z, wb.y))),
                                      fmaf(wx.z, pe.x, fmaf(wy.z, pe.y, fmaf(wz.z, pe.z, wb.z))), fmaf(wx.w, pe.x, fmaf(wy.w, pe.y, fmaf(wz.w, pe.z, wb.w))));
        const float4 w = *reinterpret_cast<const float4*>(lds + kLdsTapW + ((wave * kRows + rr) * 2 + sv) * 4);
        const float ww[4] = {w.x, w.y, w.z, w.w};
        f32x2 lo2 = {h0.x, h0.y}, hi2 = {h0.z, h0.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 gq = tap[t];
            const f32x2 w2 = {ww[t], ww[t]};
            lo2 = __builtin_elementwise_fma(w2, f32x2{gq[0], gq[1]}, lo2);
            hi2 = __builtin_elementwise_fma(w2, f32x2{gq[2], gq[3]}, hi2);
        }
        *reinterpret_cast<float4*>(stage + rr * kStageLd + 4 * qd) =
            make_float4(fmaxf(lo2[0], 0.f), fmaxf(lo2[1], 0.f), fmaxf(hi2[0], 0.f), fmaxf(hi2[1], 0.f));
    };
    // ---- variant 10: the gather as a micro-program, one step per MFMA of a slot pair (24 steps per row group), at most two vector
    //      instructions or LDS / memory instructions per step; h goes into the wave's tile already split into fp16 hi / lo halves
    //      (row = 32 hi halves | 32 lo halves), so the B operands are plain 16-byte reads ----
    constexpr bool kSplitH = (CAR_W8_VAR == 10);
    float4 q_pe, q_w, q_wx, q_wy, q_wz, q_h;
    unsigned q_tbv = 0, q_o00 = 0, q_o10 = 0, q_h01 = 0, q_h23 = 0, q_l01 = 0, q_l23 = 0;
    float q_m0 = 0.f, q_m1 = 0.f;
    auto split_lo = [](unsigned hi, float x, float y) {
        unsigned lo;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(lo) : "v"(hi), "v"(x), "v"(y));
        return lo;
    };
    // step pp (0..23) of row group `it` of chunk (sv, c), blending `tap`; steps 20-23 re-issue `tap` for row group iit of chunk (isv, ic)
    auto gstep = [&](auto ppc, f32x4 (&tap)[4], int sv, int c, int it, bool issue, int isv, int ic, int iit) {
        constexpr int pp = decltype(ppc)::value;
        const int rr = r0 + 8 * it;
        const int rec = ((wave * kRows + rr) * 2 + sv) * 4;
        const float4* wp = reinterpret_cast<const float4*>(lds + kLdsWpt + 16 * (8 * c + qd));
        const float ww[4] = {q_w.x, q_w.y, q_w.z, q_w.w};
        auto ld = [&](unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc[isv], (int)off, 128 * ic, 0)); };
        if constexpr (pp == 0) {
            q_pe = *reinterpret_cast<const float4*>(lds + kLdsPe + rec);
            q_h = wp[3];
            if (issue) q_tbv = reinterpret_cast<const unsigned*>(lds + kLdsTapB)[(wave * kRows + r0 + 8 * iit) * 2 + isv];
        } else if constexpr (pp == 1) {
            q_wz = wp[2];
        } else if constexpr (pp == 3) {
            q_wy = wp[1];
        } else if constexpr (pp == 5) {
            q_wx = wp[0];
            if (issue) { q_o00 = q_tbv + qd16; q_o10 = q_o00 + row_step; }
        } else if constexpr (pp == 6) { q_h.x = fmaf(q_wz.x, q_pe.z, q_h.x); q_h.y = fmaf(q_wz.y, q_pe.z, q_h.y); }
        else if constexpr (pp == 7) { q_h.z = fmaf(q_wz.z, q_pe.z, q_h.z); q_h.w = fmaf(q_wz.w, q_pe.z, q_h.w); q_w = *reinterpret_cast<const float4*>(lds + kLdsTapW + rec); }
        else if constexpr (pp == 8) { q_h.x = fmaf(q_wy.x, q_pe.y, q_h.x); q_h.y = fmaf(q_wy.y, q_pe.y, q_h.y); }
        else if constexpr (pp == 9) { q_h.z = fmaf(q_wy.z, q_pe.y, q_h.z); q_h.w = fmaf(q_wy.w, q_pe.y, q_h.w); }
        else if constexpr (pp == 10) { q_h.x = fmaf(q_wx.x, q_pe.x, q_h.x); q_h.y = fmaf(q_wx.y, q_pe.x, q_h.y); }
        else if constexpr (pp == 11) { q_h.z = fmaf(q_wx.z, q_pe.x, q_h.z); q_h.w = fmaf(q_wx.w, q_pe.x, q_h.w); }
        else if constexpr (pp >= 12 && pp < 20) {
            constexpr int t = (pp - 12) >> 1;
            if constexpr ((pp & 1) == 0) { q_h.x = fmaf(ww[t], tap[t][0], q_h.x); q_h.y = fmaf(ww[t], tap[t][1], q_h.y); }
            else { q_h.z = fmaf(ww[t], tap[t][2], q_h.z); q_h.w = fmaf(ww[t], tap[t][3], q_h.w); }
        } else if constexpr (pp == 20) {
            q_m0 = fmaxf(q_h.x, 0.f); q_m1 = fmaxf(q_h.y, 0.f);
            q_h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(q_m0, q_m1));
            if (issue) tap[0] = ld(q_o00);
        } else if constexpr (pp == 21) {
            q_l01 = split_lo(q_h01, q_m0, q_m1);
            if (issue) tap[1] = ld(q_o00 + (unsigned)(kC * 4));
        } else if constexpr (pp == 22) {
            q_m0 = fmaxf(q_h.z, 0.f); q_m1 = fmaxf(q_h.w, 0.f);
            q_h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(q_m0, q_m1));
            if (issue) tap[2] = ld(q_o10);
        } else if constexpr (pp == 23) {
            q_l23 = split_lo(q_h23, q_m0, q_m1);
            *reinterpret_cast<uint2*>(stage + rr * kStageLd + 2 * qd) = make_uint2(q_h01, q_h23);
            *reinterpret_cast<uint2*>(stage + rr * kStageLd + 16 + 2 * qd) = make_uint2(q_l01, q_l23);
            if (issue) tap[3] = ld(q_o10 + (unsigned)(kC * 4));
        }
    };
    const float* lsc = lds + kLdsBias + kBiasScale;
    auto uniform = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    float e_up, e_down;
    {
        const float dW2 = lsc[kLayerW2];
        e_up = uniform(hp / dW2); e_down = uniform(dW2 * hinv);
    }
    auto read_b = [&](half8 (&bhi)[kNT], half8 (&blo)[kNT]) {
        if constexpr (kSplitH) {
#pragma unroll
            for (int st = 0; st < kNT; ++st) {
                bhi[st] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(stage + (16 * st + s) * kStageLd + 4 * q4));
                blo[st] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(stage + (16 * st + s) * kStageLd + 16 + 4 * q4));
            }
            return;
        }
#pragma unroll
        for (int st = 0; st < kNT; ++st) {
            const float4 x0 = *reinterpret_cast<const float4*>(stage + (16 * st + s) * kStageLd + 8 * q4);
            const float4 x1 = *reinterpret_cast<const float4*>(stage + (16 * st + s) * kStageLd + 8 * q4 + 4);
            const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            split8_scaled(x, bhi[st], blo[st]);
        }
    };

    // first chunk of source 0: nothing to hide it under
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        issue_row(tapA, 0, 0, it);
        if constexpr (kSplitH) {
            static_for<24>([&](auto ppc) { gstep(ppc, tapA, 0, 0, it, false, 0, 0, 0); });
        } else
        gather_row(tapA, 0, 0, it);
    }
    stream_sync();                                                     // weight chunk 0 landed
    issue_row(tapA, 0, 1, 0);                                          // pipeline prologue: chunk (0, 1), row groups 0 and 1
    issue_row(tapB, 0, 1, 1);

    // row side of the wave's tile: lane (r0, qd) owns 16 bytes of rows r0 + 8 it — eight lanes per 128-byte line of e
    float* const e_base = a.e + i_base * (2 * kE);                     // scalar
    unsigned e_off[4];                                                 // byte offset of row r0 + 8 it's channel quad qd
#pragma unroll
    for (int it = 0; it < 4; ++it) e_off[it] = (unsigned)row_rel(r0 + 8 * it) * (unsigned)(2 * kE * 4) + qd16;
    // a K step's two tiles per sample tile (32 channels of 32 rows), turned through the wave's idle h tile and out as whole lines
    auto store_tiles_at = [&](const f32x4 (&t)[kNT][kTE], int m, int col0) {
#pragma unroll
        for (int st = 0; st < kNT; ++st)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<float4*>(stage + (16 * st + s) * kStageLd + 16 * j + 4 * q4) =
                    make_float4(t[st][2 * m + j][0], t[st][2 * m + j][1], t[st][2 * m + j][2], t[st][2 * m + j][3]);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float4 v = *reinterpret_cast<const float4*>(stage + (r0 + 8 * it) * kStageLd + 4 * qd);
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(e_base) + (e_off[it] + (unsigned)((col0 + 32 * m) * 4))) = v;
        }
    };

    f32x4 acc[kNT][kTE];
#if CAR_W8_VAR == 9
    long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    float m0[kNT] = {0.0f, 0.0f};
    half8 bhi[kNT], blo[kNT];
    read_b(bhi, blo);
#pragma unroll 1
    for (int sv = 0; sv < 2; ++sv) {
#pragma unroll
        for (int st = 0; st < kNT; ++st) init_bias<kTE>(acc[st], lds + kLdsBias + kBiasE, q4, e_up);
#pragma unroll 1
        for (int c = 0; c < kKS; ++c) {
            const int nsv = (c + 1 < kKS) ? sv : 1;
            const int nc = (c + 1 < kKS) ? c + 1 : 0;
            const int n2sv = (c + 2 < kKS) ? sv : 1;
            const int n2c = (c + 2 < kKS) ? c + 2 : c + 2 - kKS;
            const float* wl = lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane;
            const NextChunk nx = next_chunk_w2(a.blob, lds, g + 1);
            // 9 slots of (4 ds_read_b128 + 12 MFMAs); between them one piece of the gather / DMA issue: slots 0-4 the DMA pieces; slots 1, 3, 5, 7
            // one row group each of the next chunk — slots 1 and 3 re-issue their buffer for row groups 2 and 3 of the same chunk, slots 5 and 7
            // for row groups 0 and 1 of the chunk after (those 8 loads stay in flight over the barrier)
            auto piece = [&](int qs) {
                if (qs < kPieces) stream_issue_piece(nx, qs, lane, wave);
                if (qs == 1) { gather_row(tapA, nsv, nc, 0); issue_row(tapA, nsv, nc, 2); }
                else if (qs == 3) { gather_row(tapB, nsv, nc, 1); issue_row(tapB, nsv, nc, 3); }
                else if (qs == 5) { gather_row(tapA, nsv, nc, 2); issue_row(tapA, n2sv, n2c, 0); }
                else if (qs == 7) { gather_row(tapB, nsv, nc, 3); issue_row(tapB, n2sv, n2c, 1); }
            };
#if CAR_W8_VAR == 0
#pragma unroll
            for (int qs = 0; qs < kTE / 2; ++qs) {
                const float* w0 = wl + (2 * qs * 2) * 256;
                mfma_quad(acc[0][2 * qs], acc[0][2 * qs + 1], acc[1][2 * qs], acc[1][2 * qs + 1], w0, w0 + 512, bhi, blo);
                piece(qs);
                __builtin_amdgcn_sched_barrier(0);
            }
#elif CAR_W8_VAR == 10
            // One A buffer, refilled in place: the slot's MFMAs go (hi x hi) c00 c10 c01 c11, (hi x lo) the same, (lo x hi) the same — every
            // accumulator in the product's order — so a[0] (hi, tile 0) is last read by MFMA 5, a[1] by 7, a[2] by 9, a[3] by 11: the next slot's
            // operand is read into the register right after, six or more MFMAs ahead of its use.
            half8 a4[4];
            load_a(wl, a4);                                            // slot 0: read after the barrier that published the buffer
            static_for<12 * (kTE / 2)>([&](auto pc) {
                constexpr int p = decltype(pc)::value, qs = p / 12, k = p % 12;
                {
                    constexpr int st = k % 2, j = (k % 4) / 2;
                    acc[st][2 * qs + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[(k >= 8 ? 2 : 0) + j], (k >= 4 && k < 8) ? blo[st] : bhi[st], acc[st][2 * qs + j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (qs + 1 < kTE / 2 && (k == 6 || k == 8 || k == 10 || k == 11)) {
                    constexpr int ai = k == 6 ? 0 : k == 8 ? 1 : k == 10 ? 2 : 3;
                    const float* wn = wl + (2 * (qs + 1) * 2) * 256 + (ai == 0 ? 0 : ai == 1 ? 512 : ai == 2 ? 256 : 768);
                    a4[ai] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wn));
                }
                if constexpr (qs < kPieces && k == 4) stream_issue_piece(nx, qs, lane, wave);
                if constexpr (qs < 8) {
                    constexpr int it = qs / 2;
                    const std::integral_constant<int, p - 24 * it> ppc;
                    if constexpr (it == 0) gstep(ppc, tapA, nsv, nc, 0, true, nsv, nc, 2);
                    else if constexpr (it == 1) gstep(ppc, tapB, nsv, nc, 1, true, nsv, nc, 3);
                    else if constexpr (it == 2) gstep(ppc, tapA, nsv, nc, 2, true, n2sv, n2c, 0);
                    else gstep(ppc, tapB, nsv, nc, 3, true, n2sv, n2c, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#elif CAR_W8_VAR == 9
            // shader-clock sums per section of the chunk loop (results stay right except pixel_val, which receives the sums)
#pragma unroll
            for (int qs = 0; qs < kTE / 2; ++qs) {
                const float* w0 = wl + (2 * qs * 2) * 256;
                const long long t0 = __builtin_amdgcn_s_memtime();
                mfma_quad(acc[0][2 * qs], acc[0][2 * qs + 1], acc[1][2 * qs], acc[1][2 * qs + 1], w0, w0 + 512, bhi, blo);
                __builtin_amdgcn_sched_barrier(0);
                const long long t1 = __builtin_amdgcn_s_memtime();
                if (qs < kPieces) stream_issue_piece(nx, qs, lane, wave);
                __builtin_amdgcn_sched_barrier(0);
                const long long t2 = __builtin_amdgcn_s_memtime();
                if (qs == 1) gather_row(tapA, nsv, nc, 0); else if (qs == 3) gather_row(tapB, nsv, nc, 1);
                else if (qs == 5) gather_row(tapA, nsv, nc, 2); else if (qs == 7) gather_row(tapB, nsv, nc, 3);
                __builtin_amdgcn_sched_barrier(0);
                const long long t3 = __builtin_amdgcn_s_memtime();
                if (qs == 1) issue_row(tapA, nsv, nc, 2); else if (qs == 3) issue_row(tapB, nsv, nc, 3);
                else if (qs == 5) issue_row(tapA, n2sv, n2c, 0); else if (qs == 7) issue_row(tapB, n2sv, n2c, 1);
                __builtin_amdgcn_sched_barrier(0);
                const long long t4 = __builtin_amdgcn_s_memtime();
                tsum[0] += t1 - t0; tsum[1] += t2 - t1; tsum[2] += t3 - t2; tsum[3] += t4 - t3;
            }
            {
                const long long t0 = __builtin_amdgcn_s_memtime();
                read_b(bhi, blo);
                __builtin_amdgcn_sched_barrier(0);
                const long long t1 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                const long long t2 = __builtin_amdgcn_s_memtime();
                __syncthreads();
                const long long t3 = __builtin_amdgcn_s_memtime();
                tsum[4] += t1 - t0; tsum[5] += t2 - t1; tsum[6] += t3 - t2; tsum[7] += 1;
            }
            ++g;
            continue;
#else
            // variants: 1 the next slot's A operands are read right after this slot's MFMAs have issued (same registers), under the piece;
            // 2 double-buffered A operands (read a whole slot ahead); 3 = 1 with the piece in front of the MFMAs
            half8 a0[4], a1[4];
            load_a(wl, a0);
#pragma unroll
            for (int qs = 0; qs < kTE / 2; ++qs) {
                const float* wn = wl + (2 * (qs + 1) * 2) * 256;
#if CAR_W8_VAR == 2
                if (qs + 1 < kTE / 2) { if (qs & 1) load_a(wn, a0); else load_a(wn, a1); }
                if (qs & 1) mma_quad(acc[0][2 * qs], acc[0][2 * qs + 1], acc[1][2 * qs], acc[1][2 * qs + 1], a1, bhi, blo);
                else mma_quad(acc[0][2 * qs], acc[0][2 * qs + 1], acc[1][2 * qs], acc[1][2 * qs + 1], a0, bhi, blo);
                piece(qs);
#elif CAR_W8_VAR == 3
                piece(qs);
                mma_quad(acc[0][2 * qs], acc[0][2 * qs + 1], acc[1][2 * qs], acc[1][2 * qs + 1], a0, bhi, blo);
                if (qs + 1 < kTE / 2) load_a(wn, a0);
#else
                mma_quad(acc[0][2 * qs], acc[0][2 * qs + 1], acc[1][2 * qs], acc[1][2 * qs + 1], a0, bhi, blo);
                if (qs + 1 < kTE / 2) load_a(wn, a0);
                piece(qs);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            read_b(bhi, blo);
            stream_sync<0, 8>();
            ++g;
        }
#pragma unroll
        for (int st = 0; st < kNT; ++st) scale_acc<kTE>(acc[st], e_down);
        if (sv == 0) {
#pragma unroll
            for (int st = 0; st < kNT; ++st) m0[st] = sample_max<kTE, false>(acc[st]);
#pragma unroll
            for (int m = 0; m < kTE / 2; ++m) store_tiles_at(acc, m, 0);
        }
    }
#if CAR_W8_VAR == 9
    const long long t_passes = __builtin_amdgcn_s_memtime() - t_begin;
#endif
    // ---- k1 = Wk1 [e_0 ; e_1] + bk1 ----
    float p[kNT], pinv[kNT];
    f32x4 k1[kNT][kTD];
#pragma unroll
    for (int st = 0; st < kNT; ++st) {
        pow2_scale(fmaxf(fmaxf(m0[st], sample_max<kTE, false>(acc[st])), 1e-30f), p[st], pinv[st]);
        init_bias<kTD>(k1[st], lds + kLdsBias + kBiasK1, q4, p[st] / lsc[kLayerK1]);
    }
    auto store_tiles = [&](int m) { store_tiles_at(acc, m, kE); };
    chained_layer2<kTE, false, kG_K1b, 4>(k1, acc, p, a.blob, lds, lane, wave, store_tiles);
    {
        // lane (r0, qd) fetches 16 bytes of row rr = r0 + 8 it; a row's eight segments are stored rotated by (rr >> 1) & 7
        unsigned esrc[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) esrc[it] = e_off[it] - qd16 + 16u * (unsigned)(qd ^ (((r0 + 8 * it) >> 1) & 7));
        float* const ebuf[2] = {stage, lds + kLdsE0 + wave * 1024};
        auto issue_e0 = [&](int m) {                                   // K step m (channels 32 m .. 32 m + 31) -> buffer m & 1: four loads
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(ebuf[m & 1] + it * 256));
                const unsigned voff = esrc[it] + 128u * (unsigned)m;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(e_base) : "memory");
            }
        };
        auto wait_vm = [&](auto n) {
            constexpr int N = decltype(n)::value;
            if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        constexpr int kSteps = kTE / 2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's e_0 / e_1 stores are out before their lines are read back
        issue_e0(0);
        issue_e0(1);
#pragma unroll
        for (int c = 0; c < kChK1; ++c) {
            const int gk = kG_K1a + c;
            const float* wl = lds + kLdsW + (gk & 1) * kChunkTiles * kTile + 4 * lane;
            const NextChunk nx = next_chunk(a.blob, lds, gk + 1);
#pragma unroll
            for (int kl = 0; kl < 2; ++kl) {
                const int m = 2 * c + kl;
                if (m < kSteps) {
                    // younger than e_0(m)'s four loads: e_0(m + 1)'s four, and for the second K step of a chunk its five weight pieces in between
                    if (kl == 0) { if (m + 1 < kSteps) wait_vm(std::integral_constant<int, 4>()); else wait_vm(std::integral_constant<int, 0>()); }
                    else { if (m + 1 < kSteps) wait_vm(std::integral_constant<int, 9>()); else wait_vm(std::integral_constant<int, 5>()); }
                    half8 ehi[kNT], elo[kNT];
#pragma unroll
                    for (int st = 0; st < kNT; ++st) {
                        const float* eb = ebuf[m & 1] + (16 * st + s) * 32;
                        const int rot = (s >> 1) & 7;
                        const float4 x0 = *reinterpret_cast<const float4*>(eb + 4 * (q4 ^ rot));
                        const float4 x1 = *reinterpret_cast<const float4*>(eb + 4 * ((4 + q4) ^ rot));
                        const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        split8(x, p[st], ehi[st], elo[st]);
                    }
#pragma unroll
                    for (int q = 0; q < kTD / 2; ++q) {
                        const float* w0 = wl + ((kl * kTD + 2 * q) * 2) * 256;
                        mfma_quad(k1[0][2 * q], k1[0][2 * q + 1], k1[1][2 * q], k1[1][2 * q + 1], w0, w0 + 512, ehi, elo);
                        if (kl == 0) {
                            if (q < 3) stream_issue_piece(nx, q, lane, wave);
                            else { stream_issue_piece(nx, 3, lane, wave); stream_issue_piece(nx, 4, lane, wave); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (m + 2 < kSteps) issue_e0(m + 2);
                }
            }
            if (2 * c + 2 < kSteps && 2 * c + 3 < kSteps) wait_vm(std::integral_constant<int, 8>());
            else if (2 * c + 2 < kSteps) wait_vm(std::integral_constant<int, 4>());
            else wait_vm(std::integral_constant<int, 0>());
            __syncthreads();
        }
    }
    f32x4 key[kNT][kTD];
#pragma unroll
    for (int st = 0; st < kNT; ++st) {
        scale_acc<kTD>(k1[st], lsc[kLayerK1] * pinv[st]);
        pow2_scale(fmaxf(sample_max<kTD, true>(k1[st]), 1e-30f), p[st], pinv[st]);
        init_bias<kTD>(key[st], lds + kLdsBias + kBiasK2, q4, p[st] / lsc[kLayerK2]);
    }
    chained_layer2<kTD, true, kG_K2>(key, k1, p, a.blob, lds, lane, wave);
#pragma unroll
    for (int st = 0; st < kNT; ++st) scale_acc<kTD>(key[st], lsc[kLayerK2] * pinv[st]);

    // ---- qry = Wq2 relu(Wq1 g + bq1) + bq2 ;  logit = <key, qry>/16 ----
    half8 ghi[kNT], glo[kNT];
#pragma unroll
    for (int st = 0; st < kNT; ++st) {
        const float* gl = lds + kLdsG + (wave * kRows + 16 * st + s) * 16 + 8 * (q4 & 1);
        float gx8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) gx8[k] = q4 < 2 ? gl[k] : (q4 == 2 && k == 0) ? 1.0f : 0.0f;
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(gx8[k]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        pow2_scale(m, p[st], pinv[st]);
        split8(gx8, p[st], ghi[st], glo[st]);
    }
    // k1's registers are free: t1 takes them over
    f32x4 (&t1)[kNT][kTD] = k1;
#pragma unroll
    for (int st = 0; st < kNT; ++st)
#pragma unroll
        for (int t = 0; t < kTD; ++t) t1[st][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    stream_issue_all(a.blob, lds, kG_Q1 + 1, lane, wave);
    {
        const float* wl = lds + kLdsW + (kG_Q1 & 1) * kChunkTiles * kTile + 4 * lane;
#pragma unroll
        for (int q = 0; q < kTD / 2; ++q) {
            const float* w0 = wl + (2 * q * 2) * 256;
            mfma_quad(t1[0][2 * q], t1[0][2 * q + 1], t1[1][2 * q], t1[1][2 * q + 1], w0, w0 + 512, ghi, glo);
        }
    }
    stream_sync();
    f32x4 qv[kNT][kTD];
#pragma unroll
    for (int st = 0; st < kNT; ++st) {
        scale_acc<kTD>(t1[st], lsc[kLayerQ1] * pinv[st]);
        pow2_scale(fmaxf(sample_max<kTD, true>(t1[st]), 1e-30f), p[st], pinv[st]);
        init_bias<kTD>(qv[st], lds + kLdsBias + kBiasQ2, q4, p[st] / lsc[kLayerQ2]);
    }
    chained_layer2<kTD, true, kG_Q2>(qv, t1, p, a.blob, lds, lane, wave);
#pragma unroll
    for (int st = 0; st < kNT; ++st) {
        scale_acc<kTD>(qv[st], lsc[kLayerQ2] * pinv[st]);
        float dot = 0.0f;
#pragma unroll
        for (int t = 0; t < kTD; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fmaf(key[st][t][r], qv[st][t][r], dot);
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        if (row_live(16 * st + s)) {
            const long i = i_base + row_rel(16 * st + s);
            store_rows<kTD>(qv[st], a.qry + i * kD, q4);
            if (q4 == 0) a.logit[i] = dot / 16.0f;
        }
    }
#if CAR_W8_VAR == 9
    if (lane == 0) {
        long long* out = reinterpret_cast<long long*>(a.pixel_val) + ((long)blk * kWaves + wave) * 10;
        for (int k = 0; k < 8; ++k) out[k] = tsum[k];
        out[8] = t_passes;
        out[9] = __builtin_amdgcn_s_memtime() - t_begin;
    }
#endif
}

}  // namespace

extern "C" int car_fused_samples_ws(const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h, int lat_w,
                                    int lat_pad, const float* gmeta, const float* wpt, const float* blob, const float* bias, int b, int V, int R, int P,
                                    int H, int W, int no_sample, float* e, float* qry, float* g, float* logit, float* pt, float* pixel_val, void* stream) {
    CAR_REQUIRE(poses && rays && steps && lattice && gmeta && wpt && blob && bias, "car_fused_samples_ws: null input");
    CAR_REQUIRE(e && qry && g && logit && pt && pixel_val, "car_fused_samples_ws: null output");
    CAR_REQUIRE(V == 2, "car_fused_samples_ws: built for V = 2 (got %d)", V);
    CAR_REQUIRE(b > 0 && R > 0 && P > 0 && H > 1 && W > 1, "car_fused_samples_ws: bad sizes");
    CAR_REQUIRE((long)lat_h * lat_w * (kC * 4) < kMaxMapBytes, "car_fused_samples_ws: lattice too large");
    CAR_REQUIRE((long)kTileRays * P * (2 * kE * 4) < 0x7fffffffL, "car_fused_samples_ws: a ray bundle's rows of e exceed 32-bit offsets");
    FusedArgs a;
    a.poses = (const CarPose*)poses; a.rays = (const CarRay*)rays; a.steps = steps;
    a.lattice = lattice; a.lh = lat_h; a.lw = lat_w; a.pad = lat_pad;
    a.map_bytes = (unsigned)((long)lat_h * lat_w * (kC * 4));
    a.sx = (float)((lat_w - 2 * lat_pad + 1) / 2); a.sy = (float)((lat_h - 2 * lat_pad + 1) / 2);
    a.gmeta = gmeta; a.wpt = wpt; a.blob = blob; a.bias = bias;
    a.b = b; a.V = V; a.R = R; a.P = P; a.H = H; a.W = W;
    a.no_sample = no_sample != 0;
    a.S = (long)b * V * R * P;
    a.e = e; a.qry = qry; a.g = g; a.logit = logit; a.pt = pt; a.pixel_val = pixel_val;
    const long groups = (long)b * V * car_div_up(R, kTileRays) * car_div_up(P, kTileSteps);
    hipError_t e1 = hipFuncSetAttribute((const void*)fused_kernel_w8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e1 != hipSuccess) { car_set_error("car_fused_samples_ws: cannot reserve %zu bytes of LDS: %s", kLdsBytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(fused_kernel_w8, dim3((unsigned)groups), dim3(kThreads), kLdsBytes, (hipStream_t)stream, a);
    CAR_CHECK_LAUNCH("car_fused_samples_ws");
    return CAR_OK;
}
