// car_fused_w32.hip — development-only candidate for the fused per-sample kernel, round 6 experiment (a) (never part of the product; built by
// tools/build_dev.py with CAR_DEV_UNIT=car_fused_w32.hip, compared with the product by tools/bench_fused.py 400; numbers in
// profiles/round6_fused_closing.md).  The two source passes — 75 % of the kernel's matrix work — on v_mfma_f32_32x32x16_f16 tiles:
//   * TWO waves per SIMD (8 waves, 256 registers each), each wave owning 32 samples = the 32 columns of a 32 x 32 tile;
//   * a slot = 32 output channels x 32 samples x K = 32: SIX 32-clock MFMAs (two K halves x hi*hi, hi*lo, lo*hi) where the product kernel issues
//     twenty-four 16-clock ones for the same block (12 for its 16 samples): half the MFMA instructions, and per multiply-accumulate half the
//     A-operand LDS bytes and half the operand-register reads of v_mfma_f32_16x16x32_f16;
//   * the W2 region of the weight blob re-laid for the 32 x 32 x 16 A operand by the bench script (a permutation of the product's bytes):
//     [K step][32-channel tile][K half][hi | lo][lane][8 halves], lane l = row l % 32, k = 16 half + 8 (l / 32) + e;
//   * e leaves the accumulators (lane (s, h) register 4 g + r = channel 32 T + 8 g + 4 h + r) through the wave's LDS tile as before; the key
//     layer reads BOTH halves of its input as rows — e_1 from that tile while it is being stored, e_0 back from the output tensor by LDS-DMA —
//     and stays, like the closing layers, on the product's 16 x 16 x 32 tiles and packed weights.
// Results equal the product's to fp32 rounding, not bit for bit: a K = 32 step is two K = 16 MFMAs here (another summation order).
#include "car_common.h"
#include "car_geom.h"
#include <type_traits>
#include <utility>

namespace {

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
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kT32 = kE / 32;                      // 9 output tiles of 32 channels

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
    float* g;
    float* logit;
    float* pt;
    float* pixel_val;
};

// chunk order (car_fused.hip):  W2 x18 | W2 x18 | K1 over e_1 x5 | K1 over e_0 x5 | Q1 | M x2
constexpr int kChK1 = 5;
constexpr int kG_W2b = kKS, kG_K1b = 2 * kKS, kG_K1a = kG_K1b + kChK1, kG_Q1 = kG_K1a + kChK1, kG_M = kG_Q1 + 1;
static_assert(kG_M + 2 == kNumChunks, "chunk count");
__device__ __forceinline__ constexpr int chunk_tile_offset(int g) {
    if (g < kG_W2b) return kOffW2 + g * kTE;
    if (g < kG_K1b) return kOffW2 + (g - kG_W2b) * kTE;
    if (g < kG_K1a) return kOffK1 + 9 * kTD + (g - kG_K1b) * 2 * kTD;
    if (g < kG_Q1) return kOffK1 + (g - kG_K1a) * 2 * kTD;
    if (g < kG_M) return kOffQ1;
    return kOffM + (g - kG_M) * 2 * kTD;
}
__device__ __forceinline__ constexpr int chunk_tiles(int g) {
    if (g < kG_K1b) return kTE;
    if (g == kG_K1a - 1 || g == kG_Q1 - 1 || g == kG_Q1) return kTD;
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

// an A-operand pair (two 16-channel output tiles, hi and lo halves: four LDS reads) against the B operands of BOTH 16-sample tiles: twelve
// 16 x 16 x 32 MFMAs (the layers behind the source passes)
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

// chained layer over both sample tiles (car_fused_mma.h chained_layer, one A read per pair of sample tiles)
template <int NSRC, bool RELU, int G0>
__device__ __forceinline__ void chained_layer2(f32x4 (&acc)[kNT][kTD], const f32x4 (&src)[kNT][NSRC], const float (&p)[kNT],
                                               const float* __restrict__ blob, float* lds, int lane, int wave) {
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
            }
        }
        stream_sync<0, 0>();
    }
}

__global__ void __launch_bounds__(kThreads) fused_kernel_w32(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, q4 = lane >> 4;                           // 16 x 16 x 32 side: row s of sample tile st, lane group q4
    const int s32 = lane & 31, h = lane >> 5;                          // 32 x 32 x 16 side: sample row s32, lane half h
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    {
        const int q8 = nblk / 8, r8 = nblk % 8, xcd = blk % 8, idx = blk / 8;
        blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int pgs = (a.P + kTileSteps - 1) / kTileSteps, bundles = (a.R + kTileRays - 1) / kTileRays;
    const int nsets = a.b * a.V;                                       // the product's step-major order (car_fused.hip)
    const int pg = blk / (nsets * bundles), nn = (blk / bundles) % nsets, bun = blk % bundles;
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
            a.pixel_val[2 * gi] = smp.grid[0]; a.pixel_val[2 * gi + 1] = smp.grid[1];
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
    const float* lsc = lds + kLdsBias + kBiasScale;
    auto uniform = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    float e_up, e_down;
    {
        const float dW2 = lsc[kLayerW2];
        e_up = uniform(hp / dW2); e_down = uniform(dW2 * hinv);
    }
    // B operands of the 32 x 32 x 16 tiles: lane (s32, h) holds channels 16 kh + 8 h .. + 7 of its sample's row of h, for both K halves kh
    auto read_b = [&](half8 (&bhi)[2], half8 (&blo)[2]) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const float4 x0 = *reinterpret_cast<const float4*>(stage + s32 * kStageLd + 16 * kh + 8 * h);
            const float4 x1 = *reinterpret_cast<const float4*>(stage + s32 * kStageLd + 16 * kh + 8 * h + 4);
            const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            split8_scaled(x, bhi[kh], blo[kh]);
        }
    };

    // first chunk of source 0: nothing to hide it under
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        issue_row(tapA, 0, 0, it);
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
    // a 32-channel tile of e (all 32 rows), turned through the wave's idle h tile and out as whole lines; the tile stays in LDS afterwards
    auto store_tile = [&](const f32x16& t, int m, int col0) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(stage + s32 * kStageLd + 8 * gq + 4 * h) = make_float4(t[4 * gq], t[4 * gq + 1], t[4 * gq + 2], t[4 * gq + 3]);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float4 v = *reinterpret_cast<const float4*>(stage + (r0 + 8 * it) * kStageLd + 4 * qd);
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(e_base) + (e_off[it] + (unsigned)((col0 + 32 * m) * 4))) = v;
        }
    };

    f32x16 acc[kT32];
    float m0 = 0.0f;                                                   // largest |e_0| of this lane's sample (row s32)
    half8 bhi[2], blo[2];
    read_b(bhi, blo);
#pragma unroll 1
    for (int sv = 0; sv < 2; ++sv) {
#pragma unroll
        for (int T = 0; T < kT32; ++T) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 b4 = *reinterpret_cast<const float4*>(lds + kLdsBias + kBiasE + 32 * T + 8 * gq + 4 * h);
                acc[T][4 * gq] = b4.x * e_up; acc[T][4 * gq + 1] = b4.y * e_up; acc[T][4 * gq + 2] = b4.z * e_up; acc[T][4 * gq + 3] = b4.w * e_up;
            }
        }
#pragma unroll 1
        for (int c = 0; c < kKS; ++c) {
            const int nsv = (c + 1 < kKS) ? sv : 1;
            const int nc = (c + 1 < kKS) ? c + 1 : 0;
            const int n2sv = (c + 2 < kKS) ? sv : 1;
            const int n2c = (c + 2 < kKS) ? c + 2 : c + 2 - kKS;
            const float* wl = lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane;
            const NextChunk nx = next_chunk_w2(a.blob, lds, g + 1);
            // 9 slots of (4 ds_read_b128 + 6 MFMAs of 32 clocks); between them one piece of the gather / DMA issue, as in the 16 x 16 x 32 candidate
            auto piece = [&](int qs) {
                if (qs < kPieces) stream_issue_piece(nx, qs, lane, wave);
                if (qs == 1) { gather_row(tapA, nsv, nc, 0); issue_row(tapA, nsv, nc, 2); }
                else if (qs == 3) { gather_row(tapB, nsv, nc, 1); issue_row(tapB, nsv, nc, 3); }
                else if (qs == 5) { gather_row(tapA, nsv, nc, 2); issue_row(tapA, n2sv, n2c, 0); }
                else if (qs == 7) { gather_row(tapB, nsv, nc, 3); issue_row(tapB, n2sv, n2c, 1); }
            };
#pragma unroll
            for (int T = 0; T < kT32; ++T) {
                const float* w0 = wl + (T * 4) * 256;                  // [T][K half][hi | lo]: 1 KB each
                const half8 ah0 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0));
                const half8 al0 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 256));
                const half8 ah1 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 512));
                const half8 al1 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 768));
                acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bhi[0], acc[T], 0, 0, 0);
                acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, blo[0], acc[T], 0, 0, 0);
                acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bhi[0], acc[T], 0, 0, 0);
                acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bhi[1], acc[T], 0, 0, 0);
                acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, blo[1], acc[T], 0, 0, 0);
                acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bhi[1], acc[T], 0, 0, 0);
                piece(T);
                __builtin_amdgcn_sched_barrier(0);
            }
            read_b(bhi, blo);
            stream_sync<0, 8>();
            ++g;
        }
#pragma unroll
        for (int T = 0; T < kT32; ++T) acc[T] *= e_down;
        if (sv == 0) {
#pragma unroll
            for (int T = 0; T < kT32; ++T)
#pragma unroll
                for (int r = 0; r < 16; ++r) m0 = fmaxf(m0, fabsf(acc[T][r]));
#pragma unroll
            for (int m = 0; m < kT32; ++m) store_tile(acc[m], m, 0);
        }
    }
    // ---- k1 = Wk1 [e_0 ; e_1] + bk1 on 16 x 16 x 32 tiles: per-sample power of two from the largest |e| of the sample (its row's two lane halves) ----
    float p[kNT], pinv[kNT];
    {
        float m1 = m0;
#pragma unroll
        for (int T = 0; T < kT32; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) m1 = fmaxf(m1, fabsf(acc[T][r]));
        m1 = fmaxf(m1, __shfl_xor(m1, 32, 64));                        // sample s32's maximum, in lanes s32 and s32 + 32
#pragma unroll
        for (int st = 0; st < kNT; ++st) pow2_scale(fmaxf(__shfl(m1, 16 * st + s, 64), 1e-30f), p[st], pinv[st]);      // row 16 st + s of the 16-row tiles
    }
    f32x4 k1[kNT][kTD];
#pragma unroll
    for (int st = 0; st < kNT; ++st) init_bias<kTD>(k1[st], lds + kLdsBias + kBiasK1, q4, p[st] / lsc[kLayerK1]);
    // the e_1 half: K step m = tile m of the accumulators, written to the wave's LDS tile (and from there to memory as whole lines) and read
    // back as the B operands of both 16-row tiles in the chained K order the packed key_map expects (channels 4 q4 .. + 3 and 16 + 4 q4 .. + 3)
    {
        constexpr int kSteps = kT32;
#pragma unroll
        for (int m0_ = 0; m0_ < kSteps; m0_ += 2) {
            const int nks = m0_ + 1 < kSteps ? 2 : 1;
            const int gk = kG_K1b + m0_ / 2;
            const float* wl = lds + kLdsW + (gk & 1) * kChunkTiles * kTile + 4 * lane;
            const NextChunk nx = next_chunk(a.blob, lds, gk + 1);
#pragma unroll
            for (int kl = 0; kl < 2; ++kl) {
                if (kl < nks) {
                    const int m = m0_ + kl;
                    store_tile(acc[m], m, kE);
                    half8 ehi[kNT], elo[kNT];
#pragma unroll
                    for (int st = 0; st < kNT; ++st) {
                        const float4 x0 = *reinterpret_cast<const float4*>(stage + (16 * st + s) * kStageLd + 4 * q4);
                        const float4 x1 = *reinterpret_cast<const float4*>(stage + (16 * st + s) * kStageLd + 16 + 4 * q4);
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
                }
            }
            // the second K step's four stores of e_1 are younger than the chunk's weight pieces and may stay in flight
            if (nks == 2) stream_sync<0, 4>(); else stream_sync<0, 0>();
        }
    }
    {
        // the e_0 half: lane (r0, qd) fetches 16 bytes of row rr = r0 + 8 it; a row's eight segments are stored rotated by (rr >> 1) & 7
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
#pragma unroll
    for (int st = 0; st < kNT; ++st) scale_acc<kTD>(k1[st], lsc[kLayerK1] * pinv[st]);

    // ---- logit = <key, qry> / 16 as r^T (M x + v) + u^T x + c (car_fused_layout.h) ----
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
    f32x4 t1[kNT][kTD], mt[kNT][kTD];
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
#pragma unroll
    for (int st = 0; st < kNT; ++st) {
        scale_acc<kTD>(t1[st], lsc[kLayerQ1] * pinv[st]);
        pow2_scale(fmaxf(sample_max<kTD, true>(t1[st]), 1e-30f), p[st], pinv[st]);
        init_bias<kTD>(mt[st], lds + kLdsBias + kBiasV, q4, p[st] / lsc[kLayerM]);
    }
    chained_layer2<kTD, true, kG_M>(mt, t1, p, a.blob, lds, lane, wave);
#pragma unroll
    for (int st = 0; st < kNT; ++st) {
        scale_acc<kTD>(mt[st], lsc[kLayerM] * pinv[st]);
        float dot = 0.0f;
#pragma unroll
        for (int t = 0; t < kTD; ++t) {
            const float4 u4 = *reinterpret_cast<const float4*>(lds + kLdsBias + kBiasU + 16 * t + 4 * q4);
            const float uu[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dot = fmaf(fmaxf(k1[st][t][r], 0.0f), mt[st][t][r], dot);
                dot = fmaf(uu[r], fmaxf(t1[st][t][r], 0.0f), dot);
            }
        }
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        dot += lds[kLdsBias + kBiasConst];
        if (row_live(16 * st + s) && q4 == 0) a.logit[i_base + row_rel(16 * st + s)] = dot / 16.0f;
    }
}

}  // namespace

// same arguments as car_fused_samples; `blob` must carry the W2 region in the 32 x 32 x 16 order (tools/bench_fused.py relays it)
extern "C" int car_fused_samples_w32(const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h, int lat_w,
                                     int lat_pad, const float* gmeta, const float* wpt, const float* blob, const float* bias, int b, int V, int R, int P,
                                     int H, int W, int no_sample, float* e, float* g, float* logit, float* pt, float* pixel_val, void* stream) {
    CAR_REQUIRE(poses && rays && steps && lattice && gmeta && wpt && blob && bias, "car_fused_samples_w32: null input");
    CAR_REQUIRE(e && g && logit && pt && pixel_val, "car_fused_samples_w32: null output");
    CAR_REQUIRE(V == 2, "car_fused_samples_w32: built for V = 2 (got %d)", V);
    CAR_REQUIRE(b > 0 && R > 0 && P > 0 && H > 1 && W > 1, "car_fused_samples_w32: bad sizes");
    CAR_REQUIRE((long)lat_h * lat_w * (kC * 4) < kMaxMapBytes, "car_fused_samples_w32: lattice too large");
    CAR_REQUIRE((long)kTileRays * P * (2 * kE * 4) < 0x7fffffffL, "car_fused_samples_w32: a ray bundle's rows of e exceed 32-bit offsets");
    FusedArgs a;
    a.poses = (const CarPose*)poses; a.rays = (const CarRay*)rays; a.steps = steps;
    a.lattice = lattice; a.lh = lat_h; a.lw = lat_w; a.pad = lat_pad;
    a.map_bytes = (unsigned)((long)lat_h * lat_w * (kC * 4));
    a.sx = (float)((lat_w - 2 * lat_pad + 1) / 2); a.sy = (float)((lat_h - 2 * lat_pad + 1) / 2);
    a.gmeta = gmeta; a.wpt = wpt; a.blob = blob; a.bias = bias;
    a.b = b; a.V = V; a.R = R; a.P = P; a.H = H; a.W = W;
    a.no_sample = no_sample != 0;
    a.S = (long)b * V * R * P;
    a.e = e; a.g = g; a.logit = logit; a.pt = pt; a.pixel_val = pixel_val;
    const long groups = (long)b * V * car_div_up(R, kTileRays) * car_div_up(P, kTileSteps);
    hipError_t e1 = hipFuncSetAttribute((const void*)fused_kernel_w32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e1 != hipSuccess) { car_set_error("car_fused_samples_w32: cannot reserve %zu bytes of LDS: %s", kLdsBytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(fused_kernel_w32, dim3((unsigned)groups), dim3(kThreads), kLdsBytes, (hipStream_t)stream, a);
    CAR_CHECK_LAUNCH("car_fused_samples_w32");
    return CAR_OK;
}
