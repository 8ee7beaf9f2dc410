// car_fused2.hip — second generation of the fused per-sample kernel (car_fused.hip has the algorithm and the reference
// line numbers; this file changes only the CDNA4 mapping).
//
// PMC counters of car_fused.hip (profiles/r06): one wave per SIMD, matrix pipe busy 25 % of the wave's cycles, 29 % parked
// in s_waitcnt/barriers, 26 % issuing VALU (gather FMAs, fp16 splits) that a single in-order wave cannot overlap with its
// own MFMAs.  The cure is a second wave per SIMD, i.e. at most 256 registers per wave instead of 464:
//   * one workgroup = 8 waves = 128 consecutive samples (same weight stream, same LDS budget), each wave owns 16 samples;
//   * v_mfma_f32_16x16x32_f16 (K = 32 = one whole weight chunk per instruction): weights are the A operand (16 output
//     channels per tile), samples the B operand; a 288-wide accumulator set is 18 tiles x 4 registers = 72 registers,
//     the 128-wide ones 32;
//   * all layers run split-fp16 (x = hi + lo, three products per term, weights pre-scaled by 2^kWShift; see
//     car_fused.hip), including the two K = 16 layers fed by the geometric query;
//   * chained layers: the C/D layout of a 16x16 tile is  channel = 16 t + 4 (lane >> 4) + r , so two source tiles give a
//     lane its 8 B values of one K = 32 step; the host bakes  k = 16 (2 m + e/4) + 4 (lane >> 4) + e%4  into the packed
//     weights (engine.pack_fused2_weights, tests/test_fused_pack.py).
#include "car_common.h"
#include "car_geom.h"
#include <stdlib.h>

namespace {

constexpr int kWaves = 8, kRows = 16, kGroup = kWaves * kRows;       // 128 samples per workgroup

constexpr int kPieces = 5;                         // LDS-DMA pieces per chunk: 8 waves x 1 KB each

#include "car_fused16.h"

constexpr int kLdsStage = kLdsW + 2 * kChunkTiles * kTile;      // [8][16][36]            h tiles, wave private   18 KB
constexpr int kLdsTapI = kLdsStage + kGroup * kStageLd;         // [128][2][3][4] uint    tap byte offsets into a level's map   12 KB
constexpr int kLdsTapW = kLdsTapI + kGroup * 24;                // [128][2][3][4]         tap weights             12 KB
constexpr int kLdsPe = kLdsTapW + kGroup * 24;                  // [128][2][4]            tanh(pt_s/5)             4 KB
constexpr int kLdsWpt = kLdsPe + kGroup * 8;                    // [576][4]               (W1[:,C:C+3], b1)        9 KB
constexpr int kLdsBias = kLdsWpt + kC * 4;                      // [672]
constexpr int kLdsG = kLdsBias + kBiasFloats;                   // [128][16]              geometric query g per sample 8 KB
constexpr int kLdsFloats = kLdsG + kGroup * 16;
constexpr size_t kLdsBytes = (size_t)kLdsFloats * sizeof(float);

struct Fused2Args {
    const CarPose* poses;
    const CarRay* rays;
    const float* steps;
    const float* gmap[3];
    int gh[3], gw[3];
    const float* wpt;
    const float* blob;
    const float* bias;
    int b, V, R, P, H, W;
    int xcd_bands;
    int ray_major;
    long S;
    float* e;
    float* qry;
    float* ug;
    float* logit;
    float* pt;
    float* pixel_val;
};

// chunk order:  W2 x18 (source 0) | K1 over e_0 x5 (2,2,2,2,1 K steps) | W2 x18 (source 1) | K1 over e_1 x5 | K2 x2 | Q1 | Q2 x2 | UG
constexpr int kChK1 = 5;
constexpr int kG_K1a = kKS, kG_W2b = kG_K1a + kChK1, kG_K1b = kG_W2b + kKS, kG_K2 = kG_K1b + kChK1, kG_Q1 = kG_K2 + 2,
              kG_Q2 = kG_Q1 + 1, kG_UG = kG_Q2 + 2;
__device__ __forceinline__ int chunk_tile_offset(int g) {
    if (g < kG_K1a) return kOffW2 + g * kTE;
    if (g < kG_W2b) return kOffK1 + (g - kG_K1a) * 2 * kTD;
    if (g < kG_K1b) return kOffW2 + (g - kG_W2b) * kTE;
    if (g < kG_K2) return kOffK1 + 9 * kTD + (g - kG_K1b) * 2 * kTD;
    if (g < kG_Q1) return kOffK2 + (g - kG_K2) * 2 * kTD;
    if (g < kG_Q2) return kOffQ1;
    if (g < kG_UG) return kOffQ2 + (g - kG_Q2) * 2 * kTD;
    return kOffUG;
}
__device__ __forceinline__ int chunk_tiles(int g) {
    if (g < kG_K1a || (g >= kG_W2b && g < kG_K1b)) return kTE;
    if (g == kG_W2b - 1 || g == kG_K2 - 1 || g == kG_Q1 || g == kG_UG) return kTD;       // odd last K1 step, Q1, UG
    return 2 * kTD;
}

// ABL > 0: timing-only ablations (wrong results): 1 no tap loads, 2 no gather work, 3 = 2 + no weight DMA / barriers,
// 4 no MFMAs in the e path (gather, DMA and barriers only)
// SCHED 1: the tap loads run as a continuous two-buffer pipeline across chunk boundaries (16 loads per wave always in
// flight, also over the barrier); SCHED 0: every chunk's gather is issued and consumed inside the previous chunk.
// (Hand-counted s_waitcnt for the taps, so that a tap wait does not also wait for the LDS-DMA pieces the compiler cannot
// see, was measured too: 4.9 vs 4.9 ms, no gain — the waves are not held up there.)
template <int ABL, int SCHED = 1>
__global__ void __launch_bounds__(512) fused2_kernel(const Fused2Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 15, q4 = lane >> 4;
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    if (a.xcd_bands) {                                                 // contiguous band of sample groups per XCD, see car_fused.hip
        const int q8 = nblk / 8, r8 = nblk % 8, xcd = blk % 8, idx = blk / 8;
        blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    // Which 128 samples: with ray_major a workgroup takes 16 consecutive rays x 8 consecutive steps, wave = step, lane = ray, so
    // the 16 rows a wave gathers together are the SAME step of neighbouring rays: their taps fall on neighbouring texels
    // (shared 128-byte rows -> L1 hits instead of 16 unrelated misses spread along one epipolar line).
    long i;
    bool live;
    if (a.ray_major) {
        const int pgs = (a.P + kWaves - 1) / kWaves, bundles = (a.R + kRows - 1) / kRows;
        const int pg = blk % pgs, bun = (blk / pgs) % bundles, nn = blk / (pgs * bundles);
        const int ray = bun * kRows + s, pp = pg * kWaves + wave;
        live = ray < a.R && pp < a.P;
        i = ((long)nn * a.R + (ray < a.R ? ray : a.R - 1)) * a.P + (pp < a.P ? pp : a.P - 1);
    } else {
        const long i_raw = (long)blk * kGroup + wave * kRows + s;
        live = i_raw < a.S;
        i = live ? i_raw : a.S - 1;
    }

    for (int k = tid; k < kC; k += 512) *reinterpret_cast<float4*>(lds + kLdsWpt + 4 * k) = *reinterpret_cast<const float4*>(a.wpt + 4 * k);
    for (int k = tid; k < kBiasFloats; k += 512) lds[kLdsBias + k] = a.bias[k];
    int g = 0;
    stream_issue_all<ABL>(a.blob, lds, 0, lane, wave);

    // ---- geometry of this lane's sample (the four lane groups repeat it); lane group 0 / 1 prepares source view 0 / 1 ----
    const int P = a.P, V = a.V;
    const int p = (int)(i % P);
    const long nr = i / P;
    const int n = (int)(nr / a.R);
    const int v = n % V, sc = n / V;
    {
        const CarPose& Ps = a.poses[n];
        const CarRay ray = a.rays[nr];
        CarSample smp;
        for (int k = 0; k < 2; ++k) smp.grid[k] = ray.start[k] + (ray.end[k] - ray.start[k]) * a.steps[p];
        car_sample_setup(Ps, a.poses + sc * 2, ray, 2, a.H, a.W, &smp);
        if (q4 < 2) {
            const int sv = q4;
            float gx, gy;
            int mode, m;
            if (sv == v) { gx = smp.grid[0]; gy = smp.grid[1]; mode = 0; m = n; }
            else { gx = sv == 0 ? smp.grid_in[0][0] : smp.grid_in[1][0]; gy = sv == 0 ? smp.grid_in[0][1] : smp.grid_in[1][1]; mode = 1; m = sc * V + sv; }
            int* ti = reinterpret_cast<int*>(lds + kLdsTapI) + ((wave * kRows + s) * 2 + sv) * 12;
            float* tw = lds + kLdsTapW + ((wave * kRows + s) * 2 + sv) * 12;
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                int idx[4];
                float w[4];
                car_bilinear_taps(gx, gy, a.gw[l], a.gh[l], mode, idx, w);
#pragma unroll
                for (int t = 0; t < 4; ++t) { ti[4 * l + t] = (int)((unsigned)(m * a.gh[l] * a.gw[l] + idx[t]) * (unsigned)(kC * 4)); tw[4 * l + t] = w[t]; }
            }
            float* pe = lds + kLdsPe + ((wave * kRows + s) * 2 + sv) * 4;
#pragma unroll
            for (int k = 0; k < 3; ++k) pe[k] = tanhf((sv == 0 ? smp.pt_in[0][k] : smp.pt_in[1][k]) / 5.0f);
            pe[3] = 0.0f;
        }
        if (live && q4 == 0) {
            a.pixel_val[2 * i] = smp.grid[0]; a.pixel_val[2 * i + 1] = smp.grid[1];
            a.pt[3 * i + 0] = smp.pt[0]; a.pt[3 * i + 1] = smp.pt[1]; a.pt[3 * i + 2] = smp.pt[2];
        }
        // g is needed again only after the e path: parked in LDS instead of eight registers
        if (q4 < 2) {
            float* gl = lds + kLdsG + (wave * kRows + s) * 16 + 8 * q4;
#pragma unroll
            for (int k = 0; k < 8; ++k) gl[k] = q4 == 0 ? smp.g[k] : smp.g[8 + k];
        }
    }
    __syncthreads();                                                   // tables and tap records visible

    // ---- gather machinery: lane owns rows rr = (lane>>3) + 8*it (it = 0, 1) and channel quad qd = lane & 7 of a chunk ----
    const int qd = lane & 7, r0 = lane >> 3;
    float* stage = lds + kLdsStage + wave * kRows * kStageLd;
    float4 hacc[2];
    f32x4 tapA[8], tapB[8];

    const unsigned qd16 = 16u * qd;
    auto issue_row = [&](f32x4 (&tap)[8], int sv, int c, int l, int it) {
        if constexpr (ABL == 1 || ABL == 2 || ABL == 3) return;
        // wave-uniform base + 32-bit per-lane byte offset: one v_add per load instead of a 64-bit multiply-add
        const char* base = reinterpret_cast<const char*>(a.gmap[l] + 32 * c);
        const unsigned* ti = reinterpret_cast<const unsigned*>(lds + kLdsTapI) + ((wave * kRows + r0 + 8 * it) * 2 + sv) * 12 + 4 * l;
        const uint4 id = *reinterpret_cast<const uint4*>(ti);
        tap[4 * it + 0] = *reinterpret_cast<const f32x4*>(base + (id.x + qd16));
        tap[4 * it + 1] = *reinterpret_cast<const f32x4*>(base + (id.y + qd16));
        tap[4 * it + 2] = *reinterpret_cast<const f32x4*>(base + (id.z + qd16));
        tap[4 * it + 3] = *reinterpret_cast<const f32x4*>(base + (id.w + qd16));
    };
    auto blend_row = [&](const f32x4 (&tap)[8], int sv, int l, int it) {
        if constexpr (ABL == 2 || ABL == 3) return;
        const float4 w = *reinterpret_cast<const float4*>(lds + kLdsTapW + ((wave * kRows + r0 + 8 * it) * 2 + sv) * 12 + 4 * l);
        const float ww[4] = {w.x, w.y, w.z, w.w};
        f32x2 lo2 = {hacc[it].x, hacc[it].y}, hi2 = {hacc[it].z, hacc[it].w};          // v_pk_fma_f32: two FMAs per instruction
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 gq = tap[4 * it + t];
            const f32x2 w2 = {ww[t], ww[t]};
            lo2 = __builtin_elementwise_fma(w2, f32x2{gq[0], gq[1]}, lo2);
            hi2 = __builtin_elementwise_fma(w2, f32x2{gq[2], gq[3]}, hi2);
        }
        hacc[it] = make_float4(lo2[0], lo2[1], hi2[0], hi2[1]);
    };
    auto affine_row = [&](int sv, int c, int it) {
        if constexpr (ABL == 2 || ABL == 3) return;
        const int rr = r0 + 8 * it;
        const float4 pe = *reinterpret_cast<const float4*>(lds + kLdsPe + ((wave * kRows + rr) * 2 + sv) * 4);
        const float4* wp = reinterpret_cast<const float4*>(lds + kLdsWpt + 4 * (32 * c + 4 * qd));
        const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
        hacc[it] = make_float4(fmaf(w0.z, pe.z, fmaf(w0.y, pe.y, w0.x * pe.x)) + w0.w,
                               fmaf(w1.z, pe.z, fmaf(w1.y, pe.y, w1.x * pe.x)) + w1.w,
                               fmaf(w2.z, pe.z, fmaf(w2.y, pe.y, w2.x * pe.x)) + w2.w,
                               fmaf(w3.z, pe.z, fmaf(w3.y, pe.y, w3.x * pe.x)) + w3.w);
    };
    auto finish_row = [&](int it) {
        if constexpr (ABL == 2 || ABL == 3) return;
        const float4 o = hacc[it];
        *reinterpret_cast<float4*>(stage + (r0 + 8 * it) * kStageLd + 4 * qd) =
            make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    };
    auto read_b = [&](half8& bhi, half8& blo) {                        // this lane's 8 channels of the wave's h tile, split
        const float4 x0 = *reinterpret_cast<const float4*>(stage + s * kStageLd + 8 * q4);
        const float4 x1 = *reinterpret_cast<const float4*>(stage + s * kStageLd + 8 * q4 + 4);
        const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        split8(x, bhi, blo);
    };

    // first chunk of source 0: nothing to hide it under
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        affine_row(0, 0, it);
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            issue_row(tapA, 0, 0, l, it);
            blend_row(tapA, 0, l, it);
        }
        finish_row(it);
    }
    stream_sync();                                                     // weight chunk 0 landed
    constexpr bool kTapsLive = (ABL == 0 || ABL == 4);
    if constexpr (SCHED >= 1) {                                        // pipeline prologue: the two big levels of chunk (0, 1),
        issue_row(tapB, 0, 1, 1, 0); issue_row(tapB, 0, 1, 1, 1);      // in the steady-state order (tapB older than tapA)
        issue_row(tapA, 0, 1, 2, 0); issue_row(tapA, 0, 1, 2, 1);
    }

    const float up = (float)(1 << kWShift), down = 1.0f / (float)(1 << kWShift);
    f32x4 k1[kTD];
    init_bias<kTD>(k1, lds + kLdsBias + kBiasK1, q4, up);
    f32x4 acc[kTE];
    half8 bhi, blo;
    read_b(bhi, blo);
#pragma unroll 1
    for (int sv = 0; sv < 2; ++sv) {
        init_bias<kTE>(acc, lds + kLdsBias + kBiasE, q4, up);
#pragma unroll 1
        for (int c = 0; c < kKS; ++c) {
            // chunk being gathered: m+1 = (nsv, nc); in the pipelined schedule also m+2 = (n2sv, n2c).  Branch-free on purpose:
            // past the last chunk the gather harmlessly re-reads chunks of source 1.
            const int nsv = (c + 1 < kKS) ? sv : 1;
            const int nc = (c + 1 < kKS) ? c + 1 : 0;
            const int n2sv = (c + 2 < kKS) ? sv : 1;
            const int n2c = (c + 2 < kKS) ? c + 2 : c + 2 - kKS;
            const float* wl = lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane;
            const NextChunk nx = next_chunk(a.blob, lds, g + 1);
            // 9 slots of (4 ds_read_b128 + 6 MFMAs of 16 cycles); between them one piece of the gather / DMA issue.
            // SCHED 1 (levels: 2 = full resolution):
            //   0-4 DMA pieces   0 affine start (m+1)   2 blend L2(m+1), issue L0(m+1) -> tapA   4 blend L1(m+1), issue L1(m+2) -> tapB
            //   6 blend L0(m+1), ReLU, LDS write, issue L2(m+2) -> tapA
            // SCHED 0:
            //   0 issue L2 -> tapA   1 issue L1 -> tapB   1-5 DMA pieces   2 affine start   4 blend L2   5 issue L0 -> tapA
            //   6 blend L1           8 blend L0, ReLU, LDS write
            auto piece = [&](int qs) {
                if constexpr (SCHED == 1) {
                    if (qs < kPieces) stream_issue_piece<ABL>(nx, qs, lane, wave);
                    if (qs == 0) { affine_row(nsv, nc, 0); affine_row(nsv, nc, 1); }
                    else if (qs == 2) {
                        blend_row(tapA, nsv, 2, 0); blend_row(tapA, nsv, 2, 1);
                        issue_row(tapA, nsv, nc, 0, 0); issue_row(tapA, nsv, nc, 0, 1);
                    } else if (qs == 4) {
                        blend_row(tapB, nsv, 1, 0); blend_row(tapB, nsv, 1, 1);
                        issue_row(tapB, n2sv, n2c, 1, 0); issue_row(tapB, n2sv, n2c, 1, 1);
                    } else if (qs == 6) {
                        blend_row(tapA, nsv, 0, 0); blend_row(tapA, nsv, 0, 1); finish_row(0); finish_row(1);
                        issue_row(tapA, n2sv, n2c, 2, 0); issue_row(tapA, n2sv, n2c, 2, 1);
                    }
                } else {
                    if (qs >= 1 && qs < 1 + kPieces) stream_issue_piece<ABL>(nx, qs - 1, lane, wave);
                    if (qs == 0) { issue_row(tapA, nsv, nc, 2, 0); issue_row(tapA, nsv, nc, 2, 1); }
                    else if (qs == 1) { issue_row(tapB, nsv, nc, 1, 0); issue_row(tapB, nsv, nc, 1, 1); }
                    else if (qs == 2) { affine_row(nsv, nc, 0); affine_row(nsv, nc, 1); }
                    else if (qs == 4) { blend_row(tapA, nsv, 2, 0); blend_row(tapA, nsv, 2, 1); }
                    else if (qs == 5) { issue_row(tapA, nsv, nc, 0, 0); issue_row(tapA, nsv, nc, 0, 1); }
                    else if (qs == 6) { blend_row(tapB, nsv, 1, 0); blend_row(tapB, nsv, 1, 1); }
                    else if (qs == 8) { blend_row(tapA, nsv, 0, 0); blend_row(tapA, nsv, 0, 1); finish_row(0); finish_row(1); }
                }
            };
            AHi an = load_ahi(wl);
#pragma unroll
            for (int qs = 0; qs < kTE / 2; ++qs) {
                const AHi ac = an;
                if (qs + 1 < kTE / 2) an = load_ahi(wl + (2 * (qs + 1) * 2) * 256);          // next slot's hi halves first
                if constexpr (ABL != 4) mfma_ahead(acc[2 * qs], acc[2 * qs + 1], ac, wl + (2 * qs * 2) * 256, bhi, blo);
                piece(qs);
                __builtin_amdgcn_sched_barrier(0);
            }
            read_b(bhi, blo);                                          // next chunk's B operand (own LDS tile, in-order LDS)
            // pipelined: the 16 tap loads issued in slots 4 and 6 (after the last DMA piece) stay in flight over the barrier
            stream_sync<ABL, (SCHED >= 1 && kTapsLive) ? 16 : 0>();
            ++g;
        }
        scale_acc<kTE>(acc, down);
        chained_layer<kTE, false, ABL>(k1, acc, a.blob, lds, g, lane, wave);
        if (live) store_rows<kTE>(acc, a.e + i * (2 * kE) + sv * kE, q4);
    }
    scale_acc<kTD>(k1, down);
    f32x4 key[kTD];
    init_bias<kTD>(key, lds + kLdsBias + kBiasK2, q4, up);
    chained_layer<kTD, true, ABL>(key, k1, a.blob, lds, g, lane, wave);
    scale_acc<kTD>(key, down);

    // ---- qry = Wq2 relu(Wq1 g + bq1) + bq2 ;  logit = <key, qry>/16 ;  ug = Wr1[:,128:] g + br1 ---------------------
    half8 ghi, glo;                                                    // B operand of the two layers fed by g (k = 16: folded bias)
    {
        const float* gl = lds + kLdsG + (wave * kRows + s) * 16 + 8 * (q4 & 1);
        float gx8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) gx8[k] = q4 < 2 ? gl[k] : (q4 == 2 && k == 0) ? 1.0f : 0.0f;
        split8(gx8, ghi, glo);
    }
    f32x4 t1[kTD], qv[kTD];
#pragma unroll
    for (int t = 0; t < kTD; ++t) t1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    stream_issue_all<ABL>(a.blob, lds, g + 1, lane, wave);
    small_layer(t1, ghi, glo, lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane);          // q1
    stream_sync<ABL>();
    ++g;
    scale_acc<kTD>(t1, down);
    init_bias<kTD>(qv, lds + kLdsBias + kBiasQ2, q4, up);
    chained_layer<kTD, true, ABL>(qv, t1, a.blob, lds, g, lane, wave);
    scale_acc<kTD>(qv, down);
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
#pragma unroll
    for (int t = 0; t < kTD; ++t) t1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    small_layer(t1, ghi, glo, lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane);          // ug
    scale_acc<kTD>(t1, down);
    if (live) store_rows<kTD>(t1, a.ug + i * kD, q4);
}

}  // namespace

extern "C" size_t car_fused2_blob_floats(void) { return (size_t)kBlobTiles * kTile; }

extern "C" int car_fused_samples_v2(const float* poses, const float* rays, const float* steps, const float* const* gmaps,
                                    const int* level_h, const int* level_w, int n_levels, int C, const float* wpt,
                                    const float* blob, const float* bias, int b, int V, int R, int P, int H, int W,
                                    float* e, float* qry, float* ug, float* logit, float* pt, float* pixel_val, void* stream) {
    CAR_REQUIRE(poses && rays && steps && gmaps && level_h && level_w && wpt && blob && bias, "car_fused_samples_v2: null input");
    CAR_REQUIRE(e && qry && ug && logit && pt && pixel_val, "car_fused_samples_v2: null output");
    CAR_REQUIRE(n_levels == 3 && C == kC && V == 2, "car_fused_samples_v2: built for 3 pyramid levels, C = %d, V = 2 (got %d, %d, %d)", kC, n_levels, C, V);
    CAR_REQUIRE(b > 0 && R > 0 && P > 0 && H > 1 && W > 1, "car_fused_samples_v2: bad sizes");
    Fused2Args a;
    a.poses = (const CarPose*)poses; a.rays = (const CarRay*)rays; a.steps = steps;
    for (int l = 0; l < 3; ++l) {
        a.gmap[l] = gmaps[l]; a.gh[l] = level_h[l]; a.gw[l] = level_w[l];
        CAR_REQUIRE(a.gmap[l] && a.gh[l] > 0 && a.gw[l] > 0 && (long)b * V * a.gh[l] * a.gw[l] * (kC * 4) < 4294967296L, "car_fused_samples_v2: bad level %d (a level's map must stay below 4 GiB)", l);
    }
    a.wpt = wpt; a.blob = blob; a.bias = bias;
    a.b = b; a.V = V; a.R = R; a.P = P; a.H = H; a.W = W;
    a.S = (long)b * V * R * P;
    const char* xb = getenv("CAR_FUSED_XCD_BANDS");
    a.xcd_bands = xb ? atoi(xb) : 1;
    const char* rm = getenv("CAR_FUSED_RAY_MAJOR");
    a.ray_major = rm ? atoi(rm) : 1;
    const long groups = a.ray_major ? (long)b * V * car_div_up(R, kRows) * car_div_up(P, kWaves) : car_div_up(a.S, kGroup);
    a.e = e; a.qry = qry; a.ug = ug; a.logit = logit; a.pt = pt; a.pixel_val = pixel_val;
    const char* abl_env = getenv("CAR_FUSED_ABLATE");
    const int abl = abl_env ? atoi(abl_env) : 0;
    const char* sch_env = getenv("CAR_FUSED_SCHED");
    const int sch = sch_env ? atoi(sch_env) : 1;
    void (*kern)(const Fused2Args) = abl == 1 ? fused2_kernel<1> : abl == 2 ? fused2_kernel<2> : abl == 3 ? fused2_kernel<3> : abl == 4 ? fused2_kernel<4>
                                     : sch == 0 ? fused2_kernel<0, 0> : fused2_kernel<0, 1>;
    hipError_t e1 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e1 != hipSuccess) { car_set_error("car_fused_samples_v2: cannot reserve %zu bytes of LDS: %s", kLdsBytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(512), kLdsBytes, (hipStream_t)stream, a);
    CAR_CHECK_LAUNCH("car_fused_samples_v2");
    return CAR_OK;
}
