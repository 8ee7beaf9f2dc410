// car_fused_ws.hip — EXPERIMENT, not product (built only by tools/build_dev.py, timed by tools/bench_fused.py 200): the fused per-sample
// kernel with SPECIALISED waves (same arithmetic, same outputs as csrc/car_fused.hip, bit for bit).  Measured slower than the product
// kernel (3.59 vs 3.32 ms per 8192 rays): profiles/round3_fused_experiments.md section 11 has the numbers and the reason.
//
// Why: in car_fused.hip every wave gathers its own 16 samples AND runs their matrix work.  Its waves stall in the ISSUE of their
// vector-memory instructions (the texture path takes a 1 KB instruction every ~20 clocks per CU; twelve waves offer their tap loads
// and DMA pieces in the same slots of the same chunk), an in-order wave cannot issue its MFMAs meanwhile, and the waves reach the
// chunk barrier apart: matrix pipe 41 % busy, texture path 52 % busy, LDS 43 % busy — nothing saturated
// (profiles/round3_fused_experiments.md, sections 9 and 10).  Here the two kinds of work run in different waves:
//   * 12 MATRIX waves (three per SIMD), 16 samples each: per chunk one B operand (two ds_read_b128 of ready-made fp16 halves) and
//     54 MFMAs; their only vector memory instructions inside the two source passes are three LDS-DMA pieces of the weight stream;
//   * 4 GATHER waves (one per SIMD), 48 samples each: all tap loads (three row groups = 12 x 16 B per lane in flight, half a chunk
//     ahead), the blend, the affine start values, ReLU and the fp16 hi/lo split of h, written to a double-buffered B-operand stage
//     in LDS.  They issue nothing but tap loads, so the compiler's own vmcnt waits are exact, and they end after the source passes.
// -DCAR_WS_STAMP / _NOTAPS / _NOGATHER / _L2TEST / _RING=n: timing variants of the experiment.
// 16 waves = four per SIMD, so every wave has 128 registers: the key layer therefore takes BOTH halves of its input ([e_0 ; e_1])
// back from the output tensor (stored by the same lane moments earlier: L2) instead of chaining e_1 from the accumulators.
// One barrier per chunk as before; a chunk's h tile and weights are produced during the previous chunk's period.
#include "car_common.h"
#include "car_geom.h"

namespace {

constexpr int kMatrix = 12, kGather = 4, kRows = 16, kGroup = kMatrix * kRows;       // 192 samples per workgroup
constexpr int kWaves = kMatrix;                    // DMA participants (car_fused_mma.h)
constexpr int kStepsPerGroup = 4, kBundles = kMatrix / kStepsPerGroup;
constexpr int kThreads = 64 * (kMatrix + kGather);
constexpr int kPieces = 3;                         // LDS-DMA pieces per chunk and matrix wave: 36 KB / (12 waves x 1 KB)
constexpr int kRowGroups = kGroup / kGather / 8;   // 6 row groups of 8 rows per gather wave and chunk
#ifndef CAR_WS_RING
#define CAR_WS_RING 6
#endif
constexpr int kRing = CAR_WS_RING;                 // row groups of tap loads in flight per gather wave (divides 6: ring slots are static)
constexpr unsigned kDeadTap = 0xc0000000u;
constexpr long kMaxMapBytes = 0x80000000L;

#include "car_fused_mma.h"

constexpr int kStageWave = 512;                                 // floats per wave tile: [hi | lo][16 rows][32 halves]   2 KB
constexpr int kLdsStage = kLdsW + 2 * kChunkTiles * kTile;      // [2][12][512]           B operands of a chunk, ready split   48 KB
constexpr int kLdsTapB = kLdsStage + 2 * kMatrix * kStageWave;  // [192][2] uint
constexpr int kLdsTapW = kLdsTapB + kGroup * 2;                 // [192][2][4]
constexpr int kLdsPe = kLdsTapW + kGroup * 8;                   // [192][2][4]
constexpr int kLdsWpt = kLdsPe + kGroup * 8;                    // [576][4]
constexpr int kLdsBias = kLdsWpt + kC * 4;                      // [672]
constexpr int kLdsG = kLdsBias + kBiasFloats;                   // [192][16]
constexpr int kLdsFloats = kLdsG + kGroup * 16;
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
    long S;
    float* e;
    float* qry;
    float* g;
    float* logit;
    float* pt;
    float* pixel_val;
};

// chunk order (as car_fused.hip):  W2 x18 (source 0) | W2 x18 (source 1) | K1 over e_1 x5 | K1 over e_0 x5 | K2 x2 | Q1 | Q2 x2
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

// the chunk after g inside the two source passes (g + 1 in [1, 36]): selects only, no branch tree in the hot loop
__device__ __forceinline__ NextChunk next_chunk_w2(const float* __restrict__ blob, float* lds, int gn) {
    const bool w2 = gn < kG_K1b;
    const int step = gn >= kG_W2b ? gn - kG_W2b : gn;
    NextChunk n;
    n.src = blob + (long)(w2 ? kOffW2 + step * kTE : chunk_tile_offset(kG_K1b)) * kTile;
    n.dst = lds + kLdsW + (gn & 1) * kChunkTiles * kTile;
    n.nkb = w2 ? 2 * kTE : 2 * chunk_tiles(kG_K1b);
    return n;
}

template <int KEEP>
__device__ __forceinline__ void chunk_end() {                          // this wave's DMA pieces of the next chunk have landed; chunk retired
    if constexpr (KEEP == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// A layer with 128 outputs on the matrix waves: weight chunks of two K steps (G0 = the layer's first chunk; everything about a
// chunk is a constant here), the DMA pieces of the following chunk issued in the first K step.
// fetch(m, x): the 8 source values of this lane for K step m (tiles 2m and 2m+1, channels 4 q .. 4 q + 3 of each); HOOK_LOADS = the
// vector loads it issues per call AFTER handing out x (they are younger than the chunk's DMA pieces when issued in the second K
// step, and may stay in flight across the chunk barrier).
template <int NSRC, int G0, int HOOK_LOADS, class Fetch>
__device__ __forceinline__ void matrix_layer(f32x4 (&acc)[kTD], float p, const float* __restrict__ blob, float* lds, int lane, int wave, Fetch fetch) {
    constexpr int kSteps = NSRC / 2;
#pragma unroll
    for (int m0 = 0; m0 < kSteps; m0 += 2) {
        const int g = G0 + m0 / 2;
        const float* wl = lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane;
        const bool more = g + 1 < kNumChunks;
        const NextChunk nx = next_chunk(blob, lds, g + 1);
#pragma unroll
        for (int kl = 0; kl < 2; ++kl) {
            if (m0 + kl < kSteps) {
                float x[8];
                fetch(m0 + kl, x);
                half8 bhi, blo;
                split8(x, p, bhi, blo);
#pragma unroll
                for (int q = 0; q < kTD / 2; ++q) {
                    const float* w0 = wl + ((kl * kTD + 2 * q) * 2) * 256;
                    mfma_pair(acc[2 * q], acc[2 * q + 1], w0, w0 + 512, bhi, blo);
                    if (kl == 0 && q < kPieces && more) stream_issue_piece(nx, q, lane, wave);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (more) { if (m0 + 1 < kSteps) chunk_end<HOOK_LOADS>(); else chunk_end<0>(); }
    }
}

__global__ void __launch_bounds__(kThreads) fused_ws_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: the role split below is a uniform branch
#ifdef CAR_WS_STAMP
    const long long t_begin = (long long)__builtin_amdgcn_s_memtime();
#endif
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    {   // workgroup b runs on XCD b % 8 (observed, speed only): give every XCD a contiguous band of sample groups
        const int q8 = nblk / 8, r8 = nblk % 8, xcd = blk % 8, idx = blk / 8;
        blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int pgs = (a.P + kStepsPerGroup - 1) / kStepsPerGroup, bundles = (a.R + kBundles * kRows - 1) / (kBundles * kRows);
    const int pg = blk % pgs, bun = (blk / pgs) % bundles, nn = blk / (pgs * bundles);

    // hp: the power of two of the first layer's output; folded into the tap weights, the point terms and the bias where they are made
    float hp, hinv;
    pow2_scale(fmaxf(a.gmeta[0] + a.bias[kBiasScale + 5], 1e-30f), hp, hinv);
    for (int k = tid; k < kC; k += kThreads) {                        // per quad of channels: [x0..x3 | y0..y3 | z0..z3 | b0 hp..b3 hp]
        const float4 v = *reinterpret_cast<const float4*>(a.wpt + 4 * k);
        float* q = lds + kLdsWpt + 16 * (k >> 2) + (k & 3);
        q[0] = v.x; q[4] = v.y; q[8] = v.z; q[12] = v.w * hp;
    }
    for (int k = tid; k < kBiasFloats; k += kThreads) lds[kLdsBias + k] = a.bias[k];
    if (wave < kMatrix) stream_issue_all(a.blob, lds, 0, lane, wave);                   // weight chunk 0: nothing else touches the buffers yet

    // ---- geometry: one sample per lane of waves 0-2 (as car_fused.hip) ----
    if (wave < kGroup / 64) {
        const int P = a.P, V = a.V;
        const int sg = wave * 64 + lane, gwv = sg >> 4, gs = sg & 15;
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
#pragma unroll
        for (int sv = 0; sv < 2; ++sv) {
            float gx, gy;
            int mode;
            if (sv == v) { gx = smp.grid[0]; gy = smp.grid[1]; mode = 0; }
            else { gx = sv == 0 ? smp.grid_in[0][0] : smp.grid_in[1][0]; gy = sv == 0 ? smp.grid_in[0][1] : smp.grid_in[1][1]; mode = 1; }
            int node, flags;
            float w[4];
            car_lattice_taps(gx, gy, a.lw, a.lh, a.pad, a.sx, a.sy, &node, &flags, w);
            const bool dead = mode == 1 && (flags & 4);                // zeros padding, beyond the outer ring: exact zeros, no memory touched
            reinterpret_cast<unsigned*>(lds + kLdsTapB)[sg * 2 + sv] = dead ? kDeadTap : (unsigned)node * (unsigned)(kC * 4);
            *reinterpret_cast<float4*>(lds + kLdsTapW + (sg * 2 + sv) * 4) = dead ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(w[0] * hp, w[1] * hp, w[2] * hp, w[3] * hp);
            const float px = sv == 0 ? smp.pt_in[0][0] : smp.pt_in[1][0], py = sv == 0 ? smp.pt_in[0][1] : smp.pt_in[1][1],
                        pz = sv == 0 ? smp.pt_in[0][2] : smp.pt_in[1][2];
            *reinterpret_cast<float4*>(lds + kLdsPe + (sg * 2 + sv) * 4) = make_float4(tanhf(px / 5.0f) * hp, tanhf(py / 5.0f) * hp, tanhf(pz / 5.0f) * hp, 0.0f);
        }
        if (g_live) {
#ifndef CAR_WS_STAMP
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
    __syncthreads();                                                   // tables and tap records visible

    const float* lsc = lds + kLdsBias + kBiasScale;
    auto uniform = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    float e_up, e_down;
    {
        const float dW2 = lsc[kLayerW2];
        e_up = uniform(hp / dW2); e_down = uniform(dW2 * hinv);
    }
#ifdef CAR_WS_STAMP
    long long t_work = 0, t_bar = 0, t_aux = 0, ph[5] = {0, 0, 0, 0, 0};
    auto tick = [&]() -> long long { return (long long)__builtin_amdgcn_s_memtime(); };
    auto dump = [&]() {
        if (lane == 0) {
            long long* out = reinterpret_cast<long long*>(a.pixel_val) + ((long)blk * (kMatrix + kGather) + wave) * 8;
            out[0] = t_work; out[1] = t_bar; out[2] = t_aux;
            for (int k = 0; k < 5; ++k) out[3 + k] = ph[k];
        }
    };
    ph[0] = t_begin; ph[1] = tick();
#else
    long long t_work = 0, t_bar = 0, t_aux = 0, ph[5] = {0, 0, 0, 0, 0};
    auto tick = [&]() -> long long { return 0; };
    auto dump = [&]() {};
#endif
    if (wave >= kMatrix) {
        // =========================== gather wave: 48 samples, the tap loads of a whole chunk in flight ===========================
        const int gw = wave - kMatrix;
#ifndef CAR_WS_NOPRIO
        __builtin_amdgcn_s_setprio(3);                                  // the gather wave is the critical path of its SIMD: issue it first
#endif
        const int qd = lane & 7, r0 = lane >> 3;
        const unsigned qd16 = 16u * qd;
        const unsigned row_step = (unsigned)a.lw * (kC * 4);
        const int v_own = nn % a.V, sc_own = nn / a.V;
        const long map_floats = (long)a.lh * a.lw * kC;
        const __amdgpu_buffer_rsrc_t rsrc[2] = {
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lattice + ((long)(sc_own * a.V + 0) * 2 + (v_own == 0 ? 0 : 1)) * map_floats), 0,
                                              (int)a.map_bytes, 0x00027000),
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lattice + ((long)(sc_own * a.V + 1) * 2 + (v_own == 1 ? 0 : 1)) * map_floats), 0,
                                              (int)a.map_bytes, 0x00027000)};
        f32x4 ring[kRing][4];
        // Row group rg of this wave = rows r0 + 8 (rg & 1) of matrix wave 3 gw + (rg >> 1) = sample 48 gw + r0 + 8 rg of the workgroup:
        // every per-sample record sits at a compile-time distance from the wave's first one.
        const int smp0 = 48 * gw + r0;
        const unsigned* tapb = reinterpret_cast<const unsigned*>(lds + kLdsTapB) + smp0 * 2;
        const float* tapw = lds + kLdsTapW + smp0 * 8;
        const float* pes = lds + kLdsPe + smp0 * 8;
        // The wave walks the (chunk m, row group rg) positions of both source passes in order — m = 18 sv + c, position 6 m + rg —
        // finishing one position per step and issuing the tap loads of the position kRing ahead into the ring slot just freed.
        // A gather wave is alone on its SIMD with the LDS busy under the matrix waves' operand reads, so nothing it reads from LDS
        // may sit in the step's dependency chain: the records of a position (point term, tap weights) and the tap offsets of the
        // position to issue are read TWO steps ahead, the chunk's four weight rows one period ahead.
        float4 peq[3], twq[3], wq[4];
        unsigned tbq[3];
        auto pos_m = [](int m, int rg, int ahead) { return m + (rg + ahead) / kRowGroups; };
        auto pos_rg = [](int rg, int ahead) { return (rg + ahead) % kRowGroups; };
        auto src_of = [](int m) { return m >= kKS ? 1 : 0; };
        auto fetch_records = [&](int slot, int m, int rg) {           // for the finish of position (m, rg); past the end: source 1 again
            const int sv = src_of(m);
            peq[slot] = *reinterpret_cast<const float4*>(pes + (rg * 16 + sv) * 4);      // sample smp0 + 8 rg, record 2 sample + sv
            twq[slot] = *reinterpret_cast<const float4*>(tapw + (rg * 16 + sv) * 4);
        };
        auto fetch_offsets = [&](int slot, int m, int rg) { tbq[slot] = tapb[rg * 16 + src_of(m)]; };
        auto fetch_weights = [&](int m) {
            const int c = m < kG_K1b ? m - src_of(m) * kKS : 0;
            const float4* wp = reinterpret_cast<const float4*>(lds + kLdsWpt + 16 * (8 * c + qd));
            wq[0] = wp[0]; wq[1] = wp[1]; wq[2] = wp[2]; wq[3] = wp[3];      // x | y | z | b of the lane's four channels
        };
        auto issue = [&](int slot, unsigned tbv, int m) {
#if defined(CAR_WS_NOTAPS) || defined(CAR_WS_NOGATHER)   // timing only (wrong results)
            ring[slot][0] = ring[slot][1] = ring[slot][2] = ring[slot][3] = f32x4{(float)tbv, 0.f, 0.f, 0.f};
            return;
#endif
            const int sv = src_of(m), chunk_off = 128 * (m - sv * kKS);
#ifdef CAR_WS_L2TEST   // timing only (wrong results): every tap inside a 1 MB window of its lattice, i.e. resident in the XCD's L2
            const unsigned o00 = (tbv & 0xfff00u) + qd16, dx = (unsigned)(kC * 4), dy = 4u * (kC * 4);
#else
            const unsigned o00 = tbv + qd16, dx = (unsigned)(kC * 4), dy = row_step;
#endif
            auto ld = [&](unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc[sv], (int)off, chunk_off, 0)); };
            ring[slot][0] = ld(o00);
            ring[slot][1] = ld(o00 + dx);
            ring[slot][2] = ld(o00 + dy);
            ring[slot][3] = ld(o00 + dx + dy);
        };
        // h rows of position (m, rg): affine start, the four taps, ReLU, fp16 hi / lo halves -> B-operand stage m & 1
        auto finish = [&](int slot, int rslot, int m, int rg) {
#ifdef CAR_WS_NOGATHER
            return;
#endif
            const float4 pe = peq[rslot];
            const float4 wx = wq[0], wy = wq[1], wz = wq[2], wb = wq[3];
            const float4 h0 = make_float4(fmaf(wx.x, pe.x, fmaf(wy.x, pe.y, fmaf(wz.x, pe.z, wb.x))), fmaf(wx.y, pe.x, fmaf(wy.y, pe.y, fmaf(wz.y, pe.z, wb.y))),
                                          fmaf(wx.z, pe.x, fmaf(wy.z, pe.y, fmaf(wz.z, pe.z, wb.z))), fmaf(wx.w, pe.x, fmaf(wy.w, pe.y, fmaf(wz.w, pe.z, wb.w))));
            const float ww[4] = {twq[rslot].x, twq[rslot].y, twq[rslot].z, twq[rslot].w};
            f32x2 lo2 = {h0.x, h0.y}, hi2 = {h0.z, h0.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 gq = ring[slot][t];
                const f32x2 w2_ = {ww[t], ww[t]};
                lo2 = __builtin_elementwise_fma(w2_, f32x2{gq[0], gq[1]}, lo2);
                hi2 = __builtin_elementwise_fma(w2_, f32x2{gq[2], gq[3]}, hi2);
            }
            unsigned ha, la, hb, lb;                                     // already h * hp: ReLU, then the fp16 halves
            split_pair(fmaxf(lo2[0], 0.f), fmaxf(lo2[1], 0.f), ha, la);
            split_pair(fmaxf(hi2[0], 0.f), fmaxf(hi2[1], 0.f), hb, lb);
            // matrix wave 3 gw + (rg >> 1), row rr = r0 + 8 (rg & 1), channels 4 qd .. 4 qd + 3: four halves = 8 bytes of the row's 64;
            // the row's four 16-byte slots are rotated by rr >> 2 so that the matrix wave's ds_read_b128 (16 rows per pass) is conflict-free
            const int rr = r0 + 8 * (rg & 1);
            float* st = lds + kLdsStage + ((m & 1) * kMatrix + 3 * gw + (rg >> 1)) * kStageWave + rr * 16 + 4 * ((qd >> 1) ^ ((rr >> 2) & 3)) + 2 * (qd & 1);
            *reinterpret_cast<uint2*>(st) = make_uint2(ha, hb);
            *reinterpret_cast<uint2*>(st + 256) = make_uint2(la, lb);
        };
        static_assert(kRowGroups % 3 == 0 && kRowGroups % kRing == 0, "record / ring slots are static");
        // prologue: the first kRing positions' taps, the records of positions 0 and 1, the offsets of the first two positions to issue
        fetch_weights(0);
#pragma unroll
        for (int rg = 0; rg < kRing; ++rg) issue(rg, tapb[rg * 16], 0);
        fetch_records(0, 0, 0);
        fetch_records(1, 0, 1);
        fetch_offsets(0, pos_m(0, 0, kRing), pos_rg(0, kRing));
        fetch_offsets(1, pos_m(0, 1, kRing), pos_rg(1, kRing));
#pragma unroll 1
        for (int m = 0; m < kG_K1b; ++m) {                              // chunk m is finished during period m - 1 (m = 0: before the first)
            const long long t0 = tick();
#pragma unroll
            for (int rg = 0; rg < kRowGroups; ++rg) {
                fetch_records((rg + 2) % 3, pos_m(m, rg, 2), pos_rg(rg, 2));
                fetch_offsets((rg + 2) % 3, pos_m(m, rg, 2 + kRing), pos_rg(rg, 2 + kRing));
#ifdef CAR_WS_STAMP
                const long long ta = tick();
                finish(rg % kRing, rg % 3, m, rg);
                const long long tb = tick();
                const int m2 = pos_m(m, rg, kRing);
                if (m2 < kG_K1b) issue(rg % kRing, tbq[rg % 3], m2);
                t_aux += tb - ta; ph[2] += tick() - tb;
#else
                finish(rg % kRing, rg % 3, m, rg);
                const int m2 = pos_m(m, rg, kRing);
                if (m2 < kG_K1b) issue(rg % kRing, tbq[rg % 3], m2);
#endif
                if (rg == kRowGroups - 1) fetch_weights(m + 1);
            }
            const long long t1 = tick();
            __syncthreads();                                            // chunk m - 1 retired, chunk m ready
            t_work += t1 - t0; t_bar += tick() - t1;
        }
        dump();
        return;                                                         // the later layers need no gather: 12 waves remain
    }

    // =========================== matrix wave: 16 samples, MFMAs and LDS reads only in the source passes ===========================
    const int s = lane & 15, q4 = lane >> 4;
    const int ray_i = bun * (kBundles * kRows) + (wave / kStepsPerGroup) * kRows + s, pp = pg * kStepsPerGroup + wave % kStepsPerGroup;
    const bool live = ray_i < a.R && pp < a.P;
    const long i = ((long)nn * a.R + (ray_i < a.R ? ray_i : a.R - 1)) * a.P + (pp < a.P ? pp : a.P - 1);
    float* erow = a.e + i * (2 * kE) + 4 * q4;
    chunk_end<0>();                                                     // weight chunk 0 landed; chunk 0 ready

    float msrc[2];
    {
        f32x4 acc[kTE];
#pragma unroll 1
        for (int sv = 0; sv < 2; ++sv) {
            init_bias<kTE>(acc, lds + kLdsBias + kBiasE, q4, e_up);
#pragma unroll 1
            for (int c = 0; c < kKS; ++c) {
                const int g = sv * kKS + c;
                const float* st = lds + kLdsStage + ((g & 1) * kMatrix + wave) * kStageWave + s * 16 + 4 * (q4 ^ ((s >> 2) & 3));
                const half8 bhi = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(st));
                const half8 blo = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(st + 256));
                const float* wl = lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane;
                const NextChunk nx = next_chunk_w2(a.blob, lds, g + 1);
                const long long t0 = tick();
#pragma unroll
                for (int qs = 0; qs < kTE / 2; ++qs) {
                    const float* w0 = wl + (2 * qs * 2) * 256;
                    mfma_pair(acc[2 * qs], acc[2 * qs + 1], w0, w0 + 512, bhi, blo);
                    if (qs < kPieces) stream_issue_piece(nx, qs, lane, wave);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const long long t1 = tick();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const long long t2 = tick();
                __syncthreads();
                t_work += t1 - t0; t_aux += t2 - t1; t_bar += tick() - t2;
            }
            scale_acc<kTE>(acc, e_down);
            msrc[sv] = sample_max<kTE, false>(acc);
            // unconditional (a lane past the end works on a clamped duplicate of a live sample and writes its values again)
            store_rows<kTE>(acc, erow + sv * kE - 4 * q4, q4);
        }
    }
    ph[2] = tick();
    // ---- k1 = Wk1 [e_0 ; e_1] + bk1: the e_1 half first, then e_0 (the order of car_fused.hip); both halves come back from the
    //      output tensor, two tiles per K step, the next step's pair in flight under this step's MFMAs ----
    float p, pinv;
    pow2_scale(fmaxf(fmaxf(msrc[0], msrc[1]), 1e-30f), p, pinv);
    f32x4 k1[kTD];
    init_bias<kTD>(k1, lds + kLdsBias + kBiasK1, q4, p / lsc[kLayerK1]);
    {
        f32x4 nxt[2];
        auto load_pair = [&](const float* row, int m) {
            nxt[0] = *reinterpret_cast<const f32x4*>(row + 32 * m);
            nxt[1] = *reinterpret_cast<const f32x4*>(row + 32 * m + 16);
        };
        load_pair(erow + kE, 0);
        matrix_layer<kTE, kG_K1b, 2>(k1, p, a.blob, lds, lane, wave, [&](int m, float (&x)[8]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = nxt[e >> 2][e & 3];
            if (m + 1 < kTE / 2) load_pair(erow + kE, m + 1); else load_pair(erow, 0);
        });
        matrix_layer<kTE, kG_K1a, 2>(k1, p, a.blob, lds, lane, wave, [&](int m, float (&x)[8]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = nxt[e >> 2][e & 3];
            if (m + 1 < kTE / 2) load_pair(erow, m + 1);
        });
    }
    ph[3] = tick();
    scale_acc<kTD>(k1, lsc[kLayerK1] * pinv);
    f32x4 key[kTD];
    pow2_scale(fmaxf(sample_max<kTD, true>(k1), 1e-30f), p, pinv);
    init_bias<kTD>(key, lds + kLdsBias + kBiasK2, q4, p / lsc[kLayerK2]);
    matrix_layer<kTD, kG_K2, 0>(key, p, a.blob, lds, lane, wave, [&](int m, float (&x)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = fmaxf(k1[2 * m + (e >> 2)][e & 3], 0.f);
    });
    scale_acc<kTD>(key, lsc[kLayerK2] * pinv);

    // ---- qry = Wq2 relu(Wq1 g + bq1) + bq2 ;  logit = <key, qry>/16 ----
    half8 ghi, glo;
    {
        const float* gl = lds + kLdsG + (wave * kRows + s) * 16 + 8 * (q4 & 1);
        float gx8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) gx8[k] = q4 < 2 ? gl[k] : (q4 == 2 && k == 0) ? 1.0f : 0.0f;
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(gx8[k]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        pow2_scale(m, p, pinv);
        split8(gx8, p, ghi, glo);
    }
    f32x4 t1[kTD], qv[kTD];
#pragma unroll
    for (int t = 0; t < kTD; ++t) t1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    stream_issue_all(a.blob, lds, kG_Q1 + 1, lane, wave);
    small_layer(t1, ghi, glo, lds + kLdsW + (kG_Q1 & 1) * kChunkTiles * kTile + 4 * lane);
    chunk_end<0>();                                                     // chunk Q1 retired
    scale_acc<kTD>(t1, lsc[kLayerQ1] * pinv);
    pow2_scale(fmaxf(sample_max<kTD, true>(t1), 1e-30f), p, pinv);
    init_bias<kTD>(qv, lds + kLdsBias + kBiasQ2, q4, p / lsc[kLayerQ2]);
    matrix_layer<kTD, kG_Q2, 0>(qv, p, a.blob, lds, lane, wave, [&](int m, float (&x)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = fmaxf(t1[2 * m + (e >> 2)][e & 3], 0.f);
    });
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
    ph[4] = tick();
    dump();
}

}  // namespace

extern "C" int car_fused_samples_ws(const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h, int lat_w,
                                    int lat_pad, const float* gmeta, const float* wpt, const float* blob, const float* bias, int b, int V, int R,
                                    int P, int H, int W, float* e, float* qry, float* g, float* logit, float* pt, float* pixel_val, void* stream) {
    CAR_REQUIRE(poses && rays && steps && lattice && gmeta && wpt && blob && bias, "car_fused_samples: null input");
    CAR_REQUIRE(e && qry && g && logit && pt && pixel_val, "car_fused_samples: null output");
    CAR_REQUIRE(V == 2, "car_fused_samples: built for V = 2 (got %d)", V);
    CAR_REQUIRE(b > 0 && R > 0 && P > 0 && H > 1 && W > 1, "car_fused_samples: bad sizes");
    CAR_REQUIRE(lat_pad >= 2 && lat_h > 2 * lat_pad + 1 && lat_w > 2 * lat_pad + 1 && ((lat_h - 2 * lat_pad) & 1) && ((lat_w - 2 * lat_pad) & 1),
                "car_fused_samples: bad lattice %d x %d, pad %d (car_lattice_shape)", lat_h, lat_w, lat_pad);
    CAR_REQUIRE((long)lat_h * lat_w * (kC * 4) < kMaxMapBytes, "car_fused_samples: a lattice of %d x %d nodes exceeds 4 GiB per view", lat_h, lat_w);
    FusedArgs a;
    a.poses = (const CarPose*)poses; a.rays = (const CarRay*)rays; a.steps = steps;
    a.lattice = lattice; a.lh = lat_h; a.lw = lat_w; a.pad = lat_pad;
    a.map_bytes = (unsigned)((long)lat_h * lat_w * (kC * 4));
    a.sx = (float)((lat_w - 2 * lat_pad + 1) / 2); a.sy = (float)((lat_h - 2 * lat_pad + 1) / 2);
    a.gmeta = gmeta; a.wpt = wpt; a.blob = blob; a.bias = bias;
    a.b = b; a.V = V; a.R = R; a.P = P; a.H = H; a.W = W;
    a.S = (long)b * V * R * P;
    a.e = e; a.qry = qry; a.g = g; a.logit = logit; a.pt = pt; a.pixel_val = pixel_val;
    const long groups = (long)b * V * car_div_up(R, kBundles * kRows) * car_div_up(P, kStepsPerGroup);
    hipError_t e1 = hipFuncSetAttribute((const void*)fused_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e1 != hipSuccess) { car_set_error("car_fused_samples: cannot reserve %zu bytes of LDS: %s", kLdsBytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(fused_ws_kernel, dim3((unsigned)groups), dim3(kThreads), kLdsBytes, (hipStream_t)stream, a);
    CAR_CHECK_LAUNCH("car_fused_samples");
    return CAR_OK;
}
