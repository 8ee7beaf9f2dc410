// car_fused_v2.hip — the fused per-sample kernel with 2-D wave tiles in the two source passes (development unit, evaluated against
// csrc/car_fused.hip by tools/bench_fused.py 200: outputs must be equal bit for bit).
//
// csrc/car_fused.hip gives every wave 16 samples x all 288 output channels of e: per 32-channel chunk a wave reads the WHOLE 36 KB
// weight chunk from LDS (36 ds_read_b128) for 54 MFMAs — 432 KB of LDS reads per chunk and CU, which cost both LDS time and, more, power
// (the kernel runs at ~1.7 of 2.4 GHz; without the A-operand reads it clocks 12 % higher: profiles/round4_fused_experiments.md).
// Here a wave owns 48 samples x 96 channels (3 x 6 tiles of 16 x 16: the same 72 accumulator registers): per chunk 12 A-operand reads
// (its third of the weights) + 6 B-operand reads for the same 54 MFMAs.  The B operands (the first layer's activations h) therefore have to
// be shared between waves: the gathering lanes split h into fp16 hi / lo halves themselves and write them into a workgroup-wide,
// double-buffered tile in LDS (the chunk barrier that retires the weight buffer also publishes it).  e leaves through the same
// whole-line path as before; the key layer takes BOTH halves of its input back from the output tensor by LDS-DMA (the e_0 path of the
// product kernel, run twice), since a sample's e row is now spread over three waves; the per-sample power of two of that layer comes
// from an LDS max over the three waves' partial maxima.  Everything a value goes through is unchanged (same MFMA sequence per
// accumulator, same splits), so the outputs equal the product kernel's bit for bit.
#include "car_common.h"
#include "car_geom.h"
#include <type_traits>

namespace {

constexpr int kWaves = 12, kRows = 16, kGroup = kWaves * kRows;      // 192 samples per workgroup
constexpr int kWaveRays = 8, kWaveSteps = kRows / kWaveRays;
constexpr int kTileSteps = 8, kStepWaves = kTileSteps / kWaveSteps, kRayWaves = kWaves / kStepWaves, kTileRays = kRayWaves * kWaveRays;
__device__ __forceinline__ int tile_ray(int w, int s) { return (w / kStepWaves) * kWaveRays + (s & (kWaveRays - 1)); }
__device__ __forceinline__ int tile_step(int w, int s) { return (w % kStepWaves) * kWaveSteps + s / kWaveRays; }
constexpr int kThreads = 64 * kWaves;

constexpr int kPieces = 3;
constexpr unsigned kDeadTap = 0xc0000000u;
constexpr long kMaxMapBytes = 0x80000000L;

#include "car_fused_mma.h"
__device__ __forceinline__ int h_rot(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// matrix tiles of a wave in the source passes: kCT channel tiles x kST sample tiles
constexpr int kCT = 9, kST = 2, kChanWaves = kTE / kCT, kSampWaves = kWaves / kChanWaves;
static_assert(kChanWaves * kCT == kTE && kSampWaves * kST * 16 == kGroup, "2-D wave tiles");

constexpr int kHRow = 32;                                           // floats per row of the shared h tile: 8 x 16 bytes = [hi 0-3 | lo 4-7], rotated per row
constexpr int kLdsH = kLdsW + 2 * kChunkTiles * kTile;              // [2][192][32]        h, fp16 hi / lo halves         48 KB
constexpr int kLdsBias = kLdsH + 2 * kGroup * kHRow;                // [688]
constexpr int kLdsG = kLdsBias + kBiasFloats;                       // [192][16]           geometric query g per sample   12 KB
constexpr int kLdsTapB = kLdsG + kGroup * 16;                       // [192][2] uint
constexpr int kLdsTapW = kLdsTapB + kGroup * 2;                     // [192][2][4]
constexpr int kLdsPe = kLdsTapW + kGroup * 8;                       // [192][2][4]
constexpr int kLdsWpt = kLdsPe + kGroup * 8;                        // [144][4][4]
constexpr int kLdsMax = kLdsWpt + kC * 4;                           // [192] uint          max |e| per sample (float bits; both sources)
constexpr int kLdsFloats = kLdsMax + kGroup;
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

// chunk order (as in csrc/car_fused.hip):  W2 x18 (source 0) | W2 x18 (source 1) | K1 over e_1 x5 | K1 over e_0 x5 | K2 x2 | Q1 | Q2 x2
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

__global__ void __launch_bounds__(kThreads) fused_kernel_v2(const FusedArgs a) {
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
    // Rows of the output tensors: row0 (wave-uniform, lives in scalar registers) is the tile's first sample; a sample's row is row0 + a
    // 32-bit offset — clamped past the end of the rays / steps (a duplicate of a live sample).  Every per-sample address below is a
    // scalar base + one 32-bit vector offset (half the address registers of 64-bit row pointers).
    const int ray0 = bun * kTileRays, pp0 = pg * kTileSteps;
    const long row0 = ((long)nn * a.R + ray0) * a.P + pp0;
    auto rel_of = [&](int gw, int r) -> int {
        const int ray_r = ray0 + tile_ray(gw, r), pp_r = pp0 + tile_step(gw, r);
        return ((ray_r < a.R ? ray_r : a.R - 1) - ray0) * a.P + ((pp_r < a.P ? pp_r : a.P - 1) - pp0);
    };
    auto row_of = [&](int gw, int r) -> long { return row0 + rel_of(gw, r); };
    float* const e0 = a.e + row0 * (2 * kE);
    const int ray_i = ray0 + tile_ray(wave, s), pp = pp0 + tile_step(wave, s);
    const bool live = ray_i < a.R && pp < a.P;

#ifdef CAR_STAMP_ALL
    constexpr bool kStamp = true;
#else
    constexpr bool kStamp = false;
#endif
    long long stamp[12];
    auto mark = [&](int k) { if constexpr (kStamp) stamp[k] = (long long)__builtin_amdgcn_s_memtime(); };
    mark(0);
    float hp, hinv;
    pow2_scale(fmaxf(a.gmeta[0] + a.bias[kBiasScale + 5], 1e-30f), hp, hinv);
    for (int k = tid; k < kC; k += kThreads) {
        const float4 v = *reinterpret_cast<const float4*>(a.wpt + 4 * k);
        float* q = lds + kLdsWpt + 16 * (k >> 2) + (k & 3);
        q[0] = v.x; q[4] = v.y; q[8] = v.z; q[12] = v.w * hp;
    }
    for (int k = tid; k < kBiasFloats; k += kThreads) lds[kLdsBias + k] = a.bias[k];
    if (tid < kGroup) reinterpret_cast<unsigned*>(lds + kLdsMax)[tid] = 0u;
    int g = 0;
    stream_issue_all<0>(a.blob, lds, 0, lane, wave);

    // ---- geometry: one sample per lane of waves 0-2, both source views ----
    const int P = a.P, V = a.V;
    if (wave < kGroup / 64) {
        const int sg = wave * 64 + lane, gwv = sg >> 4, gs = sg & 15;
        const int g_ray = bun * kTileRays + tile_ray(gwv, gs), g_pp = pg * kTileSteps + tile_step(gwv, gs);
        const bool g_live = g_ray < a.R && g_pp < a.P;
        const long gi = row_of(gwv, gs);
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
            const bool dead = mode == 1 && (flags & 4);
            reinterpret_cast<unsigned*>(lds + kLdsTapB)[sg * 2 + sv] = dead ? kDeadTap : (unsigned)node * (unsigned)(kC * 4);
            *reinterpret_cast<float4*>(lds + kLdsTapW + (sg * 2 + sv) * 4) =
                dead ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(w[0] * hp, w[1] * hp, w[2] * hp, w[3] * hp);
            const float px = sv == 0 ? smp.pt_in[0][0] : smp.pt_in[1][0], py = sv == 0 ? smp.pt_in[0][1] : smp.pt_in[1][1],
                        pz = sv == 0 ? smp.pt_in[0][2] : smp.pt_in[1][2];
            *reinterpret_cast<float4*>(lds + kLdsPe + (sg * 2 + sv) * 4) = make_float4(tanhf(px / 5.0f) * hp, tanhf(py / 5.0f) * hp, tanhf(pz / 5.0f) * hp, 0.0f);
        }
        if (g_live) {
            if constexpr (!kStamp) { a.pixel_val[2 * gi] = smp.grid[0]; a.pixel_val[2 * gi + 1] = smp.grid[1]; }
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
    mark(1);

    // ---- gather machinery (as in the product kernel): lane owns rows rr = (lane>>3) + 8*it and channel quad qd = lane & 7 of a chunk ----
    const int qd = lane & 7, r0 = lane >> 3;
    f32x4 bufA[4], bufB[4];
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
    // one row group of chunk (sv, c): start values (point term + bias), the four taps blended in, ReLU, fp16 hi / lo split, out to the
    // shared h tile `hb` — rows of 8 x 16 bytes [hi of channels 0-7 | 8-15 | 16-23 | 24-31 | lo likewise], the 16-byte segments rotated by
    // (row >> 1) & 7 so that the B-operand reads (16 rows, 128 bytes apart) spread over the banks
    const int segx = (qd >> 1) ^ h_rot(r0);
    auto gather_row = [&](const f32x4 (&tap)[4], int sv, int c, int it, float* hb) {
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
        unsigned ha, la, hb_, lb;
        split_pair(fmaxf(lo2[0], 0.f), fmaxf(lo2[1], 0.f), ha, la);
        split_pair(fmaxf(hi2[0], 0.f), fmaxf(hi2[1], 0.f), hb_, lb);
        float* row = hb + (wave * kRows + rr) * kHRow + 2 * (qd & 1);
        // h_rot(r0 + 8) = h_rot(r0) ^ 2: one lane constant (segx) and immediates
        *reinterpret_cast<uint2*>(row + 4 * (segx ^ (2 * it))) = make_uint2(ha, hb_);
        *reinterpret_cast<uint2*>(row + 4 * (segx ^ (4 + 2 * it))) = make_uint2(la, lb);
    };
    const float* lsc = lds + kLdsBias + kBiasScale;
    // rotation of a row's eight 16-byte segments: distinct for the 8 even and the 8 odd rows of a 16-row tile (B-operand reads: 16 rows x
    // one segment) and opposite halves for rows two apart (the gather's 8-byte writes: 4 rows x 8 lanes per LDS pass)
    auto uniform = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    float e_up, e_down;
    {
        const float dW2 = lsc[kLayerW2];
        e_up = uniform(hp / dW2); e_down = uniform(dW2 * hinv);
    }
    auto hbuf = [&](int k) -> float* { return lds + kLdsH + (k & 1) * (kGroup * kHRow); };      // the two shared h tiles

    // first chunk of source 0: nothing to hide it under
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        issue_row(bufA, 0, 0, it);
        gather_row(bufA, 0, 0, it, hbuf(0));
    }
    stream_sync();                                                     // weight chunk 0 landed, h chunk 0 published
    mark(2);
    issue_row(bufA, 0, 1, 0);
    issue_row(bufB, 0, 1, 1);

    // matrix side: wave (cw, sw) owns channels 96 cw .. + 95 of samples 48 sw .. + 47
    const int sw = wave % kSampWaves, cw = wave / kSampWaves;
    f32x4 acc[kCT * kST];                                              // [ct][st]
    const int rot_s = h_rot(s);
    const int boff_hi = (sw * kST * 16 + s) * kHRow + 4 * (q4 ^ rot_s), boff_lo = (sw * kST * 16 + s) * kHRow + 4 * ((4 + q4) ^ rot_s);
#pragma unroll 1
    for (int sv = 0; sv < 2; ++sv) {
#pragma unroll
        for (int ct = 0; ct < kCT; ++ct) {
            const float4 b4 = *reinterpret_cast<const float4*>(lds + kLdsBias + kBiasE + 16 * (kCT * cw + ct) + 4 * q4);
#pragma unroll
            for (int st = 0; st < kST; ++st) acc[ct * kST + st] = f32x4{b4.x * e_up, b4.y * e_up, b4.z * e_up, b4.w * e_up};
        }
#pragma unroll 1
        for (int c = 0; c < kKS; ++c) {
            const int nsv = (c + 1 < kKS) ? sv : 1;
            const int nc = (c + 1 < kKS) ? c + 1 : 0;
            const int n2sv = (c + 2 < kKS) ? sv : 1;
            const int n2c = (c + 2 < kKS) ? c + 2 : c + 2 - kKS;
            const float* wl = lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane + (kCT * cw) * kTile;
            const float* hb = hbuf(g);
            float* hn = hbuf(g + 1);
            const NextChunk nx = next_chunk_w2(a.blob, lds, g + 1);
            half8 bhi[kST], blo[kST];
#pragma unroll
            for (int st = 0; st < kST; ++st) {
                bhi[st] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(hb + boff_hi + st * 16 * kHRow));
                blo[st] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(hb + boff_lo + st * 16 * kHRow));
            }
            // six slots of (2 ds_read_b128 + 9 MFMAs); between them: the DMA pieces (slots 0-2), one row group of the gather each in
            // slots 2 and 4 — blend, split, publish, re-issue
#pragma unroll
            for (int ct = 0; ct < kCT; ++ct) {
                const half8 ah = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wl + ct * kTile));
                const half8 al = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wl + ct * kTile + 256));
#pragma unroll
                for (int st = 0; st < kST; ++st) acc[ct * kST + st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bhi[st], acc[ct * kST + st], 0, 0, 0);
#pragma unroll
                for (int st = 0; st < kST; ++st) acc[ct * kST + st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, blo[st], acc[ct * kST + st], 0, 0, 0);
#pragma unroll
                for (int st = 0; st < kST; ++st) acc[ct * kST + st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bhi[st], acc[ct * kST + st], 0, 0, 0);
                if (ct < kPieces) stream_issue_piece<0>(nx, ct, lane, wave);
#ifndef CAR_V2_SLOT_A
#define CAR_V2_SLOT_A 3
#define CAR_V2_SLOT_B 6
#endif
                if (ct == CAR_V2_SLOT_A) { gather_row(bufA, nsv, nc, 0, hn); issue_row(bufA, n2sv, n2c, 0); }
                else if (ct == CAR_V2_SLOT_B) { gather_row(bufB, nsv, nc, 1, hn); issue_row(bufB, n2sv, n2c, 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            stream_sync<0, 8>();                                       // the 8 tap loads of slots 2 and 4 stay in flight over the barrier
            ++g;
        }
        // ---- end of a source pass: e_sv = acc * e_down; partial max |e| per sample; out as whole 128-byte lines (turned through this
        //      wave's own 16 rows of the h buffer no chunk reads any more: hbuf[g & 1] holds the NEXT pass's first chunk, the other is free)
#pragma unroll
        for (int t = 0; t < kCT * kST; ++t) acc[t] *= e_down;
#pragma unroll
        for (int st = 0; st < kST; ++st) {
            float m = 0.0f;
#pragma unroll
            for (int ct = 0; ct < kCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(acc[ct * kST + st][r]));
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            if (q4 == 0) atomicMax(reinterpret_cast<unsigned*>(lds + kLdsMax) + (sw * kST + st) * 16 + s, __float_as_uint(m));
        }
        {
            // a wave's 144 channels are 4.5 lines of 128 bytes: eight tiles leave as four whole lines (turned through LDS), the tile next to
            // the other wave's half (the last one of wave cw = 0, the first one of cw = 1) as 16 half lines straight from the accumulators.
            // Addresses: scalar base (this pass's columns of the tile's slab of e) + one 32-bit byte offset per row, made HERE from an
            // opaque copy of the lane index — hoisted out of the pass loop they were spilled, and every reload between two stores waits for
            // all the stores before it (scratch and global memory share vmcnt)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int qd_o = lane_o & 7, r0_o = lane_o >> 3, s_o = lane_o & 15, q4_o = lane_o >> 4, rot_o = h_rot(s_o);
            float* turn = hbuf(g + 1) + wave * kRows * kHRow;
            char* const eb = reinterpret_cast<char*>(e0 + sv * kE + 16 * kCT * cw);
            auto store_all = [&](auto first_) {
                constexpr int kFirst = decltype(first_)::value;                    // index of the first paired tile: 0 (cw = 0) or 1 (cw = 1)
                constexpr int kSingle = kFirst == 0 ? kCT - 1 : 0;
#pragma unroll
                for (int st = 0; st < kST; ++st) {
                    const int gw = sw * kST + st;
                    unsigned o_r[2];                                               // byte offset of (row, channel quad qd) inside the slab
#pragma unroll
                    for (int it = 0; it < 2; ++it) o_r[it] = (unsigned)rel_of(gw, r0_o + 8 * it) * (unsigned)(2 * kE * 4) + 16u * qd_o;
                    const unsigned o_s = (unsigned)rel_of(gw, s_o) * (unsigned)(2 * kE * 4) + 16u * q4_o;
#pragma unroll
                    for (int m = 0; m < kCT / 2; ++m) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x4 v = acc[(kFirst + 2 * m + j) * kST + st];
                            *reinterpret_cast<float4*>(turn + s_o * kHRow + 4 * ((4 * j + q4_o) ^ rot_o)) = make_float4(v[0], v[1], v[2], v[3]);
                        }
#pragma unroll
                        for (int it = 0; it < 2; ++it) {
                            const int rr = r0_o + 8 * it;
                            const float4 v = *reinterpret_cast<const float4*>(turn + rr * kHRow + 4 * (qd_o ^ h_rot(rr)));
                            *reinterpret_cast<float4*>(eb + o_r[it] + 64 * (kFirst + 2 * m)) = v;
                        }
                    }
                    const f32x4 v = acc[kSingle * kST + st];
                    *reinterpret_cast<float4*>(eb + o_s + 64 * kSingle) = make_float4(v[0], v[1], v[2], v[3]);
                }
            };
            if (cw == 0) store_all(std::integral_constant<int, 0>()); else store_all(std::integral_constant<int, 1>());
        }
        mark(3 + sv);
    }
    // every wave's e rows are out (vmcnt(0)) before any wave reads them back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- k1 = Wk1 [e_0 ; e_1] + bk1: both halves take their B operands back from the output tensor (L2) by LDS-DMA — whole 128-byte
    //      lines, two K steps ahead, into this wave's own rows of the two h buffers.  Order and arithmetic as in the product kernel: the
    //      e_1 half first, then e_0, one per-sample power of two for both.
    float p, pinv;
    pow2_scale(fmaxf(__uint_as_float(reinterpret_cast<const unsigned*>(lds + kLdsMax)[wave * kRows + s]), 1e-30f), p, pinv);
    f32x4 k1[kTD];
    init_bias<kTD>(k1, lds + kLdsBias + kBiasK1, q4, p / lsc[kLayerK1]);
    unsigned o_row[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) o_row[it] = (unsigned)rel_of(wave, r0 + 8 * it) * (unsigned)(2 * kE);
    auto ebuf = [&](int k) -> float* { return hbuf(k) + wave * kRows * kHRow; };
    auto wait_vm = [&](auto n) {
        constexpr int N = decltype(n)::value;
        if constexpr (N >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    constexpr int kSteps = kTE / 2;                                    // 9 K steps per half, two per weight chunk
    auto k1_half = [&](auto g0_, int col0) {
        constexpr int G0 = decltype(g0_)::value;
        unsigned esrc[2];                                              // byte offset of the lane's 16 bytes inside the tile's slab of e
#pragma unroll
        for (int it = 0; it < 2; ++it) esrc[it] = 4u * (o_row[it] + (unsigned)(col0 + 4 * (qd ^ h_rot(r0 + 8 * it))));
        auto issue_e = [&](int m) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(ebuf(m) + it * 256));
                const float* gbase = e0 + 32 * m;                      // scalar base + the lane's 32-bit offset
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(esrc[it]), "s"(lds_dst), "s"(gbase) : "memory");
            }
        };
        issue_e(0);
        issue_e(1);
#pragma unroll
        for (int c = 0; c < kChK1; ++c) {
            const int gg = G0 + c;
            const float* wl = lds + kLdsW + (gg & 1) * kChunkTiles * kTile + 4 * lane;
            const NextChunk nx = next_chunk(a.blob, lds, gg + 1);
#pragma unroll
            for (int kl = 0; kl < 2; ++kl) {
                const int m = 2 * c + kl;
                if (m < kSteps) {
                    if (kl == 0) { if (m + 1 < kSteps) wait_vm(std::integral_constant<int, 2>()); else wait_vm(std::integral_constant<int, 0>()); }
                    else { if (m + 1 < kSteps) wait_vm(std::integral_constant<int, 5>()); else wait_vm(std::integral_constant<int, 3>()); }
                    const float* eb = ebuf(m) + s * 32;
                    const float4 x0 = *reinterpret_cast<const float4*>(eb + 4 * (q4 ^ rot_s));
                    const float4 x1 = *reinterpret_cast<const float4*>(eb + 4 * ((4 + q4) ^ rot_s));
                    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    half8 bhi, blo;
                    split8(x, p, bhi, blo);
#pragma unroll
                    for (int q = 0; q < kTD / 2; ++q) {
                        const float* w0 = wl + ((kl * kTD + 2 * q) * 2) * 256;
                        mfma_pair<0>(k1[2 * q], k1[2 * q + 1], w0, w0 + 512, bhi, blo);
                        if (kl == 0 && q < kPieces) stream_issue_piece<0>(nx, q, lane, wave);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (m + 2 < kSteps) issue_e(m + 2);
                }
            }
            if (2 * c + 2 < kSteps && 2 * c + 3 < kSteps) wait_vm(std::integral_constant<int, 4>());
            else if (2 * c + 2 < kSteps) wait_vm(std::integral_constant<int, 2>());
            else wait_vm(std::integral_constant<int, 0>());
            __syncthreads();
        }
    };
    k1_half(std::integral_constant<int, kG_K1b>(), kE);
    mark(7);
    k1_half(std::integral_constant<int, kG_K1a>(), 0);
    mark(8);
    scale_acc<kTD>(k1, lsc[kLayerK1] * pinv);
    f32x4 key[kTD];
    pow2_scale(fmaxf(sample_max<kTD, true>(k1), 1e-30f), p, pinv);
    init_bias<kTD>(key, lds + kLdsBias + kBiasK2, q4, p / lsc[kLayerK2]);
    chained_layer<kTD, true, 0, kG_K2>(key, k1, p, a.blob, lds, lane, wave);
    scale_acc<kTD>(key, lsc[kLayerK2] * pinv);
    mark(5);

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
    stream_issue_all<0>(a.blob, lds, kG_Q1 + 1, lane, wave);
    small_layer(t1, ghi, glo, lds + kLdsW + (kG_Q1 & 1) * kChunkTiles * kTile + 4 * lane);
    stream_sync<0>();
    scale_acc<kTD>(t1, lsc[kLayerQ1] * pinv);
    pow2_scale(fmaxf(sample_max<kTD, true>(t1), 1e-30f), p, pinv);
    init_bias<kTD>(qv, lds + kLdsBias + kBiasQ2, q4, p / lsc[kLayerQ2]);
    chained_layer<kTD, true, 0, kG_Q2>(qv, t1, p, a.blob, lds, lane, wave);
    scale_acc<kTD>(qv, lsc[kLayerQ2] * pinv);
    float dot = 0.0f;
#pragma unroll
    for (int t = 0; t < kTD; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dot = fmaf(key[t][r], qv[t][r], dot);
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if (live) {
        const unsigned rel = (unsigned)rel_of(wave, s);
        store_rows<kTD>(qv, a.qry + row0 * kD + rel * (unsigned)kD, q4);
        if (q4 == 0) (a.logit + row0)[rel] = dot / 16.0f;
    }
    mark(6);
    if constexpr (kStamp) {
        if (tid == 0) {
            long long* out = reinterpret_cast<long long*>(a.pixel_val) + (long)blk * 16;
            for (int k = 0; k < 10; ++k) out[k] = stamp[k];
        }
    }
}

}  // namespace

extern "C" int car_fused_samples_ws(const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h, int lat_w,
                                    int lat_pad, const float* gmeta, const float* wpt, const float* blob, const float* bias, int b, int V, int R, int P,
                                    int H, int W, int no_sample, float* e, float* qry, float* g, float* logit, float* pt, float* pixel_val, void* stream) {
    CAR_REQUIRE(poses && rays && steps && lattice && gmeta && wpt && blob && bias, "car_fused_samples_v2: null input");
    CAR_REQUIRE(!no_sample, "car_fused_samples_v2: epipolar sampling only");
    CAR_REQUIRE(e && qry && g && logit && pt && pixel_val, "car_fused_samples_v2: null output");
    CAR_REQUIRE(V == 2, "car_fused_samples_v2: built for V = 2 (got %d)", V);
    CAR_REQUIRE((long)lat_h * lat_w * (kC * 4) < kMaxMapBytes, "car_fused_samples_v2: lattice too large");
    FusedArgs a;
    a.poses = (const CarPose*)poses; a.rays = (const CarRay*)rays; a.steps = steps;
    a.lattice = lattice; a.lh = lat_h; a.lw = lat_w; a.pad = lat_pad;
    a.map_bytes = (unsigned)((long)lat_h * lat_w * (kC * 4));
    a.sx = (float)((lat_w - 2 * lat_pad + 1) / 2); a.sy = (float)((lat_h - 2 * lat_pad + 1) / 2);
    a.gmeta = gmeta; a.wpt = wpt; a.blob = blob; a.bias = bias;
    a.b = b; a.V = V; a.R = R; a.P = P; a.H = H; a.W = W;
    a.S = (long)b * V * R * P;
    a.e = e; a.qry = qry; a.g = g; a.logit = logit; a.pt = pt; a.pixel_val = pixel_val;
    const long groups = (long)b * V * car_div_up(R, kTileRays) * car_div_up(P, kTileSteps);
    hipError_t e1 = hipFuncSetAttribute((const void*)fused_kernel_v2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e1 != hipSuccess) { car_set_error("car_fused_samples_v2: cannot reserve %zu bytes of LDS: %s", kLdsBytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(fused_kernel_v2, dim3((unsigned)groups), dim3(kThreads), kLdsBytes, (hipStream_t)stream, a);
    CAR_CHECK_LAUNCH("car_fused_samples_v2");
    return CAR_OK;
}
