// car_fused.hip — the per-sample part of the render forward as ONE kernel (first half of a14).
//
// For every epipolar sample (V = 2 context views) it computes, without touching HBM in between:
//   geometry   pixel_val, fp64 Pluecker point, cross-view projections, geometric query g      (car_geom.h; models.py:261-331, 494-528)
//   encode     h_s = relu(sum_l bilinear(G_l) + Wpt tanh(pt_s/5) + b1) for both source views s (per-texel first layer, see car_encode.hip)
//   e          = [W2 h_0 + b2 ; W2 h_1 + b2]                      576 -> 288 per source        (models.py:333-344)
//   qry        = Wq2 relu(Wq1 g + bq1) + bq2                      16 -> 128 -> 128             (models.py:529)
//   ug         = Wr1[:,128:] g + br1                              local half of round 2's query (models.py:552-553)
//   key        = Wk2 relu(Wk1 e + bk1) + bk2                      576 -> 128 -> 128            (models.py:491)
//   logit      = <key, qry> / 16                                                              (models.py:532)
// and writes e, qry, ug, logit, pt, pixel_val.  The per-ray softmax / reductions stay in car_attention.hip.
//
// CDNA4 mapping.  One workgroup = 4 waves = 128 consecutive samples, ONE wave per SIMD (up to 512 VGPRs): each wave owns
// 32 samples for the whole chain.  All layers use v_mfma_f32_32x32x2_f32 with weights as the A operand and samples as the
// B operand (see car_linear.hip), so a layer's accumulators (lane = sample, registers = channels) are exactly the next
// layer's B operands: e, key and qry never leave the register file ("chained" layers; the K index of a chained layer is
// permuted to the accumulator layout k = 32*T + (r&3) + 8*(r>>2) + 4*(lane>>5), which the host bakes into the packed
// weights).  Only the encode output has to be transposed (gather lanes own channels, MFMA lanes own samples); it goes
// through a wave-private 32x32 LDS tile per K chunk.  The 12 tap loads per output float4 of chunk c+1 are issued in three
// level-sized batches between the three MFMA sub-blocks of chunk c, so their L2 latency hides under ~3000 matrix cycles.
// The weights of all six layers (1.1 MB) stream L2 -> LDS by LDS-DMA, double buffered, one barrier per 32-wide K chunk,
// shared by the four waves.  fp32 MFMA issues one 32x32x2 per 64 cycles per SIMD: everything else (gather FMAs, tanh,
// fp64 geometry, LDS traffic) fits in its shadow; the kernel is matrix-pipe bound on 0.89 MFLOP per sample.
#include "car_common.h"
#include "car_geom.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

// PREC = 1: the 576 -> 288 layer runs on the f16 matrix pipe with both operands split into two fp16 halves,
//   x = hi + lo,  hi = fp16(x), lo = fp16(x - hi),      x*w ~= hi_x*hi_w + hi_x*lo_w + lo_x*hi_w
// (three v_mfma_f32_32x32x16_f16 per 16-wide K step instead of eight v_mfma_f32_32x32x2_f32: every fp16 x fp16 product is
// exact in the fp32 accumulator, the dropped lo*lo term is 2^-22 relative, so the result is fp32-class — it is checked
// against the oracle at 1e-4 like the fp32 kernel, which stays selectable).  The weights are pre-scaled by 2^kWShift on
// the host so their low halves stay out of the fp16 subnormal range; the accumulators are scaled back exactly.
constexpr int kWShift = 8;

constexpr int kC = 576;            // feature channels = width of h
constexpr int kE = 288;            // per-source width of e
constexpr int kD = 128;            // hidden width of the key / query MLPs
constexpr int kKT = kC / 32;       // 18 K-chunks of the 576 -> 288 layer
constexpr int kNTE = kE / 32;      // 9 output tiles
constexpr int kNTD = kD / 32;      // 4 output tiles
constexpr int kTile = 1024;        // packed floats per (chunk, tile)
constexpr int kStageLd = 36;       // row stride of the wave-private h tile (floats): conflict-free b128 reads and writes

// ---- packed-weight blob: tile offsets (units of kTile floats) of each layer, in consumption order -------------------
constexpr int kOffW2 = 0;                          // 18 chunks x 9 tiles, standard K mapping
constexpr int kOffQ1 = kOffW2 + kKT * kNTE;        // 1 chunk x 4 tiles, standard, bias folded at k = 16
constexpr int kOffQ2 = kOffQ1 + kNTD;              // 4 chunks x 4 tiles, chained
constexpr int kOffUG = kOffQ2 + 4 * kNTD;          // 1 chunk x 4 tiles, standard, bias folded
constexpr int kOffK1 = kOffUG + kNTD;              // 18 chunks x 4 tiles, chained over [e_0 ; e_1]
constexpr int kOffK2 = kOffK1 + 2 * kNTE * kNTD;   // 4 chunks x 4 tiles, chained
constexpr int kBlobTiles = kOffK2 + 4 * kNTD;
constexpr int kNumChunks = 2 * kKT + 2 * 5 + 2 + 1 + 2 + 1;           // 52 weight chunks per 128-sample pass

// bias table (floats): b2[288] | bq2[128] | bk1[128] | bk2[128]
constexpr int kBiasE = 0, kBiasQ2 = kE, kBiasK1 = kE + kD, kBiasK2 = kE + 2 * kD, kBiasFloats = kE + 3 * kD;

// ---- dynamic LDS carve-up (floats) --------------------------------------------------------------------------------
constexpr int kLdsW = 0;                                   // [2][9][1024]          weight chunks           72 KB
constexpr int kLdsStage = kLdsW + 2 * kNTE * kTile;        // [4][32][36]           h tiles, wave private    18 KB
constexpr int kLdsTapI = kLdsStage + 4 * 32 * kStageLd;    // [4][32][2][3][4] int  tap texel indices        12 KB
constexpr int kLdsTapW = kLdsTapI + 4 * 32 * 2 * 3 * 4;    // [4][32][2][3][4]      tap weights              12 KB
constexpr int kLdsPe = kLdsTapW + 4 * 32 * 2 * 3 * 4;      // [4][32][2][4]         tanh(pt_s/5)              4 KB
constexpr int kLdsWpt = kLdsPe + 4 * 32 * 2 * 4;           // [576][4]              (W1[:,C:C+3], b1)         9 KB
constexpr int kLdsBias = kLdsWpt + kC * 4;                 // [672]                                           2.6 KB
constexpr int kLdsFloats = kLdsBias + kBiasFloats;
constexpr size_t kLdsBytes = (size_t)kLdsFloats * sizeof(float);

struct FusedArgs {
    const CarPose* poses;
    const CarRay* rays;
    const float* steps;
    const float* gmap[3];
    int gh[3], gw[3];
    const float* wpt;        // [576][4]
    const float* blob;       // kBlobTiles * 1024 floats
    const float* bias;       // kBiasFloats
    int b, V, R, P, H, W;
    int xcd_bands;           // 1: remap workgroups so each XCD gets a contiguous band of rays
    int prec;                // 0: fp32 MFMA for the 576->288 layer; 1: split-fp16 MFMA (blob packed accordingly)
    long S;                  // b*V*R*P samples
    float* e;                // [S, 576]
    float* qry;              // [S, 128]
    float* ug;               // [S, 128]
    float* logit;            // [S]
    float* pt;               // [S, 3]
    float* pixel_val;        // [S, 2]
};

// consumption order (a "chunk" is what one LDS buffer holds between two barriers):
//   W2 x18 (source 0, 9 tiles) | K1 over e_0 x5 (source tiles {0,1},{2,3},{4,5},{6,7},{8}: 8,8,8,8,4 tiles) | W2 x18 (source 1) |
//   K1 over e_1 x5 | K2 x2 (8 tiles) | Q1 (4) | Q2 x2 (8) | UG (4)
// The chained layers take two 32-row source tiles per chunk so that a chunk lasts >= 8192 matrix cycles: an LDS-DMA needs
// ~1.1 us from issue to landing, which a 4096-cycle chunk cannot hide.
constexpr int kChK1 = 5;                                   // chunks per K1 half
constexpr int kG_K1a = kKT, kG_W2b = kG_K1a + kChK1, kG_K1b = kG_W2b + kKT, kG_K2 = kG_K1b + kChK1, kG_Q1 = kG_K2 + 2,
              kG_Q2 = kG_Q1 + 1, kG_UG = kG_Q2 + 2;
__device__ __forceinline__ int chunk_tile_offset(int g) {
    if (g < kG_K1a) return kOffW2 + g * kNTE;
    if (g < kG_W2b) return kOffK1 + (g - kG_K1a) * 2 * kNTD;
    if (g < kG_K1b) return kOffW2 + (g - kG_W2b) * kNTE;
    if (g < kG_K2) return kOffK1 + kNTE * kNTD + (g - kG_K1b) * 2 * kNTD;
    if (g < kG_Q1) return kOffK2 + (g - kG_K2) * 2 * kNTD;
    if (g < kG_Q2) return kOffQ1;
    if (g < kG_UG) return kOffQ2 + (g - kG_Q2) * 2 * kNTD;
    return kOffUG;
}
__device__ __forceinline__ int chunk_tiles(int g) {
    if (g < kG_K1a || (g >= kG_W2b && g < kG_K1b)) return kNTE;
    if (g == kG_W2b - 1 || g == kG_K2 - 1 || g == kG_Q1 || g == kG_UG) return kNTD;       // odd last K1 tile, Q1, UG
    return 2 * kNTD;
}

// LDS-DMA of weight chunk g into buffer (g & 1): every wave copies a quarter of every tile (see car_linear.hip for why
// this is inline asm).  Nothing is issued past the last chunk.
__device__ __forceinline__ void stream_issue(const float* __restrict__ blob, float* lds, int g, int tid, int wave) {
    if (g >= kNumChunks) return;
    const float* src = blob + (long)chunk_tile_offset(g) * kTile;
    float* dst = lds + kLdsW + (g & 1) * kNTE * kTile;
    const int nt = chunk_tiles(g);
    for (int t = 0; t < nt; ++t) {
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(dst + 4 * (t * 256 + wave * 64)));
        const float* gsrc = src + 4 * (t * 256 + tid);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
}
// Descriptor of the weight chunk to prefetch, resolved ONCE per chunk (scalar code with branches) so that the per-tile
// issue below is straight-line code: a branch inside the unrolled MFMA body splits it into basic blocks and pins the
// gather/DMA pieces (and their waits) to the block boundaries.
struct NextChunk { const float* src; float* dst; int nt; };
__device__ __forceinline__ NextChunk next_chunk(const float* __restrict__ blob, float* lds, int g) {
    const int ge = g < kNumChunks ? g : kNumChunks - 1;            // past the end: re-copy the last chunk onto itself (same bytes)
    NextChunk n;
    n.src = blob + (long)chunk_tile_offset(ge) * kTile;
    n.dst = lds + kLdsW + (ge & 1) * kNTE * kTile;
    n.nt = chunk_tiles(ge);
    return n;
}
// one tile (4 KB) of the next weight chunk; t may exceed the chunk's tile count, then an earlier tile is copied again
// (identical bytes, harmless).  Must only run after the barrier that ended the chunk which last used that buffer.
template <int ABL = 0>
__device__ __forceinline__ void stream_issue_tile(const NextChunk& n, int t, int tid, int wave) {
    if constexpr (ABL == 3) return;
    int te = t < n.nt ? t : t - n.nt;
    te = te < n.nt ? te : te - n.nt;
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(n.dst + 4 * (te * 256 + wave * 64)));
    const float* gsrc = n.src + 4 * (te * 256 + tid);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// end of a chunk: the DMA of the next chunk has landed and every wave is done reading the current one
template <int ABL = 0>
__device__ __forceinline__ void stream_sync() {
    if constexpr (ABL == 3) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// 16 MFMA steps of one chunk on tiles [T0, T0+NTS) of acc, A from the LDS weight buffer, B from 16 registers
template <int NT, int T0, int NTS>
__device__ __forceinline__ void mfma_tiles(f32x16 (&acc)[NT], const float (&bv)[16], const float* wl) {
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
#pragma unroll
        for (int t = T0; t < T0 + NTS; ++t) {
            const float4 a = *reinterpret_cast<const float4*>(wl + (t * 4 + j4) * 256);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv[4 * j4 + 0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv[4 * j4 + 1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bv[4 * j4 + 2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bv[4 * j4 + 3], acc[t], 0, 0, 0);
        }
    }
}

// accumulators start at the layer's bias: lane (s, h) register r of tile t holds channel 32 t + (r&3) + 8 (r>>2) + 4 h
template <int NT>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NT], const float* lbias, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = lbias[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
}

template <int NT>
__device__ __forceinline__ void store_rows(const f32x16 (&acc)[NT], float* row, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(row + 32 * t + 8 * g + 4 * h) =
                make_float4(acc[t][4 * g + 0], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
}

// one chained layer with 128 outputs: B operands are the NSRC x 16 registers of `src` (optionally through ReLU)
template <int NSRC, bool RELU, int ABL = 0, int PREC = 0>
__device__ __forceinline__ void chained_layer(f32x16 (&acc)[kNTD], const f32x16 (&src)[NSRC], const float* __restrict__ blob,
                                              float* lds, int& g, int tid, int wave, int lane) {
#pragma unroll
    for (int T0 = 0; T0 < NSRC; T0 += 2) {
        constexpr int kG = 4 * kNTD;                                   // MFMA groups per source tile
        const int nsrc = T0 + 1 < NSRC ? 2 : 1;
        const float* wl = lds + kLdsW + (g & 1) * kNTE * kTile + 4 * lane;
        const NextChunk nx = next_chunk(blob, lds, g + 1);
        // slots are pinned with sched_barrier(0) (hipcc otherwise regroups the pieces and shrinks the latency slack they
        // were placed for); the A operand of the next group is therefore read one slot ahead by hand
        if constexpr (PREC == 1) {
            // split-fp16 path: per source tile, the 16 accumulator values of this lane become two 8-wide K groups (hi, lo);
            // packed tile (Tl, t): [kg][hi|lo][lane][8 halves].  3 MFMAs of 32 cycles per (t, kg).
#pragma unroll
            for (int Tl = 0; Tl < 2; ++Tl) {
                if (Tl < nsrc) {
                    const f32x16& sv = src[T0 + Tl < NSRC ? T0 + Tl : NSRC - 1];
                    half8 bhi[2], blo[2];
#pragma unroll
                    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
                        for (int e8 = 0; e8 < 8; ++e8) {
                            float x = sv[8 * kg + e8];
                            if (RELU) x = fmaxf(x, 0.f);
                            const _Float16 hi = (_Float16)x;
                            bhi[kg][e8] = hi;
                            blo[kg][e8] = (_Float16)(x - (float)hi);
                        }
#pragma unroll
                    for (int t = 0; t < kNTD; ++t)
#pragma unroll
                        for (int kg = 0; kg < 2; ++kg) {
                            const float* wt = wl + (Tl * kNTD + t) * kTile;
                            const half8 ah = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wt + ((kg * 2 + 0) * 64) * 4));
                            const half8 al = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wt + ((kg * 2 + 1) * 64) * 4));
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhi[kg], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blo[kg], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhi[kg], acc[t], 0, 0, 0);
                            const int gq = (Tl * kNTD + t) * 2 + kg;
                            if (gq < kNTE) stream_issue_tile<ABL>(nx, gq, tid, wave);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            }
            // a single-source-tile chunk has only 8 slots: issue the rest of a 9-tile successor here
#pragma unroll
            for (int gq = nsrc * 2 * kNTD; gq < kNTE; ++gq) stream_issue_tile<ABL>(nx, gq, tid, wave);
        } else {
        float4 aw[kNTD];
#pragma unroll
        for (int t = 0; t < kNTD; ++t) aw[t] = *reinterpret_cast<const float4*>(wl + (t * 4) * 256);
        // (source tile Tl, step group j4, step e, out tile t): round-robin over the 4 out tiles, see the e-path loop
#pragma unroll
        for (int Tl = 0; Tl < 2; ++Tl)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < kNTD; ++t) {
            const int mi = Tl * 4 * kG + j4 * 4 * kNTD + e * kNTD + t;
            if (mi < nsrc * 4 * kG) {
            const f32x16& sv = src[T0 + Tl < NSRC ? T0 + Tl : NSRC - 1];
            float bq = sv[4 * j4 + e];
            if (RELU) bq = fmaxf(bq, 0.f);
            const float av = e == 0 ? aw[t].x : e == 1 ? aw[t].y : e == 2 ? aw[t].z : aw[t].w;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq, acc[t], 0, 0, 0);
            if (e == 3) {                                              // refresh tile t's A operand for its next step group
                const int nj = j4 + 1 < 4 ? j4 + 1 : 0, nT = j4 + 1 < 4 ? Tl : Tl + 1;
                if (nT < nsrc) aw[t] = *reinterpret_cast<const float4*>(wl + ((nT * kNTD + t) * 4 + nj) * 256);
            }
            if (mi % 4 == 3) {
                const int gq = mi / 4;
                if (gq < kNTE) stream_issue_tile<ABL>(nx, gq, tid, wave);            // next chunk may have up to 9 tiles
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        }
        }
        stream_sync<ABL>();
        ++g;
    }
}

// scale helpers of the split-fp16 layers: accumulators start at bias * 2^kWShift and are scaled back exactly
template <int NT>
__device__ __forceinline__ void scale_acc(f32x16 (&acc)[NT], float f) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] *= f;
}

// ABL > 0 are timing-only ablations (results are wrong by construction), selected with CAR_FUSED_ABLATE for tools/bench_fused.py:
//   1 no tap loads   2 no gather work at all   3 = 2 + no weight DMA / barriers   4 = 0 but without the chained layers
//   5 / 6 / 7 no tap loads of pyramid level 2 / 1 / 0
template <int ABL, int SCHED = 0, int PREC = 0>
__global__ void __launch_bounds__(256, 1) fused_sample_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 31, h = lane >> 5;
    // XCD-aware placement: workgroup b runs on XCD b % 8 (observed dispatch order, used for speed only), so give every XCD
    // a contiguous band of sample groups (= consecutive rays = neighbouring epipolar lines): each private L2 then holds
    // its own band's texels instead of one eighth of everybody's.
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    if (a.xcd_bands) {
        const int q8 = nblk / 8, r8 = nblk % 8, xcd = blk % 8, idx = blk / 8;
        blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;     // bijective for any grid size
    }
    const long i_raw = (long)blk * 128 + wave * 32 + s;
    const bool live = i_raw < a.S;
    const long i = live ? i_raw : a.S - 1;

    // ---- tables shared by the workgroup -----------------------------------------------------------------------------
    for (int k = tid; k < kC; k += 256) *reinterpret_cast<float4*>(lds + kLdsWpt + 4 * k) = *reinterpret_cast<const float4*>(a.wpt + 4 * k);
    for (int k = tid; k < kBiasFloats; k += 256) lds[kLdsBias + k] = a.bias[k];
    int g = 0;                                     // index of the weight chunk being consumed
    stream_issue(a.blob, lds, 0, tid, wave);

    // ---- geometry of this lane's sample; lane half h prepares source view h ---------------------------------------
    const int P = a.P, V = a.V;
    const int p = (int)(i % P);
    const long nr = i / P;
    const int n = (int)(nr / a.R);
    const int v = n % V, sc = n / V;
    const CarPose& Ps = a.poses[n];
    const CarRay ray = a.rays[nr];
    CarSample smp;
    for (int k = 0; k < 2; ++k) smp.grid[k] = ray.start[k] + (ray.end[k] - ray.start[k]) * a.steps[p];
    car_sample_setup(Ps, a.poses + sc * 2, ray, 2, a.H, a.W, &smp);   // V == 2: literal so the per-view loops unroll (no scratch)
    {
        const int sv = h;                                              // this lane prepares source view sv of sample s
        float gx, gy;
        int mode, m;
        if (sv == v) { gx = smp.grid[0]; gy = smp.grid[1]; mode = 0; m = n; }
        else { gx = sv == 0 ? smp.grid_in[0][0] : smp.grid_in[1][0]; gy = sv == 0 ? smp.grid_in[0][1] : smp.grid_in[1][1]; mode = 1; m = sc * V + sv; }
        int* ti = reinterpret_cast<int*>(lds + kLdsTapI) + ((wave * 32 + s) * 2 + sv) * 12;
        float* tw = lds + kLdsTapW + ((wave * 32 + s) * 2 + sv) * 12;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            int idx[4];
            float w[4];
            car_bilinear_taps(gx, gy, a.gw[l], a.gh[l], mode, idx, w);
#pragma unroll
            for (int t = 0; t < 4; ++t) { ti[4 * l + t] = m * a.gh[l] * a.gw[l] + idx[t]; tw[4 * l + t] = w[t]; }
        }
        float* pe = lds + kLdsPe + ((wave * 32 + s) * 2 + sv) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) pe[k] = tanhf((sv == 0 ? smp.pt_in[0][k] : smp.pt_in[1][k]) / 5.0f);
        pe[3] = 0.0f;
    }
    if (live && h == 0) {
        a.pixel_val[2 * i] = smp.grid[0]; a.pixel_val[2 * i + 1] = smp.grid[1];
        a.pt[3 * i + 0] = smp.pt[0]; a.pt[3 * i + 1] = smp.pt[1]; a.pt[3 * i + 2] = smp.pt[2];
    }
    // B operand of the two K=16 layers fed by g (standard mapping: lanes 0-31 carry k = 0..15, lanes 32-63 carry the
    // folded bias input k = 16 and zeros)
    float gb[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) gb[k] = h == 0 ? smp.g[k] : (k == 0 ? 1.0f : 0.0f);
    __syncthreads();                                                   // tables and tap records visible

    // ---- gather machinery: lane owns rows rr = (lane>>3) + 8*it (it = 0..3) and channel quad qd = lane & 7 of a chunk ----
    const int qd = lane & 7, r0 = lane >> 3;
    float* stage = lds + kLdsStage + wave * 32 * kStageLd;
    float4 hacc[4];                                                    // h of the chunk being gathered, 4 rows x float4
    float4 tapA[16], tapB[16];                                         // two levels' worth of taps (4 rows x 4 taps) in flight

    // The gather of chunk c+1 is cut into per-row pieces that are dealt out between the 36 MFMA groups of chunk c (see
    // the slot table in the chunk loop): a wave has ONE instruction stream, so anything not sitting between two MFMAs
    // leaves the matrix pipe idle.
    auto issue_row = [&](float4 (&tap)[16], int sv, int c, int l, int it) {          // 4 tap loads of one row
        if constexpr (ABL == 1 || ABL == 2 || ABL == 3) return;
        if constexpr (ABL >= 5 && ABL <= 7) { if (l == 7 - ABL) return; }              // 5: no level-2 taps, 6: no level 1, 7: no level 0
        const float* base = a.gmap[l] + 32 * c + 4 * qd;
        const int* ti = reinterpret_cast<const int*>(lds + kLdsTapI) + ((wave * 32 + r0 + 8 * it) * 2 + sv) * 12 + 4 * l;
        const int4 id = *reinterpret_cast<const int4*>(ti);
        tap[4 * it + 0] = *reinterpret_cast<const float4*>(base + (long)id.x * kC);
        tap[4 * it + 1] = *reinterpret_cast<const float4*>(base + (long)id.y * kC);
        tap[4 * it + 2] = *reinterpret_cast<const float4*>(base + (long)id.z * kC);
        tap[4 * it + 3] = *reinterpret_cast<const float4*>(base + (long)id.w * kC);
    };
    auto blend_row = [&](const float4 (&tap)[16], int sv, int l, int it) {           // hacc[it] += sum_t w_t tap_t
        if constexpr (ABL == 2 || ABL == 3) return;
        const float4 w = *reinterpret_cast<const float4*>(lds + kLdsTapW + ((wave * 32 + r0 + 8 * it) * 2 + sv) * 12 + 4 * l);
        const float ww[4] = {w.x, w.y, w.z, w.w};
        float4 acc4 = hacc[it];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 gq = tap[4 * it + t];
            acc4.x = fmaf(ww[t], gq.x, acc4.x); acc4.y = fmaf(ww[t], gq.y, acc4.y);
            acc4.z = fmaf(ww[t], gq.z, acc4.z); acc4.w = fmaf(ww[t], gq.w, acc4.w);
        }
        hacc[it] = acc4;
    };
    auto affine_row = [&](int sv, int c, int it) {                     // hacc[it] = Wpt tanh(pt_sv/5) + b1 (start value)
        if constexpr (ABL == 2 || ABL == 3) return;
        const int rr = r0 + 8 * it;
        const float4 pe = *reinterpret_cast<const float4*>(lds + kLdsPe + ((wave * 32 + rr) * 2 + sv) * 4);
        const float4* wp = reinterpret_cast<const float4*>(lds + kLdsWpt + 4 * (32 * c + 4 * qd));
        const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
        hacc[it] = make_float4(fmaf(w0.z, pe.z, fmaf(w0.y, pe.y, w0.x * pe.x)) + w0.w,
                               fmaf(w1.z, pe.z, fmaf(w1.y, pe.y, w1.x * pe.x)) + w1.w,
                               fmaf(w2.z, pe.z, fmaf(w2.y, pe.y, w2.x * pe.x)) + w2.w,
                               fmaf(w3.z, pe.z, fmaf(w3.y, pe.y, w3.x * pe.x)) + w3.w);
    };
    auto finish_row = [&](int it) {                                    // ReLU, into the wave's h tile
        if constexpr (ABL == 2 || ABL == 3) return;
        const float4 o = hacc[it];
        *reinterpret_cast<float4*>(stage + (r0 + 8 * it) * kStageLd + 4 * qd) =
            make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    };

    // ---- e_s = W2 h_s + b2, then immediately its share of k1 = Wk1 [e_0 ; e_1] + bk1 (chained on the accumulators), then e_s
    //      is stored and its registers are reused for the other source.  The gather of chunk cc+1 hides under the MFMAs of cc.
    // first chunk of source 0: nothing to hide it under
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        affine_row(0, 0, it);
#pragma unroll
        for (int l = 0; l < 3; ++l) { issue_row(tapA, 0, 0, l, it); blend_row(tapA, 0, l, it); }
        finish_row(it);
    }
    stream_sync();                                                     // weight chunk 0 landed

    f32x16 k1[kNTD];
    init_bias<kNTD>(k1, lds + kLdsBias + kBiasK1, h);
    if constexpr (PREC == 1) scale_acc<kNTD>(k1, (float)(1 << kWShift));
    f32x16 acc[kNTE];
    float bv[16];                                                      // B operands of the current chunk (this lane's 16 channels)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(stage + s * kStageLd + 16 * h + 4 * q);
        bv[4 * q + 0] = x.x; bv[4 * q + 1] = x.y; bv[4 * q + 2] = x.z; bv[4 * q + 3] = x.w;
    }
#pragma unroll 1
    for (int sv = 0; sv < 2; ++sv) {
        init_bias<kNTE>(acc, lds + kLdsBias + kBiasE, h);
        if constexpr (PREC == 1) {
#pragma unroll
            for (int t = 0; t < kNTE; ++t) acc[t] *= (float)(1 << kWShift);        // the packed weights carry 2^kWShift
        }
#pragma unroll 1
        for (int c = 0; c < kKT; ++c) {
            // next chunk to gather: (sv, c+1), or (1, 0) after the last chunk of source 0.  Kept branch-free on purpose (a
            // conditional gather makes hipcc copy the in-flight tap registers at the block boundary, i.e. wait for them
            // before the MFMAs): after the very last chunk the gather harmlessly re-reads chunk (1, 0).
            const int nsv = (c + 1 < kKT) ? sv : 1;
            const int nc = (c + 1 < kKT) ? c + 1 : 0;
            const float* wl = lds + kLdsW + (g & 1) * kNTE * kTile + 4 * lane;
            const NextChunk nx = next_chunk(a.blob, lds, g + 1);
            // 36 groups of (one ds_read_b128 of weights + 4 dependent MFMAs); between them, one piece of the next chunk's
            // gather or of the next weight chunk's DMA issue.  Slot table (level 2 = full resolution, the slowest to
            // arrive, goes first):
            //   0-3 issue L2 -> tapA    4-12 DMA tile 0..8    4-7 affine start values    12-15 issue L1 -> tapB
            //   16-19 blend L2 (tapA)   20-23 issue L0 -> tapA    28-31 blend L1 (tapB)   32-35 blend L0, ReLU, LDS write
            auto piece = [&](int gq) {            // slot gq = 0..35 of the chunk: one piece of the next chunk's gather / DMA issue
                if constexpr (SCHED == 0) {
                    if (gq >= 4 && gq < 13) stream_issue_tile<ABL>(nx, gq - 4, tid, wave);
                    if (gq < 4) issue_row(tapA, nsv, nc, 2, gq);
                    else if (gq < 8) affine_row(nsv, nc, gq - 4);
                    else if (gq >= 12 && gq < 16) issue_row(tapB, nsv, nc, 1, gq - 12);
                    else if (gq >= 16 && gq < 20) blend_row(tapA, nsv, 2, gq - 16);
                    else if (gq >= 20 && gq < 24) issue_row(tapA, nsv, nc, 0, gq - 20);
                    else if (gq >= 28 && gq < 32) blend_row(tapB, nsv, 1, gq - 28);
                    else if (gq >= 32) { blend_row(tapA, nsv, 0, gq - 32); finish_row(gq - 32); }
                } else if constexpr (SCHED == 1) {
                    // both big levels ahead of the DMA in the wave's in-order VMEM queue
                    if (gq >= 8 && gq < 17) stream_issue_tile<ABL>(nx, gq - 8, tid, wave);
                    if (gq < 4) issue_row(tapA, nsv, nc, 2, gq);
                    else if (gq < 8) issue_row(tapB, nsv, nc, 1, gq - 4);
                    else if (gq < 12) affine_row(nsv, nc, gq - 8);
                    else if (gq >= 16 && gq < 20) blend_row(tapA, nsv, 2, gq - 16);
                    else if (gq >= 20 && gq < 24) issue_row(tapA, nsv, nc, 0, gq - 20);
                    else if (gq >= 24 && gq < 28) blend_row(tapB, nsv, 1, gq - 24);
                    else if (gq >= 32) { blend_row(tapA, nsv, 0, gq - 32); finish_row(gq - 32); }
                } else if constexpr (SCHED == 3) {
                    // one tap set in flight at a time (16 loads + the DMA): 8 slots of slack each
                    if (gq >= 4 && gq < 13) stream_issue_tile<ABL>(nx, gq - 4, tid, wave);
                    if (gq < 4) { issue_row(tapA, nsv, nc, 2, gq); affine_row(nsv, nc, gq); }
                    else if (gq >= 8 && gq < 12) blend_row(tapA, nsv, 2, gq - 8);
                    else if (gq >= 12 && gq < 16) issue_row(tapB, nsv, nc, 1, gq - 12);
                    else if (gq >= 20 && gq < 24) blend_row(tapB, nsv, 1, gq - 20);
                    else if (gq >= 24 && gq < 28) issue_row(tapA, nsv, nc, 0, gq - 24);
                    else if (gq >= 32) { blend_row(tapA, nsv, 0, gq - 32); finish_row(gq - 32); }
                } else {
                    // DMA first (3 tiles per slot), every tap batch 16 slots ahead of its consumer except level 0
                    if (gq < 3) { stream_issue_tile<ABL>(nx, 3 * gq, tid, wave); stream_issue_tile<ABL>(nx, 3 * gq + 1, tid, wave); stream_issue_tile<ABL>(nx, 3 * gq + 2, tid, wave); }
                    else if (gq < 7) issue_row(tapA, nsv, nc, 2, gq - 3);
                    else if (gq < 11) issue_row(tapB, nsv, nc, 1, gq - 7);
                    else if (gq < 15) affine_row(nsv, nc, gq - 11);
                    else if (gq >= 20 && gq < 24) blend_row(tapA, nsv, 2, gq - 20);
                    else if (gq >= 24 && gq < 28) { issue_row(tapA, nsv, nc, 0, gq - 24); blend_row(tapB, nsv, 1, gq - 24); }
                    else if (gq >= 32) { blend_row(tapA, nsv, 0, gq - 32); finish_row(gq - 32); }
                }
            };
            if constexpr (PREC == 1) {
                // split this lane's 16 activations (two 8-wide K groups) into fp16 high and low halves
                half8 bhi[2], blo[2];
#pragma unroll
                for (int kg = 0; kg < 2; ++kg)
#pragma unroll
                    for (int e8 = 0; e8 < 8; ++e8) {
                        const float x = bv[8 * kg + e8];
                        const _Float16 hi = (_Float16)x;
                        bhi[kg][e8] = hi;
                        blo[kg][e8] = (_Float16)(x - (float)hi);
                    }
                // packed tile: [kg][hi|lo][lane][8 halves]; 18 slots of (two ds_read_b128, three MFMAs), two pieces each
#pragma unroll
                for (int t = 0; t < kNTE; ++t)
#pragma unroll
                    for (int kg = 0; kg < 2; ++kg) {
                        const float4 ahf = *reinterpret_cast<const float4*>(wl + t * kTile + ((kg * 2 + 0) * 64) * 4);
                        const float4 alf = *reinterpret_cast<const float4*>(wl + t * kTile + ((kg * 2 + 1) * 64) * 4);
                        const half8 ah = __builtin_bit_cast(half8, ahf), al = __builtin_bit_cast(half8, alf);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhi[kg], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blo[kg], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhi[kg], acc[t], 0, 0, 0);
                        piece(2 * (t * 2 + kg));
                        piece(2 * (t * 2 + kg) + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else {
            // MFMA order: for each 4-step group j4, steps e = 0..3 round-robin over the 9 output tiles, so consecutive MFMAs
            // never share an accumulator (a VALU instruction between two MFMAs on the SAME accumulator costs ~40 extra
            // cycles; between independent ones it is nearly free).  The A operand of tile t (4 steps = one ds_read_b128)
            // is refreshed in place right after its last use, 8 MFMAs before it is needed again.
            float4 aw[kNTE];
#pragma unroll
            for (int t = 0; t < kNTE; ++t) aw[t] = *reinterpret_cast<const float4*>(wl + (t * 4) * 256);
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
            for (int qs = 0; qs < kNTE; ++qs) {                          // 9 slots per step group, 4 MFMAs each
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int e = (4 * qs + k4) / kNTE, t = (4 * qs + k4) % kNTE;
                    const float av = e == 0 ? aw[t].x : e == 1 ? aw[t].y : e == 2 ? aw[t].z : aw[t].w;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[4 * j4 + e], acc[t], 0, 0, 0);
                    if (e == 3 && j4 < 3) aw[t] = *reinterpret_cast<const float4*>(wl + (t * 4 + j4 + 1) * 256);
                }
                piece(j4 * kNTE + qs);
                // Pin the slot (see chained_layer).  Measured on MI355X: with one wave per SIMD the gather's VALU/VMEM issue time
                // is NOT hidden under the wave's own MFMAs (SQ_ACTIVE_INST_VALU adds 1:1 to SQ_WAVE_CYCLES), and spreading a
                // piece over the four MFMA gaps with sched_group_barrier is slower than leaving it as one lump (9.4 vs 9.0 ms).
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            // B operands of the next chunk: this wave's own LDS tile, written just above (LDS ops of a wave are in order);
            // read before the barrier so the latency overlaps it
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x = *reinterpret_cast<const float4*>(stage + s * kStageLd + 16 * h + 4 * q);
                bv[4 * q + 0] = x.x; bv[4 * q + 1] = x.y; bv[4 * q + 2] = x.z; bv[4 * q + 3] = x.w;
            }
            stream_sync<ABL>();
            ++g;
        }
        if constexpr (PREC == 1) {
#pragma unroll
            for (int t = 0; t < kNTE; ++t) acc[t] *= 1.0f / (float)(1 << kWShift);  // exact power of two
        }
        if constexpr (ABL == 4) { g += kChK1; }
        else chained_layer<kNTE, false, ABL, PREC>(k1, acc, a.blob, lds, g, tid, wave, lane);
        if (live) store_rows<kNTE>(acc, a.e + i * (2 * kE) + sv * kE, h);
    }
    f32x16 key[kNTD];
    init_bias<kNTD>(key, lds + kLdsBias + kBiasK2, h);
    if constexpr (PREC == 1) { scale_acc<kNTD>(k1, 1.0f / (float)(1 << kWShift)); scale_acc<kNTD>(key, (float)(1 << kWShift)); }
    if constexpr (ABL != 4) chained_layer<kNTD, true, ABL, PREC>(key, k1, a.blob, lds, g, tid, wave, lane);
    else g += 2;
    if constexpr (PREC == 1) scale_acc<kNTD>(key, 1.0f / (float)(1 << kWShift));

    // ---- qry = Wq2 relu(Wq1 g + bq1) + bq2 ;  logit = <key, qry>/16 ;  ug = Wr1[:,128:] g + br1 ---------------------
    f32x16 t1[kNTD], qv[kNTD];
#pragma unroll
    for (int t = 0; t < kNTD; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) t1[t][r] = 0.0f;
    stream_issue(a.blob, lds, g + 1, tid, wave);
    mfma_tiles<kNTD, 0, kNTD>(t1, gb, lds + kLdsW + (g & 1) * kNTE * kTile + 4 * lane);          // q1 (bias folded)
    stream_sync();
    ++g;
    init_bias<kNTD>(qv, lds + kLdsBias + kBiasQ2, h);
    if constexpr (PREC == 1) scale_acc<kNTD>(qv, (float)(1 << kWShift));
    if constexpr (ABL != 4) chained_layer<kNTD, true, ABL, PREC>(qv, t1, a.blob, lds, g, tid, wave, lane);   // qry
    else g += 2;
    if constexpr (PREC == 1) scale_acc<kNTD>(qv, 1.0f / (float)(1 << kWShift));
    float dot = 0.0f;
#pragma unroll
    for (int t = 0; t < kNTD; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dot = fmaf(key[t][r], qv[t][r], dot);
    dot += __shfl_xor(dot, 32, 64);                                    // the other lane half holds the other channels
    if (live) {
        store_rows<kNTD>(qv, a.qry + i * kD, h);
        if (h == 0) a.logit[i] = dot / 16.0f;
    }
#pragma unroll
    for (int t = 0; t < kNTD; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) t1[t][r] = 0.0f;
    stream_issue(a.blob, lds, g + 1, tid, wave);                       // past the last chunk: no-op
    mfma_tiles<kNTD, 0, kNTD>(t1, gb, lds + kLdsW + (g & 1) * kNTE * kTile + 4 * lane);          // ug (bias folded)
    if (live) store_rows<kNTD>(t1, a.ug + i * kD, h);
}

}  // namespace

extern "C" size_t car_fused_blob_floats(void) { return (size_t)kBlobTiles * kTile; }
extern "C" size_t car_fused_bias_floats(void) { return (size_t)kBiasFloats; }

extern "C" int car_fused_samples(const float* poses, const float* rays, const float* steps, const float* const* gmaps,
                                 const int* level_h, const int* level_w, int n_levels, int C, const float* wpt,
                                 const float* blob, const float* bias, int b, int V, int R, int P, int H, int W,
                                 float* e, float* qry, float* ug, float* logit, float* pt, float* pixel_val, int prec,
                                 void* stream) {
    CAR_REQUIRE(poses && rays && steps && gmaps && level_h && level_w && wpt && blob && bias, "car_fused_samples: null input");
    CAR_REQUIRE(e && qry && ug && logit && pt && pixel_val, "car_fused_samples: null output");
    CAR_REQUIRE(n_levels == 3 && C == kC && V == 2, "car_fused_samples: built for 3 pyramid levels, C = %d, V = 2 (got %d, %d, %d)", kC, n_levels, C, V);
    CAR_REQUIRE(b > 0 && R > 0 && P > 0 && H > 1 && W > 1, "car_fused_samples: bad sizes");
    FusedArgs a;
    a.poses = (const CarPose*)poses; a.rays = (const CarRay*)rays; a.steps = steps;
    for (int l = 0; l < 3; ++l) {
        a.gmap[l] = gmaps[l]; a.gh[l] = level_h[l]; a.gw[l] = level_w[l];
        CAR_REQUIRE(a.gmap[l] && a.gh[l] > 0 && a.gw[l] > 0 && (long)b * V * a.gh[l] * a.gw[l] < 2147483647L, "car_fused_samples: bad level %d", l);
    }
    a.wpt = wpt; a.blob = blob; a.bias = bias;
    a.b = b; a.V = V; a.R = R; a.P = P; a.H = H; a.W = W;
    a.S = (long)b * V * R * P;
    const char* xb = getenv("CAR_FUSED_XCD_BANDS");
    a.xcd_bands = xb ? atoi(xb) : 1;
    a.e = e; a.qry = qry; a.ug = ug; a.logit = logit; a.pt = pt; a.pixel_val = pixel_val;
    CAR_REQUIRE(prec == 0 || prec == 1, "car_fused_samples: prec must be 0 (fp32 MFMA) or 1 (split fp16 MFMA)");
    a.prec = prec;
    const char* abl_env = getenv("CAR_FUSED_ABLATE");
    const int abl = abl_env ? atoi(abl_env) : 0;
    const char* sch_env = getenv("CAR_FUSED_SCHED");
    const int sch = sch_env ? atoi(sch_env) : 1;
    void (*kern)(const FusedArgs) = (prec == 1 && abl == 0) ? fused_sample_kernel<0, 1, 1> : (prec == 1 && abl == 1) ? fused_sample_kernel<1, 1, 1>
                                    : (prec == 1 && abl == 2) ? fused_sample_kernel<2, 1, 1> : (prec == 1 && abl == 3) ? fused_sample_kernel<3, 1, 1>
                                    : (prec == 1 && abl == 4) ? fused_sample_kernel<4, 1, 1> : (abl == 0 && sch == 1) ? fused_sample_kernel<0, 1> : (abl == 0 && sch == 2) ? fused_sample_kernel<0, 2> : (abl == 0 && sch == 0) ? fused_sample_kernel<0, 0> : (abl == 0 && sch == 3) ? fused_sample_kernel<0, 3> : abl == 1 ? fused_sample_kernel<1> : abl == 2 ? fused_sample_kernel<2> : abl == 3 ? fused_sample_kernel<3>
                                    : abl == 4 ? fused_sample_kernel<4> : abl == 5 ? fused_sample_kernel<5> : abl == 6 ? fused_sample_kernel<6>
                                    : abl == 7 ? fused_sample_kernel<7> : fused_sample_kernel<0>;
    hipError_t e1 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e1 != hipSuccess) { car_set_error("car_fused_samples: cannot reserve %zu bytes of LDS: %s", kLdsBytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(car_div_up(a.S, 128)), dim3(256), kLdsBytes, (hipStream_t)stream, a);
    CAR_CHECK_LAUNCH("car_fused_samples");
    return CAR_OK;
}
