// car_fused.hip — the fused per-sample kernel (SURVEY.md §8a rows a6-a13 + the logits of a14; reference models.py:261-344,
// 487-532).  Per sample, without touching HBM in between: geometry (car_geom.h, fp64 Pluecker intersection) -> per 32-channel
// chunk a FOUR-tap gather of the per-texel projected pyramid: the first point-MLP layer is applied once per texel and ALL levels are
// summed once per stereo pair on the integer lattice their texel centres share (car_geom.h car_lattice_taps), so
// grid_sample over three levels is one bilinear lookup ->
// e_s = W2 relu(h_s) + b2 for both source views -> k1 = Wk1 [e_0 ; e_1] -> logit = <key, qry>/16 as a bilinear form of relu(k1) and
// relu(query_embed(g)) (one folded 128 x 128 layer, car_fused_layout.h; neither key nor qry is formed).  Every layer
// runs on the f16 matrix pipe as three v_mfma_f32_16x16x32_f16 products of fp16 hi/lo operand halves (car_fused_mma.h); a
// layer's accumulators are the next layer's B operands.
//
// Three waves per SIMD (<= 168 registers per wave), which needs
//   * the key layer's accumulators (k1, 32 registers) out of the e path: k1 = Wk1 [e_0 ; e_1] is computed after both
//     sources, chained from the accumulators for e_1; e_0 comes back from the output tensor it was just stored to (L2) by
//     LDS-DMA, straight into B-operand tiles in LDS;
//   * two tap batches (one row group each: 4 float4) in flight, each issued a whole chunk before it is blended;
//   * no read-ahead of the weight operands (the third wave hides the LDS latency instead).
// One workgroup = 12 waves = 192 samples = 24 consecutive rays x 8 consecutive steps (wave = 8 rays x 2 steps); the weight
// stream is shared by 192 samples.  The tap tables are stored compactly (byte offset of the north-west node, four weights).
// Every vector instruction in the chunk loop costs matrix-pipe time (a SIMD issues either), and every memory instruction ahead of a
// chunk barrier is waited for (in-order vmcnt): tap addresses need no decode, the first layer's scale is folded into its operands,
// e leaves as whole 128-byte lines (turned through the wave's LDS tile).  profiles/round3_fused_experiments.md has the measurements.
// The geometric query g (16 floats per sample) is written out for the second attention round (car_round2.hip recomputes the
// 16 -> 128 half of query_repeat_embed from it instead of reading a 128-wide row back).
#include "car_common.h"
#include "car_geom.h"
#include <type_traits>

namespace {

constexpr int kWaves = 12, kRows = 16, kGroup = kWaves * kRows;      // 192 samples per workgroup
// A workgroup's 192 samples = kTileRays consecutive rays x kTileSteps consecutive steps; a wave = 8 rays x 2 steps (rows 0-7: the
// first step, rows 8-15: the second), so the 8 rows of a tap instruction are one step of neighbouring rays.  24 x 8 touches 22 % fewer
// lattice nodes per workgroup than 48 x 4 (751 vs 964 over the four (view, source) maps of the bench frame; 12 x 16: 804, 96 x 2:
// 1374 — profiles/round3_fused_experiments.md section 15), i.e. more of a workgroup's tap lines come from its own L1.
#ifndef CAR_TILE_STEPS
#define CAR_TILE_STEPS 8
#endif
constexpr int kWaveRays = 8, kWaveSteps = kRows / kWaveRays;          // a wave's 16 rows
constexpr int kTileSteps = CAR_TILE_STEPS, kStepWaves = kTileSteps / kWaveSteps, kRayWaves = kWaves / kStepWaves, kTileRays = kRayWaves * kWaveRays;
static_assert(kStepWaves * kRayWaves == kWaves && kTileSteps % kWaveSteps == 0, "tile shape");
// row s of wave w: ray (w / kStepWaves) * 8 + (s & 7), step (w % kStepWaves) * 2 + (s >> 3) of the tile
__device__ __forceinline__ int tile_ray(int w, int s) { return (w / kStepWaves) * kWaveRays + (s & (kWaveRays - 1)); }
__device__ __forceinline__ int tile_step(int w, int s) { return (w % kStepWaves) * kWaveSteps + s / kWaveRays; }
constexpr int kThreads = 64 * kWaves;

constexpr int kPieces = 3;                         // LDS-DMA pieces per chunk: 12 waves x 1 KB each
constexpr unsigned kDeadTap = 0xc0000000u;         // tap-table entry of a sample that reads exact zeros: beyond any map, and no wrap with + row + node + column
constexpr long kMaxMapBytes = 0x80000000L;         // the lattice of one (view, padding mode) stays below it

#include "car_fused_mma.h"

constexpr int kLdsStage = kLdsW + 2 * kChunkTiles * kTile;      // [12][16][36]           h tiles, wave private     27 KB
constexpr int kLdsBias = kLdsStage + kGroup * kStageLd;         // [672]
constexpr int kLdsG = kLdsBias + kBiasFloats;                   // [192][16]              geometric query g per sample 12 KB
// the gather's tables: dead once the two source passes are over — the key layer's second e_0 buffer (kLdsE0) lies over them
constexpr int kLdsTapB = kLdsG + kGroup * 16;                   // [192][2] uint          byte offset of the nw node (or kDeadTap)   1.5 KB
constexpr int kLdsTapW = kLdsTapB + kGroup * 2;                 // [192][2][4]            tap weights (nw, ne, sw, se) x hp    6 KB
constexpr int kLdsPe = kLdsTapW + kGroup * 8;                   // [192][2][4]            tanh(pt_s/5) x hp            6 KB
constexpr int kLdsWpt = kLdsPe + kGroup * 8;                    // [144][4][4]            (W1[:,C:C+3], b1 hp) per channel quad: x | y | z | b   9 KB
constexpr int kLdsE0 = kLdsTapB;                                // [12][512]              e_0 rows of a K step, LDS-DMA target (the other buffer: the wave's h tile)
constexpr int kLdsFloats = (kLdsWpt + kC * 4 > kLdsE0 + kWaves * 512) ? kLdsWpt + kC * 4 : kLdsE0 + kWaves * 512;
constexpr size_t kLdsBytes = (size_t)kLdsFloats * sizeof(float);
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

struct FusedArgs {
    const CarPose* poses;
    const CarRay* rays;
    const float* steps;
    const float* lattice;      // [b*V][2 padding modes][lh][lw][kC]: every pyramid level, projected, summed on the common lattice (car_project_maps)
    int lh, lw, pad;
    float sx, sy;              // lattice coordinate u = (x + 1) * sx - 1: width / height of the finest level
    unsigned map_bytes;        // one (view, padding mode) lattice: the range of a source pass's buffer loads
    const float* gmeta;        // [1] max |lattice| (car_project_maps)
    const float* wpt;
    const float* blob;
    const float* bias;
    int b, V, R, P, H, W;
    int no_sample;             // samples at the depths `steps` on the query ray (models.py:221-222) instead of along the epipolar segment
    long S;
    int blk0;                  // first sample group of this launch (0 except in the development build's partial launches)
    float* e;
    float* g;
    float* logit;
    float* pt;
    float* pixel_val;
    float* part;               // [b*V][R][ceil(P / kTileSteps)][kC]: per (ray, step group) sum_j exp(logit_j - max_j logit) e_j, or NULL
    // ROWS instance (car_fused_rows, the three-view exchange): explicit rows instead of the kernel's own geometry — row = sample * ncomp + comp
    // gathers map (row_src & 0x3fffffff), padding mode (row_src >> 30) & 1 at row_grid [2] and adds the point term of row_pe [4]; a workgroup's 192
    // rows are 24 rays x 8 steps of ONE (sample set, component): its rows share their map and padding mode
    const int* row_src; const float* row_grid; const float* row_pe; int ncomp;
#ifdef CAR_BOUNDS
    long lattice_floats;       // extent of `lattice` (0: unknown to the entry), for the debug build's range checks
#endif
};

// chunk order:  W2 x18 (source 0) | W2 x18 (source 1) | K1 over e_1 x5 (2,2,2,2,1 K steps) | K1 over e_0 x5 | Q1 | M x2
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
    if (g == kG_K1a - 1 || g == kG_Q1 - 1 || g == kG_Q1) return kTD;                      // odd last K1 step, Q1
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
#ifdef CAR_BOUNDS
    n.lim = blob + (long)kBlobTiles * kTile;
#endif
    return n;
}

// ABL > 0: timing-only ablations (wrong results), instantiated only in the -DCAR_ABLATION development build (tools/):
// 1 no tap loads, 2 no gather work, 3 = 2 + no weight DMA / barriers, 5 no e-path MFMAs, weight DMA or barriers in the two source
// passes (the gather + the key / query layers); 11 the full kernel without the barrier of the weight stream (racy);
// 12 = 3 + no A-operand reads from LDS (matrix pipe + VALU only); 13 the full kernel without the A-operand reads;
// 4: the full kernel with shader-clock stamps at its phase boundaries (written over pixel_val);
// 20: the full kernel with shader-clock sums per piece of the chunk loop (where a wave waits inside a chunk)
// 21: the full kernel, writing each sample's north-west lattice node per source (-1: no fetch) over pixel_val (tap statistics)
// 22: every tap inside a 1 MB window of its lattice (L2 hits); 28: inside 32 nodes (L1 hits); 23: no weight DMA (barriers kept); 24 = 22 + 23;
// 30-33: the weight DMA with cache-policy bits nt / sc1 / sc0 sc1 / sc0; 40-43: waves leaving the barrier apart, taps spread over the slots;
// 50-53: the chunk loop without slot fences, vector / LDS instructions interleaved under the MFMAs by sched_group_barrier (results stay right)
// ROWS: one source pass over explicit rows, e_0 = W2 relu(h) + b2 written [row][kE], nothing else (the three-view exchange's two layers,
// models.py:345-475 through engine._encode_three_views); the chunk loop is the product kernel's own
template <int ABL, bool ROWS = false>
__global__ void __launch_bounds__(kThreads) fused_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: everything derived from it stays out of the vector ALU
    const int s = lane & 15, q4 = lane >> 4;
    const int nblk = gridDim.x;
    int blk = blockIdx.x;
    {   // workgroup b runs on XCD b % 8 (observed, speed only): give every XCD a contiguous band of sample groups so that the
        // lattice rows its workgroups share stay in one L2
        const int q8 = nblk / 8, r8 = nblk % 8, xcd = blk % 8, idx = blk / 8;
        blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx + a.blk0;
    }
    // the tile's sample of (wave, row lane & 15): tile_ray / tile_step above; the 8 rows a tap instruction gathers are one step of
    // neighbouring rays (shared lattice rows)
    const int pgs = (a.P + kTileSteps - 1) / kTileSteps, bundles = (a.R + kTileRays - 1) / kTileRays;
    // Order of the sample groups inside an XCD's band: STEP-MAJOR — (step group, sample set, ray tile), ray tiles fastest — so the ~32 workgroups
    // an XCD runs at a time are neighbouring ray tiles at ONE group of steps (a 65 536-ray frame at 64 steps: XCD k renders step group k of
    // every ray).  Round 6 measured three orders on the same box (profiles/round6_fused_closing.md): against the ray-major order of rounds 2-5
    // (a tile's 8 step groups on consecutive workgroups) this one fetches 4.5 % fewer bytes on the memory side and runs the launch 2.4 %
    // faster; putting a tile's 2 x 8 workgroups side by side fetches 5 % MORE.  A pure permutation of the work: results are unchanged.
    const int nsets = ROWS ? a.b * a.ncomp : a.b * a.V;
#if !defined(CAR_WG_ORDER) || CAR_WG_ORDER == 2
    const int pg = blk / (nsets * bundles), nset = (blk / bundles) % nsets, bun = blk % bundles;
#elif CAR_WG_ORDER == 0                                                // development build: rounds 2-5, (sample set, ray tile, step group)
    const int pg = blk % pgs, bun = (blk / pgs) % bundles, nset = blk / (pgs * bundles);
#else                                                                  // development build: (ray tile, sample set, step group)
    const int pg = blk % pgs, nset = (blk / pgs) % nsets, bun = blk / (pgs * nsets);
#endif
    const int nn = ROWS ? nset / a.ncomp : nset, comp = ROWS ? nset % a.ncomp : 0;
    const int ray_i = bun * kTileRays + tile_ray(wave, s), pp = pg * kTileSteps + tile_step(wave, s);
    const bool live = ray_i < a.R && pp < a.P;
    const long i = ((long)nn * a.R + (ray_i < a.R ? ray_i : a.R - 1)) * a.P + (pp < a.P ? pp : a.P - 1);

    // ABL 4 (development build): the full kernel, plus shader-clock stamps of wave 0 at the phase boundaries, written over pixel_val
    long long stamp[12];
#ifdef CAR_STAMP_ALL
    constexpr bool kStamp = true;
#else
    constexpr bool kStamp = (ABL == 4);
#endif
    auto mark = [&](int k) { if constexpr (kStamp) stamp[k] = (long long)__builtin_amdgcn_s_memtime(); };
    mark(0);
    // Scale of the first layer's output h (split-fp16 arithmetic, car_fused_mma.h): h is bounded by the largest lattice value plus the
    // point / bias term (the tap weights are non-negative and sum to at most one, |tanh| <= 1): one power of two hp per launch.
    // Everything that is added up into h — tap weights, point terms, the bias — is multiplied by hp ONCE, where it is made (exact:
    // a power of two), so the chunk loop never scales: h comes out of the gather as h * hp.
    float hp, hinv;
    pow2_scale(fmaxf(a.gmeta[0] + a.bias[kBiasScale + 5], 1e-30f), hp, hinv);
    // point-term table, per quad of channels: [x0..x3 | y0..y3 | z0..z3 | b0 hp..b3 hp] (operand pairs of the packed FMAs)
    for (int k = tid; k < kC; k += kThreads) {
        const float4 v = *reinterpret_cast<const float4*>(a.wpt + 4 * k);
        float* q = lds + kLdsWpt + 16 * (k >> 2) + (k & 3);
        q[0] = v.x; q[4] = v.y; q[8] = v.z; q[12] = v.w * hp;
    }
    for (int k = tid; k < kBiasFloats; k += kThreads) lds[kLdsBias + k] = a.bias[k];
    int g = 0;
    stream_issue_all<ABL>(a.blob, lds, 0, lane, wave);

    // ---- geometry: the 192 samples of the group are spread over the 192 lanes of waves 0-2 (one sample per lane, both source views),
    //      instead of every wave repeating its 16 samples in four lane groups: a third of the issue time on the critical path ----
    const int P = a.P, V = a.V;
    int rows_mode = 0, rows_map = 0;                                   // ROWS: the workgroup's (map, padding mode)
    if constexpr (ROWS) {
        if (wave < kGroup / 64) {
            const int sg = wave * 64 + lane, gwv = sg >> 4, gs = sg & 15;
            const int g_ray = bun * kTileRays + tile_ray(gwv, gs), g_pp = pg * kTileSteps + tile_step(gwv, gs);
            const long gi = ((long)nn * a.R + (g_ray < a.R ? g_ray : a.R - 1)) * a.P + (g_pp < a.P ? g_pp : a.P - 1);
            const long row = gi * a.ncomp + comp;
            const int mode = (a.row_src[row] >> 30) & 1;
            int node, flags;
            float w[4];
            car_lattice_taps(a.row_grid[2 * row], a.row_grid[2 * row + 1], a.lw, a.lh, a.pad, a.sx, a.sy, &node, &flags, w);
            const bool dead = mode == 1 && (flags & 4);
            reinterpret_cast<unsigned*>(lds + kLdsTapB)[sg * 2 + 0] = dead ? kDeadTap : (unsigned)node * (unsigned)(kC * 4);
            reinterpret_cast<unsigned*>(lds + kLdsTapB)[sg * 2 + 1] = kDeadTap;           // the loop's prefetch past the pass: no memory access
            *reinterpret_cast<float4*>(lds + kLdsTapW + (sg * 2 + 0) * 4) =
                dead ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(w[0] * hp, w[1] * hp, w[2] * hp, w[3] * hp);
            *reinterpret_cast<float4*>(lds + kLdsTapW + (sg * 2 + 1) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(lds + kLdsPe + (sg * 2 + 0) * 4) = make_float4(a.row_pe[4 * row] * hp, a.row_pe[4 * row + 1] * hp, a.row_pe[4 * row + 2] * hp, 0.0f);
            *reinterpret_cast<float4*>(lds + kLdsPe + (sg * 2 + 1) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const long gi0 = ((long)nn * a.R + (bun * kTileRays < a.R ? bun * kTileRays : a.R - 1)) * a.P + (pg * kTileSteps < a.P ? pg * kTileSteps : a.P - 1);
        const int src0 = a.row_src[gi0 * a.ncomp + comp];              // uniform over the workgroup by construction of the row list
        rows_map = __builtin_amdgcn_readfirstlane(src0 & 0x3fffffff);
        rows_mode = __builtin_amdgcn_readfirstlane((src0 >> 30) & 1);
    } else
    if (wave < kGroup / 64) {
        const int sg = wave * 64 + lane, gwv = sg >> 4, gs = sg & 15;     // sample sg belongs to row gs of matrix wave gwv
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
        } else {                                                       // geometry.get_epipolar_lines_volumetric: the ray's point at depth steps[p]
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
            // the four taps are nw + {0, 1 node} + {0, 1 row} (car_lattice_taps never returns a node of the last column / row): the
            // table holds the nw node's byte offset inside the lattice of this (source view, padding mode)
            int node, flags;
            float w[4];
            car_lattice_taps(gx, gy, a.lw, a.lh, a.pad, a.sx, a.sy, &node, &flags, w);
            // zeros padding, point on or beyond the outer ring: the four nodes are exactly zero.  Such a sample (its projection
            // misses the other view: a third of them on a wide-baseline pair) gets the out-of-range offset kDeadTap: the buffer
            // loads of its lanes return zeros without touching memory — an instruction whose lanes are all out of range costs the
            // texture path nothing (profiles/round3_fused_experiments.md) — and weight zero makes the contribution exactly +-0
            const bool dead = mode == 1 && (flags & 4);
            if constexpr (ABL == 21) { if (g_live) reinterpret_cast<int*>(a.pixel_val)[2 * gi + sv] = dead ? -1 : node; }      // tools/bench_fused.py 21: tap statistics
            unsigned tap_off = (unsigned)node * (unsigned)(kC * 4);
            if constexpr (ABL == 22 || ABL == 24) tap_off = (unsigned)(node % 448) * (unsigned)(kC * 4);       // timing probe: every tap inside a 1 MB window (L2 hits)
            if constexpr (ABL == 28) tap_off = (unsigned)(gs & 7) * (unsigned)(kC * 4);                        // timing probe: 32 nodes in all (L1 hits)
            reinterpret_cast<unsigned*>(lds + kLdsTapB)[sg * 2 + sv] = dead ? kDeadTap : tap_off;
            *reinterpret_cast<float4*>(lds + kLdsTapW + (sg * 2 + sv) * 4) =
                dead ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(w[0] * hp, w[1] * hp, w[2] * hp, w[3] * hp);
            const float px = sv == 0 ? smp.pt_in[0][0] : smp.pt_in[1][0], py = sv == 0 ? smp.pt_in[0][1] : smp.pt_in[1][1],
                        pz = sv == 0 ? smp.pt_in[0][2] : smp.pt_in[1][2];
            *reinterpret_cast<float4*>(lds + kLdsPe + (sg * 2 + sv) * 4) = make_float4(tanhf(px / 5.0f) * hp, tanhf(py / 5.0f) * hp, tanhf(pz / 5.0f) * hp, 0.0f);
        }
        if (g_live) {
            if constexpr (!kStamp && ABL != 20 && ABL != 21) { a.pixel_val[2 * gi] = smp.grid[0]; a.pixel_val[2 * gi + 1] = smp.grid[1]; }
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
    mark(1);

    // ---- gather machinery: lane owns rows rr = (lane>>3) + 8*it (it = 0, 1) and channel quad qd = lane & 7 of a chunk.
    //      A batch = the 4 tap loads of one row group `it`; two batches (bufA: it 0, bufB: it 1) are in flight, each issued right
    //      after the previous chunk's batch of the same row group has been blended: a whole chunk ahead of its use. ----
    const int qd = lane & 7, r0 = lane >> 3;
    float* stage = lds + kLdsStage + wave * kRows * kStageLd;
    float4 hacc[2];
    f32x4 bufA[4], bufB[4];
    const unsigned qd16 = 16u * qd;
    const unsigned row_step = (ABL == 22 || ABL == 24 || ABL == 28) ? 16u * (kC * 4) : (unsigned)a.lw * (kC * 4);

    // Buffer loads, range-checked against ONE (view, padding mode) lattice: the workgroup's samples all lie on the epipolar lines of
    // context view nn % V, so source view sv reads the border-padded lattice of that view when sv is the view itself and the
    // zero-padded lattice of view sv otherwise — one descriptor per source pass, 32-bit offsets inside it.
    const int v_own = nn % a.V, sc_own = nn / a.V;
    const long map_floats = (long)a.lh * a.lw * kC;
    const long lat0 = ROWS ? ((long)rows_map * 2 + rows_mode) * map_floats : ((long)(sc_own * a.V + 0) * 2 + (v_own == 0 ? 0 : 1)) * map_floats;
    const long lat1 = ROWS ? lat0 : ((long)(sc_own * a.V + 1) * 2 + (v_own == 1 ? 0 : 1)) * map_floats;
#ifdef CAR_BOUNDS
    CAR_BOUNDS_TRAP(a.lattice_floats == 0 || (lat0 >= 0 && lat0 + map_floats <= a.lattice_floats && lat1 >= 0 && lat1 + map_floats <= a.lattice_floats));
#endif
    const __amdgpu_buffer_rsrc_t rsrc[2] = {
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lattice + lat0), 0, (int)a.map_bytes, 0x00027000),
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lattice + lat1), 0, (int)a.map_bytes, 0x00027000)};
    auto issue_row = [&](f32x4 (&tap)[4], int sv, int c, int it) {
        if constexpr (ABL == 1 || ABL == 2 || ABL == 3 || ABL == 12) return;
        const int chunk_off = 128 * c;                                 // the chunk's 32 channels: scalar offset, not range-checked
        const unsigned tbv = reinterpret_cast<const unsigned*>(lds + kLdsTapB)[(wave * kRows + r0 + 8 * it) * 2 + sv];
        const unsigned o00 = tbv + qd16, o10 = o00 + row_step;          // east taps: the instruction's immediate offset (one node = 2304 B)
        // debug build: a live tap (offsets at or above 2^31 mark samples that read zeros: the hardware's range check serves them) must end
        // inside the (view, padding mode) lattice WITH the chunk's scalar offset, which the hardware does not check
        CAR_BOUNDS_TRAP(tbv >= 0x80000000u || (long)o10 + (kC * 4) + chunk_off + 16 <= (long)a.map_bytes);
        auto ld = [&](unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc[sv], (int)off, chunk_off, 0)); };
        tap[0] = ld(o00);
        tap[1] = ld(o00 + (unsigned)(kC * 4));
        tap[2] = ld(o10);
        tap[3] = ld(o10 + (unsigned)(kC * 4));
    };
    auto issue_tap = [&](f32x4 (&tap)[4], int sv, int c, int it, int t) {          // development build (ABL 43): one tap of a batch
        const int chunk_off = 128 * c;
        const unsigned tbv = reinterpret_cast<const unsigned*>(lds + kLdsTapB)[(wave * kRows + r0 + 8 * it) * 2 + sv];
        const unsigned o = tbv + qd16 + ((t & 2) ? row_step : 0u) + ((t & 1) ? (unsigned)(kC * 4) : 0u);
        tap[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc[sv], (int)o, chunk_off, 0));
    };
    auto blend_row = [&](const f32x4 (&tap)[4], int sv, int it) {
        if constexpr (ABL == 2 || ABL == 3 || ABL == 12) return;
        const float4 w = *reinterpret_cast<const float4*>(lds + kLdsTapW + ((wave * kRows + r0 + 8 * it) * 2 + sv) * 4);
        const float ww[4] = {w.x, w.y, w.z, w.w};
        f32x2 lo2 = {hacc[it].x, hacc[it].y}, hi2 = {hacc[it].z, hacc[it].w};          // v_pk_fma_f32: two FMAs per instruction
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 gq = tap[t];
            const f32x2 w2 = {ww[t], ww[t]};
            lo2 = __builtin_elementwise_fma(w2, f32x2{gq[0], gq[1]}, lo2);
            hi2 = __builtin_elementwise_fma(w2, f32x2{gq[2], gq[3]}, hi2);
        }
        hacc[it] = make_float4(lo2[0], lo2[1], hi2[0], hi2[1]);
    };
    auto affine_row = [&](int sv, int c, int it) {
        if constexpr (ABL == 2 || ABL == 3 || ABL == 12) return;
        const int rr = r0 + 8 * it;
        const float4 pe = *reinterpret_cast<const float4*>(lds + kLdsPe + ((wave * kRows + rr) * 2 + sv) * 4);
        const float4* wp = reinterpret_cast<const float4*>(lds + kLdsWpt + 16 * (8 * c + qd));
        const float4 wx = wp[0], wy = wp[1], wz = wp[2], wb = wp[3];
        // twelve FMAs: b + wz pz + wy py + wx px per channel (the bias rides in the innermost FMA: no separate multiply / add).
        // Deliberately scalar: the same sum written as six v_pk_fma_f32 with broadcast point terms produced e off by 1e-2 in this
        // kernel (cause not found — the instruction itself is fine with a destination that overlaps its broadcast source,
        // checked in isolation); this form differs from the round-2 arithmetic by rounding only (2.4e-7 on e).
        hacc[it] = make_float4(fmaf(wx.x, pe.x, fmaf(wy.x, pe.y, fmaf(wz.x, pe.z, wb.x))), fmaf(wx.y, pe.x, fmaf(wy.y, pe.y, fmaf(wz.y, pe.z, wb.y))),
                               fmaf(wx.z, pe.x, fmaf(wy.z, pe.y, fmaf(wz.z, pe.z, wb.z))), fmaf(wx.w, pe.x, fmaf(wy.w, pe.y, fmaf(wz.w, pe.z, wb.w))));
    };
    auto finish_row = [&](int it) {
        if constexpr (ABL == 2 || ABL == 3 || ABL == 12) return;
        const float4 o = hacc[it];
        *reinterpret_cast<float4*>(stage + (r0 + 8 * it) * kStageLd + 4 * qd) =
            make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    };
    // Scales of the split-fp16 arithmetic (car_fused_mma.h).  Packed weights carry 2^shift per layer (dW.. = 2^-shift, from the
    // bias table); hp: see the top of the kernel.
    const float* lsc = lds + kLdsBias + kBiasScale;
    auto uniform = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };   // keep it in an SGPR
    float e_up, e_down;
    {
        const float dW2 = lsc[kLayerW2];
        e_up = uniform(hp / dW2); e_down = uniform(dW2 * hinv);
    }
    auto read_b = [&](half8& bhi, half8& blo) {                        // this lane's 8 channels of the wave's h tile (already h * hp), split
        const float4 x0 = *reinterpret_cast<const float4*>(stage + s * kStageLd + 8 * q4);
        const float4 x1 = *reinterpret_cast<const float4*>(stage + s * kStageLd + 8 * q4 + 4);
        const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        split8_scaled(x, bhi, blo);
    };

    // first chunk of source 0: nothing to hide it under
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        affine_row(0, 0, it);
        issue_row(bufA, 0, 0, it);
        blend_row(bufA, 0, it);
        finish_row(it);
    }
    stream_sync();                                                     // weight chunk 0 landed
    mark(2);
    // development build, variants 63-65: ONE tap buffer — a row group's taps are issued four slots (not a whole chunk) before they are blended:
    // row group 0 in slot 0, blended in slot 4, where row group 1 is issued, blended in slot 8; nothing in flight over the barrier.  The 16
    // registers this frees hold the NEXT slot's A operands (64, 65): a true double buffer instead of 4 reads + wait in front of every 6 MFMAs
    constexpr bool kOneTapBuf = (ABL == 63 || ABL == 64 || ABL == 65);
    constexpr bool kTapsLive = (ABL == 0 || ABL >= 4) && !kOneTapBuf;
    if constexpr (!kOneTapBuf) {
        issue_row(bufA, 0, 1, 0);                                      // pipeline prologue: chunk (0, 1), both row groups
        issue_row(bufB, 0, 1, 1);
    }

    // ABL 20 (development build): shader-clock time the wave spends, per piece of the chunk loop, summed over the source passes and
    // written over pixel_val (tools/bench_fused.py 20); in every other variant tick() is 0 and all of this folds away
    long long t_blend = 0, t_dma = 0, t_bar = 0, t_chunks = 0, t_issue = 0, t_piece = 0, t_aff = 0, t_mfma = 0;
    auto tick = [&]() -> long long { if constexpr (ABL == 20) return (long long)__builtin_amdgcn_s_memtime(); else return 0; };
    f32x4 acc[kTE];
    float m0 = 0.0f;                                                   // largest |e_0| of this lane's sample
    half8 bhi, blo;
    read_b(bhi, blo);
#pragma unroll 1
    for (int sv = 0; sv < (ROWS ? 1 : 2); ++sv) {
        init_bias<kTE>(acc, lds + kLdsBias + kBiasE, q4, e_up);
#pragma unroll 1
        for (int c = 0; c < kKS; ++c) {
            // chunk being gathered: m+1 = (nsv, nc); chunk m+2 = (n2sv, n2c) is issued into each buffer as soon as it has been blended.
            // Branch-free on purpose: past the last chunk the gather harmlessly re-reads chunks of source 1.
            const int nsv = (c + 1 < kKS) ? sv : 1;
            const int nc = (c + 1 < kKS) ? c + 1 : 0;
            const int n2sv = (c + 2 < kKS) ? sv : 1;
            const int n2c = (c + 2 < kKS) ? c + 2 : c + 2 - kKS;
            const float* wl = lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane;
            const NextChunk nx = next_chunk_w2(a.blob, lds, g + 1);
            // 9 slots of (4 ds_read_b128 + 6 MFMAs of 16 cycles); between them one piece of the gather / DMA issue: slots 0-2 carry the
            // DMA pieces, slot 0 the affine start values, slots 3 and 6 one row group each — blend, store the h rows, re-issue.
            auto piece = [&](int qs) {
                if (qs < kPieces) { const long long t0 = tick(); stream_issue_piece<ABL>(nx, qs, lane, wave); t_piece += tick() - t0; }
                if constexpr (ABL == 43) {                             // the eight tap loads spread over slots 3-8, one or two per slot
                    if (qs == 0) { affine_row(nsv, nc, 0); affine_row(nsv, nc, 1); }
                    else if (qs == 3) { blend_row(bufA, nsv, 0); finish_row(0); issue_tap(bufA, n2sv, n2c, 0, 0); }
                    else if (qs == 4) issue_tap(bufA, n2sv, n2c, 0, 1);
                    else if (qs == 5) issue_tap(bufA, n2sv, n2c, 0, 2);
                    else if (qs == 6) { issue_tap(bufA, n2sv, n2c, 0, 3); blend_row(bufB, nsv, 1); finish_row(1); issue_tap(bufB, n2sv, n2c, 1, 0); }
                    else if (qs == 7) issue_tap(bufB, n2sv, n2c, 1, 1);
                    else if (qs == 8) { issue_tap(bufB, n2sv, n2c, 1, 2); issue_tap(bufB, n2sv, n2c, 1, 3); }
                    return;
                }
                if (qs == 0) { const long long t0 = tick(); affine_row(nsv, nc, 0); affine_row(nsv, nc, 1); t_aff += tick() - t0; }
                else if (qs == 3) { const long long t0 = tick(); blend_row(bufA, nsv, 0); const long long t1 = tick(); finish_row(0); issue_row(bufA, n2sv, n2c, 0); t_blend += t1 - t0; t_issue += tick() - t1; }
                else if (qs == 6) { const long long t0 = tick(); blend_row(bufB, nsv, 1); const long long t1 = tick(); finish_row(1); issue_row(bufB, n2sv, n2c, 1); t_blend += t1 - t0; t_issue += tick() - t1; }
            };
            if constexpr (kOneTapBuf) {
                auto load_a = [&](int qs, half8 (&a)[4]) {
                    const float* w0 = wl + (2 * qs * 2) * 256;
                    a[0] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0));
                    a[1] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 512));
                    a[2] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 256));
                    a[3] = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 768));
                };
                auto mma = [&](f32x4& c0, f32x4& c1, const half8 (&a)[4]) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bhi, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], bhi, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], blo, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], blo, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2], bhi, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[3], bhi, c1, 0, 0, 0);
                };
                auto piece1 = [&](int qs) {
                    if (qs < kPieces) stream_issue_piece<ABL>(nx, qs, lane, wave);
                    if (qs == 0) { affine_row(nsv, nc, 0); affine_row(nsv, nc, 1); issue_row(bufA, nsv, nc, 0); }
                    else if (qs == 4) { blend_row(bufA, nsv, 0); finish_row(0); issue_row(bufA, nsv, nc, 1); }
                    else if (qs == 8) { blend_row(bufA, nsv, 1); finish_row(1); }
                };
                half8 a0[4], a1[4];
                if constexpr (ABL != 63) load_a(0, a0);
#pragma unroll
                for (int qs = 0; qs < kTE / 2; ++qs) {
                    if constexpr (ABL == 63) { load_a(qs, a0); mma(acc[2 * qs], acc[2 * qs + 1], a0); piece1(qs); }
                    else {
                        if (qs + 1 < kTE / 2) { if (qs & 1) load_a(qs + 1, a0); else load_a(qs + 1, a1); }
                        if constexpr (ABL == 65) piece1(qs);               // 65: the piece under the LDS round trip, then the MFMAs
                        if (qs & 1) mma(acc[2 * qs], acc[2 * qs + 1], a1); else mma(acc[2 * qs], acc[2 * qs + 1], a0);
                        if constexpr (ABL == 64) piece1(qs);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (ABL == 60 || ABL == 61 || ABL == 62) {
                // development build: the A operands of slot qs + 1 are read right after slot qs's MFMAs have issued — into the registers those
                // MFMAs have just read — so that their LDS round trip runs under the slot's gather / DMA piece instead of in front of the next
                // MFMAs (no extra registers: the read-ahead of round 3 needed 16 and spilled).  61: the piece comes BEFORE the MFMAs instead
                // (reads issued, piece, MFMAs).  62: both (reads for the next slot after the MFMAs, and the piece first).
                auto load_a = [&](int qs, half8& ah0, half8& ah1, half8& al0, half8& al1) {
                    const float* w0 = wl + (2 * qs * 2) * 256;
                    ah0 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0));
                    ah1 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 512));
                    al0 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 256));
                    al1 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 768));
                };
                auto mma = [&](f32x4& c0, f32x4& c1, const half8& ah0, const half8& ah1, const half8& al0, const half8& al1) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bhi, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bhi, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, blo, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, blo, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bhi, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bhi, c1, 0, 0, 0);
                };
                half8 ah0, ah1, al0, al1;
                if constexpr (ABL != 61) load_a(0, ah0, ah1, al0, al1);
#pragma unroll
                for (int qs = 0; qs < kTE / 2; ++qs) {
                    if constexpr (ABL == 61) { load_a(qs, ah0, ah1, al0, al1); piece(qs); }
                    if constexpr (ABL == 62) piece(qs);
                    mma(acc[2 * qs], acc[2 * qs + 1], ah0, ah1, al0, al1);
                    if constexpr (ABL != 61) { if (qs + 1 < kTE / 2) load_a(qs + 1, ah0, ah1, al0, al1); }
                    if constexpr (ABL == 60) piece(qs);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
            for (int qs = 0; qs < kTE / 2; ++qs) {
                const float* w0 = wl + (2 * qs * 2) * 256;
                { const long long t0 = tick(); if constexpr (ABL != 5) mfma_pair<ABL>(acc[2 * qs], acc[2 * qs + 1], w0, w0 + 512, bhi, blo); t_mfma += tick() - t0; }
                piece(qs);
                if constexpr (ABL < 50 || ABL > 53) __builtin_amdgcn_sched_barrier(0);
            }
            }
            // development build, variants 50-53: no slot fences — the scheduler is asked to put the chunk's vector / LDS instructions UNDER
            // the MFMAs instead (two vector instructions per 16-clock MFMA are free: profiles/round4_fused_experiments.md section 5)
            if constexpr (ABL >= 50 && ABL <= 53) {
#pragma unroll
                for (int k = 0; k < 54; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              // one MFMA
                    if constexpr (ABL == 50 || ABL == 52) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
                    __builtin_amdgcn_sched_group_barrier(0x002, ABL == 51 || ABL == 53 ? 1 : 2, 0);  // one / two vector instructions
                    if constexpr (ABL == 51 || ABL == 53) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if constexpr (ABL >= 52) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // a vector-memory read where one is due
                }
            }
            read_b(bhi, blo);                                          // next chunk's B operand (own LDS tile, in-order LDS)
            // the 8 tap loads issued in slots 3 and 6 stay in flight over the barrier (they are younger than every DMA piece)
            if constexpr (ABL == 20) {
                const long long t0 = tick();
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                const long long t1 = tick();
                __syncthreads();
                t_dma += t1 - t0; t_bar += tick() - t1; t_chunks += 1;
            } else
            stream_sync<ABL, kTapsLive ? 8 : 0>();
            if constexpr (ABL == 40 || ABL == 41 || ABL == 42) {       // development build: the three waves of a SIMD (w, w + 4, w + 8) leave the barrier apart
                constexpr int kS = ABL == 40 ? 4 : ABL == 41 ? 8 : 2;
                if (wave >= 4) __builtin_amdgcn_s_sleep(kS);
                if (wave >= 8) __builtin_amdgcn_s_sleep(kS);
            }
            ++g;
        }
        scale_acc<kTE>(acc, e_down);
        mark(3 + sv);
        if (sv == 0) {
            m0 = sample_max<kTE, false>(acc);
            // e_0 out, as whole lines: the wave's h tile is free between the B-operand read that closed the pass and the next chunk's
            // first h rows, so each pair of tiles is turned through it (accumulator layout: four lanes x 16 bytes per row; row layout:
            // eight lanes per 128-byte line).  The 18 half-line stores this replaces stood in front of the next pass's first barrier.
#pragma unroll
            for (int m = 0; m < kTE / 2; ++m) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<float4*>(stage + s * kStageLd + 16 * j + 4 * q4) = make_float4(acc[2 * m + j][0], acc[2 * m + j][1], acc[2 * m + j][2], acc[2 * m + j][3]);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int rr = r0 + 8 * it;
                    const int ray_r = bun * kTileRays + tile_ray(wave, rr), pp_r = pg * kTileSteps + tile_step(wave, rr);
                    const long i_r = ((long)nn * a.R + (ray_r < a.R ? ray_r : a.R - 1)) * a.P + (pp_r < a.P ? pp_r : a.P - 1);
                    const float4 v = *reinterpret_cast<const float4*>(stage + rr * kStageLd + 4 * qd);
                    if constexpr (ROWS) *reinterpret_cast<float4*>(a.e + (i_r * a.ncomp + comp) * kE + 32 * m + 4 * qd) = v;
                    else
                    *reinterpret_cast<float4*>(a.e + i_r * (2 * kE) + 32 * m + 4 * qd) = v;      // rows past the end: duplicates, same values
                }
            }
        }
    }
    if constexpr (ROWS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the prefetched taps of a pass that does not exist
        return;
    }
    // ---- k1 = Wk1 [e_0 ; e_1] + bk1: first the e_1 half, chained from the accumulators (each K step's two tiles are stored as soon
    //      as they are consumed); then the e_0 half, whose B operands come back from the output tensor (written by this wave one source
    //      pass ago: L2) by LDS-DMA — whole 128-byte lines, no registers, two K steps ahead — instead of 16 half lines per load into
    //      the accumulator registers (the e traffic made this layer texture-path bound: profiles/round3_fused_experiments.md §16).
    //      Both halves accumulate into the same registers, so they share one per-sample power of two (from max |e_0|, |e_1|).
    float p, pinv;
    pow2_scale(fmaxf(fmaxf(m0, sample_max<kTE, false>(acc)), 1e-30f), p, pinv);
    f32x4 k1[kTD];
    init_bias<kTD>(k1, lds + kLdsBias + kBiasK1, q4, p / lsc[kLayerK1]);
    // row side of the wave's tile: lane (r0, qd) owns 16 bytes (channel quad qd of a 32-channel pair) of rows r0 and r0 + 8 — eight
    // lanes per 128-byte line of e.  Clamped like i above: a row past the end of the rays / steps is a duplicate of a live sample.
    long i_row[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int rr = r0 + 8 * it;
        const int ray_r = bun * kTileRays + tile_ray(wave, rr), pp_r = pg * kTileSteps + tile_step(wave, rr);
        i_row[it] = ((long)nn * a.R + (ray_r < a.R ? ray_r : a.R - 1)) * a.P + (pp_r < a.P ? pp_r : a.P - 1);
    }
    // A K step's two tiles of e_1 are stored as soon as they are consumed — turned through the wave's (idle) h tile, so that the two
    // store instructions write 8 whole lines each instead of 16 half lines (accumulator layout: four lanes x 16 bytes per row).
    auto store_tiles = [&](int m) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<float4*>(stage + s * kStageLd + 16 * j + 4 * q4) = make_float4(acc[2 * m + j][0], acc[2 * m + j][1], acc[2 * m + j][2], acc[2 * m + j][3]);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            // unconditional (duplicates write their sample's own values again) — the chunk barrier counts on exactly two memory
            // instructions per call
            const float4 v = *reinterpret_cast<const float4*>(stage + (r0 + 8 * it) * kStageLd + 4 * qd);
            *reinterpret_cast<float4*>(a.e + i_row[it] * (2 * kE) + kE + 32 * m + 4 * qd) = v;
        }
    };
    chained_layer<kTE, false, ABL, kG_K1b, 2>(k1, acc, p, a.blob, lds, lane, wave, store_tiles);
    mark(7);
    {
        constexpr bool kStream = !(ABL == 3 || ABL == 12 || ABL == 5);
        // lane (r0, qd) fetches 16 bytes of row rr = r0 + 8 it of the wave's tile; a row's eight 16-byte segments are stored rotated
        // by f(rr) = (rr >> 1) & 7 (the lane asks for segment qd ^ f(rr)), so that the B-operand reads below — 16 rows per pass,
        // 128 bytes apart — spread over the banks
        const float* esrc[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) esrc[it] = a.e + i_row[it] * (2 * kE) + 4 * (qd ^ (((r0 + 8 * it) >> 1) & 7));
        float* const ebuf[2] = {stage, lds + kLdsE0 + wave * 512};
        auto issue_e0 = [&](int m) {                                   // K step m (channels 32 m .. 32 m + 31) -> buffer m & 1
            if constexpr (!kStream) return;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(ebuf[m & 1] + it * 256));
                const float* gsrc = esrc[it] + 32 * m;
                CAR_BOUNDS_TRAP(gsrc >= a.e && gsrc + 4 <= a.e + a.S * kC);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
            }
        };
        auto wait_vm = [&](auto n) {                                   // at most n of this wave's vector memory operations outstanding
            if constexpr (!kStream) return;
            constexpr int N = decltype(n)::value;
            if constexpr (N >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        constexpr int kSteps = kTE / 2;                                // 9 K steps, two per weight chunk
        issue_e0(0);
        issue_e0(1);
#pragma unroll
        for (int c = 0; c < kChK1; ++c) {
            const int g = kG_K1a + c;
            const float* wl = lds + kLdsW + (g & 1) * kChunkTiles * kTile + 4 * lane;
            const NextChunk nx = next_chunk(a.blob, lds, g + 1);
#pragma unroll
            for (int kl = 0; kl < 2; ++kl) {
                const int m = 2 * c + kl;
                if (m < kSteps) {
                    // vector memory operations issued after e_0(m)'s two: e_0(m + 1)'s, and for the second K step of a chunk the
                    // chunk's three weight pieces in between
                    if (kl == 0) { if (m + 1 < kSteps) wait_vm(std::integral_constant<int, 2>()); else wait_vm(std::integral_constant<int, 0>()); }
                    else { if (m + 1 < kSteps) wait_vm(std::integral_constant<int, 5>()); else wait_vm(std::integral_constant<int, 3>()); }
                    const float* eb = ebuf[m & 1] + s * 32;
                    const int rot = (s >> 1) & 7;
                    const float4 x0 = *reinterpret_cast<const float4*>(eb + 4 * (q4 ^ rot));
                    const float4 x1 = *reinterpret_cast<const float4*>(eb + 4 * ((4 + q4) ^ rot));
                    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    half8 bhi, blo;
                    split8(x, p, bhi, blo);
#pragma unroll
                    for (int q = 0; q < kTD / 2; ++q) {
                        const float* w0 = wl + ((kl * kTD + 2 * q) * 2) * 256;
                        mfma_pair<ABL>(k1[2 * q], k1[2 * q + 1], w0, w0 + 512, bhi, blo);
                        if (kl == 0 && q < kPieces) stream_issue_piece<ABL>(nx, q, lane, wave);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (m + 2 < kSteps) issue_e0(m + 2);              // into the buffer just read
                }
            }
            // the weight pieces of the next chunk have landed: behind them only the e_0 rows issued after the K steps of this chunk
            if (2 * c + 2 < kSteps && 2 * c + 3 < kSteps) wait_vm(std::integral_constant<int, 4>());
            else if (2 * c + 2 < kSteps) wait_vm(std::integral_constant<int, 2>());
            else wait_vm(std::integral_constant<int, 0>());
            if constexpr (kStream && ABL != 11) __syncthreads();
        }
    }
    mark(8);
    scale_acc<kTD>(k1, lsc[kLayerK1] * pinv);                          // k1 = key_map([e_0 ; e_1]); r = relu(k1) is used straight from these registers
    mark(5);
    // ---- logit = <key, qry> / 16 as the bilinear form r^T (M x + v) + u^T x + c of r = relu(k1) and x = relu(Wq1 g + bq1)
    //      (car_fused_layout.h): one 128 x 128 layer instead of key_map_2 and query_embed_2 ------------------------------------------
    half8 ghi, glo;                                                    // B operand of the layer fed by g (k = 16: folded bias)
    {
        const float* gl = lds + kLdsG + (wave * kRows + s) * 16 + 8 * (q4 & 1);
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
    f32x4 t1[kTD], mt[kTD];
#pragma unroll
    for (int t = 0; t < kTD; ++t) t1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    stream_issue_all<ABL>(a.blob, lds, kG_Q1 + 1, lane, wave);
    small_layer(t1, ghi, glo, lds + kLdsW + (kG_Q1 & 1) * kChunkTiles * kTile + 4 * lane);          // q1
    stream_sync<ABL>();
    scale_acc<kTD>(t1, lsc[kLayerQ1] * pinv);
    pow2_scale(fmaxf(sample_max<kTD, true>(t1), 1e-30f), p, pinv);
    init_bias<kTD>(mt, lds + kLdsBias + kBiasV, q4, p / lsc[kLayerM]);
    chained_layer<kTD, true, ABL, kG_M>(mt, t1, p, a.blob, lds, lane, wave);
    scale_acc<kTD>(mt, lsc[kLayerM] * pinv);                           // M x + v
    float dot = 0.0f;
#pragma unroll
    for (int t = 0; t < kTD; ++t) {
        const float4 u4 = *reinterpret_cast<const float4*>(lds + kLdsBias + kBiasU + 16 * t + 4 * q4);
        const float uu[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dot = fmaf(fmaxf(k1[t][r], 0.0f), mt[t][r], dot);
            dot = fmaf(uu[r], fmaxf(t1[t][r], 0.0f), dot);
        }
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    dot += lds[kLdsBias + kBiasConst];
    // ---- first attention round, this workgroup's share (models.py:533-541): for each of its rays the step group's
    //      sum_j exp(logit_j - m) e_j with m = max_j logit_j over the group's kTileSteps samples, read back from the rows of e this
    //      workgroup has written (L2) — 1/kTileSteps of the bytes the attention launch would otherwise stream from HBM; that launch
    //      (car_attend_parts) folds the groups of a ray together with exp(m - M) / L.  A wave owns kTileRays / kWaves rays; a 16-lane
    //      group reads a whole 2304-byte row as nine float4 per lane, the wave's four groups take four rows at a time.
    //      Every wave's stores of e were drained (vmcnt 0) in front of a chunk barrier layers ago, so the logits' hand-over barrier is
    //      all the ordering the read-back needs (no wait for the logit stores issued just before it); the row loads of ALL the
    //      wave's rays are in flight together.
    constexpr int kRaysPerWave = kTileRays / kWaves, kRowIts = kTileSteps / 4;
    static_assert(kTileRays % kWaves == 0 && kTileSteps % 4 == 0, "partial sums: whole rays per wave, four rows at a time");
    f32x4 xr[kRaysPerWave][kRowIts][kC / 64];
    float pw[kRaysPerWave][kRowIts];
    const int sub = lane & 15, grp = lane >> 4;
    if (live) {
        if (q4 == 0) a.logit[i] = dot / 16.0f;
    }
    if (a.part) {
        float* lgt = lds + kLdsG;                                      // the wave's g rows are dead: row (wave, s) keeps its logit in float 0
        if (q4 == 0) lgt[(wave * kRows + s) * 16] = live ? dot / 16.0f : -INFINITY;
        __syncthreads();                                               // the logit stores above stay in flight across it
        auto lgt_of = [&](int rr, int k) -> float {                    // logit of the tile's (ray rr, step k): inverse of tile_ray / tile_step
            const int w_ = (rr / kWaveRays) * kStepWaves + k / kWaveSteps, s_ = (rr % kWaveRays) + kWaveRays * (k % kWaveSteps);
            return lgt[(w_ * kRows + s_) * 16];
        };
#pragma unroll
        for (int rw = 0; rw < kRaysPerWave; ++rw) {
            const int rr = wave * kRaysPerWave + rw, ray_g = bun * kTileRays + rr;
            const int ray_c = ray_g < a.R ? ray_g : a.R - 1;           // a ray past the end: loads a live ray's rows, stores nothing
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < kTileSteps; ++k) mx = fmaxf(mx, lgt_of(rr, k));
#pragma unroll
            for (int it = 0; it < kRowIts; ++it) {
                const int k = 4 * it + grp, pp_k = pg * kTileSteps + k;
                pw[rw][it] = expf(lgt_of(rr, k) - mx);                 // a step past P carries -inf: weight 0 (its row is a duplicate of step P - 1)
                const f32x4* rowp = reinterpret_cast<const f32x4*>(a.e + (((long)nn * a.R + ray_c) * a.P + (pp_k < a.P ? pp_k : a.P - 1)) * kC + 4 * sub);
                CAR_BOUNDS_TRAP(reinterpret_cast<const float*>(rowp) >= a.e && reinterpret_cast<const float*>(rowp + 16 * (kC / 64 - 1) + 1) <= a.e + a.S * kC);
#pragma unroll
                for (int j = 0; j < kC / 64; ++j) xr[rw][it][j] = __builtin_nontemporal_load(rowp + 16 * j);   // read once, written by other waves: past L1
            }
        }
    }
    mark(6);
    if (a.part) {
#pragma unroll
        for (int rw = 0; rw < kRaysPerWave; ++rw) {
            const int ray_g = bun * kTileRays + wave * kRaysPerWave + rw;
            float* out = a.part + (((long)nn * a.R + ray_g) * pgs + pg) * kC + 4 * sub;
#pragma unroll
            for (int j = 0; j < kC / 64; ++j) {
                f32x4 acc = xr[rw][0][j] * pw[rw][0];
#pragma unroll
                for (int it = 1; it < kRowIts; ++it) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaf(pw[rw][it], xr[rw][it][j][c], acc[c]);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] += __shfl_xor(acc[c], 16, 64);
                    acc[c] += __shfl_xor(acc[c], 32, 64);
                }
                if (grp == 0 && ray_g < a.R) *reinterpret_cast<f32x4*>(out + 64 * j) = acc;
            }
        }
    }
    if constexpr (ABL == 20) {
        if (lane == 0) {
            long long* out = reinterpret_cast<long long*>(a.pixel_val) + ((long)blk * kWaves + wave) * 8;
            out[0] = t_blend; out[1] = t_dma; out[2] = t_bar; out[3] = t_chunks; out[4] = t_issue; out[5] = t_piece; out[6] = t_aff; out[7] = t_mfma;
        }
    }
    if constexpr (kStamp) {
        if (tid == 0) {
            long long* out = reinterpret_cast<long long*>(a.pixel_val) + (long)blk * 16;
            for (int k = 0; k < 10; ++k) out[k] = stamp[k];
        }
    }
}


int launch_fused(int abl, int blk0, int nblk, const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h, int lat_w, int lat_pad,
                 const float* gmeta, const float* wpt, const float* blob, const float* bias, int b, int V, int R, int P, int H, int W, int no_sample, float* e,
                 float* g, float* logit, float* pt, float* pixel_val, float* part, void* stream) {
    CAR_REQUIRE(poses && rays && steps && lattice && gmeta && wpt && blob && bias, "car_fused_samples: null input");
    CAR_REQUIRE(e && g && logit && pt && pixel_val, "car_fused_samples: null output");
    CAR_REQUIRE(V == 2, "car_fused_samples: built for V = 2 (got %d)", V);
    CAR_REQUIRE(b > 0 && R > 0 && P > 0 && H > 1 && W > 1, "car_fused_samples: bad sizes");
    CAR_REQUIRE(lat_pad >= 2 && lat_h > 2 * lat_pad + 1 && lat_w > 2 * lat_pad + 1 && ((lat_h - 2 * lat_pad) & 1) && ((lat_w - 2 * lat_pad) & 1),
                "car_fused_samples: bad lattice %d x %d, pad %d (car_lattice_shape)", lat_h, lat_w, lat_pad);
    // nodes are addressed by 32-bit byte offsets inside the lattice of one (view, padding mode)
    CAR_REQUIRE((long)lat_h * lat_w * (kC * 4) < kMaxMapBytes, "car_fused_samples: a lattice of %d x %d nodes exceeds 2 GiB per view and padding mode (finest level wider than ~470 pixels): use the stage entries", lat_h, lat_w);
    FusedArgs a;
    a.poses = (const CarPose*)poses; a.rays = (const CarRay*)rays; a.steps = steps;
    a.lattice = lattice; a.lh = lat_h; a.lw = lat_w; a.pad = lat_pad;
    a.map_bytes = (unsigned)((long)lat_h * lat_w * (kC * 4));
    a.sx = (float)((lat_w - 2 * lat_pad + 1) / 2); a.sy = (float)((lat_h - 2 * lat_pad + 1) / 2);
    a.gmeta = gmeta; a.wpt = wpt; a.blob = blob; a.bias = bias;
    a.b = b; a.V = V; a.R = R; a.P = P; a.H = H; a.W = W;
    a.no_sample = no_sample != 0;
    a.S = (long)b * V * R * P;
    a.blk0 = blk0;
    a.e = e; a.g = g; a.logit = logit; a.pt = pt; a.pixel_val = pixel_val; a.part = part;
#ifdef CAR_BOUNDS
    a.lattice_floats = (long)b * V * 2 * lat_h * lat_w * kC;
#endif
    long groups = (long)b * V * car_div_up(R, kTileRays) * car_div_up(P, kTileSteps);
    if (nblk > 0) groups = (groups - blk0 < nblk) ? groups - blk0 : nblk;     // development build: a slice of the sample groups
    void (*kern)(const FusedArgs) = fused_kernel<0>;
#if defined(CAR_ABLATION) && !defined(CAR_ABLATION_NONE)       // CAR_ABLATION_NONE: the development build's entries without the timing variants
    switch (abl) {
        case 1: kern = fused_kernel<1>; break;   case 2: kern = fused_kernel<2>; break;   case 3: kern = fused_kernel<3>; break;
        case 4: kern = fused_kernel<4>; break;   case 5: kern = fused_kernel<5>; break;   case 11: kern = fused_kernel<11>; break;
        case 12: kern = fused_kernel<12>; break;   case 13: kern = fused_kernel<13>; break;   case 20: kern = fused_kernel<20>; break;
        case 21: kern = fused_kernel<21>; break;   case 22: kern = fused_kernel<22>; break;   case 23: kern = fused_kernel<23>; break;
        case 24: kern = fused_kernel<24>; break;   case 28: kern = fused_kernel<28>; break;   case 30: kern = fused_kernel<30>; break;
        case 31: kern = fused_kernel<31>; break;   case 32: kern = fused_kernel<32>; break;   case 33: kern = fused_kernel<33>; break;
        case 40: kern = fused_kernel<40>; break;   case 41: kern = fused_kernel<41>; break;   case 42: kern = fused_kernel<42>; break;
        case 43: kern = fused_kernel<43>; break;   case 50: kern = fused_kernel<50>; break;   case 51: kern = fused_kernel<51>; break;
        case 52: kern = fused_kernel<52>; break;   case 53: kern = fused_kernel<53>; break;   case 60: kern = fused_kernel<60>; break;
        case 61: kern = fused_kernel<61>; break;   case 62: kern = fused_kernel<62>; break;   case 63: kern = fused_kernel<63>; break;
        case 64: kern = fused_kernel<64>; break;   case 65: kern = fused_kernel<65>; break;
        default: break;
    }
#else
    (void)abl;
#endif
    hipError_t e1 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e1 != hipSuccess) { car_set_error("car_fused_samples: cannot reserve %zu bytes of LDS: %s", kLdsBytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(kThreads), kLdsBytes, (hipStream_t)stream, a);
    CAR_CHECK_LAUNCH("car_fused_samples");
    return CAR_OK;
}

}  // namespace

extern "C" size_t car_fused_blob_floats(void) { return (size_t)kBlobTiles * kTile; }
extern "C" size_t car_fused_bias_floats(void) { return (size_t)(kBiasFloats + kBiasScratch); }      // the table + car_fused_pack's scratch

extern "C" int car_fused_samples(const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h, int lat_w,
                                 int lat_pad, const float* gmeta, const float* wpt, const float* blob, const float* bias, int b, int V, int R, int P,
                                 int H, int W, int no_sample, float* e, float* g, float* logit, float* pt, float* pixel_val, void* stream) {
    return launch_fused(0, 0, 0, poses, rays, steps, lattice, lat_h, lat_w, lat_pad, gmeta, wpt, blob, bias, b, V, R, P, H, W, no_sample, e, g, logit, pt,
                        pixel_val, nullptr, stream);
}

// the same launch, which also leaves the first attention round's per-step-group partial sums in `part`
// [b*V][R][ceil(P / car_fused_tile_steps())][576] for car_attend_parts
extern "C" int car_fused_tile_steps(void) { return kTileSteps; }
extern "C" int car_fused_samples_parts(const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h, int lat_w,
                                       int lat_pad, const float* gmeta, const float* wpt, const float* blob, const float* bias, int b, int V, int R,
                                       int P, int H, int W, int no_sample, float* e, float* g, float* logit, float* pt, float* pixel_val,
                                       float* part, void* stream) {
    CAR_REQUIRE(part, "car_fused_samples_parts: null output");
    return launch_fused(0, 0, 0, poses, rays, steps, lattice, lat_h, lat_w, lat_pad, gmeta, wpt, blob, bias, b, V, R, P, H, W, no_sample, e, g, logit, pt,
                        pixel_val, part, stream);
}

#ifdef CAR_ABLATION
// development build only (tools/build_dev.py): timing-only variants of the kernel, results are wrong by construction
extern "C" int car_fused_samples_ablate(int abl, const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h,
                                        int lat_w, int lat_pad, const float* gmeta, const float* wpt, const float* blob, const float* bias, int b,
                                        int V, int R, int P, int H, int W, int no_sample, float* e, float* g, float* logit, float* pt,
                                        float* pixel_val, void* stream) {
    return launch_fused(abl, 0, 0, poses, rays, steps, lattice, lat_h, lat_w, lat_pad, gmeta, wpt, blob, bias, b, V, R, P, H, W, no_sample, e, g, logit, pt,
                        pixel_val, nullptr, stream);
}
// the same launch cut into slices of `nblk` sample groups (one kernel launch each): every slice starts its workgroups in phase
extern "C" int car_fused_samples_sliced(int nblk, const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h,
                                        int lat_w, int lat_pad, const float* gmeta, const float* wpt, const float* blob, const float* bias, int b,
                                        int V, int R, int P, int H, int W, int no_sample, float* e, float* g, float* logit, float* pt,
                                        float* pixel_val, void* stream) {
    const long groups = (long)b * V * car_div_up(R, kTileRays) * car_div_up(P, kTileSteps);
    for (long b0 = 0; b0 < groups; b0 += nblk) {
        const int rc = launch_fused(0, (int)b0, nblk, poses, rays, steps, lattice, lat_h, lat_w, lat_pad, gmeta, wpt, blob, bias, b, V, R, P, H, W, no_sample, e, g,
                                    logit, pt, pixel_val, nullptr, stream);
        if (rc != CAR_OK) return rc;
    }
    return CAR_OK;
}
#endif

// The three-view exchange's two layers for explicit rows (models.py:345-475 through engine._encode_three_views): the ROWS instance of the fused
// per-sample kernel — its source pass, one per (sample set, component), nothing behind it.  rows = n_sets * R * P * ncomp, row = sample * ncomp +
// comp; e [rows][288].  lattice [n_maps][2][lat_h][lat_w][576] (car_merge_lattice), gmeta [1] = its largest magnitude; blob / bias / wpt:
// car_fused_pack_rows.  Every 24-ray x 8-step tile of a (set, component) must share its (map, padding mode) — true for the exchange's row list.
extern "C" int car_fused_rows(const float* lattice, int lat_h, int lat_w, int lat_pad, const float* gmeta, const float* wpt, const float* blob,
                              const float* bias, const int* row_src, const float* row_grid, const float* row_pe, int n_sets, int R, int P, int ncomp,
                              float* e, void* stream) {
    CAR_REQUIRE(lattice && gmeta && wpt && blob && bias && row_src && row_grid && row_pe && e, "car_fused_rows: null pointer");
    CAR_REQUIRE(n_sets > 0 && R > 0 && P > 0 && ncomp > 0, "car_fused_rows: bad sizes");
    CAR_REQUIRE(lat_pad >= 2 && lat_h > 2 * lat_pad + 1 && lat_w > 2 * lat_pad + 1 && ((lat_h - 2 * lat_pad) & 1) && ((lat_w - 2 * lat_pad) & 1),
                "car_fused_rows: bad lattice %d x %d, pad %d (car_merge_lattice)", lat_h, lat_w, lat_pad);
    CAR_REQUIRE((long)lat_h * lat_w * (kC * 4) < kMaxMapBytes, "car_fused_rows: a lattice of %d x %d nodes exceeds 2 GiB per map and padding mode", lat_h, lat_w);
    FusedArgs a{};
    a.lattice = lattice; a.lh = lat_h; a.lw = lat_w; a.pad = lat_pad;
    a.map_bytes = (unsigned)((long)lat_h * lat_w * (kC * 4));
    a.sx = (float)((lat_w - 2 * lat_pad + 1) / 2); a.sy = (float)((lat_h - 2 * lat_pad + 1) / 2);
    a.gmeta = gmeta; a.wpt = wpt; a.blob = blob; a.bias = bias;
    a.b = n_sets; a.V = 1; a.R = R; a.P = P;
    a.S = (long)n_sets * R * P;
    a.e = e;
    a.row_src = row_src; a.row_grid = row_grid; a.row_pe = row_pe; a.ncomp = ncomp;
    const long groups = (long)n_sets * ncomp * car_div_up(R, kTileRays) * car_div_up(P, kTileSteps);
    void (*kern)(const FusedArgs) = fused_kernel<0, true>;
    hipError_t e1 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e1 != hipSuccess) { car_set_error("car_fused_rows: cannot reserve %zu bytes of LDS: %s", kLdsBytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(kThreads), kLdsBytes, (hipStream_t)stream, a);
    CAR_CHECK_LAUNCH("car_fused_rows");
    return CAR_OK;
}
