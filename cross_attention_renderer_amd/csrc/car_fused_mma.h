// car_fused_mma.h — matrix-pipe pieces of the fused per-sample kernel (car_fused.hip), built on v_mfma_f32_16x16x32_f16 tiles:
// layer / blob constants, the LDS-DMA weight stream, the split-fp16 helpers and the chained layers.
// Included INSIDE the kernel file's anonymous namespace, after it has defined kWaves (waves per workgroup = DMA participants)
// and kPieces (LDS-DMA pieces per chunk = ceil(36 / kWaves)); the kernel file then defines chunk_tile_offset / chunk_tiles
// (its order of the weight chunks) and carves its own LDS beyond the two weight buffers.
//
// Split-fp16 arithmetic: x = hi + lo with fp16 halves, three products per term (hi*hi + hi*lo + lo*hi, fp32 accumulate; the
// dropped lo*lo is 2^-22 relative).  fp16 keeps 11 bits only between 2^-14 and 65504, so both operands are moved into that
// window by powers of two, undone exactly on the fp32 side: a layer's weights carry 2^shift chosen at pack time from the layer's
// largest weight (car_plan_build), the activations a power of two chosen from their largest magnitude — per sample for the
// chained layers (the whole input vector sits in registers), per launch for the first layer (bounded by the maxima of the
// projected maps) — so that the largest value lands in [2^13, 2^14): no overflow, and everything within 2^-17 of the largest
// value keeps its full 22 bits.
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

#include "car_fused_layout.h"

constexpr int kStageLd = 36;       // row stride of the wave-private h tile (floats)

// ---- dynamic LDS carve-up (floats) --------------------------------------------------------------------------------
constexpr int kLdsW = 0;                                        // [2][18][512]           weight chunks          72 KB

__device__ __forceinline__ constexpr int chunk_tile_offset(int g);      // defined by the including kernel file
__device__ __forceinline__ constexpr int chunk_tiles(int g);

// Descriptor of the chunk to prefetch, resolved once per chunk with scalar branches so the per-piece issue is straight-line
struct NextChunk {
    const float* src; float* dst; int nkb;
#ifdef CAR_BOUNDS
    const float* lim = nullptr;        // one past the packed array the chunk lies in (nullptr: unknown to this kernel)
#endif
};
__device__ __forceinline__ NextChunk next_chunk(const float* __restrict__ blob, float* lds, int g) {
    const int ge = g < kNumChunks ? g : kNumChunks - 1;            // past the end: re-copy the last chunk onto itself
    NextChunk n;
    n.src = blob + (long)chunk_tile_offset(ge) * kTile;
    n.dst = lds + kLdsW + (ge & 1) * kChunkTiles * kTile;
    n.nkb = 2 * chunk_tiles(ge);
#ifdef CAR_BOUNDS
    n.lim = blob + (long)kBlobTiles * kTile;
#endif
    return n;
}
// piece p of the next chunk: wave w copies KB number kWaves p + w (wrapped into the chunk: re-copying identical bytes is harmless).
// LDS-DMA in inline asm, see car_linear.hip.  Must only run after the barrier that retired the buffer's previous chunk.
template <int ABL = 0>
__device__ __forceinline__ void stream_issue_piece(const NextChunk& n, int p, int lane, int wave) {
    if constexpr (ABL == 3 || ABL == 12 || ABL == 5 || ABL == 23 || ABL == 24) return;      // 23 / 24: no weight DMA, barriers kept
    int kb = kWaves * p + wave;
    kb = kb < n.nkb ? kb : kb - n.nkb;
    kb = kb < n.nkb ? kb : kb - n.nkb;
    // a piece index that two subtractions do not bring back into the chunk would copy 1 KB from BEHIND it (car_linear16.hip before 0d74f26:
    // three pieces issued for a four-KB chunk); and no piece may leave the packed array
    CAR_BOUNDS_TRAP(kb >= 0 && kb < n.nkb);
#ifdef CAR_BOUNDS
    {   // 32-bit arithmetic on the distance (a 64-bit pointer compare would move the address computation below into vector registers)
        const int room = n.lim ? (int)(n.lim - n.src) : 0x7fffffff;
        CAR_BOUNDS_TRAP((kb + 1) * 256 <= room);
    }
#endif
    // `wave` is scalar, so both addresses are: scalar base + the lane's 16 bytes (no 64-bit vector address arithmetic per piece)
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(n.dst + kb * 256));
    const float* gsrc = n.src + kb * 256;
#ifdef CAR_BOUNDS
    {   // the checks' branches make the compiler lose sight of the address being wave-uniform: say so again for the "s" operand below
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)gsrc), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)gsrc >> 32));
        gsrc = reinterpret_cast<const float*>(((uintptr_t)hi << 32) | lo);
    }
#endif
    const unsigned voff = 16u * (unsigned)lane;
    unsigned keep;
#define CAR_DMA_PIECE(POLICY) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3" POLICY "\n\ts_mov_b32 m0, %0" \
                                           : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gsrc) : "memory")
    if constexpr (ABL == 30) CAR_DMA_PIECE(" nt");                    // development build: cache-policy probes of the weight stream
    else if constexpr (ABL == 31) CAR_DMA_PIECE(" sc1");
    else if constexpr (ABL == 32) CAR_DMA_PIECE(" sc0 sc1");
    else if constexpr (ABL == 33) CAR_DMA_PIECE(" sc0");
    else CAR_DMA_PIECE("");
#undef CAR_DMA_PIECE
}
template <int ABL = 0>
__device__ __forceinline__ void stream_issue_all(const float* __restrict__ blob, float* lds, int g, int lane, int wave) {
    if (g >= kNumChunks) return;
    const NextChunk n = next_chunk(blob, lds, g);
#pragma unroll
    for (int p = 0; p < kPieces; ++p) stream_issue_piece<ABL>(n, p, lane, wave);
}
// end of a chunk: the DMA of the next chunk has landed and every wave is done reading the current one.  KEEP = number of
// vector loads this wave issued AFTER its last DMA piece and wants to leave in flight across the barrier (loads return in
// order, so "at most KEEP outstanding" still means every DMA piece has landed).
template <int ABL = 0, int KEEP = 0>
__device__ __forceinline__ void stream_sync() {
    if constexpr (ABL == 3 || ABL == 12 || ABL == 5) return;
    if constexpr (KEEP == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (KEEP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (KEEP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (ABL != 11) __syncthreads();                          // 11 (development build): the weight stream without its barrier
}

#include "car_split.h"

// largest magnitude of a sample's values: the sample's row is spread over NT x 4 registers of the four lanes (s, q = 0..3)
template <int NT, bool RELU>
__device__ __forceinline__ float sample_max(const f32x4 (&v)[NT]) {
    float m = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, RELU ? v[t][r] : fabsf(v[t][r]));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    return fmaxf(m, __shfl_xor(m, 32, 64));
}

// two output tiles x three split products, interleaved so consecutive MFMAs never share an accumulator
template <int ABL = 0>
__device__ __forceinline__ void mfma_pair(f32x4& c0, f32x4& c1, const float* w0, const float* w1, const half8& bhi, const half8& blo) {
    half8 ah0, ah1, al0, al1;
    if constexpr (ABL == 12 || ABL == 13) { ah0 = bhi; ah1 = blo; al0 = blo; al1 = bhi; }        // development build: no A-operand reads from LDS
    else {
        ah0 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0));
        ah1 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w1));
        al0 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 256));
        al1 = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w1 + 256));
    }
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bhi, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bhi, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, blo, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, blo, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bhi, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bhi, c1, 0, 0, 0);
}

// accumulators start at bias * scale: lane (s, q) register r of tile t holds channel 16 t + 4 q + r
template <int NT>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[NT], const float* lbias, int q, float scale) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 b4 = *reinterpret_cast<const float4*>(lbias + 16 * t + 4 * q);
        acc[t][0] = b4.x * scale; acc[t][1] = b4.y * scale; acc[t][2] = b4.z * scale; acc[t][3] = b4.w * scale;
    }
}
template <int NT>
__device__ __forceinline__ void scale_acc(f32x4 (&acc)[NT], float f) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] *= f;
}
template <int NT>
__device__ __forceinline__ void store_rows(const f32x4 (&acc)[NT], float* row, int q) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
        *reinterpret_cast<float4*>(row + 16 * t + 4 * q) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
}

// one chained layer with 128 outputs over NSRC source tiles (two per K step), weight chunks of two K steps; the source values
// are multiplied by the power of two p on their way into the fp16 split.  `after(m)` runs once K step m has been issued (its two
// source tiles are consumed by then); HOOK_OPS = the number of vector memory instructions it issues per call (they are younger
// than the chunk's DMA pieces and may stay in flight across the chunk barrier).
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
// G0: index of the layer's first weight chunk.  The chunk order is static, so every chunk's place in the blob, its size and its LDS
// buffer are constants here (resolving them at run time cost a chain of ~20 scalar branches per chunk).
template <int NSRC, bool RELU, int ABL, int G0, int HOOK_OPS = 0, class Hook = NoHook>
__device__ __forceinline__ void chained_layer(f32x4 (&acc)[kTD], const f32x4 (&src)[NSRC], float p, const float* __restrict__ blob,
                                              float* lds, int lane, int wave, Hook after = Hook()) {
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
                    const float* w0 = wl + ((kl * kTD + 2 * q) * 2) * 256;
                    mfma_pair<ABL>(acc[2 * q], acc[2 * q + 1], w0, w0 + 512, bhi, blo);
                    // every DMA piece of the successor goes out in the first K step's four slots, ahead of the hook's memory
                    // operations: the chunk's closing "at most HOOK_OPS outstanding" must cover all of them
                    if (kl == 0) {
                        if (q < 3) { if (q < kPieces) stream_issue_piece<ABL>(nx, q, lane, wave); }
                        else {
#pragma unroll
                            for (int p_ = 3; p_ < kPieces; ++p_) stream_issue_piece<ABL>(nx, p_, lane, wave);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                after(m);
            }
        }
        if (nks == 2) stream_sync<ABL, 2 * HOOK_OPS>(); else stream_sync<ABL, HOOK_OPS>();
    }
}

// a K = 16 (+ folded bias) layer with 128 outputs: one K step, B operand (ghi, glo) prepared by the caller
__device__ __forceinline__ void small_layer(f32x4 (&acc)[kTD], const half8& ghi, const half8& glo, const float* wl) {
#pragma unroll
    for (int q = 0; q < kTD / 2; ++q) {
        const float* w0 = wl + (2 * q * 2) * 256;
        mfma_pair(acc[2 * q], acc[2 * q + 1], w0, w0 + 512, ghi, glo);
    }
}

