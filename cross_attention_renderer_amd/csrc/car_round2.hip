// car_round2.hip — per-sample part of the second attention round in one kernel (reference
// models.py:548-556):
//     ug    = Wr1[:,128:] g + br1                  g  = the sample's 16-channel geometric query (from the fused kernel)
//     q2    = Wr2 relu(ug + uh[ray]) + br2         uh = Wr1[:,:128] encode_latent(z1)   (per ray, car_ray_mid)
//     logit = <q2, qry> / 16
// Neither ug nor q2 exists in memory: ug is produced in the MFMA accumulators from the 64-byte g row (weights as A operand,
// samples as B operand), its registers — plus uh, loaded in the same accumulator layout — are the B operand of the second layer
// (the accumulator's channel permutation is baked into the packed Wr2, "chained" K order), and q2 is dotted with the sample's
// qry row straight out of the accumulators.  f16 matrix pipe with fp16 hi/lo operand splits (three products per term, see
// car_fused_mma.h); the activations of the second layer are scaled per sample by a power of two so that the split stays inside
// fp16's normal range whatever their magnitude.  Both layers (4 KB + 64 KB packed) sit in LDS, loaded once per workgroup
// (8 waves): no weight stream and no barrier in the main loop.  HBM-bound on the qry rows (512 B per sample) + g (64 B): they
// are loaded coalesced (8 lanes x 16 B per row = one 128-byte line) and turned through a wave-private LDS tile.
#include "car_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
constexpr int kD = 128, kNT = 4, kTile = 1024, kChunks = 4;
constexpr int kWaves = 8, kStageLd = 36;
constexpr int kLdsW1 = kChunks * kNT * kTile;                       // Wr1g tiles after the 64 KB of Wr2: [4 tiles][hi|lo][64 lanes][8 halves]
constexpr int kLdsBias = kLdsW1 + kNT * 512;                        // br1 [128] | br2 [128] | 2^-shift of Wr1g, Wr2
constexpr int kLdsStage = kLdsBias + 2 * kD + 4;                    // [8 waves][32 rows][36]
constexpr size_t kLdsBytes = (size_t)(kLdsStage + kWaves * 32 * kStageLd) * sizeof(float);
// G instance (car_round2_logits_from_g): no qry rows.  <q2, qry> with q2 = Wr2 y + br2, qry = Wq2 x + bq2 (y = relu(Wr1g g + br1 + uh),
// x = relu(Wq1 g + bq1), models.py:529, 549-556) is the bilinear form y^T (M x + v) + u^T x + c with M = Wr2^T Wq2, v = Wr2^T bq2,
// u = Wq2^T br2, c = <br2, bq2> folded once per checkpoint (car_round2q_pack): ONE 128 x 128 layer per sample, as many matrix operations as
// the stored-query form needs for q2 alone, and nothing 128 wide is read.  Packed: M (chained over x, 64 KB) | Wr1g | Wq1 (8 KB each).
constexpr int kLdsWq1 = kLdsBias;                                   // Wq1 tiles behind Wr1g's
constexpr int kLdsBiasG = kLdsWq1 + kNT * 512;                      // br1 | v | bq1 | u | 2^-shift of Wr1g, M, Wq1 | c
constexpr int kBiasFloatsG = 4 * kD + 8;
constexpr int kScratchG = kD * kD;                                  // car_round2q_pack's scratch behind the bias table: M in fp32
constexpr size_t kLdsBytesG = (size_t)(kLdsBiasG + kBiasFloatsG) * sizeof(float);

#include "car_split.h"

// acc = W1 x16 + b1 for the wave's 32 samples (K = 16, one v_mfma_f32_32x32x16_f16 step, fp16 hi / lo operand halves), times `up`:
// lane (s, h) register 4 g' + r of tile t = channel 32 t + 8 g' + 4 h + r
template <bool ZERO = false>
__device__ __forceinline__ void first_layer(const half8& ghi, const half8& glo, const float* lw1, const float* lb1, float up, int lane, int h,
                                            f32x16 (&ug)[kNT]) {
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
        // ZERO: the products alone, from a zero accumulator (the matrix instruction takes the constant: no 64 register writes, no 64 multiplies
        // by `up`); the caller adds the bias after undoing the scale
#pragma unroll
        for (int r = 0; r < 16; ++r) ug[t][r] = ZERO ? 0.0f : lb1[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h] * up;
        const float* w1 = lw1 + t * 512 + 4 * lane;
        const half8 ah = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w1));
        const half8 al = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w1 + 256));
        ug[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ghi, ug[t], 0, 0, 0);
        ug[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, glo, ug[t], 0, 0, 0);
        ug[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ghi, ug[t], 0, 0, 0);
    }
}
// acc = W2 x + b2 with x in the accumulator layout above (its registers are the B operands: K step (source tile c, group kg) takes
// registers 8 kg .. 8 kg + 7 of x's tile c), times xp / down2
template <bool ZERO = false>
__device__ __forceinline__ void second_layer(const f32x16 (&x)[kNT], float xp, const float* lw2, const float* lb2, float up2, int lane, int h,
                                             f32x16 (&acc)[kNT]) {
#pragma unroll
    for (int t = 0; t < kNT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = ZERO ? 0.0f : lb2[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h] * up2;
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
        half8 bhi[2], blo[2];
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            float x8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x8[e] = x[c][8 * kg + e];
            split8(x8, xp, bhi[kg], blo[kg]);
        }
        const float* wl = lw2 + c * kNT * kTile + 4 * lane;
        // the A operands of step (t, kg) + 1 are read before the three products of step (t, kg) are issued (car_raychain.hip::mma_chunk)
        auto read = [&](int i, half8& ah, half8& al) {
            const int t = i >> 1, kg = i & 1;
            ah = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wl + t * kTile + ((kg * 2 + 0) * 64) * 4));
            al = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wl + t * kTile + ((kg * 2 + 1) * 64) * 4));
        };
        half8 ah[2], al[2];
        read(0, ah[0], al[0]);
#pragma unroll
        for (int i = 0; i < 2 * kNT; ++i) {
            const int t = i >> 1, kg = i & 1, cur = i & 1;
            if (i + 1 < 2 * kNT) read(i + 1, ah[cur ^ 1], al[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bhi[kg], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], blo[kg], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], bhi[kg], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// G = false: qry rows are read (the staged routes, which keep them).  G = true: the bilinear form above, from g and uh alone.
template <bool G>
__global__ void __launch_bounds__(512) round2_kernel(const float* __restrict__ g, const float* __restrict__ uh,
                                                     const float* __restrict__ qry, const float* __restrict__ wpacked,
                                                     const float* __restrict__ bias, int V, int R, int P, long S,
                                                     float* __restrict__ logit) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 31, h = lane >> 5;
    const int qd = lane & 7, r8 = lane >> 3;                        // coalesced side: row r8 + 8 it, channel quad qd of a 32-wide chunk
    constexpr int kW = G ? kLdsBiasG : kLdsBias, kB = G ? kBiasFloatsG : 2 * kD + 4;
    for (int k = tid; k < kW / 4; k += 512)
        *reinterpret_cast<float4*>(lds + 4 * k) = *reinterpret_cast<const float4*>(wpacked + 4 * k);
    for (int k = tid; k < kB; k += 512) lds[kW + k] = bias[k];
    __syncthreads();
    const float* lb = lds + kW;
    const float down1 = lb[(G ? 4 : 2) * kD], down2 = lb[(G ? 4 : 2) * kD + 1];
    float* stage = lds + kLdsStage + wave * 32 * kStageLd;          // G: unused

    for (long row0 = ((long)blockIdx.x * kWaves + wave) * 32; row0 < S; row0 += (long)gridDim.x * kWaves * 32) {
        // the HBM stream first: qry rows, coalesced (every load instruction covers 8 whole 128-byte lines)
        float4 qs[G ? 1 : kChunks][G ? 1 : 4];
        if constexpr (!G) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const long row = row0 + r8 + 8 * it < S ? row0 + r8 + 8 * it : S - 1;
#pragma unroll
                for (int c = 0; c < kChunks; ++c) {                  // read once: non-temporal
                    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(qry + row * kD + 32 * c + 4 * qd));
                    qs[c][it] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        // this lane's sample: its g half (B operand of the first layers) and the row of uh its ray owns
        const long srow = row0 + s < S ? row0 + s : S - 1;
        const long nr = srow / P;                                  // (scene-view n, ray r)
        const float* uhrow = uh + (((nr / R) / V) * R + nr % R) * kD;   // (scene, ray): uh is shared by the views
        half8 ghi, glo;
        float gp, ginv;
        {
            const float4 g0 = *reinterpret_cast<const float4*>(g + srow * 16 + 8 * h);
            const float4 g1 = *reinterpret_cast<const float4*>(g + srow * 16 + 8 * h + 4);
            const float gx[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            float m = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(gx[k]));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            pow2_scale(fmaxf(m, 1e-30f), gp, ginv);
            split8(gx, gp, ghi, glo);
        }
        // ug = Wr1g g + br1; y = relu(ug + uh) in place.  G: the biases of all three layers are added AFTER the products, in true units (one
        // fma each, where the product's scale is undone anyway), so the accumulators start from the matrix instruction's zero
        f32x16 ug[kNT];
        first_layer<G>(ghi, glo, lds + kLdsW1, lb, gp / down1, lane, h, ug);
        const float undo1 = down1 * ginv;
        float xm = 0.0f;
#pragma unroll
        for (int t = 0; t < kNT; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 u = *reinterpret_cast<const float4*>(uhrow + 32 * t + 8 * gq + 4 * h);
                float uu[4] = {u.x, u.y, u.z, u.w};
                if constexpr (G) {
                    const float4 b = *reinterpret_cast<const float4*>(lb + 32 * t + 8 * gq + 4 * h);
                    uu[0] += b.x; uu[1] += b.y; uu[2] += b.z; uu[3] += b.w;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = fmaxf(fmaf(ug[t][4 * gq + r], undo1, uu[r]), 0.0f);
                    ug[t][4 * gq + r] = x;
                    xm = fmaxf(xm, x);
                }
            }
        float dot = 0.0f;
        if constexpr (G) {
            // x = relu(Wq1 g + bq1); t = M x + v; <q2, qry> = y^T t + u^T x + c
            const float downq = lb[4 * kD + 2];
            f32x16 xq[kNT];
            first_layer<true>(ghi, glo, lds + kLdsWq1, lb + 2 * kD, gp / downq, lane, h, xq);
            const float undoq = downq * ginv;
            const float* lu = lb + 3 * kD;
            const float* lq = lb + 2 * kD;
            float xqm = 0.0f, dot_u = 0.0f;
#pragma unroll
            for (int t = 0; t < kNT; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 u = *reinterpret_cast<const float4*>(lu + 32 * t + 8 * gq + 4 * h);
                    const float4 b = *reinterpret_cast<const float4*>(lq + 32 * t + 8 * gq + 4 * h);
                    const float uu[4] = {u.x, u.y, u.z, u.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = fmaxf(fmaf(xq[t][4 * gq + r], undoq, bb[r]), 0.0f);
                        xq[t][4 * gq + r] = x;
                        xqm = fmaxf(xqm, x);
                        dot_u = fmaf(uu[r], x, dot_u);
                    }
                }
            xqm = fmaxf(xqm, __shfl_xor(xqm, 32, 64));
            float xp, xinv;
            pow2_scale(fmaxf(xqm, 1e-30f), xp, xinv);
            f32x16 acc[kNT];
            second_layer<true>(xq, xp, lds, lb + kD, xp / down2, lane, h, acc);
            // y^T (M x) in the product's units, y^T v in true ones
            const float* lv = lb + kD;
            float dot_v = 0.0f;
#pragma unroll
            for (int t = 0; t < kNT; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 v4 = *reinterpret_cast<const float4*>(lv + 32 * t + 8 * gq + 4 * h);
                    const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dot = fmaf(ug[t][4 * gq + r], acc[t][4 * gq + r], dot);
                        dot_v = fmaf(ug[t][4 * gq + r], vv[r], dot_v);
                    }
                }
            dot = fmaf(dot, down2 * xinv, dot_u + dot_v);
            dot += __shfl_xor(dot, 32, 64);
            dot += lb[4 * kD + 3];
        } else {
            xm = fmaxf(xm, __shfl_xor(xm, 32, 64));
            float xp, xinv;
            pow2_scale(fmaxf(xm, 1e-30f), xp, xinv);
            f32x16 acc[kNT];
            second_layer(ug, xp, lds, lb + kD, xp / down2, lane, h, acc);
            // <q2, qry>: lane (s, h) holds channels 32 t + 8 g' + 4 h + (0..3) of its sample in acc[t][4g'..4g'+3]; qry comes through
            // the wave's tile, 32 channels at a time
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
#pragma unroll
                for (int it = 0; it < 4; ++it) *reinterpret_cast<float4*>(stage + (r8 + 8 * it) * kStageLd + 4 * qd) = qs[t][it];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 qv = *reinterpret_cast<const float4*>(stage + s * kStageLd + 8 * gq + 4 * h);
                    dot = fmaf(acc[t][4 * gq + 0], qv.x, dot); dot = fmaf(acc[t][4 * gq + 1], qv.y, dot);
                    dot = fmaf(acc[t][4 * gq + 2], qv.z, dot); dot = fmaf(acc[t][4 * gq + 3], qv.w, dot);
                }
            }
            dot += __shfl_xor(dot, 32, 64);
            dot *= down2 * xinv;
        }
        if (h == 0 && row0 + s < S) logit[row0 + s] = dot / 16.0f;
    }
}

}  // namespace

extern "C" size_t car_round2_packed_floats(void) { return (size_t)kLdsBias; }
extern "C" size_t car_round2_bias_floats(void) { return (size_t)(2 * kD + 4); }

extern "C" size_t car_round2q_packed_floats(void) { return (size_t)kLdsBiasG; }
extern "C" size_t car_round2q_bias_floats(void) { return (size_t)(kBiasFloatsG + kScratchG); }      // the table + car_round2q_pack's scratch

static int launch_round2(bool qg, const float* g, const float* uh, const float* qry, const float* wpacked, const float* bias,
                         int b, int V, int R, int P, float* logit, void* stream, const char* who) {
    CAR_REQUIRE(g && uh && (qg || qry) && wpacked && bias && logit, "%s: null pointer", who);
    CAR_REQUIRE(b > 0 && V > 0 && R > 0 && P > 0, "%s: bad sizes", who);
    const long S = (long)b * V * R * P;
    void (*kern)(const float*, const float*, const float*, const float*, const float*, int, int, int, long, float*) =
        qg ? round2_kernel<true> : round2_kernel<false>;
    const size_t lds_bytes = qg ? kLdsBytesG : kLdsBytes;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { car_set_error("%s: cannot reserve LDS: %s", who, hipGetErrorString(e)); return CAR_E_LAUNCH; }
    const long groups = (S + 255) / 256;
    const unsigned blocks = (unsigned)(groups < 1024 ? groups : 1024);       // one 8-wave workgroup per CU x 256 CUs x 4: grid-stride
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds_bytes, (hipStream_t)stream, g, uh, qry, wpacked, bias, V, R, P, S, logit);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { car_set_error("%s: launch failed: %s", who, hipGetErrorString(e_)); return CAR_E_LAUNCH; }
    return CAR_OK;
}

extern "C" int car_round2_logits(const float* g, const float* uh, const float* qry, const float* wpacked, const float* bias,
                                 int b, int V, int R, int P, float* logit, void* stream) {
    return launch_round2(false, g, uh, qry, wpacked, bias, b, V, R, P, logit, stream, "car_round2_logits");
}

// The same logits without the qry rows: the bilinear form of y and x = relu(query_embed(g)) described above.  wpacked / bias: car_round2q_pack.
extern "C" int car_round2_logits_from_g(const float* g, const float* uh, const float* wpacked, const float* bias, int b, int V, int R, int P,
                                        float* logit, void* stream) {
    return launch_round2(true, g, uh, nullptr, wpacked, bias, b, V, R, P, logit, stream, "car_round2_logits_from_g");
}
