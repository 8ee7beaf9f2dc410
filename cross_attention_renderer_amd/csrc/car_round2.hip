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

#include "car_split.h"

__global__ void __launch_bounds__(512) round2_kernel(const float* __restrict__ g, const float* __restrict__ uh,
                                                     const float* __restrict__ qry, const float* __restrict__ wpacked,
                                                     const float* __restrict__ bias, int V, int R, int P, long S,
                                                     float* __restrict__ logit) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 31, h = lane >> 5;
    const int qd = lane & 7, r8 = lane >> 3;                        // coalesced side: row r8 + 8 it, channel quad qd of a 32-wide chunk
    for (int k = tid; k < (kLdsBias - 0) / 4; k += 512)
        *reinterpret_cast<float4*>(lds + 4 * k) = *reinterpret_cast<const float4*>(wpacked + 4 * k);
    if (tid < 2 * kD + 4) lds[kLdsBias + tid] = bias[tid];
    __syncthreads();
    const float* lb1 = lds + kLdsBias;
    const float* lb2 = lds + kLdsBias + kD;
    const float down1 = lds[kLdsBias + 2 * kD], down2 = lds[kLdsBias + 2 * kD + 1];
    float* stage = lds + kLdsStage + wave * 32 * kStageLd;

    for (long row0 = ((long)blockIdx.x * kWaves + wave) * 32; row0 < S; row0 += (long)gridDim.x * kWaves * 32) {
        // the HBM stream first: qry rows, coalesced (every load instruction covers 8 whole 128-byte lines)
        float4 qs[kChunks][4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long row = row0 + r8 + 8 * it < S ? row0 + r8 + 8 * it : S - 1;
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {                      // read once: non-temporal
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(qry + row * kD + 32 * c + 4 * qd));
                qs[c][it] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        // this lane's sample: its g half (B operand of the first layer) and the row of uh its ray owns
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
        // ug = Wr1g g + br1, accumulator layout: lane (s, h) register 4 g' + r of tile t = channel 32 t + 8 g' + 4 h + r
        f32x16 ug[kNT];
        const float up1 = gp / down1;
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ug[t][r] = lb1[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h] * up1;
            const float* w1 = lds + kLdsW1 + t * 512 + 4 * lane;
            const half8 ah = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w1));
            const half8 al = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w1 + 256));
            ug[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ghi, ug[t], 0, 0, 0);
            ug[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, glo, ug[t], 0, 0, 0);
            ug[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ghi, ug[t], 0, 0, 0);
        }
        // x = relu(ug + uh) in place, and its largest magnitude over the sample's 128 channels
        const float undo1 = down1 * ginv;
        float xm = 0.0f;
#pragma unroll
        for (int t = 0; t < kNT; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 u = *reinterpret_cast<const float4*>(uhrow + 32 * t + 8 * gq + 4 * h);
                const float uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = fmaxf(fmaf(ug[t][4 * gq + r], undo1, uu[r]), 0.0f);
                    ug[t][4 * gq + r] = x;
                    xm = fmaxf(xm, x);
                }
            }
        xm = fmaxf(xm, __shfl_xor(xm, 32, 64));
        float xp, xinv;
        pow2_scale(fmaxf(xm, 1e-30f), xp, xinv);
        // q2 = Wr2 x + br2: K step (source tile c, group kg) takes registers 8 kg .. 8 kg + 7 of x's tile c
        f32x16 acc[kNT];
        const float up2 = xp / down2;
#pragma unroll
        for (int t = 0; t < kNT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = lb2[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h] * up2;
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            half8 bhi[2], blo[2];
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) {
                float x8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x8[e] = ug[c][8 * kg + e];
                split8(x8, xp, bhi[kg], blo[kg]);
            }
            const float* wl = lds + c * kNT * kTile + 4 * lane;
#pragma unroll
            for (int t = 0; t < kNT; ++t)
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) {
                    const half8 ah = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wl + t * kTile + ((kg * 2 + 0) * 64) * 4));
                    const half8 al = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wl + t * kTile + ((kg * 2 + 1) * 64) * 4));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhi[kg], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blo[kg], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhi[kg], acc[t], 0, 0, 0);
                }
        }
        // <q2, qry>: lane (s, h) holds channels 32 t + 8 g' + 4 h + (0..3) of its sample in acc[t][4g'..4g'+3]; qry comes through
        // the wave's tile, 32 channels at a time
        float dot = 0.0f;
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
        if (h == 0 && row0 + s < S) logit[row0 + s] = dot * (down2 * xinv) / 16.0f;
    }
}

}  // namespace

extern "C" size_t car_round2_packed_floats(void) { return (size_t)kLdsBias; }
extern "C" size_t car_round2_bias_floats(void) { return (size_t)(2 * kD + 4); }

extern "C" int car_round2_logits(const float* g, const float* uh, const float* qry, const float* wpacked, const float* bias,
                                 int b, int V, int R, int P, float* logit, void* stream) {
    CAR_REQUIRE(g && uh && qry && wpacked && bias && logit, "car_round2_logits: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && R > 0 && P > 0, "car_round2_logits: bad sizes");
    const long S = (long)b * V * R * P;
    hipError_t e = hipFuncSetAttribute((const void*)round2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) { car_set_error("car_round2_logits: cannot reserve LDS: %s", hipGetErrorString(e)); return CAR_E_LAUNCH; }
    const long groups = (S + 255) / 256;
    const unsigned blocks = (unsigned)(groups < 1024 ? groups : 1024);       // one 8-wave workgroup per CU x 256 CUs x 4: grid-stride
    (void)hipGetLastError();
    hipLaunchKernelGGL(round2_kernel, dim3(blocks), dim3(512), kLdsBytes, (hipStream_t)stream, g, uh, qry, wpacked, bias, V, R,
                       P, S, logit);
    CAR_CHECK_LAUNCH("car_round2_logits");
    return CAR_OK;
}
