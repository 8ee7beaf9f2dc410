// car_round2.hip — per-sample part of the second attention round in one kernel (reference
// models.py:548-556):
//     q2    = Wr2 relu(ug + uh[ray]) + br2         ug = Wr1[:,128:] g + br1 (per sample, from the fused kernel),
//                                                  uh = Wr1[:,:128] encode_latent(z1)   (per ray)
//     logit = <q2, qry> / 16
// q2 never exists in memory: it is produced in the MFMA accumulators (weights as A operand, samples as B operand, see
// car_linear.hip; f16 matrix pipe with fp16 hi/lo operand splits as in car_fused.hip) and immediately dotted with the
// sample's qry row.  The 128x128 layer (64 KB packed) is loaded into LDS once per workgroup (8 waves), so there is no
// weight stream and no barrier in the main loop.  HBM-bound: it reads ug and qry (2 x 512 B per sample) and writes 4 B per
// sample.  The MFMA wants lane = sample, memory wants lanes along a row: every 32-row x 32-channel tile is loaded
// coalesced (8 lanes x 16 B per row = one 128-byte line) and turned through a wave-private LDS tile; row-per-lane loads
// straight from HBM ran at 2.2 TB/s (3.9 ms per frame).
#include "car_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int kWShift = 8;          // the packed fp16 hi/lo weights carry 2^8 (see car_fused.hip, PREC = 1)
constexpr int kD = 128, kNT = 4, kTile = 1024, kChunks = 4;
constexpr int kWaves = 8, kStageLd = 36;
constexpr int kLdsBias = kChunks * kNT * kTile;                     // after the 64 KB of weights
constexpr int kLdsStage = kLdsBias + kD;                            // [8 waves][32 rows][36]
constexpr size_t kLdsBytes = (size_t)(kLdsStage + kWaves * 32 * kStageLd) * sizeof(float);

__global__ void __launch_bounds__(512) round2_kernel(const float* __restrict__ ug, const float* __restrict__ uh,
                                                     const float* __restrict__ qry, const float* __restrict__ wpacked,
                                                     const float* __restrict__ bias, int V, int R, int P, long S,
                                                     float* __restrict__ logit) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 31, h = lane >> 5;
    const int qd = lane & 7, r8 = lane >> 3;                        // coalesced side: row r8 + 8 it, channel quad qd of a 32-wide chunk
    for (int k = tid; k < kChunks * kNT * kTile / 4; k += 512)
        *reinterpret_cast<float4*>(lds + 4 * k) = *reinterpret_cast<const float4*>(wpacked + 4 * k);
    if (tid < kD) lds[kLdsBias + tid] = bias[tid];
    __syncthreads();
    const float* lbias = lds + kLdsBias;
    float* stage = lds + kLdsStage + wave * 32 * kStageLd;

    for (long row0 = ((long)blockIdx.x * kWaves + wave) * 32; row0 < S; row0 += (long)gridDim.x * kWaves * 32) {
        // all inputs of the 32 rows first: 16 + 16 + 16 float4 per lane, every load instruction covers 8 whole 128-byte lines
        float4 xs[kChunks][4], us[kChunks][4], qs[kChunks][4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const long row = row0 + r8 + 8 * it < S ? row0 + r8 + 8 * it : S - 1;
            const long nr = row / P;                               // (scene-view n, ray r)
            const long ray = ((nr / R) / V) * R + nr % R;          // (scene, ray): uh is shared by the views
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                xs[c][it] = *reinterpret_cast<const float4*>(ug + row * kD + 32 * c + 4 * qd);
                us[c][it] = *reinterpret_cast<const float4*>(uh + ray * kD + 32 * c + 4 * qd);
                qs[c][it] = *reinterpret_cast<const float4*>(qry + row * kD + 32 * c + 4 * qd);
            }
        }
        f32x16 acc[kNT];
#pragma unroll
        for (int t = 0; t < kNT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = lbias[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h] * (float)(1 << kWShift);
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            // relu(ug + uh) of this chunk through the wave's tile: written along rows, read back lane = sample
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float4 x = xs[c][it], u = us[c][it];
                *reinterpret_cast<float4*>(stage + (r8 + 8 * it) * kStageLd + 4 * qd) =
                    make_float4(fmaxf(x.x + u.x, 0.f), fmaxf(x.y + u.y, 0.f), fmaxf(x.z + u.z, 0.f), fmaxf(x.w + u.w, 0.f));
            }
            float bv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x = *reinterpret_cast<const float4*>(stage + s * kStageLd + 16 * h + 4 * q);
                bv[4 * q + 0] = x.x; bv[4 * q + 1] = x.y; bv[4 * q + 2] = x.z; bv[4 * q + 3] = x.w;
            }
            // fp16 hi/lo split of the activations, three exact products per term on the f16 matrix pipe
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
        // <q2, qry>: lane (s, h) holds channels 32 t + 8 g + 4 h + (0..3) of its sample in acc[t][4g..4g+3]; qry comes through
        // the same tile, 32 channels at a time
        float dot = 0.0f;
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
#pragma unroll
            for (int it = 0; it < 4; ++it) *reinterpret_cast<float4*>(stage + (r8 + 8 * it) * kStageLd + 4 * qd) = qs[t][it];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 qv = *reinterpret_cast<const float4*>(stage + s * kStageLd + 8 * g + 4 * h);
                dot = fmaf(acc[t][4 * g + 0], qv.x, dot); dot = fmaf(acc[t][4 * g + 1], qv.y, dot);
                dot = fmaf(acc[t][4 * g + 2], qv.z, dot); dot = fmaf(acc[t][4 * g + 3], qv.w, dot);
            }
        }
        dot += __shfl_xor(dot, 32, 64);
        if (h == 0 && row0 + s < S) logit[row0 + s] = dot * (1.0f / (float)(1 << kWShift)) / 16.0f;
    }
}

}  // namespace

extern "C" int car_round2_logits(const float* ug, const float* uh, const float* qry, const float* wpacked, const float* bias,
                                 int b, int V, int R, int P, float* logit, void* stream) {
    CAR_REQUIRE(ug && uh && qry && wpacked && bias && logit, "car_round2_logits: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && R > 0 && P > 0, "car_round2_logits: bad sizes");
    const long S = (long)b * V * R * P;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)round2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
        if (e != hipSuccess) { car_set_error("car_round2_logits: cannot reserve LDS: %s", hipGetErrorString(e)); return CAR_E_LAUNCH; }
        attr = true;
    }
    const long groups = (S + 255) / 256;
    const unsigned blocks = (unsigned)(groups < 1024 ? groups : 1024);       // one 8-wave workgroup per CU x 256 CUs x 4: grid-stride
    (void)hipGetLastError();
    hipLaunchKernelGGL(round2_kernel, dim3(blocks), dim3(512), kLdsBytes, (hipStream_t)stream, ug, uh, qry, wpacked, bias, V, R,
                       P, S, logit);
    CAR_CHECK_LAUNCH("car_round2_logits");
    return CAR_OK;
}
