// issue_probe.hip — development microbenchmark (not product): do MFMA and ordinary vector ALU instructions of DIFFERENT waves on the
// same gfx950 SIMD overlap, or do they take turns?  One workgroup per CU, 8 waves = 2 per SIMD: waves 0-3 ("matrix") run back-to-back
// v_mfma_f32_16x16x32_f16 on independent accumulators, waves 4-7 ("vector") run a chain-free stream of v_fma_f32 / v_pk_fma_f32 /
// ds_read_b128.  Times: matrix alone, vector alone, both.  both ~ max(...) = they overlap; both ~ sum = they share the issue slot.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/issue_probe.hip -o tools/_dev/issue_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// MODE bit 0: matrix waves work, bit 1: vector waves work; KIND: 0 v_fma_f32, 1 v_pk_fma_f32, 2 ds_read_b128, 3 v_cvt_pkrtz + v_max
template <int MODE, int KIND>
__global__ void __launch_bounds__(512) probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    lds[tid] = (float)tid; lds[tid + 512] = 1.0f;
    __syncthreads();
    if (wave < 4) {
        if (!(MODE & 1)) return;
        f32x4 acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        half8 a, b;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(lane * 0.001f + k); b[k] = (_Float16)(k * 0.5f); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
        if (s == 123.456f) out[blockIdx.x * 512 + tid] = s;
    } else {
        if (!(MODE & 2)) return;
        if constexpr (KIND == 0) {
            float x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = (float)(lane + k);
            const float m = 1.0000001f, c = 1e-7f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 16; ++k) x[k] = __builtin_fmaf(x[k], m, c);       // 16 independent chains
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += x[k];
            if (s == 123.456f) out[blockIdx.x * 512 + tid] = s;
        } else if constexpr (KIND == 1) {
            f32x2 x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = f32x2{(float)(lane + k), (float)k};
            const f32x2 m = {1.0000001f, 1.0000001f}, c = {1e-7f, 1e-7f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 16; ++k) x[k] = __builtin_elementwise_fma(x[k], m, c);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += x[k][0] + x[k][1];
            if (s == 123.456f) out[blockIdx.x * 512 + tid] = s;
        } else if constexpr (KIND == 2) {
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const f32x4 v = *reinterpret_cast<const volatile f32x4*>(lds + ((4 * lane + 256 * k) & 4095));
                    s4 += v;
                }
            }
            if (s4[0] + s4[1] + s4[2] + s4[3] == 123.456f) out[blockIdx.x * 512 + tid] = s4[0];
        } else {
            float x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = (float)(lane + k) * 0.01f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
                    const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(x[k], x[k + 1]);
                    x[k] = __builtin_fmaxf(x[k] - (float)h[0], 1e-3f) + 1.0f;
                    x[k + 1] = __builtin_fmaxf(x[k + 1] - (float)h[1], 1e-3f) + 1.0f;
                }
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += x[k];
            if (s == 123.456f) out[blockIdx.x * 512 + tid] = s;
        }
    }
}

template <int MODE, int KIND>
float run(float* out, int iters) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((probe<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, iters / 8);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((probe<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

template <int KIND>
void kind(float* out, int iters, const char* name, int vec_instr_per_iter) {
    const float m = run<1, KIND>(out, iters), v = run<2, KIND>(out, iters), both = run<3, KIND>(out, iters);
    printf("%-34s matrix alone %7.3f ms (%5.1f ns / MFMA)   vector alone %7.3f ms (%5.2f ns / instr)   both %7.3f ms   sum %7.3f   max %7.3f\n", name, m,
           m * 1e6 / (8.0 * iters), v, v * 1e6 / ((double)vec_instr_per_iter * iters), both, m + v, m > v ? m : v);
}

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
    const int iters = 20000;
    printf("one MFMA wave + one vector wave per SIMD; per iteration: 8 x v_mfma_f32_16x16x32_f16 vs the vector stream below\n");
    kind<0>(out, iters, "16 x v_fma_f32", 16);
    kind<1>(out, iters, "16 x v_pk_fma_f32", 16);
    kind<2>(out, iters, "16 x ds_read_b128 + 16 x 4 v_add", 16);
    kind<3>(out, iters, "8 x (cvt_pkrtz, 2 cvt, 2 sub, 2 max, 2 add)", 72);
    return 0;
}
