// shadow_probe.hip — development microbenchmark (not product): how many independent vector-ALU / LDS instructions of the SAME wave
// disappear in the shadow of an MFMA on a gfx950 SIMD, for v_mfma_f32_16x16x32_f16 (16 clocks) and v_mfma_f32_32x32x16_f16 (32 clocks),
// with 1, 2 or 3 waves per SIMD running the same stream.  Per MFMA the stream carries F fillers (v_fma_f32 on independent chains, or
// ds_read_b128 for KIND 1); shader-clock ticks per MFMA come from s_memtime inside the kernel (clock-independent), the wall time beside it.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/shadow_probe.hip -o tools/_dev/shadow_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int TILE, int F, int KIND>
__global__ void __launch_bounds__(768) probe(float* out, long long* ticks, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int k = tid; k < 4096; k += blockDim.x) lds[k] = (float)k;
    __syncthreads();
    half8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(lane * 0.001f + k); b[k] = (_Float16)(k * 0.5f); }
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (float)(lane + k);
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
    const float m = 1.0000001f, c = 1e-7f;
    auto fillers = [&](int j) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            if constexpr (KIND == 0) x[(j * F + f) & 7] = __builtin_fmaf(x[(j * F + f) & 7], m, c);
            else { const f32x4 v = *reinterpret_cast<const volatile f32x4*>(lds + ((4 * lane + 256 * (j * F + f)) & 4095)); s4 += v; }
        }
    };
    float s = 0.f;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    if constexpr (TILE == 16) {
        f32x4 acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
                fillers(k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 3], 0, 0, 0);
                fillers(k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][15];
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll
    for (int k = 0; k < 8; ++k) s += x[k];
    s += s4[0] + s4[1] + s4[2] + s4[3];
    if (s == 123.456f) out[blockIdx.x * 768 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) ticks[0] = t1 - t0;
}

template <int TILE, int F, int KIND>
void run(float* out, long long* ticks, int waves_per_simd, int iters) {
    const int threads = 256 * waves_per_simd;
    const size_t lds = 100 * 1024;                       // one workgroup per CU
    (void)hipFuncSetAttribute((const void*)probe<TILE, F, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((probe<TILE, F, KIND>), dim3(256), dim3(threads), lds, 0, out, ticks, iters / 8);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((probe<TILE, F, KIND>), dim3(256), dim3(threads), lds, 0, out, ticks, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    long long t = 0;
    (void)hipMemcpy(&t, ticks, sizeof(t), hipMemcpyDeviceToHost);
    const double n = 8.0 * iters;
    printf("  %dx%d  %s x %d per MFMA  %d wave(s)/SIMD: %7.1f ticks per MFMA of one wave = %6.1f per MFMA of the SIMD   (%.3f ms, %.2f GHz)\n", TILE, TILE,
           KIND == 0 ? "v_fma_f32  " : "ds_read_b128", F, waves_per_simd, t / n, t / n / waves_per_simd, best, t / (best * 1e6));
}

template <int TILE, int KIND>
void sweep(float* out, long long* ticks, int iters) {
    for (int w = 1; w <= 3; ++w) {
        run<TILE, 0, KIND>(out, ticks, w, iters); run<TILE, 1, KIND>(out, ticks, w, iters); run<TILE, 2, KIND>(out, ticks, w, iters);
        run<TILE, 3, KIND>(out, ticks, w, iters); run<TILE, 4, KIND>(out, ticks, w, iters); run<TILE, 6, KIND>(out, ticks, w, iters);
        run<TILE, 8, KIND>(out, ticks, w, iters);
    }
}

int main() {
    float* out; long long* ticks;
    if (hipMalloc(&out, 256 * 768 * 4) != hipSuccess || hipMalloc(&ticks, 64) != hipSuccess) return 1;
    const int iters = 4000;
    printf("per wave: [1 MFMA + F fillers] x 8 per iteration, sched_barrier between groups\n");
    sweep<16, 0>(out, ticks, iters);
    sweep<32, 0>(out, ticks, iters);
    sweep<16, 1>(out, ticks, iters);
    sweep<32, 1>(out, ticks, iters);
    return 0;
}
