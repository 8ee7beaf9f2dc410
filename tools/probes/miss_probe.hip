// miss_probe.hip — development microbenchmark (not product): what a vector-memory wave-instruction costs a gfx950 CU when its lines
// MISS L1, in the fused kernel's tap shape (buffer_load_dwordx4, 8 lanes x 16 B per 128-byte line, 8 lines per instruction, every
// line used once), 12 waves per CU, as a function of where the lines live (footprint F: 2 MB = every XCD's L2 holds it, 32 / 128 MB =
// memory-side cache, 2 GB = HBM) and of the loads a wave keeps in flight (D = 4, 8, 16).  The question it answers: is the ~44 clk per
// 1 KB instruction the fused kernel's source passes see (profiles/round3_fused_experiments.md, section 9) the miss path's ceiling,
// or a property of the kernel's schedule?
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/miss_probe.hip -o tools/_dev/miss_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MIX: 0 = every line random in the footprint; 1 = three of four instructions re-read a 16 KB private slice (L1 hits), the fourth
// misses — not the kernel's mix, a second point on the curve
template <int D, int MIX>
__global__ void __launch_bounds__(768) probe(const float* __restrict__ src, float* __restrict__ out, int iters, unsigned line_mask) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00027000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned h = 2654435761u * (unsigned)((blockIdx.x * 12 + wave) * 977 + (lane >> 3) * 131 + 7);
    const unsigned own = ((unsigned)blockIdx.x * 128u) & line_mask;       // private 16 KB slice for MIX 1
    f32x4 ring[D];
#pragma unroll
    for (int k = 0; k < D; ++k) ring[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            h = h * 1664525u + 1013904223u;
            unsigned line = (h >> 8) & line_mask;
            if (MIX == 1 && (k & 3) != 3) line = own + ((h >> 8) & 127u);
            const unsigned off = line * 128u + 16u * (lane & 7);
            acc += ring[k];                                                // consumes the load issued D instructions ago
            ring[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
            __builtin_amdgcn_sched_barrier(0);                             // keep wait(D - 1) -> load strictly alternating
        }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) acc += ring[k];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x * 768 + tid] = acc[0];
}

template <int D, int MIX>
double run(const float* src, float* out, int blocks, int instr_per_wave, unsigned line_mask) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = instr_per_wave / D;
    hipLaunchKernelGGL((probe<D, MIX>), dim3(blocks), dim3(768), 0, 0, src, out, iters / 4, line_mask);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<D, MIX>), dim3(blocks), dim3(768), 0, 0, src, out, iters, line_mask);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    // wave-instructions per CU: blocks / 256 workgroups per CU in sequence x 12 waves x instr_per_wave
    const double instr = (double)blocks / 256.0 * 12 * (double)(iters * D);
    return best * 1e6 / instr;                                             // ns per wave-instruction per CU
}

int main() {
    const int blocks = 256 * 2, ipw = 4096;
    const size_t max_bytes = (size_t)2 << 30;
    float* src; float* out;
    if (hipMalloc(&src, max_bytes) != hipSuccess || hipMalloc(&out, (size_t)blocks * 768 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 0, max_bytes);
    const size_t foot[5] = {(size_t)2 << 20, (size_t)32 << 20, (size_t)128 << 20, (size_t)512 << 20, (size_t)2 << 30};
    printf("ns (clk at 2.4 GHz) per 1 KB wave-instruction per CU; 12 waves per CU; every 128-byte line random in the footprint\n");
    printf("%-12s %-22s %-22s %-22s %-22s\n", "footprint", "D=4", "D=8", "D=16", "D=8, 3 of 4 instr L1 hits");
    for (int f = 0; f < 5; ++f) {
        const unsigned mask = (unsigned)(foot[f] / 128 - 1);
        const double a = run<4, 0>(src, out, blocks, ipw, mask), b = run<8, 0>(src, out, blocks, ipw, mask),
                     c = run<16, 0>(src, out, blocks, ipw, mask), d = run<8, 1>(src, out, blocks, ipw, mask);
        printf("%6zu MB    %7.2f (%6.1f)       %7.2f (%6.1f)       %7.2f (%6.1f)       %7.2f (%6.1f)\n", foot[f] >> 20, a, a * 2.4, b, b * 2.4, c, c * 2.4, d, d * 2.4);
    }
    return 0;
}
