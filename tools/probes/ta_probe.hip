// ta_probe.hip — development microbenchmark (not product): what one vector-memory wave-instruction costs the texture-address /
// L1 path of a gfx950 CU when every line hits L1, in the gather's access shape (8 lanes x 16 B per 128-B line, 8 lines per
// instruction), 12 waves per CU, 8 loads in flight per wave.  Variants:
//   0 dwordx4, all lanes live              1 dwordx4, every other instruction issued with EXEC = 0
//   2 dwordx2 (16 lanes per line)          3 dword (32 lanes per line)
//   4 dwordx4, odd instructions' lanes all out of the buffer's range (buffer_load, range-checked -> zeros)
//   5 dwordx4 buffer_load, all in range    6 global_load_lds_dwordx4 (LDS-DMA, 1 KB per instruction)
//   7 dwordx4, every other instruction skipped by a scalar branch (the floor for 1 and 4)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/ta_probe.hip -o tools/_dev/ta_probe ; run on the GPU box.
#define HIPCHK(x) (void)(x)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ void __launch_bounds__(768) probe(const float* __restrict__ src, float* __restrict__ out, int iters, unsigned region_bytes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a workgroup's own 16 KB slice (L1 resident after the first pass); rows = 128-byte lines at pseudo-random places in it
    const char* base = reinterpret_cast<const char*>(src) + (size_t)blockIdx.x * 16384;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned h = 2654435761u * (unsigned)(wave * 977 + (lane >> 3) * 131 + 7);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 16384, 0x00027000);
    for (int it = 0; it < iters; ++it) {
        unsigned off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            h = h * 1664525u + 1013904223u;
            unsigned line = (h >> 16) & 127u;                      // 128 lines of 128 B
            if (V == 2) off[k] = (((h >> 16) & 127u) ^ ((lane >> 4) & 3u)) * 128u + 8u * (lane & 15);
            else if (V == 3) off[k] = (((h >> 16) & 127u) ^ ((lane >> 5) & 1u)) * 128u + 4u * (lane & 31);
            else off[k] = line * 128u + 16u * (lane & 7);
        }
        if constexpr (V == 0 || V == 1 || V == 7) {
            f32x4 t[8];
            const char* p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = base + off[k];
            if constexpr (V == 0) {
                asm volatile(
                    "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %9, off\n\tglobal_load_dwordx4 %2, %10, off\n\tglobal_load_dwordx4 %3, %11, off\n\t"
                    "global_load_dwordx4 %4, %12, off\n\tglobal_load_dwordx4 %5, %13, off\n\tglobal_load_dwordx4 %6, %14, off\n\tglobal_load_dwordx4 %7, %15, off\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                    : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
            } else if constexpr (V == 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                unsigned long long keep;
                asm volatile(
                    "s_mov_b64 %[k], exec\n\t"
                    "global_load_dwordx4 %[t0], %[p0], off\n\ts_mov_b64 exec, 0\n\tglobal_load_dwordx4 %[t1], %[p1], off\n\ts_mov_b64 exec, %[k]\n\t"
                    "global_load_dwordx4 %[t2], %[p2], off\n\ts_mov_b64 exec, 0\n\tglobal_load_dwordx4 %[t3], %[p3], off\n\ts_mov_b64 exec, %[k]\n\t"
                    "global_load_dwordx4 %[t4], %[p4], off\n\ts_mov_b64 exec, 0\n\tglobal_load_dwordx4 %[t5], %[p5], off\n\ts_mov_b64 exec, %[k]\n\t"
                    "global_load_dwordx4 %[t6], %[p6], off\n\ts_mov_b64 exec, 0\n\tglobal_load_dwordx4 %[t7], %[p7], off\n\ts_mov_b64 exec, %[k]\n\t"
                    "s_waitcnt vmcnt(0)"
                    : [t0] "+v"(t[0]), [t1] "+v"(t[1]), [t2] "+v"(t[2]), [t3] "+v"(t[3]), [t4] "+v"(t[4]), [t5] "+v"(t[5]), [t6] "+v"(t[6]), [t7] "+v"(t[7]),
                      [k] "=&s"(keep)
                    : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]), [p7] "v"(p[7]) : "memory");
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                asm volatile(
                    "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\tglobal_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(t[0]), "=&v"(t[2]), "=&v"(t[4]), "=&v"(t[6])
                    : "v"(p[0]), "v"(p[2]), "v"(p[4]), "v"(p[6]) : "memory");
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += t[k];
        } else if constexpr (V == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(base + off[k]);
                acc[0] += v[0]; acc[1] += v[1];
            }
        } else if constexpr (V == 3) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[0] += *reinterpret_cast<const float*>(base + off[k]);
        } else if constexpr (V == 4 || V == 5) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned o = (V == 4 && (k & 1)) ? 0xfffffe00u : off[k];
                acc += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0));
            }
        } else if constexpr (V == 6) {
            // LDS-DMA: wave w copies 8 KB (8 pieces) of its workgroup's slice into its own 8 KB of LDS
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + wave * 2048 + k * 256));
                const char* g = base + ((off[k] & ~1023u) & 16383u) + 16 * lane;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if constexpr (V == 6) acc[0] += lds[tid];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x * 768 + tid] = acc[0];
}

template <int V>
float run(const float* src, float* out, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const size_t lds = V == 6 ? 12 * 8192 : 0;
    if (lds) hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(768), lds, 0, src, out, 8, 16384u);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(768), lds, 0, src, out, iters, 16384u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int blocks = 256 * 4, iters = 2000;
    float* src; float* out;
    hipMalloc(&src, (size_t)blocks * 16384); hipMalloc(&out, (size_t)blocks * 768 * 4);
    std::vector<float> h((size_t)blocks * 4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[8] = {"dwordx4 live", "dwordx4, odd instr EXEC=0", "dwordx2", "dword", "buffer dwordx4, odd instr out of range",
                            "buffer dwordx4 in range", "LDS-DMA dwordx4", "dwordx4, odd instr branched over"};
    float ms[8];
    ms[0] = run<0>(src, out, blocks, iters); ms[1] = run<1>(src, out, blocks, iters); ms[2] = run<2>(src, out, blocks, iters);
    ms[3] = run<3>(src, out, blocks, iters); ms[4] = run<4>(src, out, blocks, iters); ms[5] = run<5>(src, out, blocks, iters);
    ms[6] = run<6>(src, out, blocks, iters); ms[7] = run<7>(src, out, blocks, iters);
    for (int v = 0; v < 8; ++v) {
        // wave-instructions per CU: 4 workgroups per CU in sequence x 12 waves x 8 per iteration
        const double instr = 4.0 * 12 * 8 * iters;
        printf("variant %d %-42s %8.3f ms   %6.2f ns per wave-instruction slot per CU  (%5.1f clk at 2.4 GHz, %5.1f at 2.0)\n", v, names[v], ms[v],
               ms[v] * 1e6 / instr, ms[v] * 1e6 / instr * 2.4, ms[v] * 1e6 / instr * 2.0);
    }
    return 0;
}
