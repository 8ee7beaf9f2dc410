// cu_mask_probe.hip — which compute units does a hipExtStreamCreateWithCUMask stream use?  Launches many small workgroups on streams with
// different masks and counts the distinct (XCC, SE, CU) ids they report.  Build: hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

__global__ void who(unsigned* out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // spin a little so that the workgroups spread over every unit the stream may use
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main() {
    const int n = 8192;
    unsigned* d;
    hipMalloc(&d, 2 * n * sizeof(unsigned));
    std::vector<unsigned> h(2 * n);
    struct M { const char* name; unsigned w[8]; };
    M masks[] = {{"all", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}}, {"first 32 bits", {~0u, 0, 0, 0, 0, 0, 0, 0}}, {"first 64 bits", {~0u, ~0u, 0, 0, 0, 0, 0, 0}},
                 {"every 2nd bit", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
                 {"every 8th bit", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
                 {"bits 0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0}}, {"words 0-3", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}}};
    for (auto& m : masks) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, m.w) != hipSuccess) { printf("%s: stream creation failed\n", m.name); continue; }
        hipMemsetAsync(d, 0xff, 2 * n * sizeof(unsigned), s);
        hipLaunchKernelGGL(who, dim3(n), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, 2 * n * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::set<unsigned> cus, xccs;
        for (int i = 0; i < n; ++i) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
            xccs.insert(xcc);
        }
        printf("%-14s: %zu distinct compute units on %zu XCCs\n", m.name, cus.size(), xccs.size());
        hipStreamDestroy(s);
    }
    return 0;
}
