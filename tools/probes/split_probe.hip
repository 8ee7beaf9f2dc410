// split_probe.hip — development check (not product): the fp16 hi / lo split of car_fused_mma.h (v_cvt_pkrtz_f16_f32 + v_fma_mixlo/mixhi_f16) on its own:
// worst relative residual |x - hi - lo| / |x| over a range of magnitudes.  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/split_probe.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <cmath>
#include <cstdio>
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* out) {
    float a = x[threadIdx.x], b = x[threadIdx.x + 64];
    const fp16x2 h2 = __builtin_amdgcn_cvt_pkrtz(a, b);
    const unsigned hi = __builtin_bit_cast(unsigned, h2);
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(hi), "v"(a), "v"(b));
    out[threadIdx.x] = hi;
    out[threadIdx.x + 64] = lo;
}
int main() {
    float* x; unsigned* o; hipMalloc(&x, 512); hipMalloc(&o, 512);
    float hx[128]; for (int i = 0; i < 128; ++i) hx[i] = (i % 2 ? -1.f : 1.f) * (1.2345678f + i * 0.37f) * (i < 64 ? 1000.f : 0.01f);
    hipMemcpy(x, hx, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, o);
    unsigned ho[128]; hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 64; ++i) {
        __fp16 h[2], l[2]; memcpy(h, &ho[i], 4); memcpy(l, &ho[64 + i], 4);
        double ea = fabs((double)hx[i] - (double)h[0] - (double)l[0]) / fabs(hx[i]), eb = fabs((double)hx[64 + i] - (double)h[1] - (double)l[1]) / fabs(hx[64 + i]);
        if (ea > worst) worst = ea; if (eb > worst) worst = eb;
    }
    printf("worst relative residual %.3e (2^-20 = %.3e)\n", worst, 1.0 / (1 << 20));
    return 0;
}
