// pk_fma_probe.hip — development check (not product): does v_pk_fma_f32 with a broadcast source (op_sel_hi 0) tolerate a destination that
// overlaps that source?  (It does: identical results.)  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_fma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* out) {
    const int t = threadIdx.x;
    f32x2 a = {x[t], x[t + 64]}, b = {x[t + 128], x[t + 192]}, c = {x[t + 256], x[t + 320]};
    f32x2 d0, d1 = b;
    // reference: distinct destination
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=&v"(d0) : "v"(a), "v"(b), "v"(c));
    // destination = the broadcast source
    asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[1,0,1]" : "+v"(d1) : "v"(a), "v"(c));
    out[t] = d0[0]; out[t + 64] = d0[1]; out[t + 128] = d1[0]; out[t + 192] = d1[1];
}
int main() {
    float *x, *o; hipMalloc(&x, 384 * 4); hipMalloc(&o, 256 * 4);
    float hx[384]; for (int i = 0; i < 384; ++i) hx[i] = 0.5f + 0.013f * i;
    hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, o);
    float ho[256]; hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    int bad_lo = 0, bad_hi = 0;
    for (int t = 0; t < 64; ++t) { bad_lo += ho[t] != ho[t + 128]; bad_hi += ho[t + 64] != ho[t + 192]; }
    printf("v_pk_fma_f32 dst == broadcast source: low halves differing %d / 64, high halves differing %d / 64 (lane 0: %g vs %g; expected a.hi*b.lo+c.hi = %g, with overwritten b.lo: %g)\n",
           bad_lo, bad_hi, ho[64], ho[192], hx[64] * hx[128] + hx[320], hx[64] * (hx[0] * hx[128] + hx[256]) + hx[320]);
    return 0;
}
