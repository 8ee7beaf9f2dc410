// car_split.h — the split-fp16 operand helpers shared by the matrix kernels (car_fused_mma.h, car_round2.hip, car_raychain.hip).
// Included INSIDE the including file's anonymous namespace, after its half8 typedef.
#pragma once

typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// x = hi + lo in fp16 halves for two values already scaled into fp16's window.  hi is rounded toward zero (v_cvt_pkrtz_f16_f32:
// two values per instruction); lo = fp16(x - hi) comes from v_fma_mixlo/mixhi_f16, which take hi's fp16 halves and the fp32 value
// directly: the difference is exact in fp32 and rounded once (to nearest) into the low half, |x - hi - lo| < 2^-21 |x|.  Three
// vector instructions per pair instead of six (every vector instruction in these loops costs matrix-pipe time: MFMA and ordinary
// vector instructions of different waves do not overlap on a gfx950 SIMD, profiles/round3_fused_experiments.md section 10).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(hi), "v"(a), "v"(b));
}
__device__ __forceinline__ void split8_scaled(const float (&x)[8], half8& hi, half8& lo) {
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned hh, ll; split_pair(x[2 * e], x[2 * e + 1], hh, ll); h[e] = hh; l[e] = ll; }
    hi = __builtin_bit_cast(half8, h);
    lo = __builtin_bit_cast(half8, l);
}
// x * p = hi + lo, p a power of two chosen by the caller so that the products stay inside fp16's normal range
__device__ __forceinline__ void split8(const float (&x)[8], float p, half8& hi, half8& lo) {
    float y[8];
    const f32x2 p2 = {p, p};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const f32x2 v = f32x2{x[e], x[e + 1]} * p2;                    // v_pk_mul_f32
        y[e] = v[0]; y[e + 1] = v[1];
    }
    split8_scaled(y, hi, lo);
}
// power of two p with m p in [2^13, 2^14) for m > 0 (exponent clamped for tiny / huge m), and inv = 1 / p
__device__ __forceinline__ void pow2_scale(float m, float& p, float& inv) {
    int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
    e = e < 97 ? 97 : (e > 230 ? 230 : e);          // p in [2^-90, 2^43]: an all-zero vector or matrix must not push p_x * p_W past fp32
    p = __uint_as_float((unsigned)(267 - e) << 23);
    inv = __uint_as_float((unsigned)(e - 13) << 23);
}
