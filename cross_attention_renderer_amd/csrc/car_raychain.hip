// car_raychain.hip — the per-ray layers of the forward as two kernels (SURVEY.md §8a rows a15 (per-ray half), a17, a18; reference
// models.py:487, 548, 552, 561-565, 597-617 and resnet_block_fc.py:53-62, 132-168):
//   car_ray_mid  (after attention round 1):  z1 = Wv ebar1 + bv ;  uh = Wr1[:, :128] (We z1 + be)
//   car_ray_tail (after attention round 2):  z = (Wv ebar2 + bv) + V z1 ;  x = lin_in(coords) ;  3 x { x += lin_z_i([z, z]) ;
//                                            x += fc_1(relu(fc_0(relu(x)))) } ;  rgb = lin_out(relu(x)) valid + (1 - valid)
// instead of ~30 launches of car_linear over [rays, <= 576] matrices that are each too small to fill the chip.
//
// Mapping: weights are the A operand, rays the B operand, a wave owns 32 rays and keeps a layer's outputs in its accumulators: lane
// (ray s, half h) register r of tile T = channel 32 T + (r & 3) + 8 (r >> 2) + 4 h.  Those registers ARE the next layer's B operands
// (K step (T, kg) takes registers 8 kg .. 8 kg + 7) when the next layer's weights are packed in that K order ("chained") —
// activations never leave the register file between layers.  Arithmetic: the f16 matrix pipe with fp16 hi/lo operand halves, three
// v_mfma_f32_32x32x16_f16 products per term and fp32 accumulation (car_fused_mma.h: fp32-class accuracy, 5x the rate of the fp32
// pipe); a layer's weights carry a power of two chosen at pack time from its largest weight, its input vector one chosen per ray
// from the vector's largest magnitude, both undone exactly on the accumulators — which therefore also take the residual sums
// (x += ...) in true fp32.  A workgroup = 4 waves = 128 rays at one wave per SIMD (z, x and the residual branch need ~300
// registers); the weight chunks of all layers (K = 32 each) stream L2 -> LDS by LDS-DMA through a ring of three buffers (two chunks ahead of the products, across layer boundaries), in
// the order a host-built table lists them.
// lin_z_i sees [z, z] (the per-view replication of models.py:565, 605-606): its two 288-column halves are added once at pack time.
#include "car_common.h"
#include "car_geom.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kTileFloats = 1024;                  // packed floats per (chunk, tile): [kg (2)][hi | lo][lane (64)][8 halves]
constexpr int kMaxNT = 9;
constexpr int kBufFloats = kMaxNT * kTileFloats;   // one weight buffer: 36 KB
constexpr int kMaxChunks = 96;
constexpr int kMaxLayers = 16;

struct Chunk { unsigned off; int nt; };            // float offset of the chunk's tiles inside the weight arena, tiles in the chunk
struct ChainArgs {
    const float* arena;                            // every layer's packed tiles
    const float* bias;                             // biases of the layers, back to back in consumption order
    const float* scale;                            // [2 kMaxLayers]: 2^shift of every packed layer, then 2^-shift (car_chain_pack)
    int layer[kMaxLayers];                         // scale slot of the kernel's i-th layer
    Chunk chunk[kMaxChunks];
    int n_chunks;
    const float* x0; int ld0;                      // first layer's input rows (ebar), K0 = 576
    const float* x1; int ld1;                      // tail: decoder ray input (phi_x), K = 18
    const float* z1_in;                            // tail: z1 [rays, 288]
    float* out0;                                   // mid: z1 [rays, 288];  tail: rgb [rays, 3]
    float* out1;                                   // mid: uh [rays, 128];  tail: valid [rays]
    const CarRay* rays;                            // tail: overlaps of every view
    long M;                                        // rays
    int V, R;
    float zscale;                                  // tail: V
};

constexpr int kRing = 3;                           // weight buffers: the chunk being multiplied and the two behind it
struct Stream {
    const ChainArgs& a;
    float* lds;
    int g;
    __device__ __forceinline__ float* buffer(int gi) const { return lds + (gi % kRing) * kBufFloats; }
    __device__ __forceinline__ void issue(int gi, int tid, int wave) const {
        if (gi >= a.n_chunks) return;
        const Chunk c = a.chunk[gi];
        const float* src = a.arena + c.off;
        float* dst = buffer(gi);
        for (int t = 0; t < c.nt; ++t) {
            const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(dst + 4 * (t * 256 + wave * 64)));
            const float* gsrc = src + 4 * (t * 256 + tid);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
        }
    }
    // chunk gi has landed, in every wave's view: this wave's pieces of it are older than the one chunk issued behind it (gi + 1: a chunk is
    // nt loads per thread, nt in {1, 4, 9}), which may stay in flight.  Loads the compiler issued in between (a layer's input rows) only
    // make the count stricter; its own waits never know of these loads and are stricter for that
    __device__ __forceinline__ void landed(int gi) const {
        const int n = __builtin_amdgcn_readfirstlane(gi + 1 < a.n_chunks ? a.chunk[gi + 1].nt : 0);
        switch (n) {
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        __syncthreads();
    }
    // end of chunk g's products: chunk g + 1 must be in LDS (chunk g + 2, issued at the start of this step, may still fly), and every wave
    // is done with chunk g's buffer — which chunk g + 3 overwrites, issued at the start of the next step
    __device__ __forceinline__ void sync() const { landed(g + 1); }
};

#include "car_split.h"
template <int NT>
__device__ __forceinline__ void scale(f32x16 (&acc)[NT], float f) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] *= f;
}

// one chunk (32 values of K = two K steps) of a layer: acc[t] += W[tile t][chunk] . x, x8[kg] = this lane's 8 B-operand values of
// K step kg, already multiplied into the fp16 window by p
template <int NT>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[NT], const float* wl, const float (&x8)[2][8], float p) {
    half8 bhi[2], blo[2];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) split8(x8[kg], p, bhi[kg], blo[kg]);
    // one wave per SIMD: nobody else hides an LDS round trip, so the A operands of step (t, kg) + 1 are read BEFORE the three products of
    // step (t, kg) are issued (the compiler, left alone, reads them into the same two registers right in front of their first use and
    // waits: ~100 cycles exposed per 96 cycles of MFMA — the chains ran at a third of their matrix time)
    auto read = [&](int i, half8& ah, half8& al) {
        const int t = i >> 1, kg = i & 1;
        ah = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wl + t * kTileFloats + ((kg * 2 + 0) * 64) * 4));
        al = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(wl + t * kTileFloats + ((kg * 2 + 1) * 64) * 4));
    };
    half8 ah[2], al[2];
    read(0, ah[0], al[0]);
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
        const int t = i >> 1, kg = i & 1, cur = i & 1;
        if (i + 1 < 2 * NT) read(i + 1, ah[cur ^ 1], al[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bhi[kg], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], blo[kg], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], bhi[kg], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// acc += W x for a layer whose input rows come from global memory (standard K order: K step ks takes k = 16 ks + 8 h + e).  The row's
// largest magnitude (a pass over the row first: it comes back from L2) sets the ray's power of two; acc — already holding bias and
// whatever the layer adds to, in true units — is moved into the product's units and back, exactly.
template <int NT>
__device__ __forceinline__ void layer_global(f32x16 (&acc)[NT], const float* xrow, int K, int chunks, float dW, Stream& st, int tid, int lane,
                                             int wave) {
    const int h = lane >> 5;
    auto load = [&](int c, float (&x8)[2][8]) {
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            const int k0 = 32 * c + 16 * kg + 8 * h;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int kq = k0 + 4 * q;
                const float4 v = *reinterpret_cast<const float4*>(xrow + (kq + 4 <= K ? kq : 0));
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) x8[kg][4 * q + i] = kq + 4 <= K ? e[i] : (kq + i < K ? xrow[kq + i] : 0.0f);
            }
        }
    };
    float m = 0.0f;
    for (int c = 0; c < chunks; ++c) {
        float x8[2][8];
        load(c, x8);
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x8[kg][e]));
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float p, pinv;
    pow2_scale(fmaxf(m, 1e-30f), p, pinv);
    scale<NT>(acc, p / dW);
    float cur[2][8], nxt[2][8];
    load(0, cur);
    for (int c = 0; c < chunks; ++c) {
        st.issue(st.g + 2, tid, wave);
        if (c + 1 < chunks) load(c + 1, nxt);
        mma_chunk<NT>(acc, st.buffer(st.g) + 4 * lane, cur, p);
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int e = 0; e < 8; ++e) cur[kg][e] = nxt[kg][e];
        st.sync();
        ++st.g;
    }
    scale<NT>(acc, dW * pinv);
}
// acc += W act(src) for a layer whose input is the accumulator set of the previous layer (NSRC tiles of 32 channels), weights in the
// chained K order; scaling as above, the ray's power of two from the largest |act(src)|
template <int NSRC, int NT, bool RELU>
__device__ __forceinline__ void layer_chained(f32x16 (&acc)[NT], const f32x16 (&src)[NSRC], float dW, Stream& st, int tid, int lane, int wave) {
    float m = 0.0f;
#pragma unroll
    for (int T = 0; T < NSRC; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, RELU ? src[T][r] : fabsf(src[T][r]));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float p, pinv;
    pow2_scale(fmaxf(m, 1e-30f), p, pinv);
    scale<NT>(acc, p / dW);
#pragma unroll
    for (int T = 0; T < NSRC; ++T) {
        st.issue(st.g + 2, tid, wave);
        float x8[2][8];
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int e = 0; e < 8; ++e) x8[kg][e] = RELU ? fmaxf(src[T][8 * kg + e], 0.0f) : src[T][8 * kg + e];
        mma_chunk<NT>(acc, st.buffer(st.g) + 4 * lane, x8, p);
        st.sync();
        ++st.g;
    }
    scale<NT>(acc, dW * pinv);
}
template <int NT>
__device__ __forceinline__ void add_bias(f32x16 (&acc)[NT], const float* b, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += b[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
}
template <int NT>
__device__ __forceinline__ void zero(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}
// rows of N channels: lane (s, h) holds channels 32 t + 8 g + 4 h .. + 3 in acc[t][4 g .. 4 g + 3]
template <int NT>
__device__ __forceinline__ void store_rows(const f32x16 (&acc)[NT], float* row, int N, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = 32 * t + 8 * g + 4 * h;
            if (n0 + 4 <= N) *reinterpret_cast<float4*>(row + n0) = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
        }
}

// bias layout — mid: latent_value (288), encode_latent (128);  layers: 0 latent_value, 1 encode_latent, 2 query_repeat_embed[:, :128]
__global__ void __launch_bounds__(256, 1) ray_mid_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 31, h = lane >> 5;
    const long row = (long)blockIdx.x * 128 + wave * 32 + s;
    const long lrow = row < a.M ? row : a.M - 1;
    auto dW = [&](int i) { return a.scale[kMaxLayers + a.layer[i]]; };
    Stream st{a, lds, 0};
    st.issue(0, tid, wave);
    st.issue(1, tid, wave);
    st.landed(0);
    f32x16 z1[9];
    zero<9>(z1);
    add_bias<9>(z1, a.bias, h);
    layer_global<9>(z1, a.x0 + lrow * a.ld0, 576, 18, dW(0), st, tid, lane, wave);            // z1 = Wv ebar + bv
    if (row < a.M) store_rows<9>(z1, a.out0 + row * 288, 288, h);
    f32x16 hb[4];
    zero<4>(hb);
    add_bias<4>(hb, a.bias + 288, h);                                                         // encode_latent.bias
    layer_chained<9, 4, false>(hb, z1, dW(1), st, tid, lane, wave);
    f32x16 uh[4];
    zero<4>(uh);
    layer_chained<4, 4, false>(uh, hb, dW(2), st, tid, lane, wave);                           // Wr1[:, :128], no bias (it rides with the local half)
    if (row < a.M) store_rows<4>(uh, a.out1 + row * 128, 128, h);
}

// bias layout — tail: latent_value (288), lin_in (128), 3 x { lin_z, fc_0, fc_1 } (128 each), lin_out (padded to 32);
// layers: 0 latent_value, 1 lin_in, 2 + 3 i lin_z_i, 3 + 3 i fc_0_i, 4 + 3 i fc_1_i, 11 lin_out
__global__ void __launch_bounds__(256, 1) ray_tail_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 31, h = lane >> 5;
    const long row = (long)blockIdx.x * 128 + wave * 32 + s;
    const long lrow = row < a.M ? row : a.M - 1;
    auto dW = [&](int i) { return a.scale[kMaxLayers + a.layer[i]]; };
    Stream st{a, lds, 0};
    st.issue(0, tid, wave);
    st.issue(1, tid, wave);
    st.landed(0);
    // z = (Wv ebar2 + bv) + V z1   (models.py:561-565: "+ z_local" in every view, then the sum over the views)
    f32x16 z[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(a.z1_in + lrow * 288 + 32 * t + 8 * g + 4 * h);
            z[t][4 * g] = a.zscale * v.x; z[t][4 * g + 1] = a.zscale * v.y; z[t][4 * g + 2] = a.zscale * v.z; z[t][4 * g + 3] = a.zscale * v.w;
        }
    add_bias<9>(z, a.bias, h);
    layer_global<9>(z, a.x0 + lrow * a.ld0, 576, 18, dW(0), st, tid, lane, wave);
    // light-field decoder (resnet_block_fc.py:132-168)
    const float* bias = a.bias + 288;
    f32x16 x[4], net[4];
    zero<4>(x);
    add_bias<4>(x, bias, h);
    layer_global<4>(x, a.x1 + lrow * a.ld1, 18, 1, dW(1), st, tid, lane, wave);               // lin_in
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        add_bias<4>(x, bias + (3 * i + 1) * 128, h);
        layer_chained<9, 4, false>(x, z, dW(2 + 3 * i), st, tid, lane, wave);                 // x += lin_z_i([z, z])
        zero<4>(net);
        add_bias<4>(net, bias + (3 * i + 2) * 128, h);
        layer_chained<4, 4, true>(net, x, dW(3 + 3 * i), st, tid, lane, wave);                // net = fc_0(relu(x))
        add_bias<4>(x, bias + (3 * i + 3) * 128, h);
        layer_chained<4, 4, true>(x, net, dW(4 + 3 * i), st, tid, lane, wave);                // x += fc_1(relu(net))
    }
    f32x16 o[1];
    zero<1>(o);
    add_bias<1>(o, bias + 10 * 128, h);                                                       // lin_out.bias (padded to 32)
    layer_chained<4, 1, true>(o, x, dW(11), st, tid, lane, wave);
    // a18: rgb valid + (1 - valid), valid = any view's epipolar segment overlaps its image (models.py:614-617)
    if (row < a.M && h == 0) {
        const long sc = row / a.R, r = row % a.R;
        float ov = 0.0f;
        for (int v = 0; v < a.V; ++v) ov = fmaxf(ov, a.rays[(sc * a.V + v) * a.R + r].overlaps);
        const float valid = ov > 0.0f ? 1.0f : 0.0f;
        for (int k = 0; k < 3; ++k) a.out0[3 * row + k] = o[0][k] * valid + (1.0f - valid);
        a.out1[row] = valid;
    }
}

// scale[slot] = 2^shift with max |W (+ W2)| 2^shift in [2^13, 2^14), scale[kMaxLayers + slot] = 2^-shift.  One workgroup.
__global__ void chain_scale_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ W2, int K, int N, float* __restrict__ scale, int slot) {
    __shared__ float red[16];
    float m = 0.0f;
    for (long idx = threadIdx.x; idx < (long)N * K; idx += blockDim.x) {
        const long at = (idx / K) * ldw + idx % K;
        m = fmaxf(m, fabsf(W[at] + (W2 ? W2[at] : 0.0f)));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x / 64); ++w) m = fmaxf(m, red[w]);
        float p, inv;
        pow2_scale(fmaxf(m, 1e-30f), p, inv);
        scale[slot] = p;
        scale[kMaxLayers + slot] = inv;
    }
}
// [chunk = 32 input channels][tile = 32 outputs][kg][hi | lo][lane][8 halves]; lane l carries output 32 tile + l % 32 and
// chained: k = 32 chunk + (e & 3) + 8 (2 kg + (e >> 2)) + 4 (l >> 5) (the accumulator order of the layer before), else
// k = 32 chunk + 16 kg + 8 (l >> 5) + e; outputs >= N and inputs >= K are zero
__global__ void pack_chain_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ W2, int K, int N, int tiles, int chained,
                                  long total, const float* __restrict__ scale, int slot, _Float16* __restrict__ packed) {
    const float p = scale[slot];
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63), hl = (int)((idx >> 9) & 1), kg = (int)((idx >> 10) & 1);
        const long ct = idx >> 11;
        const int tile = (int)(ct % tiles), chunk = (int)(ct / tiles);
        const int n = 32 * tile + (lane & 31);
        const int k = chained ? 32 * chunk + (e & 3) + 8 * (2 * kg + (e >> 2)) + 4 * (lane >> 5) : 32 * chunk + 16 * kg + 8 * (lane >> 5) + e;
        float v = 0.0f;
        if (n < N && k < K) v = (W[(long)n * ldw + k] + (W2 ? W2[(long)n * ldw + k] : 0.0f)) * p;
        const _Float16 hi = (_Float16)v;
        packed[idx] = hl == 0 ? hi : (_Float16)(v - (float)hi);
    }
}

}  // namespace

// Split-fp16 tiles of one layer for the chain kernels: ceil(K/32) * ceil(N/32) * 1024 floats, [chunk][tile][kg][hi | lo][lane][8 halves].
// chained != 0: the layer's input is another layer's accumulator set (its K order); 0: input rows from memory.  W2 (optional, same
// shape and stride) is added element-wise (lin_z sees its latent twice).  The layer's power of two goes to scale[slot] (and its
// inverse to scale[16 + slot]): `scale` is a device array of 32 floats shared by the layers of a plan.
extern "C" size_t car_chain_packed_floats(int K, int N) { return (size_t)((K + 31) / 32) * ((N + 31) / 32) * kTileFloats; }
extern "C" int car_chain_pack(const float* W, int ldw, const float* W2, int K, int N, int chained, float* packed, float* scale, int slot,
                              void* stream) {
    CAR_REQUIRE(W && packed && scale && K > 0 && N > 0 && ldw >= K && slot >= 0 && slot < kMaxLayers, "car_chain_pack: bad arguments");
    const int tiles = (N + 31) / 32;
    const long total = (long)car_chain_packed_floats(K, N) * 2;              // halves
    (void)hipGetLastError();
    hipLaunchKernelGGL(chain_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, W, ldw, W2, K, N, scale, slot);
    hipLaunchKernelGGL(pack_chain_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, W, ldw, W2, K, N, tiles, chained, total, scale, slot,
                       reinterpret_cast<_Float16*>(packed));
    CAR_CHECK_LAUNCH("car_chain_pack");
    return CAR_OK;
}

namespace {
int launch_chain(bool tail, const float* arena, const unsigned* offs, const int* nts, int n_chunks, const float* bias, const float* scale,
                 const int* layers, int n_layers, const float* x0, int ld0, const float* x1, int ld1, const float* z1_in, float* out0, float* out1,
                 const float* rays, long M, int V, int R, float zscale, void* stream) {
    CAR_REQUIRE(n_chunks > 0 && n_chunks <= kMaxChunks && n_layers == (tail ? 12 : 3), "car_ray_chain: %d weight chunks, %d layers", n_chunks, n_layers);
    ChainArgs a;
    a.arena = arena; a.bias = bias; a.scale = scale; a.n_chunks = n_chunks;
    for (int i = 0; i < kMaxLayers; ++i) a.layer[i] = i < n_layers ? layers[i] : 0;
    for (int i = 0; i < n_layers; ++i) CAR_REQUIRE(layers[i] >= 0 && layers[i] < kMaxLayers, "car_ray_chain: bad scale slot");
    for (int i = 0; i < n_chunks; ++i) { a.chunk[i].off = offs[i]; a.chunk[i].nt = nts[i]; }
    a.x0 = x0; a.ld0 = ld0; a.x1 = x1; a.ld1 = ld1; a.z1_in = z1_in; a.out0 = out0; a.out1 = out1; a.rays = (const CarRay*)rays;
    a.M = M; a.V = V; a.R = R; a.zscale = zscale;
    const size_t lds_bytes = (size_t)kRing * kBufFloats * sizeof(float);
    auto kern = tail ? ray_tail_kernel : ray_mid_kernel;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { car_set_error("car_ray_chain: cannot reserve LDS: %s", hipGetErrorString(e)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(car_div_up(M, 128)), dim3(256), lds_bytes, (hipStream_t)stream, a);
    CAR_CHECK_LAUNCH("car_ray_chain");
    return CAR_OK;
}
}  // namespace

// `arena`: the layers' packed tiles; offs / nts (host arrays, n_chunks entries): float offset and tile count of every K = 32 chunk in
// the order the kernel consumes them; scale / layers: car_chain_pack's scale array and, per layer of the kernel in consumption order,
// its slot in it (car_render.hip builds all of it next to the arena).
extern "C" int car_ray_mid(const float* arena, const unsigned* offs, const int* nts, int n_chunks, const float* bias, const float* scale,
                           const int* layers, int n_layers, const float* ebar, int ld_ebar, float* z1, float* uh, long M, void* stream) {
    CAR_REQUIRE(arena && offs && nts && bias && scale && layers && ebar && z1 && uh && M > 0 && ld_ebar >= 576 && ld_ebar % 4 == 0,
                "car_ray_mid: bad arguments");
    return launch_chain(false, arena, offs, nts, n_chunks, bias, scale, layers, n_layers, ebar, ld_ebar, nullptr, 0, nullptr, z1, uh, nullptr, M, 0, 1,
                        0.0f, stream);
}
extern "C" int car_ray_tail(const float* arena, const unsigned* offs, const int* nts, int n_chunks, const float* bias, const float* scale,
                            const int* layers, int n_layers, const float* ebar, int ld_ebar, const float* phi_x, int ld_phi, const float* z1,
                            const float* rays, int b, int V, int R, float* rgb, float* valid, void* stream) {
    CAR_REQUIRE(arena && offs && nts && bias && scale && layers && ebar && phi_x && z1 && rays && rgb && valid, "car_ray_tail: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && R > 0 && ld_ebar >= 576 && ld_ebar % 4 == 0 && ld_phi >= 20 && ld_phi % 4 == 0, "car_ray_tail: bad sizes");
    return launch_chain(true, arena, offs, nts, n_chunks, bias, scale, layers, n_layers, ebar, ld_ebar, phi_x, ld_phi, z1, rgb, valid, rays, (long)b * R, V,
                        R, (float)V, stream);
}
