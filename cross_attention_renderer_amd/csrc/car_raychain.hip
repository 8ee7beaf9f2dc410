// car_raychain.hip — the per-ray layers of the forward as two kernels (SURVEY.md §8a rows a15 (per-ray half), a17, a18; reference
// models.py:487, 548, 552, 561-565, 597-617 and resnet_block_fc.py:53-62, 132-168):
//   car_ray_mid  (after attention round 1):  z1 = Wv ebar1 + bv ;  uh = Wr1[:, :128] (We z1 + be)
//   car_ray_tail (after attention round 2):  z = (Wv ebar2 + bv) + V z1 ;  x = lin_in(coords) ;  3 x { x += lin_z_i([z, z]) ;
//                                            x += fc_1(relu(fc_0(relu(x)))) } ;  rgb = lin_out(relu(x)) valid + (1 - valid)
// instead of ~30 launches of car_linear over [rays, <= 576] matrices that are each too small to fill the chip.
//
// Mapping (exact fp32: v_mfma_f32_32x32x2_f32, bitwise an fmaf chain): weights are the A operand, rays the B operand, a wave owns 32
// rays and keeps a layer's outputs in its accumulators: lane (ray s, half h) register r of tile T = channel 32 T + (r & 3) +
// 8 (r >> 2) + 4 h.  Those registers ARE the next layer's B operands, one MFMA step per register, when the next layer's weights are
// packed in that K order ("chained", car_linear_pack_chained) — activations never leave the register file between layers.  A
// workgroup = 4 waves = 128 rays at one wave per SIMD (the fp32 matrix pipe is saturated by one wave; z, x and the residual branch
// need ~300 registers); the weight chunks of all layers (K = 32 each) stream L2 -> LDS by LDS-DMA, double buffered across layer
// boundaries, in the order a host-built table lists them.
// lin_z_i sees [z, z] (the per-view replication of models.py:565, 605-606): its two 288-column halves are added once at pack time.
#include "car_common.h"
#include "car_geom.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kTileFloats = 1024;                  // packed floats per (chunk, tile): [j4 (4)][lane (64)][e (4)], MFMA step r = 4 j4 + e
constexpr int kMaxNT = 9;
constexpr int kBufFloats = kMaxNT * kTileFloats;   // one weight buffer: 36 KB
constexpr int kMaxChunks = 96;

struct Chunk { unsigned off; int nt; };            // float offset of the chunk's tiles inside the weight arena, tiles in the chunk
struct ChainArgs {
    const float* arena;                            // every layer's packed tiles
    const float* bias;                             // biases of the chained layers, back to back
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

struct Stream {
    const ChainArgs& a;
    float* lds;
    int g;
    __device__ __forceinline__ void issue(int gi, int tid, int wave) const {
        if (gi >= a.n_chunks) return;
        const Chunk c = a.chunk[gi];
        const float* src = a.arena + c.off;
        float* dst = lds + (gi & 1) * kBufFloats;
        for (int t = 0; t < c.nt; ++t) {
            const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(dst + 4 * (t * 256 + wave * 64)));
            const float* gsrc = src + 4 * (t * 256 + tid);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
        }
    }
    __device__ __forceinline__ void sync() const {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
};

// one chunk (32 values of K) of a layer: acc[t] += W[tile t][chunk] . xe, xe[r] = this lane's B operand of MFMA step r
template <int NT>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[NT], const float* wl, const float (&xe)[16]) {
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 w = *reinterpret_cast<const float4*>(wl + (t * 4 + j4) * 256);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, xe[4 * j4 + 0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, xe[4 * j4 + 1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, xe[4 * j4 + 2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, xe[4 * j4 + 3], acc[t], 0, 0, 0);
        }
}

// layer whose input rows come from global memory in the standard K order of car_linear_pack (bias folded in as column K); the
// next chunk's 16 input values are fetched before the current chunk's MFMAs
template <int NT>
__device__ __forceinline__ void layer_global(f32x16 (&acc)[NT], const float* xrow, int ldx, int K, int chunks, Stream& st, int tid, int lane, int wave) {
    const int h = lane >> 5;
    auto load = [&](int c, float4 (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = 32 * c + 16 * h + 4 * q;
            v[q] = *reinterpret_cast<const float4*>(xrow + (k0 < ldx - 4 ? k0 : ldx - 4));      // clamped address: columns >= K are replaced in fix
        }
    };
    auto fix = [&](int c, const float4 (&v)[4], float (&xe)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = 32 * c + 16 * h + 4 * q;
            const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) xe[4 * q + i] = (k0 + i) < K ? e[i] : ((k0 + i) == K ? 1.0f : 0.0f);
        }
    };
    float4 cur[4], nxt[4];
    load(0, cur);
    for (int c = 0; c < chunks; ++c) {
        st.issue(st.g + 1, tid, wave);
        if (c + 1 < chunks) load(c + 1, nxt);
        float xe[16];
        fix(c, cur, xe);
        mma_chunk<NT>(acc, st.lds + (st.g & 1) * kBufFloats + 4 * lane, xe);
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        st.sync();
        ++st.g;
    }
}
// layer whose input is the accumulator set of the previous layer (NSRC tiles of 32 channels), weights in the chained K order
template <int NSRC, int NT, bool RELU>
__device__ __forceinline__ void layer_chained(f32x16 (&acc)[NT], const f32x16 (&src)[NSRC], Stream& st, int tid, int lane, int wave) {
#pragma unroll
    for (int T = 0; T < NSRC; ++T) {
        st.issue(st.g + 1, tid, wave);
        float xe[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) xe[r] = RELU ? fmaxf(src[T][r], 0.0f) : src[T][r];
        mma_chunk<NT>(acc, st.lds + (st.g & 1) * kBufFloats + 4 * lane, xe);
        st.sync();
        ++st.g;
    }
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

__global__ void __launch_bounds__(256, 1) ray_mid_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 31, h = lane >> 5;
    const long row = (long)blockIdx.x * 128 + wave * 32 + s;
    const long lrow = row < a.M ? row : a.M - 1;
    Stream st{a, lds, 0};
    st.issue(0, tid, wave);
    st.sync();
    f32x16 z1[9];
    zero<9>(z1);
    layer_global<9>(z1, a.x0 + lrow * a.ld0, a.ld0, 576, 19, st, tid, lane, wave);          // z1 = Wv ebar + bv (bias folded)
    if (row < a.M) store_rows<9>(z1, a.out0 + row * 288, 288, h);
    f32x16 hb[4];
    zero<4>(hb);
    add_bias<4>(hb, a.bias, h);                                                               // encode_latent.bias
    layer_chained<9, 4, false>(hb, z1, st, tid, lane, wave);
    f32x16 uh[4];
    zero<4>(uh);
    layer_chained<4, 4, false>(uh, hb, st, tid, lane, wave);                                  // Wr1[:, :128], no bias (it rides with the local half)
    if (row < a.M) store_rows<4>(uh, a.out1 + row * 128, 128, h);
}

__global__ void __launch_bounds__(256, 1) ray_tail_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 31, h = lane >> 5;
    const long row = (long)blockIdx.x * 128 + wave * 32 + s;
    const long lrow = row < a.M ? row : a.M - 1;
    Stream st{a, lds, 0};
    st.issue(0, tid, wave);
    st.sync();
    // z = (Wv ebar2 + bv) + V z1   (models.py:561-565: "+ z_local" in every view, then the sum over the views)
    f32x16 z[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(a.z1_in + lrow * 288 + 32 * t + 8 * g + 4 * h);
            z[t][4 * g] = a.zscale * v.x; z[t][4 * g + 1] = a.zscale * v.y; z[t][4 * g + 2] = a.zscale * v.z; z[t][4 * g + 3] = a.zscale * v.w;
        }
    layer_global<9>(z, a.x0 + lrow * a.ld0, a.ld0, 576, 19, st, tid, lane, wave);
    // light-field decoder (resnet_block_fc.py:132-168)
    f32x16 x[4], net[4];
    zero<4>(x);
    layer_global<4>(x, a.x1 + lrow * a.ld1, a.ld1, 18, 1, st, tid, lane, wave);              // lin_in (bias folded)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        add_bias<4>(x, a.bias + (3 * i + 0) * 128, h);
        layer_chained<9, 4, false>(x, z, st, tid, lane, wave);                                // x += lin_z_i([z, z])
        zero<4>(net);
        add_bias<4>(net, a.bias + (3 * i + 1) * 128, h);
        layer_chained<4, 4, true>(net, x, st, tid, lane, wave);                               // net = fc_0(relu(x))
        add_bias<4>(x, a.bias + (3 * i + 2) * 128, h);
        layer_chained<4, 4, true>(x, net, st, tid, lane, wave);                               // x += fc_1(relu(net))
    }
    f32x16 o[1];
    zero<1>(o);
    add_bias<1>(o, a.bias + 9 * 128, h);                                                      // lin_out.bias (padded to 32)
    layer_chained<4, 1, true>(o, x, st, tid, lane, wave);
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

__global__ void pack_chained_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ W2, int K, int N, int tiles, long total,
                                    float* __restrict__ packed) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63), j4 = (int)((idx >> 8) & 3);
        const long ct = idx >> 10;
        const int tile = (int)(ct % tiles), chunk = (int)(ct / tiles);
        const int n = 32 * tile + (lane & 31), r = 4 * j4 + e;
        const int k = 32 * chunk + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = 0.0f;
        if (n < N && k < K) v = W[(long)n * ldw + k] + (W2 ? W2[(long)n * ldw + k] : 0.0f);
        packed[idx] = v;
    }
}

}  // namespace

// Weights of a layer whose input is another layer's accumulator set: [chunk = 32 input channels][tile = 32 outputs][j4][lane][e] with
// k = 32 chunk + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), r = 4 j4 + e, no bias column.  W2 (optional, same shape and stride) is added
// element-wise (lin_z sees its latent twice).  ceil(K/32) * ceil(N/32) * 1024 floats.
extern "C" size_t car_chain_packed_floats(int K, int N) { return (size_t)((K + 31) / 32) * ((N + 31) / 32) * kTileFloats; }
extern "C" int car_chain_pack(const float* W, int ldw, const float* W2, int K, int N, float* packed, void* stream) {
    CAR_REQUIRE(W && packed && K > 0 && N > 0 && ldw >= K, "car_chain_pack: bad arguments");
    const int tiles = (N + 31) / 32;
    const long total = (long)car_chain_packed_floats(K, N);
    (void)hipGetLastError();
    hipLaunchKernelGGL(pack_chained_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, W, ldw, W2, K, N, tiles, total, packed);
    CAR_CHECK_LAUNCH("car_chain_pack");
    return CAR_OK;
}

namespace {
int launch_chain(bool tail, const float* arena, const unsigned* offs, const int* nts, int n_chunks, const float* bias, const float* x0, int ld0,
                 const float* x1, int ld1, const float* z1_in, float* out0, float* out1, const float* rays, long M, int V, int R, float zscale,
                 void* stream) {
    CAR_REQUIRE(n_chunks > 0 && n_chunks <= kMaxChunks, "car_ray_chain: %d weight chunks", n_chunks);
    ChainArgs a;
    a.arena = arena; a.bias = bias; a.n_chunks = n_chunks;
    for (int i = 0; i < n_chunks; ++i) { a.chunk[i].off = offs[i]; a.chunk[i].nt = nts[i]; }
    a.x0 = x0; a.ld0 = ld0; a.x1 = x1; a.ld1 = ld1; a.z1_in = z1_in; a.out0 = out0; a.out1 = out1; a.rays = (const CarRay*)rays;
    a.M = M; a.V = V; a.R = R; a.zscale = zscale;
    const size_t lds_bytes = 2 * kBufFloats * sizeof(float);
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
// the order the kernel consumes them (car_render.hip builds them next to the arena).
extern "C" int car_ray_mid(const float* arena, const unsigned* offs, const int* nts, int n_chunks, const float* bias, const float* ebar, int ld_ebar,
                           float* z1, float* uh, long M, void* stream) {
    CAR_REQUIRE(arena && offs && nts && bias && ebar && z1 && uh && M > 0 && ld_ebar >= 576 && ld_ebar % 4 == 0, "car_ray_mid: bad arguments");
    return launch_chain(false, arena, offs, nts, n_chunks, bias, ebar, ld_ebar, nullptr, 0, nullptr, z1, uh, nullptr, M, 0, 1, 0.0f, stream);
}
extern "C" int car_ray_tail(const float* arena, const unsigned* offs, const int* nts, int n_chunks, const float* bias, const float* ebar, int ld_ebar,
                            const float* phi_x, int ld_phi, const float* z1, const float* rays, int b, int V, int R, float* rgb, float* valid,
                            void* stream) {
    CAR_REQUIRE(arena && offs && nts && bias && ebar && phi_x && z1 && rays && rgb && valid, "car_ray_tail: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && R > 0 && ld_ebar >= 576 && ld_ebar % 4 == 0 && ld_phi >= 18 && ld_phi % 4 == 0, "car_ray_tail: bad sizes");
    return launch_chain(true, arena, offs, nts, n_chunks, bias, ebar, ld_ebar, phi_x, ld_phi, z1, rgb, valid, rays, (long)b * R, V, R, (float)V, stream);
}
