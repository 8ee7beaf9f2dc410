// car_render.hip — the one-call forward of the C ABI (include/car_hip.h: car_plan_*, car_project_maps, car_render_forward).
//
// Host-side C++ (plus small re-layout / reduction kernels): it carves the caller's plan / workspace buffers, packs every layer
// into the operand order of the kernels that consume it, and issues the launch sequence of the default configuration
// (CrossAttentionRenderer(model="midas_vit", n_view=2), reference models.py:190-626) from plain pointers.  This IS the product
// path: cross_attention_renderer_amd/engine.py calls it for that configuration (its own stage-by-stage sequence covers the
// constructor variants and serves as the A/B partner in tests/test_hip_parity.py).
#include "car_common.h"
#include "car_geom.h"
#include <math.h>
#include <string.h>
#include <vector>

extern "C" size_t car_fused_blob_floats(void);
extern "C" size_t car_fused_bias_floats(void);
extern "C" int car_fused_tile_steps(void);
extern "C" size_t car_round2_packed_floats(void);
extern "C" size_t car_round2_bias_floats(void);
extern "C" size_t car_round2q_packed_floats(void);
extern "C" size_t car_round2q_bias_floats(void);
extern "C" size_t car_chain_packed_floats(int K, int N);
extern "C" int car_chain_pack(const float* W, int ldw, const float* W2, int K, int N, int chained, float* packed, float* scale, int slot,
                              void* stream);
extern "C" int car_ray_mid(const float* arena, const unsigned* offs, const int* nts, int n_chunks, const float* bias, const float* scale,
                           const int* layers, int n_layers, const float* ebar, int ld_ebar, float* z1, float* uh, long M, void* stream);
extern "C" int car_ray_tail(const float* arena, const unsigned* offs, const int* nts, int n_chunks, const float* bias, const float* scale,
                            const int* layers, int n_layers, const float* ebar, int ld_ebar, const float* phi_x, int ld_phi, const float* z1,
                            const float* rays, int b, int V, int R, float* rgb, float* valid, void* stream);

namespace {

#include "car_fused_layout.h"

constexpr int kPhiIn = 18, kPhiLd = 20, kBlocks = 3;
constexpr int kTile16 = kTile;                     // floats per (K step, 16-channel tile) of the fused kernel's blob

inline size_t up64(size_t x) { return (x + 63) & ~(size_t)63; }

// power of two p with m p in [2^13, 2^14) (the window of the split-fp16 operands, car_fused_mma.h)
__device__ __forceinline__ float pow2_for(float m) {
    int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
    e = e < 97 ? 97 : (e > 230 ? 230 : e);          // p in [2^-90, 2^43]: an all-zero vector or matrix must not push p_x * p_W past fp32
    return __uint_as_float((unsigned)(267 - e) << 23);
}
__device__ __forceinline__ float block_max(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = 0.0f;
    for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) m = fmaxf(m, red[w]);
    __syncthreads();
    return m;
}

// ---- re-layout kernels ---------------------------------------------------------------------------------------------
// Scale of a packed layer: p = 2^shift from the largest |weight| (and |bias| where the bias is folded in as a column);
// p goes to p_slot (read by the pack kernels), 1/p to down_slot (read by the consuming kernel).  One workgroup.
__global__ void layer_scale_kernel(const float* __restrict__ W, int ldw, int N, int K, const float* __restrict__ bias,
                                   float* __restrict__ p_slot, float* __restrict__ down_slot) {
    __shared__ float red[16];
    float m = 0.0f;
    for (long idx = threadIdx.x; idx < (long)N * K; idx += blockDim.x) m = fmaxf(m, fabsf(W[(idx / K) * ldw + idx % K]));
    if (bias) for (int n = threadIdx.x; n < N; n += blockDim.x) m = fmaxf(m, fabsf(bias[n]));
    m = block_max(m, red);
    if (threadIdx.x == 0) {
        const float p = pow2_for(fmaxf(m, 1e-30f));
        p_slot[0] = p;
        down_slot[0] = 1.0f / p;
    }
}
// A-operand tiles of v_mfma_f32_16x16x32_f16 with fp16 hi/lo halves: per (K step, tile) [hi|lo][lane][8 halves]; lane l carries
// output 16 t + l % 16 and k = 32 ks + 8 (l >> 4) + e (mode 0) or the accumulator order base + 16 (2 ks + e / 4) + 4 (l >> 4) + e % 4
// (mode 1); k == K selects the bias, k > K a zero.  Values are multiplied by the layer's power of two *p_slot.
__global__ void pack16_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ bias, int N, int K, int n_tiles,
                              int ksteps, int mode, int base, const float* __restrict__ p_slot, _Float16* __restrict__ out) {
    const long total = (long)ksteps * n_tiles * 512;
    const float p = p_slot[0];
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long tile = idx >> 9;
        const int t = (int)(tile % n_tiles), ks = (int)(tile / n_tiles);
        const int n = 16 * t + (lane & 15), q = lane >> 4;
        const int k = mode == 0 ? 32 * ks + 8 * q + e : base + 16 * (2 * ks + e / 4) + 4 * q + e % 4;
        float w = 0.0f;
        if (n < N) {
            if (k < K) w = W[(long)n * ldw + k] * p;
            else if (k == K && bias) w = bias[n] * p;
        }
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        _Float16* o = out + tile * 1024 + lane * 8 + e;
        o[0] = hi;
        o[512] = lo;
    }
}
// <Wa r + ba, Wb x + bb> = r^T (M x + v) + u^T x + c for two 128-wide layers that are only ever dotted with each other (the first round's
// key_map_2 / query_embed_2, the second round's query_repeat_embed_2 / query_embed_2; models.py:491, 529, 533, 553-556):
//     M[i][j] = sum_k Wa[k][i] Wb[k][j],  v[i] = sum_k Wa[k][i] bb[k],  u[j] = sum_k Wb[k][j] ba[k],  c = sum_k ba[k] bb[k]
// accumulated in fp64 in ascending k (every product of two fp32 values is exact there) and rounded once to fp32.  One workgroup per row i of M.
__global__ void bilinear_fold_kernel(const float* __restrict__ Wa, const float* __restrict__ ba, const float* __restrict__ Wb,
                                     const float* __restrict__ bb, int D, float* __restrict__ M, float* __restrict__ v, float* __restrict__ u,
                                     float* __restrict__ c) {
    const int i = blockIdx.x, j = threadIdx.x;
    if (j >= D) return;
    double m = 0.0;
    for (int k = 0; k < D; ++k) m += (double)Wa[k * D + i] * (double)Wb[k * D + j];
    M[i * D + j] = (float)m;
    if (j == 0) {
        double s = 0.0;
        for (int k = 0; k < D; ++k) s += (double)Wa[k * D + i] * (double)bb[k];
        v[i] = (float)s;
    }
    if (i == 0) {
        double s = 0.0;
        for (int k = 0; k < D; ++k) s += (double)Wb[k * D + j] * (double)ba[k];
        u[j] = (float)s;
        if (j == 0) {
            double t = 0.0;
            for (int k = 0; k < D; ++k) t += (double)ba[k] * (double)bb[k];
            c[0] = (float)t;
        }
    }
}
// A-operand tiles of v_mfma_f32_32x32x16_f16 for car_round2.hip: [chunk][tile 4][K group kgs][hi|lo][lane][8 halves], output
// 32 t + l % 32; chained = 1: k = 32 c + (e & 3) + 8 (2 kg + (e >> 2)) + 4 (l >> 5) (the accumulator order of the layer before),
// chained = 0: k = 16 c + 8 (l >> 5) + e with one K group per chunk.
__global__ void pack32_kernel(const float* __restrict__ W, int ldw, int chunks, int kgs, int chained, const float* __restrict__ p_slot,
                              _Float16* __restrict__ out) {
    const int total = chunks * 4 * kgs * 2 * 64 * 8;
    const float p = p_slot[0];
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, hl = (idx >> 9) & 1;
        int rest = idx >> 10;
        const int kg = rest % kgs; rest /= kgs;
        const int t = rest & 3, c = rest >> 2;
        const int n = 32 * t + (lane & 31);
        const int k = chained ? 32 * c + (e & 3) + 8 * (2 * kg + (e >> 2)) + 4 * (lane >> 5) : 16 * c + 8 * (lane >> 5) + e;
        const float w = W[n * ldw + k] * p;
        const _Float16 hi = (_Float16)w;
        out[idx] = hl == 0 ? hi : (_Float16)(w - (float)hi);
    }
}
// [C][4] table (W1[:, C:C+3], b1) of the per-texel first layer, and the largest row sum of magnitudes (bounds the point / bias
// term of h because |tanh| <= 1).  One workgroup of kC threads.
__global__ void wpt_kernel(const float* __restrict__ w1, const float* __restrict__ b1, float* __restrict__ wpt, float* __restrict__ bound) {
    __shared__ float red[16];
    const int ch = threadIdx.x;
    float m = 0.0f;
    if (ch < kC) {
        const float x = w1[(long)ch * (kC + 3) + kC + 0], y = w1[(long)ch * (kC + 3) + kC + 1], z = w1[(long)ch * (kC + 3) + kC + 2], b = b1[ch];
        wpt[4 * ch + 0] = x; wpt[4 * ch + 1] = y; wpt[4 * ch + 2] = z; wpt[4 * ch + 3] = b;
        m = ((fabsf(x) + fabsf(y)) + fabsf(z)) + fabsf(b);
    }
    m = block_max(m, red);
    if (threadIdx.x == 0) bound[0] = m;
}
// The merged lattice (car_geom.h car_lattice_taps): node (jy, jx) of map (m, mode) = sum over the levels of the bilinear
// interpolation of the projected level G_l[m] at lattice coordinate u = j - pad, i.e. at texel coordinate (u + 1 - r_l) / (2 r_l) of
// a level r_l times coarser than the finest, with the level's own padding rule (mode 0 border, 1 zeros).
// A 16-lane group owns one node of one map and writes BOTH padding modes of it: the taps and weights of every level are worked out
// once per node (per 16-lane group, not per float4 of channels: that arithmetic used to be most of the kernel's time); a source row
// both modes read with a non-zero weight — every tap of an interior node — is loaded once; taps of weight zero (three of four on
// the finest level's own texel centres) get an out-of-range buffer offset: the load returns zeros without touching memory, and an
// instruction whose lanes are all out of range costs the texture path nothing (profiles/round3_fused_experiments.md).  No branch
// per tap, so a whole channel step's loads are in flight together.  Lane `sub` takes the channel quads sub + 16 j: a load / store
// instruction of the group moves 256 contiguous bytes.  The sums run in the order the one-mode kernel used (levels from the last
// index down, taps nw ne sw se, one fused multiply-add each; a skipped tap had weight zero), so the values are the same up to the
// sign of an exact zero.
// gmax (optional): the largest |lattice value| goes there (atomic max of the bit pattern: non-negative floats order like integers; the
// caller zeroes it) — one atomic per workgroup of a grid-stride launch; it bounds h in the fused kernel's fp16 split.
struct MergeArgs {
    const float* g[CAR_MAX_LEVELS];   // level l of the launch's first map
    unsigned bytes[CAR_MAX_LEVELS];   // the launch's maps of level l: range of the buffer loads (< 2 GiB, car_project_maps slices the maps)
    int h[CAR_MAX_LEVELS], w[CAR_MAX_LEVELS], r[CAR_MAX_LEVELS];
    int n_levels, lh, lw, pad;
    long nodes;                       // maps of the launch * lh * lw
    long per;                         // consecutive nodes per workgroup
    int ny;                           // lattice rows a workgroup's nodes can span (its y-axis table)
    float* lat;                       // [maps][2][lh][lw][kC] of the launch's first map
};
typedef float mf32x4 __attribute__((ext_vector_type(4)));
typedef float mf32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned kNoTap = 0xc0000000u;          // beyond any sliced level: the load returns zeros
// Level l's four taps at node (jx, jy) of the launch's map m, for the lane that owns channel quad `sub`: ob = byte offset of the load
// that serves border mode — or, where border mode's weight is zero and zeros mode's is not, zeros mode's texel (e.g. on the ring just
// outside the map) — kNoTap when neither needs it; wb the border weight of what that load returns, ws the zeros weight.  Should both
// modes ever need DIFFERENT texels for one tap, `second` is set and (oz, wn) describe the extra load of the slow path.
struct LevelTaps { unsigned ob[4], oz[4]; float wb[4], ws[4], wn[4]; bool second; };
// One axis of car_bilinear_taps_px (car_geom.h): the two clamped texel indices of texel coordinate i and their weights, a weight forced to
// zero where its texel lies outside the level.  The 2-D weights are the products of the two axes' — the products car_bilinear_taps_px forms,
// zero exactly where it masks — so a node's taps come from one entry per axis, level and padding mode: 2 (lw + rows) entries per level
// instead of a page of arithmetic per node.
struct AxisTap { int c0, c1; float w0, w1; };
__device__ __forceinline__ AxisTap axis_tap(float i, int W, int mode) {
    if (mode == 0) i = fminf(fmaxf(i, 0.0f), (float)(W - 1));
    if (!(i > -4.0f)) i = -4.0f;
    if (i > (float)W + 4.0f) i = (float)W + 4.0f;
    const float f0 = floorf(i), f1 = f0 + 1.0f;
    const int x0 = (int)f0, x1 = x0 + 1;
    AxisTap t;
    t.w0 = (x0 >= 0 && x0 < W) ? f1 - i : 0.0f;
    t.w1 = (x1 >= 0 && x1 < W) ? i - f0 : 0.0f;
    t.c0 = x0 < 0 ? 0 : (x0 >= W ? W - 1 : x0);
    t.c1 = x1 < 0 ? 0 : (x1 >= W ? W - 1 : x1);
    return t;
}
__device__ __forceinline__ float lattice_to_texel(int j, int pad, int r) { return (float)(j - pad + 1 - r) / (float)(2 * r); }
__device__ __forceinline__ float4 axis_entry(const AxisTap& t) { return make_float4(__int_as_float(t.c0), __int_as_float(t.c1), t.w0, t.w1); }
__device__ __forceinline__ AxisTap axis_of(const float4& e) { return AxisTap{__float_as_int(e.x), __float_as_int(e.y), e.z, e.w}; }
// the level's four taps of both padding modes (xb / yb: border, xz / yz: zeros) -> what the node's loads and sums need
__device__ __forceinline__ LevelTaps level_taps(const AxisTap& xb, const AxisTap& yb, const AxisTap& xz, const AxisTap& yz, int W, unsigned mbase, int sub) {
    LevelTaps T;
    const int tb[4] = {yb.c0 * W + xb.c0, yb.c0 * W + xb.c1, yb.c1 * W + xb.c0, yb.c1 * W + xb.c1};
    const int tz[4] = {yz.c0 * W + xz.c0, yz.c0 * W + xz.c1, yz.c1 * W + xz.c0, yz.c1 * W + xz.c1};
    const float wz[4] = {xz.w0 * yz.w0, xz.w1 * yz.w0, xz.w0 * yz.w1, xz.w1 * yz.w1};
    T.wb[0] = xb.w0 * yb.w0; T.wb[1] = xb.w1 * yb.w0; T.wb[2] = xb.w0 * yb.w1; T.wb[3] = xb.w1 * yb.w1;
    T.second = false;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool on0 = T.wb[t] != 0.0f, on1 = wz[t] != 0.0f;
        const bool borrow = on1 && !on0, second = on1 && on0 && tz[t] != tb[t];
        T.ob[t] = (on0 || borrow) ? (mbase + (unsigned)(borrow ? tz[t] : tb[t])) * (unsigned)(kC * 4) + 16u * sub : kNoTap;
        T.oz[t] = second ? (mbase + (unsigned)tz[t]) * (unsigned)(kC * 4) + 16u * sub : kNoTap;
        T.ws[t] = (on1 && !second) ? wz[t] : 0.0f;
        T.wn[t] = second ? wz[t] : 0.0f;
        T.second = T.second || second;
    }
    return T;
}
__device__ __forceinline__ void merge_fma(float w, const mf32x4& v, mf32x2& lo, mf32x2& hi) {
    const mf32x2 w2 = {w, w};
    lo = __builtin_elementwise_fma(w2, mf32x2{v[0], v[1]}, lo);
    hi = __builtin_elementwise_fma(w2, mf32x2{v[2], v[3]}, hi);
}
// TAB: a workgroup owns `per` consecutive nodes (a few lattice rows) and keeps the axis entries of its columns and rows in LDS;
// otherwise (lattices too wide for that) the nodes are dealt out 16 at a time and every node works its entries out itself.
template <int NL, bool TAB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) merge_kernel(const MergeArgs a, unsigned* __restrict__ gmax) {
    __shared__ float red[4];
    extern __shared__ __attribute__((aligned(16))) float4 tab[];       // TAB: x axis [NL][2 modes][lw], then y axis [NL][2][a.ny]
    const int sub = threadIdx.x & 15;
    const long plane = (long)a.lh * a.lw;
    __amdgpu_buffer_rsrc_t rs[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) rs[l] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g[l]), 0, (int)a.bytes[l], 0x00027000);
    const long first = TAB ? (long)blockIdx.x * a.per : (long)blockIdx.x * 16;
    const long last = TAB ? (first + a.per < a.nodes ? first + a.per : a.nodes) : a.nodes;
    const long stride = TAB ? 16 : (long)gridDim.x * 16;
    const long row0 = first / a.lw;                                   // first lattice row (counted through the maps) of this workgroup
    float4* ytab = tab + NL * 2 * a.lw;
    if constexpr (TAB) {
        for (int i = threadIdx.x; i < NL * 2 * a.lw; i += 256) {
            const int l = i / (2 * a.lw), md = (i / a.lw) & 1, jx = i % a.lw;
            tab[i] = axis_entry(axis_tap(lattice_to_texel(jx, a.pad, a.r[l]), a.w[l], md));
        }
        for (int i = threadIdx.x; i < NL * 2 * a.ny; i += 256) {
            const int l = i / (2 * a.ny), md = (i / a.ny) & 1, jy = (int)((row0 + i % a.ny) % a.lh);
            ytab[i] = axis_entry(axis_tap(lattice_to_texel(jy, a.pad, a.r[l]), a.h[l], md));
        }
        __syncthreads();
    }
    float mx = 0.0f;
    // Tried and dropped (round 5, tools/bench_merge.py): a wave's four nodes 8 apart, so that they agree on which taps carry weight zero and a
    // dead tap is an instruction whose lanes are ALL out of range: 1.08 -> 1.5 ms (the wave's stores then fall on four distant rows); 64 lanes
    // per node (1 KB contiguous per load / store instruction, the third step a quarter full): 1.15 -> 1.46 ms = the extra instructions.
    for (long node = first + (threadIdx.x >> 4); node < last; node += stride) {
        const long row = node / a.lw;
        const int m = (int)(row / a.lh);
        const int jy = (int)(row - (long)m * a.lh), jx = (int)(node - row * a.lw);
        auto taps_of = [&](int l) {
            const unsigned mbase = (unsigned)m * (unsigned)(a.h[l] * a.w[l]);
            if constexpr (TAB) {
                const int ry = (int)(row - row0);
                return level_taps(axis_of(tab[(l * 2 + 0) * a.lw + jx]), axis_of(ytab[(l * 2 + 0) * a.ny + ry]), axis_of(tab[(l * 2 + 1) * a.lw + jx]),
                                  axis_of(ytab[(l * 2 + 1) * a.ny + ry]), a.w[l], mbase, sub);
            } else {
                const float ix = lattice_to_texel(jx, a.pad, a.r[l]), iy = lattice_to_texel(jy, a.pad, a.r[l]);
                return level_taps(axis_tap(ix, a.w[l], 0), axis_tap(iy, a.h[l], 0), axis_tap(ix, a.w[l], 1), axis_tap(iy, a.h[l], 1), a.w[l], mbase, sub);
            }
        };
        unsigned ob[NL][4];
        float wb[NL][4], ws[NL][4];
        bool second = false;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const LevelTaps T = taps_of(l);
#pragma unroll
            for (int t = 0; t < 4; ++t) { ob[l][t] = T.ob[t]; wb[l][t] = T.wb[t]; ws[l][t] = T.ws[t]; }
            second = second || T.second;
        }
        float* out0 = a.lat + (((long)m * 2 + 0) * plane + (long)jy * a.lw + jx) * kC + 4 * sub;
        float* out1 = out0 + plane * kC;
        const bool slow = __builtin_amdgcn_ballot_w64(second) != 0;   // wave-uniform
        // three channel steps per trip: a trip's loads (up to 36 per lane) are in flight together
#pragma unroll 3
        for (int j = 0; j < kC / 64; ++j) {
            mf32x2 a0l = {0.f, 0.f}, a0h = {0.f, 0.f}, a1l = {0.f, 0.f}, a1h = {0.f, 0.f};
            if (!slow) {                                               // one load per live tap serves both modes; every level's in flight together
                mf32x4 v[NL][4];
#pragma unroll
                for (int l = 0; l < NL; ++l)
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[l][t] = __builtin_bit_cast(mf32x4, __builtin_amdgcn_raw_buffer_load_b128(rs[l], (int)ob[l][t], 256 * j, 0));
#pragma unroll
                for (int l = NL - 1; l >= 0; --l)
#pragma unroll
                    for (int t = 0; t < 4; ++t) { merge_fma(wb[l][t], v[l][t], a0l, a0h); merge_fma(ws[l][t], v[l][t], a1l, a1h); }
            } else {                                                   // never seen with the two padding rules of grid_sample; kept for safety
#pragma unroll 1
                for (int l = NL - 1; l >= 0; --l) {
                    const LevelTaps T = taps_of(l);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const mf32x4 v = __builtin_bit_cast(mf32x4, __builtin_amdgcn_raw_buffer_load_b128(rs[l], (int)T.ob[t], 256 * j, 0));
                        const mf32x4 u = __builtin_bit_cast(mf32x4, __builtin_amdgcn_raw_buffer_load_b128(rs[l], (int)T.oz[t], 256 * j, 0));
                        merge_fma(T.wb[t], v, a0l, a0h); merge_fma(T.ws[t], v, a1l, a1h); merge_fma(T.wn[t], u, a1l, a1h);
                    }
                }
            }
            *reinterpret_cast<float4*>(out0 + 64 * j) = make_float4(a0l[0], a0l[1], a0h[0], a0h[1]);
            *reinterpret_cast<float4*>(out1 + 64 * j) = make_float4(a1l[0], a1l[1], a1h[0], a1h[1]);
            mx = fmaxf(fmaxf(mx, fmaxf(fmaxf(fabsf(a0l[0]), fabsf(a0l[1])), fmaxf(fabsf(a0h[0]), fabsf(a0h[1])))),
                       fmaxf(fmaxf(fabsf(a1l[0]), fabsf(a1l[1])), fmaxf(fabsf(a1h[0]), fabsf(a1h[1]))));
        }
    }
    if (gmax) {
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(gmax, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    }
}
// as many workgroups as the chip holds at once (three 4-wave workgroups per compute unit at this kernel's 168 registers: a grid that
// needs a partial second helping of workgroups per compute unit ends on a half-empty chip)
inline long merge_resident_blocks() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return 3L * cus;
}
// the merge over n_maps maps, in launches whose widest level stays below the 2 GiB a buffer load addresses
int launch_merge(const float* const* levels, const int* hs, const int* ws, const int* rs, int n_levels, int lh, int lw, int pad, int n_maps,
                 float* lattice, unsigned* gmax, hipStream_t st, const char* who) {
    long widest = 0;
    for (int l = 0; l < n_levels; ++l) widest = (long)hs[l] * ws[l] > widest ? (long)hs[l] * ws[l] : widest;
    const long per = 0x7fffffffL / (widest * kC * 4);
    CAR_REQUIRE(per >= 1, "%s: one map of the widest level exceeds 2 GiB", who);
    (void)hipGetLastError();
    for (int m0 = 0; m0 < n_maps; m0 += (int)per) {
        const int nm = n_maps - m0 < per ? n_maps - m0 : (int)per;
        MergeArgs a{};
        for (int l = 0; l < n_levels; ++l) {
            a.g[l] = levels[l] + (long)m0 * hs[l] * ws[l] * kC;
            a.bytes[l] = (unsigned)((long)nm * hs[l] * ws[l] * kC * 4);
            a.h[l] = hs[l]; a.w[l] = ws[l]; a.r[l] = rs[l];
        }
        a.n_levels = n_levels; a.lh = lh; a.lw = lw; a.pad = pad;
        a.nodes = (long)nm * lh * lw;
        a.lat = lattice + (long)m0 * 2 * lh * lw * kC;
        // consecutive nodes per workgroup (a multiple of the 16 a workgroup takes per step), the rows they span, the tables' LDS
        const long groups = (a.nodes + 15) / 16, resident = merge_resident_blocks();
        long blocks = groups < resident ? groups : resident;
        a.per = ((a.nodes + blocks - 1) / blocks + 15) / 16 * 16;
        blocks = (a.nodes + a.per - 1) / a.per;
        a.ny = (int)(a.per / lw) + 2;
        const size_t tab_bytes = (size_t)n_levels * 2 * (lw + a.ny) * sizeof(float4);
        const bool tables = tab_bytes <= 52 * 1024;                   // three workgroups per compute unit keep theirs in the 160 KB
#define CAR_MERGE_KERNEL(T) (n_levels == 1 ? merge_kernel<1, T> : n_levels == 2 ? merge_kernel<2, T> : n_levels == 3 ? merge_kernel<3, T> : merge_kernel<4, T>)
        void (*kern)(const MergeArgs, unsigned*) = tables ? CAR_MERGE_KERNEL(true) : CAR_MERGE_KERNEL(false);
#undef CAR_MERGE_KERNEL
        if (tables) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes) != hipSuccess) {
                car_set_error("%s: cannot reserve %zu bytes of LDS", who, tab_bytes);
                return CAR_E_LAUNCH;
            }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), tables ? tab_bytes : 0, st, a, gmax);
    }
    CAR_CHECK_LAUNCH(who);
    return CAR_OK;
}
// Every level is summed on the lattice: it must be an integer factor r_l coarser than the widest level, the same factor in both
// directions; lat = 2 W_max + 2 r_max + 1 nodes, pad = r_max + 1 (521 x 521 nodes, pad 5, for the 64 / 128 / 256 pyramid of a 256 x 256
// frame: 2.5 GB per scene for two views and two padding modes).
struct Lattice { int h, w, pad, r[CAR_MAX_LEVELS]; bool ok; };
Lattice lattice_of(const car_dims& d) {
    Lattice L{};
    int hm = 0, wm = 0, rmax = 1;
    for (int l = 0; l < d.n_levels; ++l) {
        hm = d.level_h[l] > hm ? d.level_h[l] : hm; wm = d.level_w[l] > wm ? d.level_w[l] : wm;
    }
    L.ok = d.n_levels > 0;
    for (int l = 0; l < d.n_levels && L.ok; ++l) {
        const int h = d.level_h[l], w = d.level_w[l];
        L.ok = h > 0 && w > 0 && hm % h == 0 && wm % w == 0 && hm / h == wm / w;
        if (L.ok) { L.r[l] = hm / h; rmax = L.r[l] > rmax ? L.r[l] : rmax; }
    }
    L.pad = rmax + 1;
    L.h = 2 * hm + 2 * rmax + 1; L.w = 2 * wm + 2 * rmax + 1;
    return L;
}
// ---- plan layout ---------------------------------------------------------------------------------------------------
struct Plan {
    // offsets in floats.  latent_value ... lout_c: car_chain_pack tiles of the per-ray chains (car_raychain.hip; latent_value and lin_in
    // read their input rows from memory, the *_c layers the previous layer's accumulators); chain_scale: their powers of two;
    // mid_bias / tail_bias: their biases in consumption order
    size_t steps, blob, fbias, wpt, r2qw, r2qb, proj[CAR_MAX_LEVELS], proj16[CAR_MAX_LEVELS], latent_value, lin_in, enc_c, qreh_c, lz_c[kBlocks], fc0_c[kBlocks],
        fc1_c[kBlocks], lout_c, chain_scale, mid_bias, tail_bias, total;
};
// car_linear_x3 wants rows of whole float4s and a K worth its 32-wide chunks; narrower levels stay on car_linear
inline bool level_on_f16_pipe(int c) { return c % 4 == 0 && c >= 32; }
Plan plan_layout(const car_dims& d) {
    Plan p;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += up64(n); return at; };
    p.steps = take((size_t)d.P);
    p.blob = take(car_fused_blob_floats());
    p.fbias = take(car_fused_bias_floats());
    p.wpt = take((size_t)kC * 4);
    p.r2qw = take(car_round2q_packed_floats());
    p.r2qb = take(car_round2q_bias_floats());
    for (int l = 0; l < CAR_MAX_LEVELS; ++l) p.proj[l] = l < d.n_levels ? take(car_linear_packed_floats(d.level_c[l], kC)) : 0;
    // the same slices for the split-fp16 kernel (car_linear_x3), which car_project_maps takes for the levels it serves
    for (int l = 0; l < CAR_MAX_LEVELS; ++l) p.proj16[l] = (l < d.n_levels && level_on_f16_pipe(d.level_c[l])) ? take(car_linear_x3_packed_floats(d.level_c[l], kC)) : 0;
    p.latent_value = take(car_chain_packed_floats(kC, kE));
    p.lin_in = take(car_chain_packed_floats(kPhiIn, kD));
    p.enc_c = take(car_chain_packed_floats(kE, kD));
    p.qreh_c = take(car_chain_packed_floats(kD, kD));
    for (int i = 0; i < kBlocks; ++i) {
        p.lz_c[i] = take(car_chain_packed_floats(kE, kD));
        p.fc0_c[i] = take(car_chain_packed_floats(kD, kD));
        p.fc1_c[i] = take(car_chain_packed_floats(kD, kD));
    }
    p.lout_c = take(car_chain_packed_floats(kD, 3));
    p.chain_scale = take(32);
    p.mid_bias = take(kE + kD);
    p.tail_bias = take(kE + kD + 3 * kBlocks * kD + 32);
    p.total = o;
    return p;
}

int check_dims(const car_dims* d, const char* who) {
    CAR_REQUIRE(d, "%s: null dims", who);
    CAR_REQUIRE(d->b > 0 && d->R > 0 && d->P > 1 && d->H > 1 && d->W > 1, "%s: bad sizes", who);
    CAR_REQUIRE(d->V == 2 && d->n_levels == 3, "%s: the one-call forward covers n_view = 2 and three pyramid levels (got %d, %d); "
                "other configurations run through the stage entries", who, d->V, d->n_levels);
    int csum = 0;
    for (int l = 0; l < d->n_levels; ++l) {
        CAR_REQUIRE(d->level_h[l] > 0 && d->level_w[l] > 0 && d->level_c[l] > 0, "%s: bad level %d", who, l);
        csum += d->level_c[l];
    }
    CAR_REQUIRE(csum == kC, "%s: the levels' channels must add up to %d (got %d)", who, kC, csum);
    CAR_REQUIRE(2 * d->P <= 128 * 3, "%s: too many samples per ray", who);
    CAR_REQUIRE(lattice_of(*d).ok, "%s: every pyramid level must be an integer factor coarser than the widest one, the same factor in both "
                "directions (the fused kernel gathers them from their common lattice); other pyramids run through the stage entries", who);
    return CAR_OK;
}

// ---- workspace layout ----------------------------------------------------------------------------------------------
struct Work {
    size_t rays, phi_x, e, g, logit, logit2, pt, pixel_val, coords, at_wt, at_wt2, amax, depth, ebar, z1, uh, valid, part, total;                                  // offsets in floats
};
// step groups per (view, ray) of the first round's partial sums (car_fused_samples_parts)
inline size_t step_groups(const car_dims& d) { const int ts = car_fused_tile_steps(); return (size_t)((d.P + ts - 1) / ts); }
Work work_layout(const car_dims& d) {
    Work w;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += up64(n); return at; };
    const size_t n = (size_t)d.b * d.V, S = n * d.R * d.P, BR = (size_t)d.b * d.R;
    w.rays = take(n * d.R * CAR_RAY_FLOATS);
    w.phi_x = take(BR * kPhiLd);
    w.e = take(S * kC);
    w.g = take(S * CAR_G_DIM);
    w.logit = take(S);
    w.logit2 = take(S);
    w.pt = take(S * 3);
    w.pixel_val = take(S * 2);
    w.coords = take(n * d.R * 9);
    w.at_wt = take(S);
    w.at_wt2 = take(S);
    w.amax = take(n * d.R);
    w.depth = take(BR);
    w.ebar = take(BR * kC);
    w.z1 = take(BR * kE);
    w.uh = take(BR * kD);
    w.valid = take(BR);
    w.part = take(n * d.R * step_groups(d) * kC);
    w.total = o;
    return w;
}

#define CAR_TRY(call)                 \
    do {                              \
        const int rc_ = (call);       \
        if (rc_ != CAR_OK) return rc_; \
    } while (0)

// ---- stage timing --------------------------------------------------------------------------------------------------
struct StageRec { const char* name; hipEvent_t start, stop; };
struct Profile {
    bool on = false;
    std::vector<StageRec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        return e;
    }
    void clear() {
        for (StageRec& r : recs) { pool.push_back(r.start); pool.push_back(r.stop); }
        recs.clear();
    }
};
thread_local Profile g_prof;
// brackets the launches of one stage with two events on the caller's stream (when profiling is on)
struct Stage {
    hipStream_t st;
    bool live;
    Stage(const char* name, hipStream_t s) : st(s), live(g_prof.on) {
        if (!live) return;
        StageRec r{name, g_prof.event(), g_prof.event()};
        if (!r.start || !r.stop) { live = false; return; }
        (void)hipEventRecord(r.start, st);
        g_prof.recs.push_back(r);
    }
    ~Stage() { if (live) (void)hipEventRecord(g_prof.recs.back().stop, st); }
};

}  // namespace

extern "C" void car_profile_enable(int on) { g_prof.clear(); g_prof.on = on != 0; }
extern "C" void car_profile_reset(void) { g_prof.clear(); }
extern "C" int car_profile_count(void) { return (int)g_prof.recs.size(); }
extern "C" int car_profile_read(int i, const char** name, float* ms) {
    CAR_REQUIRE(i >= 0 && i < (int)g_prof.recs.size() && name && ms, "car_profile_read: no stage %d", i);
    const StageRec& r = g_prof.recs[i];
    if (hipEventSynchronize(r.stop) != hipSuccess || hipEventElapsedTime(ms, r.start, r.stop) != hipSuccess) {
        car_set_error("car_profile_read: %s", hipGetErrorString(hipGetLastError()));
        return CAR_E_LAUNCH;
    }
    *name = r.name;
    return CAR_OK;
}

// torch.linspace (CPU, fp32): step = (b - a) / (n - 1); first half a + step * i, second half b - step * (n - 1 - i)
extern "C" void car_linspace(float a, float b, int n, float* out) {
    if (n == 1) { out[0] = a; return; }
    const float step = (b - a) / (float)(n - 1);
    const int half = n / 2;
    for (int i = 0; i < n; ++i) out[i] = i < half ? a + step * (float)i : b - step * (float)(n - 1 - i);
}

extern "C" size_t car_plan_bytes(const car_dims* dims) {
    if (check_dims(dims, "car_plan_bytes") != CAR_OK) return 0;
    return plan_layout(*dims).total * sizeof(float);
}
extern "C" size_t car_workspace_bytes(const car_dims* dims) {
    if (check_dims(dims, "car_workspace_bytes") != CAR_OK) return 0;
    return work_layout(*dims).total * sizeof(float);
}

namespace {
size_t lattice_floats(const car_dims& d) { const Lattice L = lattice_of(d); return (size_t)d.b * d.V * 2 * L.h * L.w * kC; }
size_t level_floats(const car_dims& d, int l) { return (size_t)d.b * d.V * d.level_h[l] * d.level_w[l] * kC; }
// projected levels behind the lattice and gmeta (scratch of car_project_maps: the merge reads them)
size_t level_offset(const car_dims& d, int level) {
    size_t n = up64(lattice_floats(d)) + 64;
    for (int l = 0; l < level && l < d.n_levels; ++l) n += up64(level_floats(d, l));
    return n;
}
}  // namespace
extern "C" size_t car_gmeta_offset(const car_dims* dims) {
    if (check_dims(dims, "car_gmeta_offset") != CAR_OK) return 0;
    return up64(lattice_floats(*dims));
}
extern "C" size_t car_gmaps_floats(const car_dims* dims) {
    if (check_dims(dims, "car_gmaps_floats") != CAR_OK) return 0;
    return level_offset(*dims, CAR_MAX_LEVELS + 1);
}
extern "C" int car_lattice_shape(const car_dims* dims, int* lat_h, int* lat_w, int* lat_pad) {
    CAR_TRY(check_dims(dims, "car_lattice_shape"));
    CAR_REQUIRE(lat_h && lat_w && lat_pad, "car_lattice_shape: null pointer");
    const Lattice L = lattice_of(*dims);
    *lat_h = L.h; *lat_w = L.w; *lat_pad = L.pad;
    return CAR_OK;
}
extern "C" int car_workspace_find(const car_dims* dims, const char* name, size_t* offset_floats, size_t* n_floats) {
    CAR_TRY(check_dims(dims, "car_workspace_find"));
    CAR_REQUIRE(name && offset_floats && n_floats, "car_workspace_find: null pointer");
    const Work w = work_layout(*dims);
    const size_t n = (size_t)dims->b * dims->V, S = n * dims->R * dims->P, BR = (size_t)dims->b * dims->R;
    const struct { const char* name; size_t off, cnt; } tab[] = {
        {"rays", w.rays, n * dims->R * CAR_RAY_FLOATS}, {"e", w.e, S * kC}, {"g", w.g, S * CAR_G_DIM},
        {"logit", w.logit, S}, {"logit2", w.logit2, S}, {"pt", w.pt, S * 3}, {"at_wt2", w.at_wt2, S}, {"ebar", w.ebar, BR * kC},
        {"z1", w.z1, BR * kE}, {"uh", w.uh, BR * kD}, {"part", w.part, n * dims->R * step_groups(*dims) * kC}};
    for (const auto& t : tab)
        if (strcmp(t.name, name) == 0) { *offset_floats = t.off; *n_floats = t.cnt; return CAR_OK; }
    car_set_error("car_workspace_find: unknown tensor '%s'", name);
    return CAR_E_ARG;
}

// Packs the six layers of the fused per-sample kernel (csrc/car_fused.hip) into its operand order: fp16 hi/lo tiles, every layer
// times its own power of two (chosen from its largest weight; 1/p goes into the bias table).  Asynchronous, device side only.
extern "C" int car_fused_pack(const car_weights* w, float* blob_f, float* bias, float* wpt, void* stream) {
    CAR_REQUIRE(w && blob_f && bias && wpt, "car_fused_pack: null pointer");
    CAR_REQUIRE(w->query_encode_latent_w && w->query_encode_latent_b && w->query_encode_latent_2_w && w->query_encode_latent_2_b &&
                w->key_map_w && w->key_map_b && w->key_map_2_w && w->key_map_2_b && w->query_embed_w && w->query_embed_b &&
                w->query_embed_2_w && w->query_embed_2_b, "car_fused_pack: a weight pointer of the fused layers is null");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(bias, 0, sizeof(float) * (kBiasFloats + kBiasScratch), st) != hipSuccess) { car_set_error("car_fused_pack: memset failed"); return CAR_E_LAUNCH; }
    _Float16* blob = reinterpret_cast<_Float16*>(blob_f);
    float* fdown = bias + kBiasScale;
    float* pscale = bias + kBiasScale + 8;                           // pack-time scratch: 2^shift per layer
    (void)hipGetLastError();
    auto scale = [&](const float* W, int ldw, int N, int K, const float* b, int layer) {
        hipLaunchKernelGGL(layer_scale_kernel, dim3(1), dim3(1024), 0, st, W, ldw, N, K, b, pscale + layer, fdown + layer);
    };
    auto pack16 = [&](const float* W, int ldw, const float* b, int N, int K, int n_tiles, int ksteps, int mode, int kbase, int layer, int tile_off) {
        hipLaunchKernelGGL(pack16_kernel, dim3(256), dim3(256), 0, st, W, ldw, b, N, K, n_tiles, ksteps, mode, kbase, pscale + layer,
                           blob + (size_t)tile_off * kTile16 * 2);
    };
    // the closing pair key_map_2 / query_embed_2 folded into one layer (car_fused_layout.h): M in fp32 behind the bias table, v, u, c inside it
    float* Mf = bias + kBiasFloats;
    hipLaunchKernelGGL(bilinear_fold_kernel, dim3(kD), dim3(kD), 0, st, w->key_map_2_w, w->key_map_2_b, w->query_embed_2_w, w->query_embed_2_b, kD, Mf,
                       bias + kBiasV, bias + kBiasU, bias + kBiasConst);
    scale(w->query_encode_latent_2_w, kC, kE, kC, nullptr, kLayerW2);
    scale(w->query_embed_w, 16, kD, 16, w->query_embed_b, kLayerQ1);
    scale(Mf, kD, kD, kD, nullptr, kLayerM);
    scale(w->key_map_w, kC, kD, kC, nullptr, kLayerK1);
    pack16(w->query_encode_latent_2_w, kC, nullptr, kE, kC, kTE, kKS, 0, 0, kLayerW2, kOffW2);
    pack16(w->query_embed_w, 16, w->query_embed_b, kD, 16, kTD, 1, 0, 0, kLayerQ1, kOffQ1);
    pack16(Mf, kD, nullptr, kD, kD, kTD, 4, 1, 0, kLayerM, kOffM);
    pack16(w->key_map_w, kC, nullptr, kD, kC, kTD, 9, 1, 0, kLayerK1, kOffK1);
    pack16(w->key_map_w, kC, nullptr, kD, kC, kTD, 9, 1, kE, kLayerK1, kOffK1 + 9 * kTD);
    hipLaunchKernelGGL(wpt_kernel, dim3(1), dim3(kC), 0, st, w->query_encode_latent_w, w->query_encode_latent_b, wpt, fdown + 5);
    CAR_CHECK_LAUNCH("car_fused_pack");
    auto d2d = [&](float* dst, const float* src, int n) { return hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, st) == hipSuccess; };
    if (!d2d(bias + kBiasE, w->query_encode_latent_2_b, kE) || !d2d(bias + kBiasK1, w->key_map_b, kD)) {
        car_set_error("car_fused_pack: bias copy failed");
        return CAR_E_LAUNCH;
    }
    return CAR_OK;
}

// The first two point-MLP layers alone, for car_fused_rows (the three-view exchange): W2 in the fused kernel's operand tiles with its power
// of two, b2 and the scales in the bias table, the [C][4] point / bias table of the first layer with its largest row sum.  Same formats
// as car_fused_pack; the other layers' regions of blob / bias stay zero.
extern "C" int car_fused_pack_rows(const float* w1, const float* b1, const float* w2, const float* b2, float* blob_f, float* bias, float* wpt, void* stream) {
    CAR_REQUIRE(w1 && b1 && w2 && b2 && blob_f && bias && wpt, "car_fused_pack_rows: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(bias, 0, sizeof(float) * kBiasFloats, st) != hipSuccess) { car_set_error("car_fused_pack_rows: memset failed"); return CAR_E_LAUNCH; }
    _Float16* blob = reinterpret_cast<_Float16*>(blob_f);
    float* fdown = bias + kBiasScale;
    float* pscale = bias + kBiasScale + 8;
    (void)hipGetLastError();
    hipLaunchKernelGGL(layer_scale_kernel, dim3(1), dim3(1024), 0, st, w2, kC, kE, kC, (const float*)nullptr, pscale + kLayerW2, fdown + kLayerW2);
    hipLaunchKernelGGL(pack16_kernel, dim3(256), dim3(256), 0, st, w2, kC, (const float*)nullptr, kE, kC, kTE, kKS, 0, 0, pscale + kLayerW2,
                       blob + (size_t)kOffW2 * kTile16 * 2);
    hipLaunchKernelGGL(wpt_kernel, dim3(1), dim3(kC), 0, st, w1, b1, wpt, fdown + 5);
    CAR_CHECK_LAUNCH("car_fused_pack_rows");
    if (hipMemcpyAsync(bias + kBiasE, b2, sizeof(float) * kE, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        car_set_error("car_fused_pack_rows: bias copy failed");
        return CAR_E_LAUNCH;
    }
    return CAR_OK;
}

// Packs key_map_2, query_embed and query_embed_2 for car_key_query_logits (csrc/car_linear16.hip, the stage route's key / query chain): the
// fused kernel's operand formats — K2 and Q2 chained over the accumulator order of the layer before, Q1 standard with its bias folded in at
// k = 16 — in the order the kernel streams them: K2 (32 tiles) | Q1 (8) | Q2 (32).  bias: bk2 [128] | bq2 [128] | 2^-shift of K2, Q1, Q2.
extern "C" size_t car_kq_tail_floats(void);
extern "C" size_t car_kq_bias_floats(void);
extern "C" int car_kq_pack(const float* k2w, const float* k2b, const float* q1w, const float* q1b, const float* q2w, const float* q2b, float* tail,
                           float* bias, void* stream) {
    CAR_REQUIRE(k2w && k2b && q1w && q1b && q2w && q2b && tail && bias, "car_kq_pack: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)car_kq_bias_floats();
    if (hipMemsetAsync(bias, 0, sizeof(float) * nb, st) != hipSuccess) { car_set_error("car_kq_pack: memset failed"); return CAR_E_LAUNCH; }
    float* down = bias + 2 * kD;                                     // [0..2] 2^-shift of K2, Q1, Q2; [8..10] their 2^shift (pack-time scratch)
    _Float16* blob = reinterpret_cast<_Float16*>(tail);
    (void)hipGetLastError();
    hipLaunchKernelGGL(layer_scale_kernel, dim3(1), dim3(1024), 0, st, k2w, kD, kD, kD, (const float*)nullptr, down + 8, down + 0);
    hipLaunchKernelGGL(layer_scale_kernel, dim3(1), dim3(1024), 0, st, q1w, 16, kD, 16, q1b, down + 9, down + 1);
    hipLaunchKernelGGL(layer_scale_kernel, dim3(1), dim3(1024), 0, st, q2w, kD, kD, kD, (const float*)nullptr, down + 10, down + 2);
    hipLaunchKernelGGL(pack16_kernel, dim3(64), dim3(256), 0, st, k2w, kD, (const float*)nullptr, kD, kD, kTD, 4, 1, 0, down + 8, blob);
    hipLaunchKernelGGL(pack16_kernel, dim3(16), dim3(256), 0, st, q1w, 16, q1b, kD, 16, kTD, 1, 0, 0, down + 9, blob + (size_t)32 * kTile16 * 2);
    hipLaunchKernelGGL(pack16_kernel, dim3(64), dim3(256), 0, st, q2w, kD, (const float*)nullptr, kD, kD, kTD, 4, 1, 0, down + 10, blob + (size_t)40 * kTile16 * 2);
    CAR_CHECK_LAUNCH("car_kq_pack");
    if (hipMemcpyAsync(bias, k2b, sizeof(float) * kD, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(bias + kD, q2b, sizeof(float) * kD, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        car_set_error("car_kq_pack: bias copy failed");
        return CAR_E_LAUNCH;
    }
    return CAR_OK;
}

// Packs query_repeat_embed (its local_coords half, columns 128..143 of the (128, 144) matrix `wr1`) and query_repeat_embed_2 for
// csrc/car_round2.hip; same conventions as car_fused_pack.
extern "C" int car_round2_pack(const float* wr1, const float* br1, const float* wr2, const float* br2, float* wpacked, float* bias, void* stream) {
    CAR_REQUIRE(wr1 && br1 && wr2 && br2 && wpacked && bias, "car_round2_pack: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)car_round2_bias_floats();
    if (hipMemsetAsync(bias, 0, sizeof(float) * nb, st) != hipSuccess) { car_set_error("car_round2_pack: memset failed"); return CAR_E_LAUNCH; }
    float* down = bias + 2 * kD;                                     // [0] Wr1g, [1] Wr2; [2], [3]: their 2^shift (pack-time scratch)
    _Float16* out = reinterpret_cast<_Float16*>(wpacked);
    (void)hipGetLastError();
    hipLaunchKernelGGL(layer_scale_kernel, dim3(1), dim3(1024), 0, st, wr1 + kD, kD + 16, kD, 16, (const float*)nullptr, down + 2, down + 0);
    hipLaunchKernelGGL(layer_scale_kernel, dim3(1), dim3(1024), 0, st, wr2, kD, kD, kD, (const float*)nullptr, down + 3, down + 1);
    hipLaunchKernelGGL(pack32_kernel, dim3(128), dim3(256), 0, st, wr2, kD, 4, 2, 1, down + 3, out);
    hipLaunchKernelGGL(pack32_kernel, dim3(16), dim3(256), 0, st, wr1 + kD, kD + 16, 1, 1, 0, down + 2, out + 4 * 4 * 2 * 2 * 64 * 8);
    CAR_CHECK_LAUNCH("car_round2_pack");
    if (hipMemcpyAsync(bias, br1, sizeof(float) * kD, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(bias + kD, br2, sizeof(float) * kD, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        car_set_error("car_round2_pack: bias copy failed");
        return CAR_E_LAUNCH;
    }
    return CAR_OK;
}

// For car_round2_logits_from_g (csrc/car_round2.hip, G instance): query_repeat_embed_2 and query_embed_2 folded into the one layer of the
// bilinear form <q2, qry> = y^T (M x + v) + u^T x + c (M = Wr2^T Wq2, v = Wr2^T bq2, u = Wq2^T br2, c = <br2, bq2>; bilinear_fold_kernel) and the
// two 16 -> 128 layers that make y and x from g.  wpacked [car_round2q_packed_floats()] = M (chained K order) | Wr1[:, 128:] | Wq1, each laid out
// as car_round2_pack lays out its own; bias [car_round2q_bias_floats()] = br1 | v | bq1 | u | 2^-shift of Wr1g, M, Wq1 | c | scratch (their
// 2^shift, then M in fp32).
extern "C" int car_round2q_pack(const float* wr1, const float* br1, const float* wr2, const float* br2, const float* wq1, const float* bq1,
                                const float* wq2, const float* bq2, float* wpacked, float* bias, void* stream) {
    CAR_REQUIRE(wr1 && br1 && wr2 && br2 && wq1 && bq1 && wq2 && bq2 && wpacked && bias, "car_round2q_pack: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)car_round2q_bias_floats();
    if (hipMemsetAsync(bias, 0, sizeof(float) * nb, st) != hipSuccess) { car_set_error("car_round2q_pack: memset failed"); return CAR_E_LAUNCH; }
    float* down = bias + 4 * kD;                                     // [0] Wr1g, [1] M, [2] Wq1, [3] c; [4..6]: the layers' 2^shift (pack-time scratch)
    float* Mf = bias + 4 * kD + 8;                                   // scratch: M in fp32
    _Float16* out = reinterpret_cast<_Float16*>(wpacked);
    const size_t first = (size_t)4 * 4 * 2 * 2 * 64 * 8, small = (size_t)4 * 2 * 64 * 8;                  // halves: the 128 x 128 layer, a 128 x 16 layer
    (void)hipGetLastError();
    hipLaunchKernelGGL(bilinear_fold_kernel, dim3(kD), dim3(kD), 0, st, wr2, br2, wq2, bq2, kD, Mf, bias + kD, bias + 3 * kD, down + 3);
    hipLaunchKernelGGL(layer_scale_kernel, dim3(1), dim3(1024), 0, st, wr1 + kD, kD + 16, kD, 16, (const float*)nullptr, down + 4, down + 0);
    hipLaunchKernelGGL(layer_scale_kernel, dim3(1), dim3(1024), 0, st, Mf, kD, kD, kD, (const float*)nullptr, down + 5, down + 1);
    hipLaunchKernelGGL(layer_scale_kernel, dim3(1), dim3(1024), 0, st, wq1, 16, kD, 16, (const float*)nullptr, down + 6, down + 2);
    hipLaunchKernelGGL(pack32_kernel, dim3(128), dim3(256), 0, st, Mf, kD, 4, 2, 1, down + 5, out);
    hipLaunchKernelGGL(pack32_kernel, dim3(16), dim3(256), 0, st, wr1 + kD, kD + 16, 1, 1, 0, down + 4, out + first);
    hipLaunchKernelGGL(pack32_kernel, dim3(16), dim3(256), 0, st, wq1, 16, 1, 1, 0, down + 6, out + first + small);
    CAR_CHECK_LAUNCH("car_round2q_pack");
    if (hipMemcpyAsync(bias, br1, sizeof(float) * kD, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(bias + 2 * kD, bq1, sizeof(float) * kD, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        car_set_error("car_round2q_pack: bias copy failed");
        return CAR_E_LAUNCH;
    }
    return CAR_OK;
}

extern "C" int car_plan_build(const car_dims* dims, const car_weights* w, void* plan, void* stream) {
    CAR_TRY(check_dims(dims, "car_plan_build"));
    CAR_REQUIRE(w && plan, "car_plan_build: null pointer");
    const float* const* all = reinterpret_cast<const float* const*>(w);
    for (size_t k = 0; k < sizeof(car_weights) / sizeof(const float*); ++k)
        CAR_REQUIRE(all[k], "car_plan_build: weight pointer %zu of car_weights is null", k);
    CAR_REQUIRE(car_fused_blob_floats() == (size_t)kBlobTiles * kTile && car_fused_bias_floats() == (size_t)(kBiasFloats + kBiasScratch),
                "car_plan_build: the fused kernel was built with another weight layout");
    const Plan p = plan_layout(*dims);
    float* base = static_cast<float*>(plan);
    hipStream_t st = (hipStream_t)stream;
    // sample positions linspace(0, 1, P) with torch's CPU arithmetic (models.py:261)
    {
        float steps[1024];
        CAR_REQUIRE(dims->P <= 1024, "car_plan_build: P too large");
        car_linspace(0.0f, 1.0f, dims->P, steps);
        if (hipMemcpyAsync(base + p.steps, steps, sizeof(float) * dims->P, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            car_set_error("car_plan_build: upload failed: %s", hipGetErrorString(hipGetLastError()));
            return CAR_E_LAUNCH;
        }
    }
    // split-fp16 operand tiles of the fused per-sample kernel and of the round-2 kernel
    CAR_TRY(car_fused_pack(w, base + p.blob, base + p.fbias, base + p.wpt, stream));
    CAR_TRY(car_round2q_pack(w->query_repeat_embed_w, w->query_repeat_embed_b, w->query_repeat_embed_2_w, w->query_repeat_embed_2_b,
                             w->query_embed_w, w->query_embed_b, w->query_embed_2_w, w->query_embed_2_b, base + p.r2qw, base + p.r2qb, stream));
    // fp32 MFMA layers (car_linear.hip)
    int coff = 0;
    for (int l = 0; l < dims->n_levels; ++l) {
        CAR_TRY(car_linear_pack(w->query_encode_latent_w + coff, kC + 3, nullptr, dims->level_c[l], kC, base + p.proj[l], stream));
        if (level_on_f16_pipe(dims->level_c[l]))
            CAR_TRY(car_linear_x3_pack(w->query_encode_latent_w + coff, kC + 3, dims->level_c[l], kC, base + p.proj16[l], stream));
        coff += dims->level_c[l];
    }
    // the per-ray chains (car_raychain.hip), split-fp16 tiles; scale slots: 0 latent_value, 1 encode_latent, 2 query_repeat_embed[:, :128],
    // 3 lin_in, 4 + 3 i lin_z_i, 5 + 3 i fc_0_i, 6 + 3 i fc_1_i, 13 lin_out
    float* cs = base + p.chain_scale;
    CAR_TRY(car_chain_pack(w->latent_value_w, kC, nullptr, kC, kE, 0, base + p.latent_value, cs, 0, stream));
    CAR_TRY(car_chain_pack(w->encode_latent_w, kE, nullptr, kE, kD, 1, base + p.enc_c, cs, 1, stream));
    CAR_TRY(car_chain_pack(w->query_repeat_embed_w, kD + 16, nullptr, kD, kD, 1, base + p.qreh_c, cs, 2, stream));
    CAR_TRY(car_chain_pack(w->phi_lin_in_w, kPhiIn, nullptr, kPhiIn, kD, 0, base + p.lin_in, cs, 3, stream));
    for (int i = 0; i < kBlocks; ++i) {
        CAR_TRY(car_chain_pack(w->phi_lin_z_w[i], 2 * kE, w->phi_lin_z_w[i] + kE, kE, kD, 1, base + p.lz_c[i], cs, 4 + 3 * i, stream));   // [z, z]: halves added
        CAR_TRY(car_chain_pack(w->phi_fc_0_w[i], kD, nullptr, kD, kD, 1, base + p.fc0_c[i], cs, 5 + 3 * i, stream));
        CAR_TRY(car_chain_pack(w->phi_fc_1_w[i], kD, nullptr, kD, kD, 1, base + p.fc1_c[i], cs, 6 + 3 * i, stream));
    }
    CAR_TRY(car_chain_pack(w->phi_lin_out_w, kD, nullptr, kD, 3, 1, base + p.lout_c, cs, 13, stream));
    const int tail_floats = kE + kD + 3 * kBlocks * kD + 32;
    if (hipMemsetAsync(base + p.tail_bias, 0, sizeof(float) * tail_floats, st) != hipSuccess) { car_set_error("car_plan_build: memset failed"); return CAR_E_LAUNCH; }
    auto d2d = [&](float* dst, const float* src, int n) { return hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, st) == hipSuccess; };
    bool ok = d2d(base + p.mid_bias, w->latent_value_b, kE) && d2d(base + p.mid_bias + kE, w->encode_latent_b, kD);
    float* tb = base + p.tail_bias;
    ok = ok && d2d(tb, w->latent_value_b, kE) && d2d(tb + kE, w->phi_lin_in_b, kD);
    tb += kE + kD;
    for (int i = 0; i < kBlocks; ++i)
        ok = ok && d2d(tb + (3 * i + 0) * kD, w->phi_lin_z_b[i], kD) && d2d(tb + (3 * i + 1) * kD, w->phi_fc_0_b[i], kD) &&
             d2d(tb + (3 * i + 2) * kD, w->phi_fc_1_b[i], kD);
    ok = ok && d2d(tb + 3 * kBlocks * kD, w->phi_lin_out_b, 3);
    if (!ok) { car_set_error("car_plan_build: bias copy failed"); return CAR_E_LAUNCH; }
    return CAR_OK;
}

extern "C" int car_project_maps(const car_dims* dims, const void* plan, const float* const* maps, float* gmaps, void* stream) {
    CAR_TRY(check_dims(dims, "car_project_maps"));
    CAR_REQUIRE(plan && maps && gmaps, "car_project_maps: null pointer");
    const Plan p = plan_layout(*dims);
    const float* base = static_cast<const float*>(plan);
    hipStream_t st = (hipStream_t)stream;
    float* gmeta = gmaps + car_gmeta_offset(dims);
    if (hipMemsetAsync(gmeta, 0, sizeof(float) * CAR_MAX_LEVELS, st) != hipSuccess) { car_set_error("car_project_maps: memset failed"); return CAR_E_LAUNCH; }
    const Lattice L = lattice_of(*dims);
    const float* lv[CAR_MAX_LEVELS];
    for (int l = 0; l < dims->n_levels; ++l) {
        CAR_REQUIRE(maps[l], "car_project_maps: level %d is null", l);
        const long M = (long)dims->b * dims->V * dims->level_h[l] * dims->level_w[l];
        float* gl = gmaps + level_offset(*dims, l);
        // G_l = W1[:, ch_l] F_l per texel: on the f16 matrix pipe with fp16 hi / lo operand halves (car_linear_x3: fp32-class accuracy at
        // 2.3x the fp32 pipe's rate — the arithmetic of the staged route's engine._projected_maps) where the level's width allows it
        if (level_on_f16_pipe(dims->level_c[l]) && ((uintptr_t)maps[l] & 15) == 0)
            CAR_TRY(car_linear_x3(maps[l], dims->level_c[l], base + p.proj16[l], nullptr, dims->level_c[l], kC, gl, kC, M, 0, stream));
        else
            CAR_TRY(car_linear(maps[l], dims->level_c[l], base + p.proj[l], dims->level_c[l], kC, gl, kC, M, 0, stream));
        lv[l] = gl;
    }
    // the lattice, and in the same pass its largest magnitude (gmeta[0], zeroed above): it bounds h (the fused kernel scales its fp16
    // operands by it)
    return launch_merge(lv, dims->level_h, dims->level_w, L.r, dims->n_levels, L.h, L.w, L.pad, dims->b * dims->V, gmaps,
                        reinterpret_cast<unsigned*>(gmeta), st, "car_project_maps (merge)");
}

// The lattice alone, for hosts that project the levels themselves (engine.py: the three-view exchange, which has no plan): levels[l] =
// the projected level [n_maps, level_h[l], level_w[l], 576] channel-last; lattice = [n_maps][2 padding modes][lat_h][lat_w][576] (NULL: only
// the shape is returned).
extern "C" int car_merge_lattice(const float* const* levels, const int* level_h, const int* level_w, int n_levels, int n_maps, float* lattice,
                                 int* lat_h, int* lat_w, int* lat_pad, void* stream) {
    CAR_REQUIRE(levels && level_h && level_w && n_levels > 0 && n_levels <= CAR_MAX_LEVELS && n_maps > 0, "car_merge_lattice: bad arguments");
    car_dims d{};
    d.b = n_maps; d.V = 1; d.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) { d.level_h[l] = level_h[l]; d.level_w[l] = level_w[l]; }
    const Lattice L = lattice_of(d);
    CAR_REQUIRE(L.ok, "car_merge_lattice: every level must be an integer factor coarser than the widest one, the same factor in both directions");
    if (lat_h) *lat_h = L.h;
    if (lat_w) *lat_w = L.w;
    if (lat_pad) *lat_pad = L.pad;
    if (!lattice) return CAR_OK;
    for (int l = 0; l < n_levels; ++l) CAR_REQUIRE(levels[l], "car_merge_lattice: level %d is null", l);
    return launch_merge(levels, level_h, level_w, L.r, n_levels, L.h, L.w, L.pad, n_maps, lattice, nullptr, (hipStream_t)stream, "car_merge_lattice");
}
// The same, and the lattice's largest magnitude in the same pass (gmax [1]: zeroed here, then one atomic per workgroup of the merge) — what
// car_fused_rows takes as `gmeta`; no separate reduction over the gigabyte of lattice.
extern "C" int car_merge_lattice_max(const float* const* levels, const int* level_h, const int* level_w, int n_levels, int n_maps, float* lattice,
                                     float* gmax, void* stream) {
    CAR_REQUIRE(levels && level_h && level_w && n_levels > 0 && n_levels <= CAR_MAX_LEVELS && n_maps > 0 && lattice && gmax, "car_merge_lattice_max: bad arguments");
    car_dims d{};
    d.b = n_maps; d.V = 1; d.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) { d.level_h[l] = level_h[l]; d.level_w[l] = level_w[l]; }
    const Lattice L = lattice_of(d);
    CAR_REQUIRE(L.ok, "car_merge_lattice_max: every level must be an integer factor coarser than the widest one, the same factor in both directions");
    for (int l = 0; l < n_levels; ++l) CAR_REQUIRE(levels[l], "car_merge_lattice_max: level %d is null", l);
    if (hipMemsetAsync(gmax, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) { car_set_error("car_merge_lattice_max: memset failed"); return CAR_E_LAUNCH; }
    return launch_merge(levels, level_h, level_w, L.r, n_levels, L.h, L.w, L.pad, n_maps, lattice, reinterpret_cast<unsigned*>(gmax), (hipStream_t)stream,
                        "car_merge_lattice_max");
}

// the launches of one forward call in two phases: CAR_PHASE_SAMPLES = rays + the fused per-sample kernel (compute / power bound),
// CAR_PHASE_RAYS = the attention rounds and the per-ray chains (HBM bound), which only read what the first phase left in the workspace
static int render_phases(const car_dims* dims, const void* plan, const car_inputs* in, const car_outputs* out,
                         void* workspace, size_t workspace_bytes, int phases, void* stream) {
    CAR_TRY(check_dims(dims, "car_render_forward"));
    CAR_REQUIRE(plan && in && out && workspace, "car_render_forward: null pointer");
    CAR_REQUIRE(in->poses && in->uv && in->lattice && in->gmeta && out->rgb, "car_render_forward: poses, uv, lattice, gmeta and rgb are required");
    const car_dims& d = *dims;
    const Plan p = plan_layout(d);
    const Work w = work_layout(d);
    CAR_REQUIRE(workspace_bytes >= w.total * sizeof(float), "car_render_forward: workspace of %zu bytes, %zu needed", workspace_bytes,
                w.total * sizeof(float));
    const float* pl = static_cast<const float*>(plan);
    float* ws = static_cast<float*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    const int b = d.b, V = d.V, R = d.R, P = d.P;
    const bool rows_first = (phases & CAR_PHASE_ROWS_FIRST_ROUND) != 0;
    CAR_REQUIRE(!d.no_sample || in->steps, "car_render_forward: no_sample needs `steps` (the P depths along the query ray, models.py:221-222)");
    const float* steps = in->steps ? in->steps : pl + p.steps;
    const long BR = (long)b * R;
    float* coords = out->coords ? out->coords : ws + w.coords;
    float* pixel_val = out->pixel_val ? out->pixel_val : ws + w.pixel_val;
    float* at_wt = out->at_wt ? out->at_wt : ws + w.at_wt;
    float* depth = out->depth_ray ? out->depth_ray : ws + w.depth;
    float* valid = out->valid_mask ? out->valid_mask : ws + w.valid;
    int32_t* amax = out->at_wt_max ? out->at_wt_max : reinterpret_cast<int32_t*>(ws + w.amax);

    if (phases & CAR_PHASE_SAMPLES) {
    {   // a4-a6: rays, their epipolar segments, the decoder's ray input (columns 18, 19 of phi_x stay zero)
        Stage stage("ray_setup", st);
        if (hipMemsetAsync(ws + w.phi_x, 0, sizeof(float) * BR * kPhiLd, st) != hipSuccess) { car_set_error("car_render_forward: memset failed"); return CAR_E_LAUNCH; }
        CAR_TRY(car_ray_setup(in->poses, in->uv, b, V, R, d.H, d.W, P, d.no_sample != 0, steps, ws + w.rays, coords, ws + w.phi_x, kPhiLd, stream));
    }
    {   // a6-a13 + round-1 logits: the fused per-sample kernel
        Stage stage("fused_samples", st);
        const Lattice L = lattice_of(d);
        if (rows_first)
            CAR_TRY(car_fused_samples(in->poses, ws + w.rays, steps, in->lattice, L.h, L.w, L.pad, in->gmeta, pl + p.wpt, pl + p.blob,
                                      pl + p.fbias, b, V, R, P, d.H, d.W, d.no_sample != 0, ws + w.e, ws + w.g, ws + w.logit, ws + w.pt, pixel_val, stream));
        else
            CAR_TRY(car_fused_samples_parts(in->poses, ws + w.rays, steps, in->lattice, L.h, L.w, L.pad, in->gmeta, pl + p.wpt, pl + p.blob,
                                            pl + p.fbias, b, V, R, P, d.H, d.W, d.no_sample != 0, ws + w.e, ws + w.g, ws + w.logit, ws + w.pt, pixel_val,
                                            ws + w.part, stream));
    }
    }
    if (!(phases & CAR_PHASE_RAYS)) return CAR_OK;
    {   // a14 + a16: attention round 1, depth read-out, argmax.  The value average comes from the per-step-group partial sums the
        // fused kernel left behind (an eighth of the rows of e), so e itself is streamed from HBM by the second round only
        Stage stage("attend_1", st);
        if (rows_first)
            CAR_TRY(car_attend(ws + w.logit, nullptr, kD, ws + w.e, kC, b, V, R, P, nullptr, 0.0f, at_wt, ws + w.ebar, kC, 1, ws + w.pt, in->poses,
                               depth, amax, stream));
        else
            CAR_TRY(car_attend_parts(ws + w.logit, ws + w.part, car_fused_tile_steps(), kC, b, V, R, P, at_wt, ws + w.ebar, kC, 1, ws + w.pt, in->poses,
                                     depth, amax, stream));
    }
    // weight-chunk tables of the two per-ray chains (car_raychain.hip): float offset inside the plan and tile count of every K = 32
    // chunk, in the order the kernels consume them
    unsigned offs[96];
    int nts[96];
    int nch = 0;
    auto chunks = [&](size_t at, int n, int nt) { for (int c = 0; c < n; ++c) { offs[nch] = (unsigned)(at + (size_t)c * nt * 1024); nts[nch++] = nt; } };
    CAR_REQUIRE(p.total < (1ull << 32), "car_render_forward: plan too large");
    if (d.repeat_attention) {
        {   // a15, per ray: z1 = Wv ebar + bv; uh = Wr1[:, :128] encode_latent(z1)
            Stage stage("ray_layers_1", st);
            nch = 0;
            chunks(p.latent_value, 18, 9); chunks(p.enc_c, 9, 4); chunks(p.qreh_c, 4, 4);
            const int layers[3] = {0, 1, 2};
            CAR_TRY(car_ray_mid(pl, offs, nts, nch, pl + p.mid_bias, pl + p.chain_scale, layers, 3, ws + w.ebar, kC, ws + w.z1, ws + w.uh, BR, stream));
        }
        {   // a15, per sample: second-round query and logits
            Stage stage("round2_logits", st);
            // no 128-wide query rows exist on this route: <q2, qry> is a bilinear form of two hidden vectors both made from g (car_round2.hip)
            CAR_TRY(car_round2_logits_from_g(ws + w.g, ws + w.uh, pl + p.r2qw, pl + p.r2qb, b, V, R, P, ws + w.logit2, stream));
        }
        {
            Stage stage("attend_2", st);
            CAR_TRY(car_attend(ws + w.logit2, nullptr, kD, ws + w.e, kC, b, V, R, P, nullptr, 0.0f, ws + w.at_wt2, ws + w.ebar, kC, 1, nullptr,
                               nullptr, nullptr, nullptr, stream));
        }
    } else if (hipMemsetAsync(ws + w.z1, 0, sizeof(float) * BR * kE, st) != hipSuccess) {       // no second round: z = Wv ebar1 + bv
        car_set_error("car_render_forward: memset failed");
        return CAR_E_LAUNCH;
    }
    {   // z = (Wv ebar + bv) + V z1 (models.py:561-565), light-field decoder (resnet_block_fc.py:132-168), valid mask / white background
        Stage stage("ray_layers_2", st);
        nch = 0;
        chunks(p.latent_value, 18, 9); chunks(p.lin_in, 1, 4);
        for (int i = 0; i < kBlocks; ++i) { chunks(p.lz_c[i], 9, 4); chunks(p.fc0_c[i], 4, 4); chunks(p.fc1_c[i], 4, 4); }
        chunks(p.lout_c, 4, 1);
        const int layers[12] = {0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13};
        CAR_TRY(car_ray_tail(pl, offs, nts, nch, pl + p.tail_bias, pl + p.chain_scale, layers, 12, ws + w.ebar, kC, ws + w.phi_x, kPhiLd, ws + w.z1,
                             ws + w.rays, b, V, R, out->rgb, valid, stream));
    }
    return CAR_OK;
}

extern "C" int car_render_forward(const car_dims* dims, const void* plan, const car_inputs* in, const car_outputs* out,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    return render_phases(dims, plan, in, out, workspace, workspace_bytes, CAR_PHASE_SAMPLES | CAR_PHASE_RAYS, stream);
}

extern "C" int car_render_forward_phase(const car_dims* dims, const void* plan, const car_inputs* in, const car_outputs* out,
                                        void* workspace, size_t workspace_bytes, int phases, void* stream) {
    const int which = phases & ~CAR_PHASE_ROWS_FIRST_ROUND;
    CAR_REQUIRE(which == CAR_PHASE_SAMPLES || which == CAR_PHASE_RAYS || which == (CAR_PHASE_SAMPLES | CAR_PHASE_RAYS),
                "car_render_forward_phase: phases = %d (CAR_PHASE_SAMPLES, CAR_PHASE_RAYS or both, optionally | CAR_PHASE_ROWS_FIRST_ROUND)", phases);
    return render_phases(dims, plan, in, out, workspace, workspace_bytes, phases, stream);
}
