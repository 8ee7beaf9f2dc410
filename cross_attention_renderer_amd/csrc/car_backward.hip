// car_backward.hip — backward kernels of the staged render route (the reference trains with torch autograd over
// models.py:190-626, training.py:92-136).  Gradients flow to the renderer's parameters and to the feature pyramid z (from there,
// through torch autograd, into the encoder); the geometry (poses, rays, sample positions, pt, g) is not differentiated — none of it
// depends on a parameter.
//
//   car_linear_wgrad           dW += dY^T X, db += column sums of dY                       fp32 matrix pipe (v_mfma_f32_32x32x2_f32)
//   car_attend_backward        softmax-attention backward of one round (+ the depth read-out's term)      one workgroup per ray
//   car_gather_bilinear_backward   scatter-add of the gathered rows' gradients into the channel-last pyramid (grid_sample backward)
//   car_relu_mask / car_scale_rows / car_add / car_reduce_samples      the element-wise pieces between them
// The data gradients dX = dY W of the linear layers are car_linear itself with the transposed weight packed (car_linear_pack).
#include "car_common.h"
#include "car_geom.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// dW[n][k] += sum_m dY[m][n] X[m][k]  (k == K: the bias column, X = 1) on the fp32 matrix pipe: D += A B with A = dY^T (32 outputs x 2
// rows), B = X (2 rows x 32 inputs), lane l feeding dY[m + l / 32][n + l % 32] and X[m + l / 32][k + l % 32].  A workgroup tile is
// staged through LDS: 4 waves as 2 x 2 over a (2 TN 32) x (2 TK 32) tile of dW, every operand element of a 16-row block loaded ONCE
// per workgroup (float4 loads, coalesced) and read by two waves from LDS (one float per lane and MFMA operand: consecutive lanes,
// consecutive addresses).  The next block's loads are in flight under this block's MFMAs (registers -> the other LDS buffer).  The
// tile is added to dW with fp32 atomics (the row slabs of different workgroups meet there).  <3, 5>: a 192 x 320 tile, 240 accumulator
// registers per wave, for the wide layers (a 64 x 128 tile per wave, the first version, re-read the first point-MLP layer's operands
// 8352 floats per row = 19.7 GB per launch; 2320 here); <2, 2>: 128 x 128 for the 128-wide layers.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kWtRows = 16;           // rows per staged block (8 MFMA row pairs)

template <int TN, int TK>
__global__ void __launch_bounds__(256) wgrad_tile_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, long M, int N,
                                                         int K, int relu_x, long slab_rows, float* __restrict__ dW, int lddw, float* __restrict__ db) {
    constexpr int WN = 2 * TN * 32, WK = 2 * TK * 32;                 // workgroup tile
    constexpr int kA4 = kWtRows * WN / 4, kB4 = kWtRows * WK / 4;     // float4s per block
    constexpr int kPer = (kA4 + kB4 + 255) / 256;                     // float4s per thread and block
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [2][rows][WN] then [2][rows][WK]
    float* la = lds;
    float* lb = lds + 2 * kWtRows * WN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int n0 = blockIdx.x * WN, k0 = blockIdx.y * WK;
    const int Kb = db ? K + 1 : K;
    const long m_begin = (long)blockIdx.z * slab_rows;
    const long m_end = m_begin + slab_rows < M ? m_begin + slab_rows : M;
    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 stage[kPer];
    // float4 f of a block: the first kA4 belong to dY (row f / (WN / 4), columns 4 (f % (WN / 4)) ...), the rest to X.  fetch() only issues
    // the loads (from a clamped, always valid address); what a value must be replaced by — zeros past the edges, the bias column, relu — is
    // applied in commit(), after the block's MFMAs: touching the loaded registers here would put the memory latency in front of them
    // (it was 40 % of the kernel's time)
    auto fetch = [&](long m) {
#pragma unroll
        for (int p = 0; p < kPer; ++p) {
            const int f = tid + 256 * p;
            if (f < kA4) {
                const int r = f / (WN / 4), c = 4 * (f % (WN / 4));
                const long row = m + r < m_end ? m + r : m_end - 1;
                const int cc = n0 + c < ldy - 3 ? n0 + c : 0;
                stage[p] = *reinterpret_cast<const f32x4*>(dY + row * ldy + cc);
            } else if (f < kA4 + kB4) {
                const int g = f - kA4, r = g / (WK / 4), c = 4 * (g % (WK / 4));
                const long row = m + r < m_end ? m + r : m_end - 1;
                const int cc = k0 + c < ldx - 3 ? k0 + c : 0;
                stage[p] = *reinterpret_cast<const f32x4*>(X + row * ldx + cc);
            }
        }
    };
    auto commit = [&](int buf, long m) {
#pragma unroll
        for (int p = 0; p < kPer; ++p) {
            const int f = tid + 256 * p;
            f32x4 v = stage[p];
            if (f < kA4) {
                const int r = f / (WN / 4), c = 4 * (f % (WN / 4));
                const bool live = m + r < m_end && n0 + c < ldy - 3;
                for (int e = 0; e < 4; ++e) if (!live || n0 + c + e >= N) v[e] = 0.0f;
                *reinterpret_cast<f32x4*>(la + buf * kWtRows * WN + 4 * f) = v;
            } else if (f < kA4 + kB4) {
                const int g = f - kA4, r = g / (WK / 4), c = 4 * (g % (WK / 4));
                const bool in_rows = m + r < m_end, live = in_rows && k0 + c < ldx - 3;
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + c + e;
                    if (k >= K) v[e] = (k == K && db && in_rows) ? 1.0f : 0.0f;                // the bias column, then nothing
                    else if (!live) v[e] = 0.0f;
                    else if (relu_x) v[e] = fmaxf(v[e], 0.0f);
                }
                *reinterpret_cast<f32x4*>(lb + buf * kWtRows * WK + 4 * (f - kA4)) = v;
            }
        }
    };
    const int half = lane >> 5, col = lane & 31;
    if (m_begin < m_end) { fetch(m_begin); commit(0, m_begin); }
    __syncthreads();
    int buf = 0;
    for (long m = m_begin; m < m_end; m += kWtRows) {
        const bool more = m + kWtRows < m_end;
        if (more) fetch(m + kWtRows);
        const float* pa = la + buf * kWtRows * WN + wn * (TN * 32) + col;
        const float* pb = lb + buf * kWtRows * WK + wk * (TK * 32) + col;
        // the operands of step u + 1 are read while the MFMAs of step u run (one wave per SIMD: nobody else hides the LDS round trip)
        float a[2][TN], bq[2][TK];
#pragma unroll
        for (int i = 0; i < TN; ++i) a[0][i] = pa[half * WN + 32 * i];
#pragma unroll
        for (int j = 0; j < TK; ++j) bq[0][j] = pb[half * WK + 32 * j];
#pragma unroll
        for (int u = 0; u < kWtRows / 2; ++u) {
            if (u + 1 < kWtRows / 2) {
#pragma unroll
                for (int i = 0; i < TN; ++i) a[(u + 1) & 1][i] = pa[(2 * (u + 1) + half) * WN + 32 * i];
#pragma unroll
                for (int j = 0; j < TK; ++j) bq[(u + 1) & 1][j] = pb[(2 * (u + 1) + half) * WK + 32 * j];
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 1][i], bq[u & 1][j], acc[i][j], 0, 0, 0);
        }
        if (more) commit(buf ^ 1, m + kWtRows);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int k = k0 + wk * (TK * 32) + 32 * j + col;
            if (k >= Kb) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * (TN * 32) + 32 * i + 8 * (r >> 2) + 4 * half + (r & 3);
                if (n >= N) continue;
                const float v = acc[i][j][r];
                if (k < K) atomicAdd(dW + (long)n * lddw + k, v);
                else atomicAdd(db + n, v);
            }
        }
}

template <int TN, int TK>
int launch_wgrad_tile(const float* dY, int ldy, const float* X, int ldx, long M, int N, int K, int flags, float* dW, int lddw, float* db, void* stream) {
    constexpr int WN = 2 * TN * 32, WK = 2 * TK * 32;
    const int Kb = db ? K + 1 : K;
    const unsigned gx = car_div_up(N, WN), gy = car_div_up(Kb, WK);
    long want = 512 / ((long)gx * gy);                                 // ~2 workgroups per CU
    if (want < 1) want = 1;
    long slab = (M + want - 1) / want;
    // a workgroup adds its whole tile to dW with atomics whatever its slab: over few rows (the per-ray layers: 2304 rows a step) slabs of one
    // 16-row block made a launch 60 000 atomics per 16 rows — 72-141 us for a 0.8 GFLOP product; four blocks per workgroup at least
    if (slab < 4 * kWtRows) slab = 4 * kWtRows;
    slab = (slab + kWtRows - 1) / kWtRows * kWtRows;
    const unsigned gz = car_div_up(M, slab);
    CAR_REQUIRE(gz <= 65535, "car_linear_wgrad: too many row slabs");
    const size_t lds_bytes = (size_t)2 * kWtRows * (WN + WK) * sizeof(float);
    auto kern = wgrad_tile_kernel<TN, TK>;
    hipError_t e1 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e1 != hipSuccess) { car_set_error("car_linear_wgrad: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(256), lds_bytes, (hipStream_t)stream, dY, ldy, X, ldx, M, N, K, (flags & CAR_LIN_RELU_IN) ? 1 : 0, slab,
                       dW, lddw, db);
    CAR_CHECK_LAUNCH("car_linear_wgrad");
    return CAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix pipe (round 5; staging rebuilt in round 6): dW = dY^T X as three v_mfma_f32_16x16x32_bf16 products of
// bf16 hi / lo halves of BOTH operands (hi hi + hi lo + lo hi, fp32 accumulate).  Both operands are activations here — no pack-time scale to
// lean on — and bf16 keeps fp32's exponent, so no power of two has to be agreed on per launch, block or row: x = hi + lo + r with
// hi = bf16(x), lo = bf16(x - hi), both rounded to nearest, |r| <= 2^-18 |x|; the dropped lo lo term is 2^-18 relative.  Errors are
// rounding-like (unbiased), far inside the gradient tests' bound (2e-5 of the largest entry against fp64).
// A = dY^T (16 outputs n x 32 rows m), B = X (32 rows x 16 inputs k): lane (l & 15, l >> 4) feeds column l & 15 of its operand with the 8
// rows 8 (l >> 4) .. + 7 — a COLUMN of the row-major block, so the transpose happens when the block is staged: a thread takes 8 rows x 4
// columns from memory (8 float4, coalesced along the row) as two half tasks of 4 rows, splits them (v_cvt_pk_bf16_f32, two rows per
// instruction) and writes 8 bytes per column, half task and operand half into the LDS image.  512 threads = 8 waves as 2 x 4 over a
// 192 x 320 tile of dW (a wave: 96 x 80 = 6 x 5 blocks, 120 accumulator registers); waves 0-2 stage dY, waves 3-7 stage X (wave-uniform:
// operand base and stride sit in scalar registers, a thread keeps one 32-bit offset).
// Round 6 (tools/probes/wgrad_probe.hip, profiles/round6_wgrad16.md; 2.29 -> 1.18 ms on the first point-MLP layer):
//  * staging is branch-free: the slab's whole blocks run a body without row tests, columns past an operand's edge are staged as they
//    come (they only reach outputs that are never written), the bias column is one select per value;
//  * loads run TWO blocks ahead: a half task of block b + 1 is converted between the MFMAs of block b and its registers are refilled at
//    once with block b + 2, so every load has a whole iteration to land with the 32 staging registers of before;
//  * the iteration is 18 groups of five independent MFMAs (one product kind of one row of blocks) with the eight staging pieces placed
//    between them behind scheduling fences: a wave's vector work runs under its own MFMAs (the two waves of a SIMD leave every barrier in
//    phase; unfenced, both convert at the same time and the matrix pipe idles);
//  * LDS image of an operand half: blocks of 16 columns, inside a block [row octet g][column ^ g] 16-byte entries, so that lane (nl, g)
//    of an operand read takes entry 16 g + (nl ^ g) and every 16-lane group of a ds_read_b128 covers all 64 banks (the plain
//    [column][octet] order is two-way conflicted on gfx950's lane groups).
// LDS is double buffered (131 KB), one barrier per block.  Row slabs meet in dW with fp32 atomics, as in the fp32-pipe kernel.
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));
typedef float wf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));
constexpr int kW16Rows = 32, kW16N = 192, kW16K = 320, kW16Threads = 512;
constexpr int kW16QuadsA = kW16N / 4, kW16Quads = (kW16N + kW16K) / 4;                 // 48 + 80 column quads
static_assert(kW16Quads * (kW16Rows / 8) == kW16Threads, "one staging task per thread");
static_assert(kW16QuadsA * 4 % 64 == 0, "whole waves stage dY");
static_assert(kW16N % 16 == 0 && kW16K % 16 == 0, "16-column blocks");
constexpr int kW16PlaneA = kW16N * kW16Rows / 2, kW16PlaneB = kW16K * kW16Rows / 2;    // floats per (operand, half) plane
constexpr int kW16Buf = 2 * (kW16PlaneA + kW16PlaneB);                                // floats per buffer: A hi | A lo | B hi | B lo
template <bool B> struct W16Flag { static constexpr bool value = B; };

// rows 2 rp, 2 rp + 1 (rp = 0, 1) of a half task, column i of the quad: hi / lo pairs to the two planes
template <bool RELU, bool FULL>
__device__ __forceinline__ void w16_piece(const wf32x4 (&st)[4], int i, float* dst, int slot, int plane_floats, float floor, bool isone, int rows_left) {
    unsigned h[2], l[2];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        float v0 = st[2 * rp][i], v1 = st[2 * rp + 1][i];
        if (RELU) { v0 = fmaxf(v0, floor); v1 = fmaxf(v1, floor); }                   // floor = 0 for X, -inf for dY
        v0 = isone ? 1.0f : v0;                                                        // the bias column
        v1 = isone ? 1.0f : v1;
        if (!FULL) { if (2 * rp >= rows_left) v0 = 0.0f; if (2 * rp + 1 >= rows_left) v1 = 0.0f; }
        // both halves rounded to nearest: a truncated hi would leave residuals of x's own sign, and the dropped lo lo products a bias
        const bf16x2 hp = __builtin_convertvector(wf32x2{v0, v1}, bf16x2);
        h[rp] = __builtin_bit_cast(unsigned, hp);
        const float r0 = v0 - __uint_as_float(h[rp] << 16), r1 = v1 - __uint_as_float(h[rp] & 0xffff0000u);      // exact
        const bf16x2 lp = __builtin_convertvector(wf32x2{r0, r1}, bf16x2);
        l[rp] = __builtin_bit_cast(unsigned, lp);
    }
    float* d = dst + (slot ^ (4 * i));
    *reinterpret_cast<wu32x2*>(d) = wu32x2{h[0], h[1]};
    *reinterpret_cast<wu32x2*>(d + plane_floats) = wu32x2{l[0], l[1]};
}

template <bool RELU>
__global__ void __launch_bounds__(kW16Threads) wgrad16_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, long M, int N,
                                                              int K, long slab_rows, float* __restrict__ dW, int lddw, float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float lds[];                       // [2][kW16Buf]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;                                          // the wave's 96 x 80 corner of the tile
    const int n0 = blockIdx.x * kW16N, k0 = blockIdx.y * kW16K;
    const int Kb = db ? K + 1 : K;
    const long m_begin = (long)blockIdx.z * slab_rows;
    const long m_end = m_begin + slab_rows < M ? m_begin + slab_rows : M;
    // staging task: column quad q of [dY's 48 | X's 80], row octet o of the 32-row block
    const bool isA = wave < kW16QuadsA * 4 / 64;
    const int q = tid >> 2, o = tid & 3;
    const int c = isA ? 4 * q : 4 * (q - kW16QuadsA);                                 // first column inside the operand's tile
    const float* src = isA ? dY : X;
    const int ld = isA ? ldy : ldx, col0 = isA ? n0 + c : k0 + c;
    const int colc = col0 < ld - 3 ? col0 : 0;                                         // clamped: always a readable float4
    bool isone[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) isone[i] = !isA && db && col0 + i == K;
    const unsigned voff = 4u * (unsigned)(8 * o * ld + colc);                          // bytes, inside a block
    float* plane0 = lds + (isA ? 0 : 2 * kW16PlaneA);
    const int slot0 = (c >> 4) * 256 + 64 * o + 4 * (c & 15) + 4 * o;                  // entry 16 o + ((c & 15) ^ o) of block c / 16 [xor 4 i per column]
    const int plane_floats = isA ? kW16PlaneA : kW16PlaneB;
    const float floor = isA ? -__builtin_inff() : 0.0f;
    wf32x4 st[2][4];                                                                   // [half task][row]
    auto load_half = [&](int hf, long m0, bool full) {
        const char* blk = reinterpret_cast<const char*>(src + m0 * ld);
        if (full) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st[hf][r] = *reinterpret_cast<const wf32x4*>(blk + (voff + 4u * (unsigned)((4 * hf + r) * ld)));
        } else {
            const int left = (int)(m_end - m0) - 1;                                    // last row of the slab's partial block
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 8 * o + 4 * hf + r < left ? 8 * o + 4 * hf + r : left;
                st[hf][r] = *reinterpret_cast<const wf32x4*>(blk + 4u * (unsigned)(row * ld + colc));
            }
        }
    };
    auto piece = [&](int hf, int i, int buf, long m0, bool full) {
        float* dst = plane0 + buf * kW16Buf + 2 * hf;                                  // rows 4 hf .. of the octet: 8 bytes in
        if (full) w16_piece<RELU, true>(st[hf], i, dst, slot0, plane_floats, floor, isone[i], 4);
        else w16_piece<RELU, false>(st[hf], i, dst, slot0, plane_floats, floor, isone[i], (int)(m_end - m0) - (8 * o + 4 * hf));
    };
    wf32x4 acc[6][5];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};
    const int nl = lane & 15, g = lane >> 4;
    if (m_begin < m_end) {
        load_half(0, m_begin, false); load_half(1, m_begin, false);
#pragma unroll
        for (int i = 0; i < 4; ++i) { piece(0, i, 0, m_begin, false); piece(1, i, 0, m_begin, false); }
        if (m_begin + kW16Rows < m_end) { load_half(0, m_begin + kW16Rows, false); load_half(1, m_begin + kW16Rows, false); }
    }
    __syncthreads();
    int buf = 0;
    const int a_off = wn * 6 * 256 + 64 * g + 4 * (nl ^ g), b_off = 2 * kW16PlaneA + wk * 5 * 256 + 64 * g + 4 * (nl ^ g);
    // one iteration: the MFMAs of block m out of buffer `buf`; block m + 1 staged into the other buffer; the loads of block m + 2
    auto iteration = [&](long m, auto whole) {
        constexpr bool kWhole = decltype(whole)::value;
        const long m1 = m + kW16Rows, m2 = m + 2 * kW16Rows;
        const bool has1 = kWhole || m1 < m_end, full1 = kWhole || m1 + kW16Rows <= m_end, has2 = kWhole || m2 < m_end, full2 = kWhole || m2 + kW16Rows <= m_end;
        const float* pa = lds + buf * kW16Buf + a_off;
        const float* pb = lds + buf * kW16Buf + b_off;
        bf16x8 bh[5], bl[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pb + j * 256));
            bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pb + kW16PlaneB + j * 256));
        }
        bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa));
        bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + kW16PlaneA));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            bf16x8 ahn = ah, aln = al;
            if (i + 1 < 6) {
                ahn = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + (i + 1) * 256));
                aln = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + kW16PlaneA + (i + 1) * 256));
            }
#pragma unroll
            for (int kind = 0; kind < 3; ++kind) {                                     // hi hi, hi lo, lo hi: acc[i][j] takes them in this order
                const int grp = 3 * i + kind;                                          // 0 .. 17
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kind == 2 ? al : ah, kind == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
                if (kWhole) __builtin_amdgcn_sched_barrier(0);
                // staging pieces: half task 0 behind groups 1-4, its reload behind 4; half task 1 behind 9-12, its reload behind 12
                const int hf = grp < 8 ? 0 : 1, pi = grp - (grp < 8 ? 1 : 9);
                if (pi >= 0 && pi < 4) {
                    if (has1) piece(hf, pi, buf ^ 1, m1, full1);
                    if (pi == 3 && has2) load_half(hf, m2, full2);
                    if (kWhole) __builtin_amdgcn_sched_barrier(0);
                }
            }
            ah = ahn; al = aln;
        }
        __syncthreads();
        buf ^= 1;
    };
    long m = m_begin;
    for (; m + 3 * kW16Rows <= m_end; m += kW16Rows) iteration(m, W16Flag<true>{});
    for (; m < m_end; m += kW16Rows) iteration(m, W16Flag<false>{});
    // lane (nl, g) holds, of block (i, j): dW rows n = 16 i + 4 g + r (r = 0..3), column k = 16 j + nl
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = k0 + wk * 80 + 16 * j + nl;
            if (k >= Kb) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 96 + 16 * i + 4 * g + r;
                if (n >= N) continue;
                if (k < K) atomicAdd(dW + (long)n * lddw + k, acc[i][j][r]);
                else atomicAdd(db + n, acc[i][j][r]);
            }
        }
}

int launch_wgrad16(const float* dY, int ldy, const float* X, int ldx, long M, int N, int K, int flags, float* dW, int lddw, float* db, void* stream) {
    const int Kb = db ? K + 1 : K;
    const unsigned gx = car_div_up(N, kW16N), gy = car_div_up(Kb, kW16K);
    long want = 256 / ((long)gx * gy);                                 // one 8-wave workgroup per compute unit
    if (want < 1) want = 1;
    long slab = (M + want - 1) / want;
    slab = (slab + kW16Rows - 1) / kW16Rows * kW16Rows;
    const unsigned gz = car_div_up(M, slab);
    CAR_REQUIRE(gz <= 65535, "car_linear_wgrad: too many row slabs");
    CAR_REQUIRE((long)kW16Rows * (ldy > ldx ? ldy : ldx) < (1l << 29), "car_linear_wgrad: row stride too large for the 32-bit block offsets");
    const size_t lds_bytes = (size_t)2 * kW16Buf * sizeof(float);
    const bool relu = (flags & CAR_LIN_RELU_IN) != 0;
    auto kern = relu ? wgrad16_kernel<true> : wgrad16_kernel<false>;
    static bool reserved[2][64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !reserved[relu][dev]) {
        hipError_t e1 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e1 != hipSuccess) { car_set_error("car_linear_wgrad: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e1)); return CAR_E_LAUNCH; }
        if (dev >= 0 && dev < 64) reserved[relu][dev] = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(kW16Threads), lds_bytes, (hipStream_t)stream, dY, ldy, X, ldx, M, N, K, slab, dW, lddw, db);
    CAR_CHECK_LAUNCH("car_linear_wgrad");
    return CAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// One attention round, backward.  Forward (car_attend): w = softmax_s(logit_s) over the ray's V P samples, zbar = sum_s w_s val_s,
// depth = clamp((inv_q sum_s w_s clamp(pt_s, +-100)).z, 0, 10).  Given dz [b R, D] (gradient of zbar) and optionally ddepth [b R]:
//   a_s = <dz, val_s> + ddepth [0 < depth_pre < 10] inv_q[2, :3] . clamp(pt_s)          dlogit_s = w_s (a_s - sum_t w_t a_t)
//   dval_s (+)= w_s dz
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kMaxSamples = CAR_MAX_VIEWS * 256;

__global__ void __launch_bounds__(256) attend_bwd_kernel(const float* __restrict__ w_in, const float* __restrict__ val, int D, int b, int V, int R,
                                                         int P, const float* __restrict__ dz, int ld_dz, const float* __restrict__ ddepth,
                                                         const float* __restrict__ pt, const CarPose* __restrict__ poses, float* __restrict__ dval,
                                                         int accumulate, float* __restrict__ dlogit) {
    __shared__ float s_w[kMaxSamples], s_a[kMaxSamples];
    __shared__ float s_red[4];
    const int sc = blockIdx.x / R, r = blockIdx.x % R;
    const int S = V * P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = tid & 15, grp = tid >> 4;
    auto row_of = [&](int s) -> long { return ((long)(sc * V + s / P) * R + r) * P + (s % P); };
    const float* dzr = dz + ((long)sc * R + r) * ld_dz;
    for (int s = tid; s < S; s += 256) s_w[s] = w_in[row_of(s)];
    __syncthreads();
    // depth term: the pre-clamp depth of this ray and the row of inv_q that reads it out
    float dd = 0.0f, mz[3] = {0.f, 0.f, 0.f};
    if (ddepth) {
        float acc3[3] = {0.f, 0.f, 0.f};
        if (wave == 0) {
            for (int s = lane; s < S; s += 64) {
                const float* q = pt + row_of(s) * 3;
                for (int k = 0; k < 3; ++k) acc3[k] += s_w[s] * fminf(fmaxf(q[k], -100.0f), 100.0f);
            }
            for (int k = 0; k < 3; ++k) acc3[k] = wave_sum(acc3[k]);
            if (lane == 0) {
                const float* Mi = poses[sc * V].inv_q;
                const float zc = ((acc3[0] * Mi[8] + acc3[1] * Mi[9]) + acc3[2] * Mi[10]) + Mi[11];
                s_red[0] = (zc > 0.0f && zc < 10.0f) ? ddepth[(long)sc * R + r] : 0.0f;
            }
        }
        __syncthreads();
        dd = s_red[0];
        const float* Mi = poses[sc * V].inv_q;
        mz[0] = Mi[8]; mz[1] = Mi[9]; mz[2] = Mi[10];
        __syncthreads();
    }
    // a_s: 16 lanes per sample walk the row
    for (int s = grp; s < S; s += 16) {
        const long row = row_of(s);
        const float* v = val + row * D;
        float acc = 0.0f;
        for (int k = 4 * sub; k + 3 < D; k += 64) {
            const float4 x = *reinterpret_cast<const float4*>(v + k);
            const float4 g4 = *reinterpret_cast<const float4*>(dzr + k);
            acc += x.x * g4.x + x.y * g4.y + x.z * g4.z + x.w * g4.w;
        }
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);
        if (sub == 0) {
            if (dd != 0.0f) {
                const float* q = pt + row * 3;
                acc += dd * ((mz[0] * fminf(fmaxf(q[0], -100.0f), 100.0f) + mz[1] * fminf(fmaxf(q[1], -100.0f), 100.0f)) +
                             mz[2] * fminf(fmaxf(q[2], -100.0f), 100.0f));
            }
            s_a[s] = acc;
        }
    }
    __syncthreads();
    float t = 0.0f;
    for (int s = tid; s < S; s += 256) t += s_w[s] * s_a[s];
    t = wave_sum(t);
    if (lane == 0) s_red[wave] = t;
    __syncthreads();
    t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    for (int s = tid; s < S; s += 256) dlogit[row_of(s)] = s_w[s] * (s_a[s] - t);
    // dval_s (+)= w_s dz
    for (int s = grp; s < S; s += 16) {
        const float w = s_w[s];
        float* o = dval + row_of(s) * D;
        for (int k = 4 * sub; k + 3 < D; k += 64) {
            const float4 g4 = *reinterpret_cast<const float4*>(dzr + k);
            float4 y = accumulate ? *reinterpret_cast<const float4*>(o + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            y.x = fmaf(w, g4.x, y.x); y.y = fmaf(w, g4.y, y.y); y.z = fmaf(w, g4.z, y.z); y.w = fmaf(w, g4.w, y.w);
            *reinterpret_cast<float4*>(o + k) = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// grid_sample backward with respect to the maps: the gradient of the gathered row goes to its four taps of every level with the
// forward's weights (car_bilinear_taps; zeros padding: taps outside the map have weight 0, border: clamped taps).  Same placement
// rule for the row of a point as car_gather_bilinear.  fp32 atomics (a texel is shared by many samples).
// ---------------------------------------------------------------------------------------------------------------------
struct ScatterLevels {
    float* map[CAR_MAX_LEVELS];
    int c[CAR_MAX_LEVELS], h[CAR_MAX_LEVELS], w[CAR_MAX_LEVELS];
    int q0[CAR_MAX_LEVELS + 1];
    int n_levels;
};

__global__ void __launch_bounds__(256) gather_bwd_kernel(ScatterLevels L, int n_maps, const float* __restrict__ grid, long pts, int mode, int place,
                                                         int V, const float* __restrict__ dout, int ld_out, int col_out) {
    // one thread per (point, channel), consecutive lanes on consecutive channels: an atomic instruction then covers whole 128-byte lines
    // of a texel (with a float4 of channels per thread it touched every fourth word of eight lines: 7.3 ms per launch of the training shape against 1.8)
    const int cpr = 4 * L.q0[L.n_levels];
    const long total = (long)n_maps * pts * cpr;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cpr);
        const long mp = idx / cpr;
        const int m = (int)(mp / pts);
        const long i = mp % pts;
        long row;
        if (place == CAR_PLACE_PLAIN) row = mp;
        else if (place == CAR_PLACE_OWN) row = mp * V + (m % V);
        else { const int sc = m / 2, s = m % 2; row = (((long)(sc * 2 + (1 - s))) * pts + i) * 2 + s; }
        int l = 0;
        while (l + 1 < L.n_levels && c >= 4 * L.q0[l + 1]) ++l;
        int tidx[4];
        float tw[4];
        car_bilinear_taps(grid[2 * mp], grid[2 * mp + 1], L.w[l], L.h[l], mode, tidx, tw);
        const float g = dout[row * ld_out + col_out + c];
        float* base = L.map[l] + (long)m * L.h[l] * L.w[l] * L.c[l] + (c - 4 * L.q0[l]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (tw[t] != 0.0f) atomicAdd(base + (long)tidx[t] * L.c[l], tw[t] * g);
    }
}

// ---- element-wise pieces --------------------------------------------------------------------------------------------
__global__ void relu_mask_kernel(float* __restrict__ grad, int ldg, const float* __restrict__ act, int lda, long M, int N) {
    const long total = M * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / N;
        const int n = (int)(idx % N);
        if (!(act[m * lda + n] > 0.0f)) grad[m * ldg + n] = 0.0f;
    }
}
// out[m][:] (+)= scale * s[m / group] * x[m][:]
__global__ void scale_rows_kernel(float* __restrict__ out, int ldo, const float* __restrict__ x, int ldx, const float* __restrict__ s, long group,
                                  float scale, long M, int N, int accumulate) {
    if (((N | ldo | ldx) & 3) == 0 && (((uintptr_t)out | (uintptr_t)x) & 15) == 0) {       // whole float4s (the 128-wide key / query gradients)
        const int nq = N / 4;
        const long total4 = M * nq;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (long)gridDim.x * blockDim.x) {
            const long m = idx / nq;
            const int n = 4 * (int)(idx % nq);
            const float f = scale * s[m / group];
            const float4 v = *reinterpret_cast<const float4*>(x + m * ldx + n);
            float4* o = reinterpret_cast<float4*>(out + m * ldo + n);
            float4 r = make_float4(f * v.x, f * v.y, f * v.z, f * v.w);
            if (accumulate) { const float4 p = *o; r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w; }
            *o = r;
        }
        return;
    }
    const long total = M * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / N;
        const int n = (int)(idx % N);
        const float v = scale * s[m / group] * x[m * ldx + n];
        out[m * ldo + n] = accumulate ? out[m * ldo + n] + v : v;
    }
}
// out = alpha a + beta b  (b may be NULL)
__global__ void add_kernel(float* __restrict__ out, int ldo, const float* __restrict__ a, int lda, float alpha, const float* __restrict__ bq, int ldb,
                           float beta, long M, int N) {
    const long total = M * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / N;
        const int n = (int)(idx % N);
        float v = alpha * a[m * lda + n];
        if (bq) v += beta * bq[m * ldb + n];
        out[m * ldo + n] = v;
    }
}
// du[sc][ray][c] = sum over the V P samples of the ray of d[row][c]      (backward of the per-ray broadcast of car_add_ray_bias_relu)
__global__ void __launch_bounds__(256) reduce_samples_kernel(const float* __restrict__ d, int b, int V, int R, int P, int C, float* __restrict__ du) {
    const int sc = blockIdx.x / R, r = blockIdx.x % R;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.0f;
        for (int v = 0; v < V; ++v) {
            const float* base = d + (((long)(sc * V + v) * R + r) * P) * C + c;
            for (int p = 0; p < P; ++p) acc += base[(long)p * C];
        }
        du[((long)sc * R + r) * C + c] = acc;
    }
}

unsigned grid_for(long total) {
    const long blocks = (total + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : (blocks < 65536 ? blocks : 65536));
}

}  // namespace

extern "C" int car_linear_wgrad(const float* dY, int ldy, const float* X, int ldx, long M, int N, int K, int flags, float* dW, int lddw,
                                float* db, void* stream) {
    CAR_REQUIRE(dY && X && dW, "car_linear_wgrad: null pointer");
    CAR_REQUIRE(M > 0 && N > 0 && K > 0 && ldy >= N && ldx >= K && lddw >= K, "car_linear_wgrad: bad sizes M=%ld N=%d K=%d", M, N, K);
    CAR_REQUIRE(ldy % 4 == 0 && ldx % 4 == 0, "car_linear_wgrad: row strides must be multiples of 4 floats (got %d, %d)", ldy, ldx);
    // every tile kernel below stages its operands with 16-byte loads: a column-offset view whose first element is not 16-byte aligned
    // would fault on the device, so it is refused here
    CAR_REQUIRE((((uintptr_t)dY | (uintptr_t)X) & 15) == 0, "car_linear_wgrad: dY and X must be 16-byte aligned (a column-offset view? copy it first)");
    // wide layers over many rows: the bf16 x 3 kernel (CAR_WGRAD_FP32 in flags keeps them on the fp32 pipe: A/B and tests); wide layers
    // otherwise: the fp32 pipe's 192 x 320 workgroup tile; everything else its 128 x 128 one
    // (round 6: also the 128 x 16 layers over every sample (query_embed) — a sparsely filled 192 x 320 tile on the bf16 pipe still beats the fp32
    // pipe's 128 x 128 tile, 100 against 203 us over 294 912 rows)
    if ((N > 128 || K + (db ? 1 : 0) > 128 || (long)N * K >= 32 * 32) && M >= 2048 && !(flags & CAR_WGRAD_FP32))
        return launch_wgrad16(dY, ldy, X, ldx, M, N, K, flags, dW, lddw, db, stream);
    if (N > 128 || K + (db ? 1 : 0) > 128) return launch_wgrad_tile<3, 5>(dY, ldy, X, ldx, M, N, K, flags, dW, lddw, db, stream);
    return launch_wgrad_tile<2, 2>(dY, ldy, X, ldx, M, N, K, flags, dW, lddw, db, stream);
}

extern "C" int car_attend_backward(const float* w, const float* val, int D, int b, int V, int R, int P, const float* dz, int ld_dz,
                                   const float* ddepth, const float* pt, const float* poses, float* dval, int accumulate, float* dlogit,
                                   void* stream) {
    CAR_REQUIRE(w && val && dz && dval && dlogit, "car_attend_backward: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && V <= CAR_MAX_VIEWS && R > 0 && P > 0 && V * P <= kMaxSamples, "car_attend_backward: bad sizes");
    CAR_REQUIRE(D > 0 && D % 4 == 0 && ld_dz >= D && ld_dz % 4 == 0, "car_attend_backward: D = %d must be a multiple of 4", D);
    CAR_REQUIRE(!ddepth || (pt && poses), "car_attend_backward: ddepth needs pt and poses");
    (void)hipGetLastError();
    hipLaunchKernelGGL(attend_bwd_kernel, dim3((unsigned)((long)b * R)), dim3(256), 0, (hipStream_t)stream, w, val, D, b, V, R, P, dz, ld_dz, ddepth,
                       pt, (const CarPose*)poses, dval, accumulate, dlogit);
    CAR_CHECK_LAUNCH("car_attend_backward");
    return CAR_OK;
}

extern "C" int car_gather_bilinear_backward(float* const* dmaps, const int* level_c, const int* level_h, const int* level_w, int n_levels,
                                            int n_maps, const float* grid, long pts, int mode, int place, int V, const float* dout, int ld_out,
                                            int col_out, void* stream) {
    CAR_REQUIRE(dmaps && level_c && level_h && level_w && grid && dout, "car_gather_bilinear_backward: null pointer");
    CAR_REQUIRE(n_levels > 0 && n_levels <= CAR_MAX_LEVELS && n_maps > 0 && pts > 0, "car_gather_bilinear_backward: bad sizes");
    CAR_REQUIRE(mode == 0 || mode == 1, "car_gather_bilinear_backward: mode must be 0 (border) or 1 (zeros)");
    CAR_REQUIRE(place == CAR_PLACE_PLAIN || place == CAR_PLACE_OWN || (place == CAR_PLACE_OTHER2 && V == 2 && n_maps % 2 == 0),
                "car_gather_bilinear_backward: bad placement %d for V=%d", place, V);
    ScatterLevels L;
    L.n_levels = n_levels;
    int q = 0;
    for (int l = 0; l < n_levels; ++l) {
        CAR_REQUIRE(dmaps[l] && level_c[l] > 0 && level_c[l] % 4 == 0 && level_h[l] > 0 && level_w[l] > 0,
                    "car_gather_bilinear_backward: level %d needs a channel count that is a positive multiple of 4", l);
        L.map[l] = dmaps[l]; L.c[l] = level_c[l]; L.h[l] = level_h[l]; L.w[l] = level_w[l];
        L.q0[l] = q;
        q += level_c[l] / 4;
    }
    L.q0[n_levels] = q;
    for (int l = n_levels; l < CAR_MAX_LEVELS; ++l) { L.map[l] = nullptr; L.c[l] = L.h[l] = L.w[l] = 0; if (l > n_levels) L.q0[l] = q; }
    CAR_REQUIRE(ld_out % 4 == 0 && col_out % 4 == 0 && col_out >= 0 && col_out + 4 * q <= ld_out,
                "car_gather_bilinear_backward: window [%d,%d) must be float4-aligned inside a row of %d", col_out, col_out + 4 * q, ld_out);
    (void)hipGetLastError();
    hipLaunchKernelGGL(gather_bwd_kernel, dim3(grid_for((long)n_maps * pts * q * 4)), dim3(256), 0, (hipStream_t)stream, L, n_maps, grid, pts, mode, place, V,
                       dout, ld_out, col_out);
    CAR_CHECK_LAUNCH("car_gather_bilinear_backward");
    return CAR_OK;
}

extern "C" int car_relu_mask(float* grad, int ldg, const float* act, int lda, long M, int N, void* stream) {
    CAR_REQUIRE(grad && act && M > 0 && N > 0 && ldg >= N && lda >= N, "car_relu_mask: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid_for(M * N)), dim3(256), 0, (hipStream_t)stream, grad, ldg, act, lda, M, N);
    CAR_CHECK_LAUNCH("car_relu_mask");
    return CAR_OK;
}

extern "C" int car_scale_rows(float* out, int ldo, const float* x, int ldx, const float* s, long group, float scale, long M, int N, int accumulate,
                              void* stream) {
    CAR_REQUIRE(out && x && s && M > 0 && N > 0 && group > 0 && ldo >= N && ldx >= N, "car_scale_rows: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for(M * N)), dim3(256), 0, (hipStream_t)stream, out, ldo, x, ldx, s, group, scale, M, N, accumulate);
    CAR_CHECK_LAUNCH("car_scale_rows");
    return CAR_OK;
}

extern "C" int car_add(float* out, int ldo, const float* a, int lda, float alpha, const float* b, int ldb, float beta, long M, int N, void* stream) {
    CAR_REQUIRE(out && a && M > 0 && N > 0 && ldo >= N && lda >= N && (!b || ldb >= N), "car_add: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(M * N)), dim3(256), 0, (hipStream_t)stream, out, ldo, a, lda, alpha, b, ldb, beta, M, N);
    CAR_CHECK_LAUNCH("car_add");
    return CAR_OK;
}

extern "C" int car_reduce_samples(const float* d, int b, int V, int R, int P, int C, float* du, void* stream) {
    CAR_REQUIRE(d && du && b > 0 && V > 0 && R > 0 && P > 0 && C > 0, "car_reduce_samples: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(reduce_samples_kernel, dim3((unsigned)((long)b * R)), dim3(256), 0, (hipStream_t)stream, d, b, V, R, P, C, du);
    CAR_CHECK_LAUNCH("car_reduce_samples");
    return CAR_OK;
}
