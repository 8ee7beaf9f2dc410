// wgrad_probe.hip — development microbenchmark (not product): where does wgrad16_kernel (csrc/car_backward.hip) spend its block time?
// The product kernel's body with timing-only ablations, on the training step's widest layer (dW = dY^T X, 589 824 rows, 576 x 580):
//   MODE 0 product;  1 the tiles of one row slab on ONE XCD (L2 serves the re-reads);  2 no global loads;  3 no hi / lo conversion
//   (raw bits to LDS);  4 no MFMAs;  5 = 1 + 2 (upper bound of what the memory side can give).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/wgrad_probe.hip -o tools/_dev/wgrad_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <type_traits>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
constexpr int kW16Rows = 32, kW16N = 192, kW16K = 320, kW16Threads = 512;
constexpr int kW16QuadsA = kW16N / 4, kW16Quads = (kW16N + kW16K) / 4;
constexpr int kW16PlaneA = kW16N * kW16Rows / 2, kW16PlaneB = kW16K * kW16Rows / 2;
constexpr int kW16Buf = 2 * (kW16PlaneA + kW16PlaneB);

template <int MODE>
__global__ void __launch_bounds__(kW16Threads) wgrad16_probe(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, long M, int N,
                                                             int K, int relu_x, long slab_rows, float* __restrict__ dW, int lddw, float* __restrict__ db,
                                                             int gx, int gy, int gz) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 2, wk = wave & 3;
    int bx, by, bz;
    if (MODE == 1 || MODE == 5) {
        // workgroup id b runs on XCD b % 8: the gx gy tiles of a slab take consecutive slots of one XCD
        const int b = blockIdx.x, xcd = b & 7, j = b >> 3, tiles = gx * gy;
        const int t = j % tiles;
        bz = (j / tiles) * 8 + xcd;
        bx = t % gx; by = t / gx;
        if (bz >= gz) return;
    } else { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; }
    const int n0 = bx * kW16N, k0 = by * kW16K;
    const int Kb = db ? K + 1 : K;
    const long m_begin = (long)bz * slab_rows;
    const long m_end = m_begin + slab_rows < M ? m_begin + slab_rows : M;
    const int q = tid >> 2, o = tid & 3;
    const bool isA = q < kW16QuadsA;
    const int c = isA ? 4 * q : 4 * (q - kW16QuadsA);
    const float* src = isA ? dY : X;
    const int ld = isA ? ldy : ldx, col0 = isA ? n0 + c : k0 + c;
    const int colc = col0 < ld - 3 ? col0 : 0;
    wf32x4 stage[8];
    auto fetch = [&](long m) {
        if (MODE == 2 || MODE == 5) return;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const long row = m + 8 * o + r < m_end ? m + 8 * o + r : m_end - 1;
            stage[r] = *reinterpret_cast<const wf32x4*>(src + row * ld + colc);
        }
    };
    auto commit = [&](int buf, long m) {
        float* plane = lds + buf * kW16Buf + (isA ? 0 : 2 * kW16PlaneA);
        const int plane_floats = isA ? kW16PlaneA : kW16PlaneB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = col0 + i;
            unsigned hi[8], lo[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float v = stage[r][i];
                if (MODE == 3) { hi[r] = __float_as_uint(v) >> 16; lo[r] = __float_as_uint(v) & 0xffffu; continue; }
                const bool in_rows = m + 8 * o + r < m_end;
                if (isA) { if (!in_rows || col0 >= ld - 3 || col >= N) v = 0.0f; }
                else if (col >= K) v = (col == K && db && in_rows) ? 1.0f : 0.0f;
                else if (!in_rows || col0 >= ld - 3) v = 0.0f;
                else if (relu_x) v = fmaxf(v, 0.0f);
                hi[r] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v);
                const float res = v - __uint_as_float(hi[r] << 16);
                lo[r] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)res);
            }
            const wu32x4 h4 = {hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16)};
            const wu32x4 l4 = {lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16)};
            float* dst = plane + ((c + i) * kW16Rows + 8 * o) / 2;
            *reinterpret_cast<wu32x4*>(dst) = h4;
            *reinterpret_cast<wu32x4*>(dst + plane_floats) = l4;
        }
    };
    if (MODE == 2 || MODE == 5) {
#pragma unroll
        for (int r = 0; r < 8; ++r) stage[r] = wf32x4{0.5f + tid, 0.25f, -1.5f, 3.0f * r};
    }
    wf32x4 acc[6][5];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};
    const int nl = lane & 15, g = lane >> 4;
    if (m_begin < m_end) { fetch(m_begin); commit(0, m_begin); }
    __syncthreads();
    int buf = 0;
    for (long m = m_begin; m < m_end; m += kW16Rows) {
        const bool more = m + kW16Rows < m_end;
        if (more) fetch(m + kW16Rows);
        const float* pa = lds + buf * kW16Buf + ((wn * 96 + nl) * kW16Rows + 8 * g) / 2;
        const float* pb = lds + buf * kW16Buf + 2 * kW16PlaneA + ((wk * 80 + nl) * kW16Rows + 8 * g) / 2;
        bf16x8 bh[5], bl[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pb + j * 16 * kW16Rows / 2));
            bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pb + kW16PlaneB + j * 16 * kW16Rows / 2));
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + i * 16 * kW16Rows / 2));
            const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + kW16PlaneA + i * 16 * kW16Rows / 2));
            if (MODE == 4) {
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const wf32x4 a4 = __builtin_bit_cast(wf32x4, ah), a5 = __builtin_bit_cast(wf32x4, al);
                    const wf32x4 b4 = __builtin_bit_cast(wf32x4, bh[j]), b5 = __builtin_bit_cast(wf32x4, bl[j]);
                    acc[i][j][0] += a4[0] + b4[0] + a5[1] + b5[1];
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) commit(buf ^ 1, m + kW16Rows);
        __syncthreads();
        buf ^= 1;
    }
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

// ---- candidate: branch-free staging.  Column validity / the bias column are two words per column set up once (v = (raw & keep) | one), the
// row tests exist only in the slab's last block (FULL = false), hi / lo pairs come out of v_cvt_pk_bf16_f32 two rows at a time, and the LDS
// image's columns are permuted inside aligned groups of four (col ^ ((col >> 2) & 3)) so that one ds_write_b128 of a wave covers all banks.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float wf32x2 __attribute__((ext_vector_type(2)));
template <bool RELU, bool FULL>
__device__ __forceinline__ void w16_commit(const wf32x4 (&stage)[8], float* dst, int plane_floats, const unsigned (&keep)[4], const unsigned (&one)[4],
                                           const int (&slot)[4], float floor, long rows_left) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h[4], l[4];
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            float v0 = stage[2 * rp][i], v1 = stage[2 * rp + 1][i];
            if (RELU) { v0 = fmaxf(v0, floor); v1 = fmaxf(v1, floor); }
            v0 = __uint_as_float((__float_as_uint(v0) & keep[i]) | one[i]);
            v1 = __uint_as_float((__float_as_uint(v1) & keep[i]) | one[i]);
            if (!FULL) { if (2 * rp >= rows_left) v0 = 0.0f; if (2 * rp + 1 >= rows_left) v1 = 0.0f; }
            const bf16x2 hp = __builtin_convertvector(wf32x2{v0, v1}, bf16x2);
            h[rp] = __builtin_bit_cast(unsigned, hp);
            const float r0 = v0 - __uint_as_float(h[rp] << 16), r1 = v1 - __uint_as_float(h[rp] & 0xffff0000u);      // exact
            const bf16x2 lp = __builtin_convertvector(wf32x2{r0, r1}, bf16x2);
            l[rp] = __builtin_bit_cast(unsigned, lp);
        }
        *reinterpret_cast<wu32x4*>(dst + slot[i]) = wu32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<wu32x4*>(dst + slot[i] + plane_floats) = wu32x4{l[0], l[1], l[2], l[3]};
    }
}

template <bool RELU, int VAR>
__global__ void __launch_bounds__(kW16Threads) wgrad16_v2(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, long M, int N,
                                                          int K, long slab_rows, float* __restrict__ dW, int lddw, float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 2, wk = wave & 3;
    const int n0 = blockIdx.x * kW16N, k0 = blockIdx.y * kW16K;
    const int Kb = db ? K + 1 : K;
    const long m_begin = (long)blockIdx.z * slab_rows;
    const long m_end = m_begin + slab_rows < M ? m_begin + slab_rows : M;
    const int q = tid >> 2, o = tid & 3;
    const bool isA = q < kW16QuadsA;
    const int c = isA ? 4 * q : 4 * (q - kW16QuadsA);
    const float* src = isA ? dY : X;
    const int ld = isA ? ldy : ldx, col0 = isA ? n0 + c : k0 + c;
    const bool readable = col0 < ld - 3;
    const int colc = readable ? col0 : 0;
    unsigned keep[4], one[4];
    int slot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = col0 + i;
        const bool mem = readable && (isA ? col < N : col < K);
        keep[i] = mem ? 0xffffffffu : 0u;
        one[i] = (!isA && col == K && db) ? 0x3f800000u : 0u;
        slot[i] = ((c + (i ^ ((c >> 2) & 3))) * kW16Rows + 8 * o) / 2;
    }
    const float floor = isA ? -__builtin_inff() : 0.0f;
    float* plane0 = lds + (isA ? 0 : 2 * kW16PlaneA);
    const int plane_floats = isA ? kW16PlaneA : kW16PlaneB;
    wf32x4 stage[8];
    const float* p = src + (m_begin + 8 * o) * ld + colc;
    auto fetch_full = [&]() {
#pragma unroll
        for (int r = 0; r < 8; ++r) stage[r] = *reinterpret_cast<const wf32x4*>(p + (long)r * ld);
    };
    auto fetch_tail = [&](long m) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const long row = m + 8 * o + r < m_end ? m + 8 * o + r : m_end - 1;
            stage[r] = *reinterpret_cast<const wf32x4*>(src + row * ld + colc);
        }
    };
    wf32x4 acc[6][5];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};
    const int nl = lane & 15, g = lane >> 4;
    const int nlp = nl ^ ((nl >> 2) & 3);
    if (m_begin < m_end) {
        fetch_tail(m_begin);
        w16_commit<RELU, false>(stage, plane0, plane_floats, keep, one, slot, floor, m_end - (m_begin + 8 * o));
    }
    __syncthreads();
    int buf = 0;
    for (long m = m_begin; m < m_end; m += kW16Rows) {
        const long mn = m + kW16Rows;
        const bool more = mn < m_end, full = mn + kW16Rows <= m_end;
        p += (long)kW16Rows * ld;
        if (full) fetch_full();
        else if (more) fetch_tail(mn);
        const float* pa = lds + buf * kW16Buf + ((wn * 96 + nlp) * kW16Rows + 8 * g) / 2;
        const float* pb = lds + buf * kW16Buf + 2 * kW16PlaneA + ((wk * 80 + nlp) * kW16Rows + 8 * g) / 2;
        bf16x8 bh[5], bl[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pb + j * 16 * kW16Rows / 2));
            bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pb + kW16PlaneB + j * 16 * kW16Rows / 2));
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + i * 16 * kW16Rows / 2));
            const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + kW16PlaneA + i * 16 * kW16Rows / 2));
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
            }
        }
        if (full) w16_commit<RELU, true>(stage, plane0 + (buf ^ 1) * kW16Buf, plane_floats, keep, one, slot, floor, 8);
        else if (more) w16_commit<RELU, false>(stage, plane0 + (buf ^ 1) * kW16Buf, plane_floats, keep, one, slot, floor, m_end - (mn + 8 * o));
        __syncthreads();
        buf ^= 1;
    }
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

// ---- candidate 3: loads TWO blocks ahead.  A thread's 8 rows x 4 columns are staged as two half tasks of 4 rows; the half task of block
// b + 1 is converted and written (ds_write_b64 per column and half) between the MFMA rows of block b, and its registers are refilled at once
// with the same rows of block b + 2: every load has a whole iteration to land, with the 32 staging registers of before.  Waves 0-2 stage dY,
// waves 3-7 stage X (wave-uniform: operand base and stride live in scalar registers, a thread keeps ONE 32-bit offset).  Columns past the
// operand's edge are staged as they come (clamped address): they only reach outputs that are never written.
typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));
template <bool RELU, bool FULL>
__device__ __forceinline__ void w16_half(const wf32x4 (&st)[4], float* dst, int slot, int plane_floats, bool relu, const bool (&isone)[4], int rows_left) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h[2], l[2];
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            float v0 = st[2 * rp][i], v1 = st[2 * rp + 1][i];
            if (RELU && relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
            v0 = isone[i] ? 1.0f : v0;
            v1 = isone[i] ? 1.0f : v1;
            if (!FULL) { if (2 * rp >= rows_left) v0 = 0.0f; if (2 * rp + 1 >= rows_left) v1 = 0.0f; }
            const bf16x2 hp = __builtin_convertvector(wf32x2{v0, v1}, bf16x2);
            h[rp] = __builtin_bit_cast(unsigned, hp);
            const float r0 = v0 - __uint_as_float(h[rp] << 16), r1 = v1 - __uint_as_float(h[rp] & 0xffff0000u);
            const bf16x2 lp = __builtin_convertvector(wf32x2{r0, r1}, bf16x2);
            l[rp] = __builtin_bit_cast(unsigned, lp);
        }
        float* d = dst + (slot ^ (16 * i));                                             // column i of the quad: 16 floats on, permuted
        *reinterpret_cast<wu32x2*>(d) = wu32x2{h[0], h[1]};
        *reinterpret_cast<wu32x2*>(d + plane_floats) = wu32x2{l[0], l[1]};
    }
}

template <bool RELU, int VAR>
__global__ void __launch_bounds__(kW16Threads) wgrad16_v3(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, long M, int N,
                                                          int K, long slab_rows, float* __restrict__ dW, int lddw, float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;
    const int n0 = blockIdx.x * kW16N, k0 = blockIdx.y * kW16K;
    const int Kb = db ? K + 1 : K;
    const long m_begin = (long)blockIdx.z * slab_rows;
    const long m_end = m_begin + slab_rows < M ? m_begin + slab_rows : M;
    static_assert(kW16QuadsA * 4 % 64 == 0, "whole waves stage dY");
    const bool isA = wave < kW16QuadsA * 4 / 64;
    const int q = tid >> 2, o = tid & 3;
    const int c = isA ? 4 * q : 4 * (q - kW16QuadsA);
    const float* src = isA ? dY : X;
    const int ld = isA ? ldy : ldx, col0 = isA ? n0 + c : k0 + c;
    const int colc = col0 < ld - 3 ? col0 : 0;
    bool isone[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) isone[i] = !isA && db && col0 + i == K;
    const unsigned voff = 4u * (unsigned)(8 * o * ld + colc);                          // bytes
    // float index of (column c, rows 8 o ..) inside the operand's plane, the quad's columns permuted: c 16 + 4 o, xor 16 (q & 3) [xor 16 i per column]
    float* plane0 = lds + (isA ? 0 : 2 * kW16PlaneA);
    const int slot0 = (c * kW16Rows + 8 * o) / 2 ^ (16 * ((c >> 2) & 3));
    const int plane_floats = isA ? kW16PlaneA : kW16PlaneB;
    wf32x4 st[2][4];                                                                   // [half task][row]
    auto load_half = [&](int hf, long m0, bool full) {
        const char* blk = reinterpret_cast<const char*>(src + m0 * ld);
        if (full) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st[hf][r] = *reinterpret_cast<const wf32x4*>(blk + (voff + 4u * (unsigned)((4 * hf + r) * ld)));
        } else {
            const int left = (int)(m_end - m0) - 1;                                    // last valid row of the block
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 8 * o + 4 * hf + r < left ? 8 * o + 4 * hf + r : left;
                st[hf][r] = *reinterpret_cast<const wf32x4*>(blk + 4u * (unsigned)(row * ld + colc));
            }
        }
    };
    auto commit_half = [&](int hf, int buf, long m0, bool full) {
        float* dst = plane0 + buf * kW16Buf + 2 * hf;                                  // rows 4 hf .. of the octet: 8 bytes in
        if (full) w16_half<RELU, true>(st[hf], dst, slot0, plane_floats, !isA, isone, 4);
        else w16_half<RELU, false>(st[hf], dst, slot0, plane_floats, !isA, isone, (int)(m_end - m0) - (8 * o + 4 * hf));
    };
    wf32x4 acc[6][5];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};
    const int nl = lane & 15, g = lane >> 4;
    const int nlp = nl ^ ((nl >> 2) & 3);
    if (m_begin < m_end) {
        load_half(0, m_begin, false); load_half(1, m_begin, false);
        commit_half(0, 0, m_begin, false); commit_half(1, 0, m_begin, false);
        if (m_begin + kW16Rows < m_end) { load_half(0, m_begin + kW16Rows, false); load_half(1, m_begin + kW16Rows, false); }
    }
    __syncthreads();
    int buf = 0;
    for (long m = m_begin; m < m_end; m += kW16Rows) {
        const long m1 = m + kW16Rows, m2 = m + 2 * kW16Rows;
        const bool has1 = m1 < m_end, full1 = m1 + kW16Rows <= m_end, has2 = m2 < m_end, full2 = m2 + kW16Rows <= m_end;
        const float* pa = lds + buf * kW16Buf + ((wn * 96 + nlp) * kW16Rows + 8 * g) / 2;
        const float* pb = lds + buf * kW16Buf + 2 * kW16PlaneA + ((wk * 80 + nlp) * kW16Rows + 8 * g) / 2;
        bf16x8 bh[5], bl[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pb + j * 16 * kW16Rows / 2));
            bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pb + kW16PlaneB + j * 16 * kW16Rows / 2));
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + i * 16 * kW16Rows / 2));
            const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + kW16PlaneA + i * 16 * kW16Rows / 2));
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
            }
            if (i == (VAR == 1 ? 0 : 1) || i == (VAR == 1 ? 3 : 4)) {
                const int hf = i == (VAR == 1 ? 0 : 1) ? 0 : 1;
                if (has1) commit_half(hf, buf ^ 1, m1, full1);
                if (has2) load_half(hf, m2, full2);
            }
        }
        __syncthreads();
        buf ^= 1;
    }
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

// ---- candidate 4: candidate 3 with the iteration written as 18 groups of five independent MFMAs (one product kind of one row of blocks)
// and the staging work cut into eight column pieces placed between them behind scheduling fences, so that a wave's own vector work runs
// under its own MFMAs (the two waves of a SIMD leave every barrier in phase: without this both convert at the same time and the matrix
// pipe idles).  Whole iterations (blocks b + 1 and b + 2 complete) run this body; the slab's last two blocks run the flagged one.
template <bool RELU, bool FULL>
__device__ __forceinline__ void w16_piece(const wf32x4 (&st)[4], int i, float* dst, int slot, int plane_floats, float floor, bool isone, int rows_left) {
    unsigned h[2], l[2];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        float v0 = st[2 * rp][i], v1 = st[2 * rp + 1][i];
        if (RELU) { v0 = fmaxf(v0, floor); v1 = fmaxf(v1, floor); }
        v0 = isone ? 1.0f : v0;
        v1 = isone ? 1.0f : v1;
        if (!FULL) { if (2 * rp >= rows_left) v0 = 0.0f; if (2 * rp + 1 >= rows_left) v1 = 0.0f; }
        const bf16x2 hp = __builtin_convertvector(wf32x2{v0, v1}, bf16x2);
        h[rp] = __builtin_bit_cast(unsigned, hp);
        const float r0 = v0 - __uint_as_float(h[rp] << 16), r1 = v1 - __uint_as_float(h[rp] & 0xffff0000u);
        const bf16x2 lp = __builtin_convertvector(wf32x2{r0, r1}, bf16x2);
        l[rp] = __builtin_bit_cast(unsigned, lp);
    }
    float* d = dst + (slot ^ (4 * i));
    *reinterpret_cast<wu32x2*>(d) = wu32x2{h[0], h[1]};
    *reinterpret_cast<wu32x2*>(d + plane_floats) = wu32x2{l[0], l[1]};
}

template <bool RELU, int VAR>
__global__ void __launch_bounds__(kW16Threads) wgrad16_v4(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, long M, int N,
                                                          int K, long slab_rows, float* __restrict__ dW, int lddw, float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;
    const int n0 = blockIdx.x * kW16N, k0 = blockIdx.y * kW16K;
    const int Kb = db ? K + 1 : K;
    const long m_begin = (long)blockIdx.z * slab_rows;
    const long m_end = m_begin + slab_rows < M ? m_begin + slab_rows : M;
    static_assert(kW16QuadsA * 4 % 64 == 0, "whole waves stage dY");
    const bool isA = wave < kW16QuadsA * 4 / 64;
    const int q = tid >> 2, o = tid & 3;
    const int c = isA ? 4 * q : 4 * (q - kW16QuadsA);
    const float* src = isA ? dY : X;
    const int ld = isA ? ldy : ldx, col0 = isA ? n0 + c : k0 + c;
    const int colc = col0 < ld - 3 ? col0 : 0;
    bool isone[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) isone[i] = !isA && db && col0 + i == K;
    const unsigned voff = 4u * (unsigned)(8 * o * ld + colc);
    float* plane0 = lds + (isA ? 0 : 2 * kW16PlaneA);
    // LDS image of an operand half: blocks of 16 columns, inside a block [row octet g][column ^ g] 16-byte entries: lane (nl, g) of an MFMA operand
    // read takes entry 16 g + (nl ^ g) — every 16-lane group of a ds_read_b128 covers all 64 banks (the plain [column][octet] order is 2-way)
    const int slot0 = (c >> 4) * 256 + 64 * o + 4 * (c & 15) + 4 * o;
    const float floor = isA ? -__builtin_inff() : 0.0f;
    const int plane_floats = isA ? kW16PlaneA : kW16PlaneB;
    wf32x4 st[2][4];
    auto load_half = [&](int hf, long m0, bool full) {
        const char* blk = reinterpret_cast<const char*>(src + m0 * ld);
        if (full) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st[hf][r] = *reinterpret_cast<const wf32x4*>(blk + (voff + 4u * (unsigned)((4 * hf + r) * ld)));
        } else {
            const int left = (int)(m_end - m0) - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 8 * o + 4 * hf + r < left ? 8 * o + 4 * hf + r : left;
                st[hf][r] = *reinterpret_cast<const wf32x4*>(blk + 4u * (unsigned)(row * ld + colc));
            }
        }
    };
    auto piece = [&](int hf, int i, int buf, long m0, bool full) {
        float* dst = plane0 + buf * kW16Buf + 2 * hf;
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
    // one iteration: MFMAs of block m out of buffer `buf`; block m + 1 staged into the other buffer; loads of block m + 2
    auto iteration = [&](long m, auto whole) {
        constexpr bool kWhole = decltype(whole)::value;
        const long m1 = m + kW16Rows, m2 = m + 2 * kW16Rows;
        const bool has1 = kWhole || m1 < m_end, full1 = kWhole || m1 + kW16Rows <= m_end, has2 = kWhole || m2 < m_end, full2 = kWhole || m2 + kW16Rows <= m_end;
        const float* pa = lds + buf * kW16Buf + a_off;
        const float* pb = lds + buf * kW16Buf + b_off;
        bf16x8 bh[5], bl[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pb + j * 16 * kW16Rows / 2));
            bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pb + kW16PlaneB + j * 16 * kW16Rows / 2));
        }
        bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa));
        bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + kW16PlaneA));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            bf16x8 ahn = ah, aln = al;
            if (i + 1 < 6) {
                ahn = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + (i + 1) * 16 * kW16Rows / 2));
                aln = __builtin_bit_cast(bf16x8, *reinterpret_cast<const wf32x4*>(pa + kW16PlaneA + (i + 1) * 16 * kW16Rows / 2));
            }
#pragma unroll
            for (int kind = 0; kind < 3; ++kind) {
                const int grp = 3 * i + kind;                                          // 0 .. 17
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kind == 2 ? al : ah, kind == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
                if (kWhole && VAR != 2) __builtin_amdgcn_sched_barrier(0);
                // staging pieces: half task 0 behind groups 1-4, its reload behind 4; half task 1 behind 9-12, its reload behind 12
                const int hf = grp < 8 ? 0 : 1, pi = grp - (grp < 8 ? 1 : 9);
                if (pi >= 0 && pi < 4) {
                    if (has1) piece(hf, pi, buf ^ 1, m1, full1);
                    if (pi == 3 && has2) load_half(hf, m2, full2);
                    if (kWhole && VAR != 2) __builtin_amdgcn_sched_barrier(0);
                }
            }
            ah = ahn; al = aln;
        }
        __syncthreads();
        buf ^= 1;
    };
    long m = m_begin;
    for (; m + 3 * kW16Rows <= m_end; m += kW16Rows) iteration(m, std::true_type{});
    for (; m < m_end; m += kW16Rows) iteration(m, std::false_type{});
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

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
float run(const float* dY, int ldy, const float* X, int ldx, long M, int N, int K, float* dW, float* db, int slabs_override) {
    const int Kb = K + 1;
    const int gx = (N + kW16N - 1) / kW16N, gy = (Kb + kW16K - 1) / kW16K;
    long want = 256 / ((long)gx * gy);
    if (MODE == 1 || MODE == 5) want = want / 8 * 8;
    if (slabs_override) want = slabs_override;
    long slab = (M + want - 1) / want;
    slab = (slab + kW16Rows - 1) / kW16Rows * kW16Rows;
    const int gz = (int)((M + slab - 1) / slab);
    const size_t lds_bytes = (size_t)2 * kW16Buf * sizeof(float);
    CK(hipFuncSetAttribute((const void*)wgrad16_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    dim3 grid = (MODE == 1 || MODE == 5) ? dim3(((gz + 7) / 8) * 8 * gx * gy) : dim3(gx, gy, gz);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int it = 0; it < 7; ++it) {
        CK(hipMemsetAsync(dW, 0, sizeof(float) * N * K, 0));
        CK(hipMemsetAsync(db, 0, sizeof(float) * N, 0));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(wgrad16_probe<MODE>, grid, dim3(kW16Threads), lds_bytes, 0, dY, ldy, X, ldx, M, N, K, 0, slab, dW, K, db, gx, gy, gz);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("  mode %d: %d slabs of %ld rows, grid %u x %u x %u: median %.3f ms, min %.3f ms\n", MODE, gz, slab, grid.x, grid.y, grid.z, ms[ms.size() / 2], ms[0]);
    return ms[ms.size() / 2];
}

template <bool RELU, int VAR>
float run2(const float* dY, int ldy, const float* X, int ldx, long M, int N, int K, float* dW, float* db, int slabs_override) {
    const int Kb = K + 1;
    const int gx = (N + kW16N - 1) / kW16N, gy = (Kb + kW16K - 1) / kW16K;
    long want = 256 / ((long)gx * gy);
    if (slabs_override) want = slabs_override;
    long slab = (M + want - 1) / want;
    slab = (slab + kW16Rows - 1) / kW16Rows * kW16Rows;
    const int gz = (int)((M + slab - 1) / slab);
    const size_t lds_bytes = (size_t)2 * kW16Buf * sizeof(float);
    CK(hipFuncSetAttribute((const void*)wgrad16_v2<RELU, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int it = 0; it < 7; ++it) {
        CK(hipMemsetAsync(dW, 0, sizeof(float) * N * K, 0));
        CK(hipMemsetAsync(db, 0, sizeof(float) * N, 0));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((wgrad16_v2<RELU, VAR>), dim3(gx, gy, gz), dim3(kW16Threads), lds_bytes, 0, dY, ldy, X, ldx, M, N, K, slab, dW, K, db);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("  v2<relu %d, var %d>: %d slabs of %ld rows: median %.3f ms, min %.3f ms\n", (int)RELU, VAR, gz, slab, ms[ms.size() / 2], ms[0]);
    return ms[ms.size() / 2];
}


template <bool RELU, int VAR>
float run3(const float* dY, int ldy, const float* X, int ldx, long M, int N, int K, float* dW, float* db, int slabs_override) {
    const int Kb = K + 1;
    const int gx = (N + kW16N - 1) / kW16N, gy = (Kb + kW16K - 1) / kW16K;
    long want = 256 / ((long)gx * gy);
    if (slabs_override) want = slabs_override;
    long slab = (M + want - 1) / want;
    slab = (slab + kW16Rows - 1) / kW16Rows * kW16Rows;
    const int gz = (int)((M + slab - 1) / slab);
    const size_t lds_bytes = (size_t)2 * kW16Buf * sizeof(float);
    CK(hipFuncSetAttribute((const void*)wgrad16_v3<RELU, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int it = 0; it < 7; ++it) {
        CK(hipMemsetAsync(dW, 0, sizeof(float) * N * K, 0));
        CK(hipMemsetAsync(db, 0, sizeof(float) * N, 0));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((wgrad16_v3<RELU, VAR>), dim3(gx, gy, gz), dim3(kW16Threads), lds_bytes, 0, dY, ldy, X, ldx, M, N, K, slab, dW, K, db);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("  v3<relu %d, var %d>: %d slabs of %ld rows: median %.3f ms, min %.3f ms\n", (int)RELU, VAR, gz, slab, ms[ms.size() / 2], ms[0]);
    return ms[ms.size() / 2];
}

template <bool RELU, int VAR>
float run4(const float* dY, int ldy, const float* X, int ldx, long M, int N, int K, float* dW, float* db, int slabs_override) {
    const int Kb = K + 1;
    const int gx = (N + kW16N - 1) / kW16N, gy = (Kb + kW16K - 1) / kW16K;
    long want = 256 / ((long)gx * gy);
    if (slabs_override) want = slabs_override;
    long slab = (M + want - 1) / want;
    slab = (slab + kW16Rows - 1) / kW16Rows * kW16Rows;
    const int gz = (int)((M + slab - 1) / slab);
    const size_t lds_bytes = (size_t)2 * kW16Buf * sizeof(float);
    CK(hipFuncSetAttribute((const void*)wgrad16_v4<RELU, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int it = 0; it < 7; ++it) {
        CK(hipMemsetAsync(dW, 0, sizeof(float) * N * K, 0));
        CK(hipMemsetAsync(db, 0, sizeof(float) * N, 0));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((wgrad16_v4<RELU, VAR>), dim3(gx, gy, gz), dim3(kW16Threads), lds_bytes, 0, dY, ldy, X, ldx, M, N, K, slab, dW, K, db);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("  v4<relu %d, var %d>: %d slabs of %ld rows: median %.3f ms, min %.3f ms\n", (int)RELU, VAR, gz, slab, ms[ms.size() / 2], ms[0]);
    return ms[ms.size() / 2];
}

void compare(const char* what, const float* ref, const float* dW, int N, int K) {
    std::vector<float> a((size_t)N * K), b((size_t)N * K);
    CK(hipMemcpy(a.data(), ref, sizeof(float) * N * K, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), dW, sizeof(float) * N * K, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < a.size(); ++i) { md = std::max(md, (double)fabsf(a[i] - b[i])); mx = std::max(mx, (double)fabsf(a[i])); }
    printf("  %s: max |diff| %.3e of max %.3e\n", what, md, mx);
}

int main() {
    const long M = 589824; const int N = 576, K = 579, ldx = 580;
    float *dY, *X, *dW, *db, *ref;
    CK(hipMalloc(&dY, sizeof(float) * M * N)); CK(hipMalloc(&X, sizeof(float) * M * ldx));
    CK(hipMalloc(&dW, sizeof(float) * N * K)); CK(hipMalloc(&db, sizeof(float) * N)); CK(hipMalloc(&ref, sizeof(float) * N * K));
    std::vector<float> h((size_t)M * ldx);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
    CK(hipMemcpy(X, h.data(), sizeof(float) * M * ldx, hipMemcpyHostToDevice));
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
    CK(hipMemcpy(dY, h.data(), sizeof(float) * M * N, hipMemcpyHostToDevice));
    printf("wgrad16 on M=%ld N=%d K=%d+bias\n", M, N, K);
    run<0>(dY, N, X, ldx, M, N, K, dW, db, 0);
    CK(hipMemcpy(ref, dW, sizeof(float) * N * K, hipMemcpyDeviceToDevice));
    run<1>(dY, N, X, ldx, M, N, K, dW, db, 0);
    {
        std::vector<float> a((size_t)N * K), b((size_t)N * K);
        CK(hipMemcpy(a.data(), ref, sizeof(float) * N * K, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), dW, sizeof(float) * N * K, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t i = 0; i < a.size(); ++i) { md = std::max(md, (double)fabsf(a[i] - b[i])); mx = std::max(mx, (double)fabsf(a[i])); }
        printf("  mode 1 vs mode 0: max |diff| %.3e of max %.3e\n", md, mx);
    }
    run<2>(dY, N, X, ldx, M, N, K, dW, db, 0);
    run<3>(dY, N, X, ldx, M, N, K, dW, db, 0);
    run<4>(dY, N, X, ldx, M, N, K, dW, db, 0);
    run<5>(dY, N, X, ldx, M, N, K, dW, db, 0);
    run2<false, 0>(dY, N, X, ldx, M, N, K, dW, db, 0);
    compare("v2 vs mode 0", ref, dW, N, K);
    run2<true, 0>(dY, N, X, ldx, M, N, K, dW, db, 0);
    run3<false, 0>(dY, N, X, ldx, M, N, K, dW, db, 0);
    compare("v3 vs mode 0", ref, dW, N, K);
    run3<false, 1>(dY, N, X, ldx, M, N, K, dW, db, 0);
    compare("v3 var 1 vs mode 0", ref, dW, N, K);
    run4<false, 0>(dY, N, X, ldx, M, N, K, dW, db, 0);
    compare("v4 vs mode 0", ref, dW, N, K);
    run4<false, 2>(dY, N, X, ldx, M, N, K, dW, db, 0);
    compare("v4 var 2 (no fences) vs mode 0", ref, dW, N, K);
    run4<true, 0>(dY, N, X, ldx, M, N, K, dW, db, 0);
    run4<false, 0>(dY, N, X, ldx, M - 13, N, K, dW, db, 0);
    CK(hipMemcpy(ref, dW, sizeof(float) * N * K, hipMemcpyDeviceToDevice));
    run<0>(dY, N, X, ldx, M - 13, N, K, dW, db, 0);
    compare("v4 vs mode 0, M - 13 rows", ref, dW, N, K);
    return 0;
}
