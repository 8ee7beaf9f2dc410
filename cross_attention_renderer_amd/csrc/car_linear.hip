// car_linear.hip — fp32 MFMA kernel for the 1x1 convolutions / linear layers applied to every epipolar sample
// (reference models.py:333-341, 487-491, 529, 548, 553 and
// resnet_block_fc.py:53-62, 132-168).  These layers are ~99.9 % of the path's FLOPs, so this kernel is the
// one the MFMA roofline is quoted on.
//
// Formulation (CDNA4, wave64):  Y^T[N x M] = W[N x K] . X^T[K x M]  with  v_mfma_f32_32x32x2_f32
//   A operand = weights    lane l: W[32*tile + (l&31)][k(l>>5)]
//   B operand = samples    lane l: X[row0 + (l&31)][k(l>>5)]
//   C/D       = outputs    lane l holds sample (l&31), channels 32*tile + (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15
// so the *samples* stay lane-resident across a chain of layers (a later fused kernel can feed the accumulator
// registers of one layer straight back as the B operand of the next) and an output row is written as float4s.
//
// Because the summation index k may be permuted freely, a 32-wide K chunk is split by lane half: lanes 0-31
// take k = 32c + j, lanes 32-63 take k = 32c + 16 + j for MFMA step j = 0..15.  Each lane then reads its
// sample's 16 consecutive floats straight from global memory as 4 x global_load_dwordx4 (no LDS round trip for
// X), while the weights — shared by the four waves of a workgroup — are pre-packed on the host side
// (car_linear_pack) in exactly the order the A operand wants them and streamed L2 -> LDS with
// global_load_lds_dwordx4 (double buffered), then read back as one ds_read_b128 per four MFMAs.
//
// Work decomposition: workgroup = 4 waves = 128 samples x (NT x 32) output channels; every wave owns 32 samples
// and NT accumulator tiles (NT*16 VGPRs).  fp32 MFMA runs at the fp32 vector rate (64 cycles per 32x32x2), so
// the only thing that matters is keeping the matrix pipe issuing back to back; LDS/L2 bandwidth is <15 % busy.
// Bias is folded in as an extra input column k == K whose value is the constant 1.
#include "car_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int kChunkK = 32;            // K elements per chunk (16 MFMA steps)
constexpr int kTileFloats = 1024;      // packed floats per (chunk, tile): 4 x 64 lanes x float4
constexpr int kRowsPerBlock = 128;

__host__ __device__ inline int tiles_per_block(int tiles) { return tiles >= 9 ? 9 : (tiles >= 3 ? 4 : (tiles == 2 ? 2 : 1)); }

template <int NT, bool GLDS>
__global__ void __launch_bounds__(256, (NT <= 9 ? 2 : 1))
linear_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ Wp, int K, int N, int tiles_alloc,
              int chunks, float* __restrict__ Y, int ldy, long M, int flags) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][NT][1024]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 31, h = lane >> 5;
    const long row = (long)blockIdx.x * kRowsPerBlock + wave * 32 + s;
    const long lrow = row < M ? row : M - 1;                         // clamp loads, mask stores
    const float* xrow = X + lrow * ldx;
    const int tile0 = blockIdx.y * NT;
    const bool relu_in = (flags & CAR_LIN_RELU_IN) != 0;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // raw 16-float slice of this lane's sample for chunk c (address clamped inside the row: branch-free; the
    // values of columns k >= K are replaced in fix_x, so whatever is read there does not matter)
    auto load_x = [&](int c, float4 (&xv)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int k0 = kChunkK * c + 16 * h + 4 * q;
            k0 = k0 < ldx - 4 ? k0 : ldx - 4;
            xv[q] = *reinterpret_cast<const float4*>(xrow + k0);
        }
    };
    // input activation, bias column (k == K -> 1) and zero padding (k > K); applied when the slice is consumed
    auto fix_x = [&](int c, float4 (&xv)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = kChunkK * c + 16 * h + 4 * q;
            float e[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + i;
                float t = e[i];
                if (relu_in) t = fmaxf(t, 0.0f);
                e[i] = k < K ? t : (k == K ? 1.0f : 0.0f);
            }
            xv[q] = make_float4(e[0], e[1], e[2], e[3]);
        }
    };

    float4 wreg[NT];   // only used by the register-staged (non-GLDS) variant
    auto stage_issue = [&](int c, int buf) {
        const float* src = Wp + ((long)c * tiles_alloc + tile0) * kTileFloats;
        float* dst = lds + buf * NT * kTileFloats;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int f4 = t * 256 + tid;
            if constexpr (GLDS) {
                // LDS-DMA: destination = M0 (wave-uniform LDS byte address) + lane*16, the global source is per lane.
                // Issued through inline asm on purpose: hipcc orders every ds_read behind an LDS-DMA it knows about
                // (s_waitcnt vmcnt(0) before the first ds_read of the chunk), which would serialise the weight stream
                // with the MFMAs.  Hidden from the compiler, the DMA for chunk c+1 flies under the MFMAs of chunk c;
                // the explicit vmcnt(0) + barrier at the end of the chunk orders it before the next chunk's reads.
                const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(uintptr_t)(lds_void*)(dst + 4 * (t * 256 + wave * 64)));
                const float* gsrc = src + 4 * f4;
                CAR_BOUNDS_TRAP(gsrc >= Wp && gsrc + 4 <= Wp + (long)chunks * tiles_alloc * kTileFloats);     // debug build: inside the packed layer
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
            } else {
                wreg[t] = *reinterpret_cast<const float4*>(src + 4 * f4);
            }
        }
    };
    auto stage_commit = [&](int buf) {
        if constexpr (!GLDS) {
            float* dst = lds + buf * NT * kTileFloats;
#pragma unroll
            for (int t = 0; t < NT; ++t) *reinterpret_cast<float4*>(dst + 4 * (t * 256 + tid)) = wreg[t];
        }
    };

    float4 xr[4], xn[4];
    stage_issue(0, 0);
    load_x(0, xr);
    stage_commit(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fix_x(0, xr);

    for (int c = 0; c < chunks; ++c) {
        const int cur = c & 1;
        const bool more = (c + 1) < chunks;
        if (more) {
            stage_issue(c + 1, cur ^ 1);
            load_x(c + 1, xn);
        }
        const float* wl = lds + cur * NT * kTileFloats + 4 * lane;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
            const float xe[4] = {xr[j4].x, xr[j4].y, xr[j4].z, xr[j4].w};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 a = *reinterpret_cast<const float4*>(wl + (t * 4 + j4) * 256);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xe[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xe[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xe[2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xe[3], acc[t], 0, 0, 0);
            }
        }
        if (more) {
            stage_commit(cur ^ 1);
            fix_x(c + 1, xn);
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[q] = xn[q];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // epilogue: lane (s, h) holds, for tile t and g = 0..3, channels 32*(tile0+t) + 8g + 4h .. +3 of its sample
    if (row < M) {
        const bool relu_out = (flags & CAR_LIN_RELU_OUT) != 0;
        const bool accum = (flags & CAR_LIN_ACCUM) != 0;
        const bool vec_ok = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
        float* yrow = Y + row * ldy;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = 32 * (tile0 + t) + 8 * g + 4 * h;
                if (n0 >= N) continue;
                float v[4] = {acc[t][4 * g + 0], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                if (vec_ok && n0 + 4 <= N) {
                    if (accum) {
                        const float4 o = *reinterpret_cast<const float4*>(yrow + n0);
                        v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
                    }
                    if (relu_out) { v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f); }
                    *reinterpret_cast<float4*>(yrow + n0) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (n0 + i < N) {
                            float o = v[i];
                            if (accum) o += yrow[n0 + i];
                            if (relu_out) o = fmaxf(o, 0.0f);
                            yrow[n0 + i] = o;
                        }
                    }
                }
            }
        }
    }
}

__global__ void pack_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ bias, int K, int N,
                            int tiles_alloc, long total, float* __restrict__ packed) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3);
        const int lane = (int)((idx >> 2) & 63);
        const int j4 = (int)((idx >> 8) & 3);
        const long ct = idx >> 10;                       // chunk * tiles_alloc + tile
        const int tile = (int)(ct % tiles_alloc);
        const int chunk = (int)(ct / tiles_alloc);
        const int n = 32 * tile + (lane & 31);
        const int k = kChunkK * chunk + 16 * (lane >> 5) + 4 * j4 + e;
        float v = 0.0f;
        if (n < N) {
            if (k < K) v = W[(long)n * ldw + k];
            else if (k == K && bias) v = bias[n];
        }
        packed[idx] = v;
    }
}

struct PackGeom { int chunks, tiles, nt, tiles_alloc; };
PackGeom pack_geom(int K, int N) {
    PackGeom g;
    g.chunks = (K + 1 + kChunkK - 1) / kChunkK;
    g.tiles = (N + 31) / 32;
    g.nt = tiles_per_block(g.tiles);
    g.tiles_alloc = ((g.tiles + g.nt - 1) / g.nt) * g.nt;
    return g;
}

template <int NT>
int launch_linear(const float* X, int ldx, const float* packed, int K, int N, const PackGeom& g, float* Y, int ldy,
                  long M, int flags, hipStream_t stream) {
    const dim3 grid(car_div_up(M, kRowsPerBlock), g.tiles_alloc / NT);
    const size_t lds_bytes = (size_t)2 * NT * kTileFloats * sizeof(float);
    (void)hipGetLastError();
    if (flags & CAR_LIN_NO_GLDS)
        hipLaunchKernelGGL((linear_kernel<NT, false>), grid, dim3(256), lds_bytes, stream, X, ldx, packed, K, N,
                           g.tiles_alloc, g.chunks, Y, ldy, M, flags);
    else
        hipLaunchKernelGGL((linear_kernel<NT, true>), grid, dim3(256), lds_bytes, stream, X, ldx, packed, K, N,
                           g.tiles_alloc, g.chunks, Y, ldy, M, flags);
    CAR_CHECK_LAUNCH("car_linear");
    return CAR_OK;
}

}  // namespace

extern "C" size_t car_linear_packed_floats(int K, int N) {
    if (K <= 0 || N <= 0) return 0;
    const PackGeom g = pack_geom(K, N);
    return (size_t)g.chunks * g.tiles_alloc * kTileFloats;
}

extern "C" int car_linear_pack(const float* W, int ldw, const float* bias, int K, int N, float* packed, void* stream) {
    CAR_REQUIRE(W && packed, "car_linear_pack: null pointer");
    CAR_REQUIRE(K > 0 && N > 0 && ldw >= K, "car_linear_pack: bad sizes K=%d N=%d ldw=%d", K, N, ldw);
    const PackGeom g = pack_geom(K, N);
    const long total = (long)g.chunks * g.tiles_alloc * kTileFloats;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    (void)hipGetLastError();
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, ldw, bias, K, N, g.tiles_alloc,
                       total, packed);
    CAR_CHECK_LAUNCH("car_linear_pack");
    return CAR_OK;
}

extern "C" int car_linear(const float* X, int ldx, const float* packed, int K, int N, float* Y, int ldy, long M,
                          int flags, void* stream) {
    CAR_REQUIRE(X && packed && Y, "car_linear: null pointer");
    CAR_REQUIRE(K > 0 && N > 0 && M > 0, "car_linear: bad sizes M=%ld K=%d N=%d", M, K, N);
    CAR_REQUIRE(ldx >= K && ldx % 4 == 0 && ((uintptr_t)X & 15) == 0,
                "car_linear: X needs a 16-byte aligned base and a row stride (%d) that is a multiple of 4 and >= K (%d)", ldx, K);
    CAR_REQUIRE(ldy >= N, "car_linear: ldy (%d) < N (%d)", ldy, N);
    const PackGeom g = pack_geom(K, N);
    hipStream_t st = (hipStream_t)stream;
    switch (g.nt) {
        case 9: return launch_linear<9>(X, ldx, packed, K, N, g, Y, ldy, M, flags, st);
        case 4: return launch_linear<4>(X, ldx, packed, K, N, g, Y, ldy, M, flags, st);
        case 2: return launch_linear<2>(X, ldx, packed, K, N, g, Y, ldy, M, flags, st);
        default: return launch_linear<1>(X, ldx, packed, K, N, g, Y, ldy, M, flags, st);
    }
}
