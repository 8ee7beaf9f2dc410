// car_linear16.hip — one wide linear layer of the stage entries on the f16 matrix pipe (car_linear_x3): Y = act(act_in(X) W^T + b (+ Y)).
// The staged route (constructor variants, training forward, the backward's data gradients: reference models.py:333-341, 487-491, 529,
// 548, 553) spends most of its time in 576 -> 576 / 288 / 128 layers over every epipolar sample; car_linear runs them on the fp32 matrix
// pipe (157 TFLOP/s peak).  This kernel is the fused per-sample kernel's layer machinery (car_fused_mma.h) with the B operands read
// from the rows of X: fp16 hi / lo halves of both operands, three v_mfma_f32_16x16x32_f16 products per term, fp32 accumulation
// (fp32-class accuracy, 2500 / 3 TFLOP/s peak); the layer's weights carry a power of two chosen at pack time, every row of X a power of two
// that follows the largest magnitude seen so far along the row (when a later chunk outgrows it, the row's accumulators are moved to the
// new power of two — an exact multiplication — so X is read once, not twice), both undone exactly on the accumulators.
// A workgroup = 12 waves x 16 rows; a wave keeps NT tiles of 16 outputs (NT = 18: 288 columns, 72 accumulator registers; wider layers
// run as column groups); the weight chunks (K = 32 x the group's columns) stream L2 -> LDS by LDS-DMA, double buffered.
#include "car_common.h"
#include "car_geom.h"

namespace {

constexpr int kWaves = 12, kRows = 16, kGroupRows = kWaves * kRows;
constexpr int kThreads = 64 * kWaves;
constexpr int kPieces = 3;                         // LDS-DMA pieces of the widest chunk (18 tiles = 36 KB over 12 waves); narrower column groups
                                                   // take fewer (pieces_of): a piece index past the chunk would copy from beyond it
constexpr int pieces_of(int nt) { return (2 * nt + kWaves - 1) / kWaves; }

#include "car_fused_mma.h"

// The layer stream of this kernel addresses its chunks itself (chunk_desc below); the chunk table car_fused_mma.h asks for serves the KQ
// instance's tail (car_key_query_logits): a blob of K2 = chunks 0, 1 (two K steps x 8 tiles each) | Q1 = chunk 2 (8 tiles) | Q2 = chunks 3, 4
constexpr int kG_K2 = 0, kG_Q1 = 2, kG_Q2 = 3, kTailTiles = 72;
__device__ __forceinline__ constexpr int chunk_tile_offset(int g) { return g <= 0 ? 0 : g == 1 ? 16 : g == 2 ? 32 : g == 3 ? 40 : 56; }
__device__ __forceinline__ constexpr int chunk_tiles(int g) { return g == 2 ? 8 : 16; }
constexpr int kTailBias = 2 * kD + 16;             // bk2 [128] | bq2 [128] | 2^-shift of K2, Q1, Q2 | (pack-time scratch)

struct LinArgs {
    const float* X; int ldx;
    const float* Wp; int tiles_total;      // [K step][tiles_total][512]
    const float* bias;                     // N floats or null
    const float* down;                     // 2^-shift of the packed layer
    int K, chunks;
    float* Y; int ldy;
    long M;
    int flags;
    // GATHER (car_lattice_encode_linear): the rows of X are not read, they are MADE — row i = relu(four taps of the merged lattice of map
    // (row_src[i] & 0x3fffffff), padding mode (row_src[i] >> 30) & 1, at row_grid[i] + the point term of row_pe[i]): car_encode.hip's
    // encode_kernel, one 32-channel chunk at a time, straight into the B operands
    const float* lattice; int lh, lw, lpad; float sx, sy;
    const int* row_src; const float* row_grid; const float* row_pe; const float* wpt;
    // KQ (car_key_query_logits): the layer is key_map (relu behind it); its 128 outputs stay in the accumulators and run on through
    // key_map_2, while query_embed / query_embed_2 run on the row's 16-float geometric query g; out come qry [M,128] and logit [M]
    const float* g; const float* tail; const float* tail_bias; float* qry; float* logit;
    // car_linear_x3_masked (the data gradient of a layer that sits behind a ReLU): result columns whose activation is not positive are zero
    const float* mask; int ldm;
};

template <int NT, bool GATHER = false, bool KQ = false>
__global__ void __launch_bounds__(kThreads) linear16_kernel(const LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][NT][512] (+ GATHER: the [K][4] point / bias table)
    static_assert(kWaves * pieces_of(NT) <= 3 * 2 * NT, "stream_issue_piece wraps a piece index into the chunk with two subtractions");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, q4 = lane >> 4;
    // the column groups of a block of rows read the same rows of X: they sit on ONE XCD (workgroup b runs on XCD b % 8) in consecutive
    // slots, so the second reader finds the rows in that XCD's L2 — as neighbours in launch order (rounds 4-6) they were on different
    // XCDs and a 576-wide layer fetched X twice from the fabric (HBM-bound at 1.3-1.5 ms per 589 824 x 576 x 576 layer)
    const int groups = a.tiles_total / NT;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long rblock = (long)(slot / groups) * 8 + xcd;
    if (rblock * kGroupRows >= a.M) return;                            // the grid is padded to whole rounds of eight row blocks
    const long row = rblock * kGroupRows + wave * kRows + s;
    const long lrow = row < a.M ? row : a.M - 1;
    const float* xrow = a.X + lrow * a.ldx;
    const int tile0 = (slot % groups) * NT;
    const bool relu_in = (a.flags & CAR_LIN_RELU_IN) != 0;
    const int K = a.K;

    auto chunk_desc = [&](int c) {
        const int ce = c < a.chunks ? c : a.chunks - 1;
        NextChunk n;
        n.src = a.Wp + ((long)ce * a.tiles_total + tile0) * kTile;
        n.dst = lds + (ce & 1) * NT * kTile;
        n.nkb = 2 * NT;
#ifdef CAR_BOUNDS
        n.lim = a.Wp + (long)a.chunks * a.tiles_total * kTile;
#endif
        return n;
    };
    {
        const NextChunk n0 = chunk_desc(0);
#pragma unroll
        for (int p = 0; p < pieces_of(NT); ++p) stream_issue_piece(n0, p, lane, wave);
    }
    // GATHER: the row's four lattice taps (north-west node, + one node, + one row, + both: car_lattice_taps never returns a node of the
    // last column / row), its tap weights and point coordinates stay in registers for the whole kernel; the point / bias table goes to LDS
    constexpr int kRaw = GATHER ? 8 : 2;
    const float* tap[4] = {nullptr, nullptr, nullptr, nullptr};
    float tw[4] = {0.f, 0.f, 0.f, 0.f}, pe[3] = {0.f, 0.f, 0.f};
    const float* wtab = lds + 2 * NT * kTile;
    if constexpr (GATHER) {
        for (int k = tid; k < K; k += kThreads)
            *reinterpret_cast<float4*>(lds + 2 * NT * kTile + 4 * k) = *reinterpret_cast<const float4*>(a.wpt + 4 * k);
        const int src = a.row_src[lrow];
        const int m = src & 0x3fffffff, mode = (src >> 30) & 1;
        int node, flg;
        car_lattice_taps(a.row_grid[2 * lrow], a.row_grid[2 * lrow + 1], a.lw, a.lh, a.lpad, a.sx, a.sy, &node, &flg, tw);
        const bool dead = mode == 1 && (flg & 4);                      // zeros padding, on or beyond the outer ring: exactly zero
        if (dead) { node = 0; tw[0] = tw[1] = tw[2] = tw[3] = 0.0f; }
        const float* nw = a.lattice + ((long)(dead ? 0 : (m * 2 + mode)) * a.lh * a.lw + node) * K + 8 * q4;
        tap[0] = nw; tap[1] = nw + K; tap[2] = nw + (long)a.lw * K; tap[3] = nw + (long)a.lw * K + K;
        pe[0] = a.row_pe[4 * lrow]; pe[1] = a.row_pe[4 * lrow + 1]; pe[2] = a.row_pe[4 * lrow + 2];
    }
    // this lane's eight values of chunk c: k = 32 c + 8 q4 .. + 7.  issue_x only issues the loads (from clamped, valid addresses);
    // finish_x turns them into values (zeros beyond K, relu where the layer asks for it) when they are consumed — touching the loaded
    // registers at the load would put the memory latency in front of the chunk's MFMAs
    auto issue_x = [&](int c, float4 (&raw)[kRaw]) {
        if constexpr (GATHER) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                raw[2 * t] = *reinterpret_cast<const float4*>(tap[t] + 32 * c);
                raw[2 * t + 1] = *reinterpret_cast<const float4*>(tap[t] + 32 * c + 4);
            }
        } else {
            const int k0 = 32 * c + 8 * q4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kq = k0 + 4 * h;
                raw[h] = *reinterpret_cast<const float4*>(xrow + (kq < K ? kq : 0));
            }
        }
    };
    auto finish_x = [&](int c, const float4 (&raw)[kRaw], float (&x)[8]) {
        const int k0 = 32 * c + 8 * q4;
        if constexpr (GATHER) {
            // encode_kernel's arithmetic, operation for operation (csrc/car_encode.hip): the four taps in the order nw, ne, sw, se, one fmaf
            // each from zero; then + (fmaf(wz, pz, fmaf(wy, py, wx * px)) + b); relu
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 g = raw[2 * t + h];
                    acc[0] = fmaf(tw[t], g.x, acc[0]); acc[1] = fmaf(tw[t], g.y, acc[1]);
                    acc[2] = fmaf(tw[t], g.z, acc[2]); acc[3] = fmaf(tw[t], g.w, acc[3]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 wq = *reinterpret_cast<const float4*>(wtab + 4 * (k0 + 4 * h + i));     // (wx, wy, wz, b) of the channel
                    const float pt = fmaf(wq.z, pe[2], fmaf(wq.y, pe[1], wq.x * pe[0])) + wq.w;
                    x[4 * h + i] = fmaxf(acc[i] + pt, 0.0f);
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kq = k0 + 4 * h;
                const float e[4] = {raw[h].x, raw[h].y, raw[h].z, raw[h].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = kq + i < K ? e[i] : 0.0f;
                    x[4 * h + i] = relu_in ? fmaxf(t, 0.0f) : t;
                }
            }
        }
    };
    // largest magnitude of the row's values in a chunk (the four lanes of a row hold eight each)
    auto row_max = [&](const float (&x)[8]) {
        float m = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x[e]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        return fmaxf(m, __shfl_xor(m, 32, 64));
    };
    const float dW = a.down[0];
    float xc[8];
    {
        float4 raw[kRaw];
        issue_x(0, raw);
        if constexpr (GATHER) __syncthreads();                         // the point / bias table is in LDS
        finish_x(0, raw, xc);
    }
    float mrun = fmaxf(row_max(xc), 1e-30f), p, pinv;
    pow2_scale(mrun, p, pinv);

    f32x4 acc[NT];
    if (a.bias) init_bias<NT>(acc, a.bias + 16 * tile0, q4, p / dW);
    else {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    half8 bhi, blo;
    split8(xc, p, bhi, blo);
    stream_sync();                                                     // weight chunk 0 landed

#pragma unroll 1
    for (int c = 0; c < a.chunks; ++c) {
        const float* wl = lds + (c & 1) * NT * kTile + 4 * lane;
        const NextChunk nx = chunk_desc(c + 1);
        const int cn = c + 1 < a.chunks ? c + 1 : c;
        float4 rawn[kRaw];
        issue_x(cn, rawn);                                             // next chunk's rows: in flight under this chunk's MFMAs
#pragma unroll
        for (int qs = 0; qs < NT / 2; ++qs) {
            const float* w0 = wl + (2 * qs * 2) * 256;
            mfma_pair(acc[2 * qs], acc[2 * qs + 1], w0, w0 + 512, bhi, blo);
            // nothing follows the last chunk: re-copying it onto itself would write the buffer this iteration's MFMAs are reading
            if (qs < pieces_of(NT) && c + 1 < a.chunks) stream_issue_piece(nx, qs, lane, wave);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next chunk may outgrow the row's power of two: move the row (accumulators and scale) to the smaller one, exactly
        float xn[8];
        finish_x(cn, rawn, xn);
        const float mn = row_max(xn);
        if (__builtin_amdgcn_ballot_w64(mn > mrun) != 0) {
            float pn, pninv;
            mrun = fmaxf(mrun, mn);
            pow2_scale(mrun, pn, pninv);
            scale_acc<NT>(acc, pn * pinv);
            p = pn; pinv = pninv;
        }
        split8(xn, p, bhi, blo);
        stream_sync();
    }
    if constexpr (KQ) {
        // ---- key = Wk2 relu(k1) + bk2 ; qry = Wq2 relu(Wq1 g + bq1) + bq2 ; logit = <key, qry> / 16 (models.py:487-491, 529, 533): the fused
        //      per-sample kernel's closing layers (csrc/car_fused.hip), fed by this kernel's accumulators instead of e's.  The tail's weight chunks
        //      stream through the two 36 KB buffers that lie over this layer's own (done with: the loop's last barrier has passed).
        static_assert(NT == kTD, "the key layer has 128 outputs");
        float* lb = lds + 2 * kChunkTiles * kTile;                     // the tail's bias / scale table
        for (int k = tid; k < kTailBias; k += kThreads) lb[k] = a.tail_bias[k];
        stream_issue_all(a.tail, lds, kG_K2, lane, wave);
        scale_acc<NT>(acc, dW * pinv);                                 // k1, true values (bias included)
        const float* lsc = lb + 2 * kD;
        float p2, p2inv;
        pow2_scale(fmaxf(sample_max<kTD, true>(acc), 1e-30f), p2, p2inv);
        stream_sync();                                                 // chunk 0 landed, the table is visible
        f32x4 key[kTD];
        init_bias<kTD>(key, lb, q4, p2 / lsc[0]);
        chained_layer<kTD, true, 0, kG_K2>(key, acc, p2, a.tail, lds, lane, wave);
        scale_acc<kTD>(key, lsc[0] * p2inv);
        half8 ghi, glo;                                                // B operand of the layer fed by g (k = 16: folded bias)
        {
            const float* gl = a.g + lrow * 16 + 8 * (q4 & 1);
            float gx8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) gx8[k] = q4 < 2 ? gl[k] : (q4 == 2 && k == 0) ? 1.0f : 0.0f;
            float m = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(gx8[k]));
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));                       // >= 1: the bias column
            pow2_scale(m, p2, p2inv);
            split8(gx8, p2, ghi, glo);
        }
        f32x4 t1[kTD], qv[kTD];
#pragma unroll
        for (int t = 0; t < kTD; ++t) t1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        stream_issue_all(a.tail, lds, kG_Q1 + 1, lane, wave);
        small_layer(t1, ghi, glo, lds + kLdsW + (kG_Q1 & 1) * kChunkTiles * kTile + 4 * lane);
        stream_sync();
        scale_acc<kTD>(t1, lsc[1] * p2inv);
        pow2_scale(fmaxf(sample_max<kTD, true>(t1), 1e-30f), p2, p2inv);
        init_bias<kTD>(qv, lb + kD, q4, p2 / lsc[2]);
        chained_layer<kTD, true, 0, kG_Q2>(qv, t1, p2, a.tail, lds, lane, wave);
        scale_acc<kTD>(qv, lsc[2] * p2inv);
        float dot = 0.0f;
#pragma unroll
        for (int t = 0; t < kTD; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = fmaf(key[t][r], qv[t][r], dot);
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        if (row < a.M) {
            store_rows<kTD>(qv, a.qry + row * kD, q4);
            if (q4 == 0) a.logit[row] = dot / 16.0f;
        }
        return;
    }
    if (row >= a.M) return;
    scale_acc<NT>(acc, dW * pinv);
    const bool relu_out = (a.flags & CAR_LIN_RELU_OUT) != 0, accum = (a.flags & CAR_LIN_ACCUM) != 0;
    float* yrow = a.Y + row * a.ldy + 16 * tile0 + 4 * q4;
    const float* mrow = a.mask ? a.mask + row * a.ldm + 16 * tile0 + 4 * q4 : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float4* dst = reinterpret_cast<float4*>(yrow + 16 * t);
        float4 v = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        if (accum) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        if (relu_out) v = make_float4(fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f), fmaxf(v.z, 0.0f), fmaxf(v.w, 0.0f));
        if (mrow) {                                                    // car_relu_mask's rule: keep where the activation is > 0
            const float4 k = *reinterpret_cast<const float4*>(mrow + 16 * t);
            v = make_float4(k.x > 0.0f ? v.x : 0.0f, k.y > 0.0f ? v.y : 0.0f, k.z > 0.0f ? v.z : 0.0f, k.w > 0.0f ? v.w : 0.0f);
        }
        *dst = v;
    }
}

// max |W| over the layer, as the bit pattern of a non-negative float (orders like an integer), into scale[2] (zeroed by the caller): a grid
// of workgroups, one atomic each.  (One workgroup reading the whole matrix took 25 us per layer: 0.9 ms of a training step, which re-packs
// its ~36 forward and transposed layers after every optimizer step.)
__global__ void absmax16_kernel(const float* __restrict__ W, int ldw, int K, int N, float* __restrict__ scale) {
    __shared__ float red[4];
    float m = 0.0f;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)N * K; idx += (long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(W[(idx / K) * ldw + idx % K]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x / 64); ++w) m = fmaxf(m, red[w]);
        atomicMax(reinterpret_cast<unsigned*>(scale + 2), __float_as_uint(m));
    }
}
// [K step][tile][hi | lo][lane][8 halves]: lane l carries output 16 tile + l % 16 and k = 32 step + 8 (l >> 4) + e
// scale[2] holds max |W| (absmax16_kernel): every thread derives the layer's power of two from it — scale[0] = 2^shift with
// max |W| 2^shift in [2^13, 2^14), scale[1] = 2^-shift, written by the first thread for the consuming kernel
__global__ void pack16x3_kernel(const float* __restrict__ W, int ldw, int K, int N, int tiles, long total, float* __restrict__ scale,
                                _Float16* __restrict__ out) {
    float p, inv;
    pow2_scale(fmaxf(scale[2], 1e-30f), p, inv);
    if (blockIdx.x == 0 && threadIdx.x == 0) { scale[0] = p; scale[1] = inv; }
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long tile = idx >> 9;
        const int t = (int)(tile % tiles), ks = (int)(tile / tiles);
        const int n = 16 * t + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + e;
        const float w = (n < N && k < K) ? W[(long)n * ldw + k] * p : 0.0f;
        const _Float16 hi = (_Float16)w;
        _Float16* o = out + tile * 1024 + lane * 8 + e;
        o[0] = hi;
        o[512] = (_Float16)(w - (float)hi);
    }
}

template <int NT, bool GATHER = false, bool KQ = false>
int launch16(const LinArgs& a, int groups, hipStream_t st) {
    const size_t lds_bytes = KQ ? (size_t)(2 * kChunkTiles * kTile + kTailBias) * sizeof(float)
                                : (size_t)2 * NT * kTile * sizeof(float) + (GATHER ? (size_t)a.K * 4 * sizeof(float) : 0);
    // the LDS reservation is a per-device attribute of the kernel: set it once per (instance, device), not on each of the dozens of
    // launches of a staged forward or training step
    static bool reserved[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !reserved[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)linear16_kernel<NT, GATHER, KQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) {
            car_set_error("car_linear_x3: cannot reserve %zu bytes of LDS (a gfx950-class device has 160 KB per compute unit): %s", lds_bytes, hipGetErrorString(e));
            return CAR_E_LAUNCH;
        }
        if (dev >= 0 && dev < 64) reserved[dev] = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL((linear16_kernel<NT, GATHER, KQ>), dim3((unsigned)(car_div_up(car_div_up(a.M, kGroupRows), 8) * 8 * groups)), dim3(kThreads), lds_bytes, st, a);
    CAR_CHECK_LAUNCH("car_linear_x3");
    return CAR_OK;
}

}  // namespace

// Packed weights of car_linear_x3: ceil(K / 32) x (N / 16) tiles of 512 floats, then two floats (2^shift, 2^-shift).
extern "C" size_t car_linear_x3_packed_floats(int K, int N) { return (size_t)((K + 31) / 32) * ((N + 15) / 16) * kTile + 64; }
extern "C" int car_linear_x3_pack(const float* W, int ldw, int K, int N, float* packed, void* stream) {
    CAR_REQUIRE(W && packed && K > 0 && N > 0 && ldw >= K && N % 16 == 0, "car_linear_x3_pack: bad arguments (N must be a multiple of 16)");
    const int tiles = N / 16, ksteps = (K + 31) / 32;
    float* scale = packed + (size_t)ksteps * tiles * kTile;
    (void)hipGetLastError();
    if (hipMemsetAsync(scale + 2, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) { car_set_error("car_linear_x3_pack: memset failed"); return CAR_E_LAUNCH; }
    const long elems = (long)N * K;
    hipLaunchKernelGGL(absmax16_kernel, dim3((unsigned)(elems < 65536 ? 8 : 64)), dim3(256), 0, (hipStream_t)stream, W, ldw, K, N, scale);
    hipLaunchKernelGGL(pack16x3_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, W, ldw, K, N, tiles, (long)ksteps * tiles * 512, scale,
                       reinterpret_cast<_Float16*>(packed));
    CAR_CHECK_LAUNCH("car_linear_x3_pack");
    return CAR_OK;
}

namespace {
int linear_x3(const float* X, int ldx, const float* packed, const float* bias, int K, int N, float* Y, int ldy, long M, int flags, const float* act,
              int lda, void* stream);
}
extern "C" int car_linear_x3(const float* X, int ldx, const float* packed, const float* bias, int K, int N, float* Y, int ldy, long M, int flags,
                             void* stream) {
    return linear_x3(X, ldx, packed, bias, K, N, Y, ldy, M, flags, nullptr, 0, stream);
}
// car_linear_x3 with car_relu_mask applied to the result as it is stored: Y[m][n] = act[m][n] > 0 ? (X W^T + bias (+ Y))[m][n] : 0 — the data
// gradient dX = dY W of a layer whose input went through a ReLU (torch autograd's threshold_backward over models.py:333-341, 487-491, 529),
// without the separate pass over dX and the activation.  Bit-identical to car_linear_x3 followed by car_relu_mask.
extern "C" int car_linear_x3_masked(const float* X, int ldx, const float* packed, const float* bias, int K, int N, float* Y, int ldy, long M, int flags,
                                    const float* act, int lda, void* stream) {
    CAR_REQUIRE(act && lda >= N && lda % 4 == 0 && ((uintptr_t)act & 15) == 0,
                "car_linear_x3_masked: act must be 16-byte aligned with a row stride (%d) that is a multiple of 4 and holds N = %d", lda, N);
    return linear_x3(X, ldx, packed, bias, K, N, Y, ldy, M, flags, act, lda, stream);
}
namespace {
int linear_x3(const float* X, int ldx, const float* packed, const float* bias, int K, int N, float* Y, int ldy, long M, int flags, const float* act,
              int lda, void* stream) {
    CAR_REQUIRE(X && packed && Y && M > 0 && K > 0 && N > 0, "car_linear_x3: bad arguments");
    CAR_REQUIRE(N % 32 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= ((K + 3) & ~3) && ldy >= N,
                "car_linear_x3: N = %d must be a multiple of 32, ldx = %d / ldy = %d multiples of 4 that hold a row", N, ldx, ldy);
    CAR_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0), "car_linear_x3: X, Y and bias must be 16-byte aligned");
    const int tiles = N / 16, ksteps = (K + 31) / 32;
    LinArgs a;
    a.X = X; a.ldx = ldx; a.Wp = packed; a.tiles_total = tiles; a.bias = bias;
    a.down = packed + (size_t)ksteps * tiles * kTile + 1;
    a.K = K; a.chunks = ksteps; a.Y = Y; a.ldy = ldy; a.M = M; a.flags = flags;
    a.mask = act; a.ldm = lda;
    hipStream_t st = (hipStream_t)stream;
    if (tiles % 18 == 0) return launch16<18>(a, tiles / 18, st);
    if (tiles % 8 == 0) return launch16<8>(a, tiles / 8, st);
    if (tiles % 4 == 0) return launch16<4>(a, tiles / 4, st);
    return launch16<2>(a, tiles / 2, st);
}
}  // namespace

// car_lattice_encode_rows followed by car_linear_x3 in ONE kernel (the three-view exchange, models.py:345-475 — engine._encode_three_views):
// Y[rows, N] = act(relu(rows of the merged lattice + point term) W^T + bias).  The K = 576-wide first-layer rows are never written: every
// 32-channel chunk is gathered into the lanes' B operands (the same arithmetic, operation for operation, as the two kernels it replaces).
extern "C" int car_lattice_encode_linear(const float* lattice, int lat_h, int lat_w, int lat_pad, const int* row_src, const float* row_grid,
                                         const float* row_pe, const float* wpt, int n_maps, long rows, const float* packed, const float* bias,
                                         int K, int N, float* Y, int ldy, int flags, void* stream) {
    CAR_REQUIRE(lattice && row_src && row_grid && row_pe && wpt && packed && Y, "car_lattice_encode_linear: null pointer");
    CAR_REQUIRE(K == 576 && N % 32 == 0 && N > 0 && n_maps > 0 && rows > 0, "car_lattice_encode_linear: K must be 576 and N a multiple of 32 (got %d, %d)", K, N);
    CAR_REQUIRE(lat_pad >= 2 && lat_h > 2 * lat_pad + 1 && lat_w > 2 * lat_pad + 1 && ((lat_h - 2 * lat_pad) & 1) && ((lat_w - 2 * lat_pad) & 1),
                "car_lattice_encode_linear: bad lattice %d x %d, pad %d (car_merge_lattice)", lat_h, lat_w, lat_pad);
    CAR_REQUIRE((long)n_maps * 2 * lat_h * lat_w < 2147483647L, "car_lattice_encode_linear: too many lattice nodes");
    CAR_REQUIRE(ldy % 4 == 0 && ldy >= N && ((uintptr_t)Y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) && ((uintptr_t)lattice & 15) == 0 && ((uintptr_t)wpt & 15) == 0,
                "car_lattice_encode_linear: ldy = %d must be a multiple of 4 that holds a row; Y, bias, lattice and wpt 16-byte aligned", ldy);
    CAR_REQUIRE(!(flags & CAR_LIN_RELU_IN), "car_lattice_encode_linear: the gathered rows are already rectified");
    const int tiles = N / 16, ksteps = (K + 31) / 32;
    LinArgs a{};
    a.X = nullptr; a.ldx = 0; a.Wp = packed; a.tiles_total = tiles; a.bias = bias;
    a.down = packed + (size_t)ksteps * tiles * kTile + 1;
    a.K = K; a.chunks = ksteps; a.Y = Y; a.ldy = ldy; a.M = rows; a.flags = flags;
    a.lattice = lattice; a.lh = lat_h; a.lw = lat_w; a.lpad = lat_pad;
    a.sx = (float)((lat_w - 2 * lat_pad + 1) / 2); a.sy = (float)((lat_h - 2 * lat_pad + 1) / 2);
    a.row_src = row_src; a.row_grid = row_grid; a.row_pe = row_pe; a.wpt = wpt;
    hipStream_t st = (hipStream_t)stream;
    if (tiles % 18 == 0) return launch16<18, true>(a, tiles / 18, st);
    if (tiles % 8 == 0) return launch16<8, true>(a, tiles / 8, st);
    if (tiles % 4 == 0) return launch16<4, true>(a, tiles / 4, st);
    return launch16<2, true>(a, tiles / 2, st);
}

// key_map -> relu -> key_map_2, query_embed -> relu -> query_embed_2 and the first round's logits in ONE kernel for the stage route's
// variants (n_view 1 / 3, no_latent_concat; models.py:487-491, 529, 533): the 128-wide k1, key and q1 rows are never written.
//   e [M, Ce] (row stride lde): the per-sample features; packed_k1 / bias_k1: key_map as car_linear_x3_pack lays it out (K = Ce, N = 128);
//   g [M, 16]: the geometric query; tail / tail_bias: car_kq_pack.  Out: qry [M, 128] (the second round reads it), logit [M].
extern "C" size_t car_kq_tail_floats(void) { return (size_t)kTailTiles * kTile; }
extern "C" size_t car_kq_bias_floats(void) { return (size_t)kTailBias; }
extern "C" int car_key_query_logits(const float* e, int lde, const float* packed_k1, const float* bias_k1, int Ce, const float* g, const float* tail,
                                    const float* tail_bias, long M, float* qry, float* logit, void* stream) {
    CAR_REQUIRE(e && packed_k1 && bias_k1 && g && tail && tail_bias && qry && logit, "car_key_query_logits: null pointer");
    CAR_REQUIRE(M > 0 && Ce > 0 && lde % 4 == 0 && lde >= ((Ce + 3) & ~3), "car_key_query_logits: bad sizes (Ce = %d, row stride %d)", Ce, lde);
    CAR_REQUIRE(((uintptr_t)e & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)qry & 15) == 0 && ((uintptr_t)bias_k1 & 15) == 0,
                "car_key_query_logits: e, g, qry and the bias must be 16-byte aligned");
    const int ksteps = (Ce + 31) / 32;
    LinArgs a{};
    a.X = e; a.ldx = lde; a.Wp = packed_k1; a.tiles_total = kTD; a.bias = bias_k1;
    a.down = packed_k1 + (size_t)ksteps * kTD * kTile + 1;
    a.K = Ce; a.chunks = ksteps; a.Y = nullptr; a.ldy = 0; a.M = M; a.flags = 0;
    a.g = g; a.tail = tail; a.tail_bias = tail_bias; a.qry = qry; a.logit = logit;
    return launch16<kTD, false, true>(a, 1, (hipStream_t)stream);
}
