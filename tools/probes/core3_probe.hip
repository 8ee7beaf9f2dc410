// core3_probe.hip — development microbenchmark (not product, results are not checked): the two source passes of the fused per-sample
// kernel re-cut for ONE wave per SIMD — 4 waves x 512 registers per workgroup, each wave 64 samples x 288 channels on 32 x 32 x 16 tiles
// (288 accumulator registers), the A operands double-buffered in registers, the gather's vector / LDS / texture instructions placed between
// the MFMAs of the same stream (profiles/round4_fused_experiments.md section 5: two to three of them hide under every MFMA).  It runs the
// same work per sample as csrc/car_fused.hip's source passes — 4 taps x 2304 B of a 521 x 521 x 576 lattice per sample and source at
// epipolar-like positions, start values, blend, ReLU, fp16 hi / lo split, 3 x (576 x 288) MACs, e stored — with synthetic weights and
// tables, to answer one question before anybody rewrites the kernel: how far below the product's 2.25 ms per 8192 rays do the source
// passes get in this shape?  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/core3_probe.hip -o tools/_dev/core3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kC = 576, kE = 288, kKS = 18;                 // 18 chunks of 32 channels
constexpr int kWaves = 4, kWS = 64, kGroup = kWaves * kWS;   // 256 samples per workgroup
constexpr int kCT = 9;                                       // channel tiles of 32
constexpr int kTileF = 512;                                  // floats per (k16 step, tile): [hi | lo][64 lanes][16 B]
constexpr int kChunkF = 2 * kCT * kTileF;                    // 36 KB per chunk
constexpr int kHLd = 36;                                     // row stride of a wave's h tile (floats)

constexpr int kLdsW = 0;                                     // [2][kChunkF]
constexpr int kLdsH = kLdsW + 2 * kChunkF;                   // [4 waves][64][36]
constexpr int kLdsTapB = kLdsH + kWaves * kWS * kHLd;        // [256][2] uint
constexpr int kLdsTapW = kLdsTapB + kGroup * 2;              // [256][2][4]
constexpr int kLdsPe = kLdsTapW + kGroup * 8;                // [256][2][4]
constexpr int kLdsWpt = kLdsPe + kGroup * 8;                 // [144][4][4]
constexpr int kLdsFloats = kLdsWpt + kC * 4;
constexpr size_t kLdsBytes = (size_t)kLdsFloats * 4;
static_assert(kLdsBytes <= 160 * 1024, "LDS");

__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(hi), "v"(a), "v"(b));
}

struct Args { const float* lattice; unsigned map_bytes; int lw; const float* blob; const float* wpt; float* e; int tiles_x; };

template <int MODE>   // 0 full, 1 no tap loads, 2 no gather work at all, 3 = 2 + no A-operand reads, 4 = 3 + no e stores, 5 = full without the e stores
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) core3(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = blockIdx.x;
    // tables: sample sg of the workgroup = ray (sg & 31), step (sg >> 5) of a 32-ray x 8-step tile; epipolar-like lattice positions
    {
        const int sg = tid, ray = sg & 31, step = sg >> 5;
        const int tx = blk % a.tiles_x, ty = blk / a.tiles_x;
        for (int sv = 0; sv < 2; ++sv) {
            const float x = 20.0f + 1.7f * (ray + 32 * (tx % 8)) + 7.3f * (step + 8 * (ty % 7)) + 3.0f * sv, y = 30.0f + 0.3f * ray + 3.1f * (step + 8 * (ty % 7)) + 40.0f * (tx / 8 % 8);
            const int ix = (int)x % 500, iy = (int)y % 500;
            const float fx = x - (int)x, fy = y - (int)y;
            reinterpret_cast<unsigned*>(lds + kLdsTapB)[sg * 2 + sv] = (unsigned)(iy * a.lw + ix) * (unsigned)(kC * 4);
            *reinterpret_cast<float4*>(lds + kLdsTapW + (sg * 2 + sv) * 4) = make_float4((1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy);
            *reinterpret_cast<float4*>(lds + kLdsPe + (sg * 2 + sv) * 4) = make_float4(0.1f * fx, 0.2f * fy, 0.3f, 0.0f);
        }
    }
    for (int k = tid; k < kC * 4; k += 256) lds[kLdsWpt + k] = a.wpt[k];
    // weight stream: chunk g -> buffer g & 1, 36 pieces of 1 KB, 9 per wave
    auto dma_piece = [&](int g, int p) {
        const int gg = g % kKS;
        const int kb = kWaves * p + wave;
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(lds + kLdsW + (g & 1) * kChunkF + kb * 256));
        const float* gsrc = a.blob + (long)gg * kChunkF + kb * 256;
        const unsigned voff = 16u * (unsigned)lane;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gsrc) : "memory");
    };
    for (int p = 0; p < 9; ++p) dma_piece(0, p);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lattice), 0, (int)a.map_bytes, 0x00027000);
    const int qd = lane & 7, r0 = lane >> 3;                        // gather lanes: row r0 + 8 * rg, channel quad qd
    const unsigned row_step = (unsigned)a.lw * (kC * 4);
    float* htile = lds + kLdsH + wave * kWS * kHLd;
    constexpr int kRG = kWS / 8;                                     // 8 row groups of 8 rows per wave
    constexpr int kRing = 4;                                         // row groups of taps in flight: 64 registers, half a chunk ahead
    f32x4 tap[kRing][4];
    auto issue_rg = [&](int sv, int c, int rg) {
        if constexpr (MODE >= 1 && MODE != 5) return;
        const unsigned tbv = reinterpret_cast<const unsigned*>(lds + kLdsTapB)[(wave * kWS + r0 + 8 * rg) * 2 + sv];
        const unsigned o00 = tbv + 16u * qd, o10 = o00 + row_step;
        auto ld = [&](unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 128 * c, 0)); };
        tap[rg % kRing][0] = ld(o00); tap[rg % kRing][1] = ld(o00 + kC * 4); tap[rg % kRing][2] = ld(o10); tap[rg % kRing][3] = ld(o10 + kC * 4);
    };
    auto gather_rg = [&](int sv, int c, int rg) {                     // start values, blend, ReLU -> the wave's h tile (fp32)
        if constexpr (MODE >= 2 && MODE != 5) return;
        const int rr = r0 + 8 * rg;
        const float4 pe = *reinterpret_cast<const float4*>(lds + kLdsPe + ((wave * kWS + rr) * 2 + sv) * 4);
        const float4* wp = reinterpret_cast<const float4*>(lds + kLdsWpt + 16 * (8 * c + qd));
        const float4 wx = wp[0], wy = wp[1], wz = wp[2], wb = wp[3];
        const float4 w = *reinterpret_cast<const float4*>(lds + kLdsTapW + ((wave * kWS + rr) * 2 + sv) * 4);
        f32x2 lo2 = {fmaf(wx.x, pe.x, fmaf(wy.x, pe.y, fmaf(wz.x, pe.z, wb.x))), fmaf(wx.y, pe.x, fmaf(wy.y, pe.y, fmaf(wz.y, pe.z, wb.y)))};
        f32x2 hi2 = {fmaf(wx.z, pe.x, fmaf(wy.z, pe.y, fmaf(wz.z, pe.z, wb.z))), fmaf(wx.w, pe.x, fmaf(wy.w, pe.y, fmaf(wz.w, pe.z, wb.w)))};
        const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 gq = tap[rg % kRing][t];
            const f32x2 w2 = {ww[t], ww[t]};
            lo2 = __builtin_elementwise_fma(w2, f32x2{gq[0], gq[1]}, lo2);
            hi2 = __builtin_elementwise_fma(w2, f32x2{gq[2], gq[3]}, hi2);
        }
        *reinterpret_cast<float4*>(htile + rr * kHLd + 4 * qd) = make_float4(fmaxf(lo2[0], 0.f), fmaxf(lo2[1], 0.f), fmaxf(hi2[0], 0.f), fmaxf(hi2[1], 0.f));
    };
    // B operands of a k16 step: lane (col = lane & 31, khalf = lane >> 5) takes 8 channels of sample col of each of the two 32-sample tiles
    const int col = lane & 31, kh = lane >> 5;
    auto read_b = [&](int ks, half8 (&bhi)[2], half8 (&blo)[2]) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const float* row = htile + (32 * st + col) * kHLd + 16 * ks + 8 * kh;
            const float4 x0 = *reinterpret_cast<const float4*>(row), x1 = *reinterpret_cast<const float4*>(row + 4);
            u32x4 h, l;
            unsigned hh, ll;
            split_pair(x0.x, x0.y, hh, ll); h[0] = hh; l[0] = ll;
            split_pair(x0.z, x0.w, hh, ll); h[1] = hh; l[1] = ll;
            split_pair(x1.x, x1.y, hh, ll); h[2] = hh; l[2] = ll;
            split_pair(x1.z, x1.w, hh, ll); h[3] = hh; l[3] = ll;
            bhi[st] = __builtin_bit_cast(half8, h); blo[st] = __builtin_bit_cast(half8, l);
        }
    };
    auto load_a = [&](const float* wl, int ks, int t, half8& ah, half8& al) {
        if constexpr (MODE == 3 || MODE == 4) { ah = half8{}; al = half8{}; return; }
        const float* w0 = wl + (ks * kCT + t) * kTileF + 4 * lane;
        ah = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0));
        al = __builtin_bit_cast(half8, *reinterpret_cast<const float4*>(w0 + 256));
    };

    // first chunk of source 0
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int rg = 0; rg < kRing; ++rg) issue_rg(0, 0, kRing * half + rg);
#pragma unroll
        for (int rg = 0; rg < kRing; ++rg) gather_rg(0, 0, kRing * half + rg);
    }
#pragma unroll
    for (int rg = 0; rg < kRing; ++rg) issue_rg(0, 1, rg);            // the ring's lead: row groups 0-3 of the next chunk
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                  // weight chunk 0 (older than the 16 tap loads just issued)
    __syncthreads();

    f32x16 acc[kCT][2];
    int g = 0;
#pragma unroll 1
    for (int sv = 0; sv < 2; ++sv) {
#pragma unroll
        for (int t = 0; t < kCT; ++t)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][st][r] = 0.01f * r;
#pragma unroll 1
        for (int c = 0; c < kKS; ++c) {
            const int nsv = (c + 1 < kKS) ? sv : 1, nc = (c + 1 < kKS) ? c + 1 : 0;
            const int n2sv = (c + 2 < kKS) ? sv : 1, n2c = (c + 2 < kKS) ? c + 2 : c + 2 - kKS;
            const float* wl = lds + kLdsW + (g & 1) * kChunkF;
            // 9 groups of (two A tiles, 12 MFMAs on four accumulators: a dependent MFMA is four issues away); the A operands of the next group
            // are read before this group's MFMAs; between the MFMAs the chunk's other work in small pieces: 9 DMA pieces, 8 x (blend one row
            // group of the next chunk + re-issue its ring slot)
            half8 bhi[2][2], blo[2][2], ah[2][2], al[2][2];
            read_b(0, bhi[0], blo[0]);
            read_b(1, bhi[1], blo[1]);
            load_a(wl, 0, 0, ah[0][0], al[0][0]);
            load_a(wl, 0, 1, ah[0][1], al[0][1]);
#pragma unroll
            for (int grp = 0; grp < kCT; ++grp) {
                const int cur = grp & 1;
                const int i0 = 2 * grp, i1 = 2 * grp + 1;               // (k16 step, tile) index = ks * 9 + t
                const int ks0 = i0 / kCT, t0 = i0 % kCT, ks1 = i1 / kCT, t1 = i1 % kCT;
                if (grp + 1 < kCT) {
                    load_a(wl, (i0 + 2) / kCT, (i0 + 2) % kCT, ah[cur ^ 1][0], al[cur ^ 1][0]);
                    load_a(wl, (i1 + 2) / kCT, (i1 + 2) % kCT, ah[cur ^ 1][1], al[cur ^ 1][1]);
                }
                auto mm = [&](int which, int st, const half8& av, const half8& bv) {
                    const int t = which ? t1 : t0;
                    acc[t][st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[t][st], 0, 0, 0);
                };
                mm(0, 0, ah[cur][0], bhi[ks0][0]); mm(0, 1, ah[cur][0], bhi[ks0][1]); mm(1, 0, ah[cur][1], bhi[ks1][0]); mm(1, 1, ah[cur][1], bhi[ks1][1]);
                dma_piece(g + 1, grp);
                if (grp >= 1) { const int k = grp - 1; gather_rg(nsv, nc, k); if (k + kRing < kRG) issue_rg(nsv, nc, k + kRing); else issue_rg(n2sv, n2c, k + kRing - kRG); }
                mm(0, 0, ah[cur][0], blo[ks0][0]); mm(0, 1, ah[cur][0], blo[ks0][1]); mm(1, 0, ah[cur][1], blo[ks1][0]); mm(1, 1, ah[cur][1], blo[ks1][1]);
                mm(0, 0, al[cur][0], bhi[ks0][0]); mm(0, 1, al[cur][0], bhi[ks0][1]); mm(1, 0, al[cur][1], bhi[ks1][0]); mm(1, 1, al[cur][1], bhi[ks1][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // h of the next chunk is complete (own tile), the weights of the next chunk have landed; 32 taps stay in flight
            if constexpr (MODE == 0 || MODE == 5) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ++g;
        }
        // e out: tile t, sample tile st: lane (col, kh) register r holds channel 32 t + 8 (r / 4) + 4 kh + r % 4 of sample 32 st + col
        const long row0 = ((long)blk * kGroup + wave * kWS) * (2 * kE) + sv * kE;
        if constexpr (MODE >= 4) { float sacc = 0.f; for (int t = 0; t < kCT; ++t) for (int st = 0; st < 2; ++st) sacc += acc[t][st][0] + acc[t][st][15]; if (sacc == 123.456f) a.e[row0] = sacc; }
        else
#pragma unroll
        for (int t = 0; t < kCT; ++t)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(a.e + row0 + (long)(32 * st + col) * (2 * kE) + 32 * t + 8 * q + 4 * kh) =
                        make_float4(acc[t][st][4 * q], acc[t][st][4 * q + 1], acc[t][st][4 * q + 2], acc[t][st][4 * q + 3]);
    }
}

template <int MODE>
float run(const Args& a, int groups) {
    (void)hipFuncSetAttribute((const void*)core3<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(core3<MODE>, dim3(groups), dim3(256), kLdsBytes, 0, a);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 1 && ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return best;
}

int main() {
    const int lw = 521, lh = 521;
    const size_t map_floats = (size_t)lw * lh * kC;
    const long S = 2L * 8192 * 64;                                   // samples of one 8192-ray launch of the bench frame (both context views)
    const int groups = (int)(S / kGroup);
    float *lattice, *blob, *wpt, *e;
    if (hipMalloc(&lattice, map_floats * 4) != hipSuccess || hipMalloc(&blob, (size_t)kKS * kChunkF * 4) != hipSuccess ||
        hipMalloc(&wpt, kC * 4 * 4) != hipSuccess || hipMalloc(&e, (size_t)S * 2 * kE * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f - 0.5f;
    for (size_t off = 0; off < map_floats; off += h.size())
        (void)hipMemcpy(lattice + off, h.data(), std::min(h.size(), map_floats - off) * 4, hipMemcpyHostToDevice);
    {   // weights: fp16 pairs packed into 32-bit words; any finite values do
        std::vector<unsigned> w((size_t)kKS * kChunkF);
        for (size_t i = 0; i < w.size(); ++i) w[i] = 0x2c002c00u + (unsigned)((i * 40503u) & 0x03ff03ffu);
        (void)hipMemcpy(blob, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    }
    (void)hipMemcpy(wpt, h.data(), kC * 4 * 4, hipMemcpyHostToDevice);
    Args a{lattice, (unsigned)(map_floats * 4), lw, blob, wpt, e, 16};
    printf("source passes only, %ld samples (one 8192-ray launch), %d workgroups of 4 waves x 64 samples, %zu KB LDS\n", S, groups, kLdsBytes / 1024);
    const double flop = 2.0 * S * 2 * 576 * 288;
    const float t0 = run<0>(a, groups), t1 = run<1>(a, groups), t2 = run<2>(a, groups), t3 = run<3>(a, groups), t4 = run<4>(a, groups), t5 = run<5>(a, groups);
    printf("full %.3f ms (%.0f TFLOP/s fp32-equivalent) | no tap loads %.3f | no gather work %.3f | + no A-operand reads %.3f | + no e stores %.3f | full without e stores %.3f | MFMA floor at 2.4 GHz %.3f ms\n",
           t0, flop / t0 / 1e9, t1, t2, t3, t4, t5, 3.0 * flop / 2.5e15 * 1e3);
    printf("product kernel, source passes: ~2.25 ms of its 3.16 ms per launch (profiles/round4_fused_experiments.md)\n");
    return 0;
}
