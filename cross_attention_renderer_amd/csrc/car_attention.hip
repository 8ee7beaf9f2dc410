// car_attention.hip — stage kernels for the per-ray softmax attention over the V*P epipolar samples
// (reference models.py:532-594).
//
// One 256-thread workgroup per (scene, ray).  The ray's S = V*P logits (S <= 384) live in LDS; the softmax
// is a wavefront-shuffle max/sum reduction; the value reduction sum_s w_s val[s][:] reads each value row
// once, coalesced across the D channels.  HBM-bound: (2*dq + D)*4 bytes per sample.
#include "car_common.h"
#include "car_geom.h"

namespace {

constexpr int kMaxSamples = CAR_MAX_VIEWS * 256;
constexpr int kWideSeg = 14;                     // widest row of the streaming value reduction: 14 x 64 channels (864 = three views' 288: 13.5)
constexpr int kNarrowSeg = 9;                    // the two-view routes' rows: 576 channels.  The segment count is a template argument: the
                                                 // partial sums of all kMaxSeg segments are registers (56 for 14, 36 for 9: a wave of occupancy)

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// PARTS (car_attend_parts): `val` holds, per ray and group of `tile_steps` consecutive steps of a view, the partial sum
// sum_j exp(logit_j - m_g) val_j with m_g the group's largest logit (written by the fused per-sample kernel, csrc/car_fused.hip); the value
// reduction then runs over the V * ceil(P / tile_steps) groups with the weights exp(m_g - M) / L instead of over the V * P sample rows —
// the same sum, 1 / tile_steps of the bytes.  The softmax weights themselves (w_out, depth, argmax) are computed exactly as without.
template <bool PARTS, int kMaxSeg>
__global__ void __launch_bounds__(256) attend_kernel(const float* __restrict__ qa, const float* __restrict__ qb, int dq,
                                                     const float* __restrict__ val, int D, int b, int V, int R, int P,
                                                     const float* __restrict__ zprev, float zprev_scale,
                                                     float* __restrict__ w_out, float* __restrict__ z_out, int ld_z, int reps,
                                                     const float* __restrict__ pt, const CarPose* __restrict__ poses,
                                                     float* __restrict__ depth, int32_t* __restrict__ w_argmax, int tile_steps) {
    __shared__ float s_w[kMaxSamples];
    __shared__ float s_g[PARTS ? kMaxSamples / 4 : 1];       // PARTS: the groups' weights exp(m_g - M) / L
    __shared__ float s_red[8];
    __shared__ __attribute__((aligned(16))) float s_z[4 * 64 * kMaxSeg];       // per-wave partial sums of the value reduction
    const int sc = blockIdx.x / R, r = blockIdx.x % R;
    const int S = V * P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // sample s = v*P + p of this ray lives at row ((sc*V+v)*R + r)*P + p
    auto row_of = [&](int s) -> long { return ((long)(sc * V + s / P) * R + r) * P + (s % P); };

    const int sub = tid & 15, grp = tid >> 4;            // 16 groups of 16 lanes
    // rows of the value reduction (step 3): the samples, or (PARTS) the step groups.  Their addresses do not depend on the softmax, so the
    // rows of the first iteration are requested HERE, in front of the logits -> max -> sum chain, and land while it runs: a workgroup then
    // has its 16 rows in flight for its whole life instead of the last third of it (the PARTS launch is ONE iteration per workgroup and ran
    // at 4.3 TB/s, latency-bound on two dependent round trips to memory)
    const int pgs = PARTS ? (P + tile_steps - 1) / tile_steps : 0;
    const int NV = PARTS ? V * pgs : S;
    auto vrow_of = [&](int i) -> long {
        if constexpr (PARTS) return ((long)(sc * V + i / pgs) * R + r) * pgs + (i % pgs);
        else return row_of(i);
    };
    const bool streaming = D % 4 == 0 && D >= 64 && D <= 64 * kMaxSeg;
    const int nseg = (D + 63) / 64;
    float4 part[kMaxSeg];
    // (PARTS only: the 36 registers held across the softmax cost the sample-row instance a wave of occupancy — 3.27 -> 3.58 ms on the second
    // round's 19 GB, which runs eight iterations per workgroup and is bandwidth-bound, not latency-bound)
    if (PARTS && streaming) {
        const float* rowp = val + vrow_of(grp < NV ? grp : NV - 1) * D + 4 * sub;
#pragma unroll
        for (int j = 0; j < kMaxSeg; ++j) {
            part[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < nseg && 64 * j + 4 * sub < D) {
                // streamed once per round (19 GB per frame): non-temporal, so the rows do not displace what the next kernel reads
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rowp + 64 * j));
                part[j] = make_float4(xv[0], xv[1], xv[2], xv[3]);
            }
        }
    }
    // 1. logits: 16 lanes per sample, float4 per lane per step
    if (qb == nullptr) {                                 // logits were computed upstream (car_fused_samples)
        for (int s = tid; s < S; s += 256) s_w[s] = qa[row_of(s)];
    } else
    for (int s = grp; s < S; s += 16) {
        const long row = row_of(s);
        const float* a = qa + row * dq;
        const float* c = qb + row * dq;
        float acc = 0.0f;
        for (int k = 4 * sub; k < dq; k += 64) {
            const float4 x = *reinterpret_cast<const float4*>(a + k);
            const float4 y = *reinterpret_cast<const float4*>(c + k);
            acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);
        if (sub == 0) s_w[s] = acc / 16.0f;
    }
    __syncthreads();
    // 2. softmax over the S samples
    float m = -INFINITY;
    for (int s = tid; s < S; s += 256) m = fmaxf(m, s_w[s]);
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    if constexpr (PARTS) {                                  // exp(m_g - M) of every group, from the raw logits (before they are overwritten)
        for (int gi = tid; gi < V * pgs; gi += 256) {
            const int v = gi / pgs, p0 = (gi % pgs) * tile_steps;
            float mg = -INFINITY;
            for (int p = p0; p < p0 + tile_steps && p < P; ++p) mg = fmaxf(mg, s_w[v * P + p]);
            s_g[gi] = expf(mg - m);
        }
        __syncthreads();
    }
    float sum = 0.0f;
    for (int s = tid; s < S; s += 256) { const float e = expf(s_w[s] - m); s_w[s] = e; sum += e; }
    sum = wave_sum(sum);
    if (lane == 0) s_red[4 + wave] = sum;
    __syncthreads();
    sum = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    for (int s = tid; s < S; s += 256) {
        const float w = s_w[s] / sum;
        s_w[s] = w;
        w_out[row_of(s)] = w;
    }
    if constexpr (PARTS) { for (int gi = tid; gi < V * pgs; gi += 256) s_g[gi] = s_g[gi] / sum; }
    __syncthreads();
    const float* wv = PARTS ? s_g : s_w;
    // 3. z = sum_s w_s val[s] (+ scale * zprev), replicated `reps` times
    if (streaming) {
        // Streaming form for wide rows: a 16-lane group reads a whole row as ceil(D/64) float4 loads (16 B per lane, the row's D*4
        // bytes contiguous; the lanes past the end of a last, partial segment sit out), the 16 groups of the workgroup take samples
        // s = 16 it + group; every load instruction moves up to 1 KB and 4 iterations are in flight.  Partial sums meet in LDS.
        if constexpr (PARTS) {   // the first iteration's rows were loaded in front of the softmax: w * x (what fmaf(w, x, 0) gave)
            const float w = grp < NV ? wv[grp] : 0.0f;
#pragma unroll
            for (int j = 0; j < kMaxSeg; ++j) { part[j].x *= w; part[j].y *= w; part[j].z *= w; part[j].w *= w; }
        } else {
#pragma unroll
            for (int j = 0; j < kMaxSeg; ++j) part[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int s0 = PARTS ? 16 : 0; s0 < NV; s0 += 16) {
            const int sidx = s0 + grp;
            const bool on = sidx < NV;
            const float w = on ? wv[sidx] : 0.0f;
            const float* rowp = val + vrow_of(on ? sidx : NV - 1) * D + 4 * sub;
#pragma unroll
            for (int j = 0; j < kMaxSeg; ++j) {
                if (j < nseg && 64 * j + 4 * sub < D) {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    const f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rowp + 64 * j));
                    const float4 x = make_float4(xv[0], xv[1], xv[2], xv[3]);
                    part[j].x = fmaf(w, x.x, part[j].x); part[j].y = fmaf(w, x.y, part[j].y);
                    part[j].z = fmaf(w, x.z, part[j].z); part[j].w = fmaf(w, x.w, part[j].w);
                }
            }
        }
        // the four groups of a wave hold different samples of the same channels: fold them, then the four waves through LDS
#pragma unroll
        for (int j = 0; j < kMaxSeg; ++j) {
            if (j < nseg) {
                float4 v4 = part[j];
                v4.x += __shfl_xor(v4.x, 16, 64); v4.y += __shfl_xor(v4.y, 16, 64); v4.z += __shfl_xor(v4.z, 16, 64); v4.w += __shfl_xor(v4.w, 16, 64);
                v4.x += __shfl_xor(v4.x, 32, 64); v4.y += __shfl_xor(v4.y, 32, 64); v4.z += __shfl_xor(v4.z, 32, 64); v4.w += __shfl_xor(v4.w, 32, 64);
                if (lane < 16) *reinterpret_cast<float4*>(s_z + wave * (64 * kMaxSeg) + 64 * j + 4 * sub) = v4;
            }
        }
        __syncthreads();
        for (int d = tid; d < D; d += 256) {
            float acc = (s_z[d] + s_z[64 * kMaxSeg + d]) + (s_z[2 * 64 * kMaxSeg + d] + s_z[3 * 64 * kMaxSeg + d]);
            if (zprev) acc += zprev_scale * zprev[((long)sc * R + r) * D + d];
            for (int k = 0; k < reps; ++k) z_out[((long)sc * R + r) * ld_z + (long)k * D + d] = acc;
        }
    } else
    for (int d = tid; d < D; d += 256) {
        float acc = 0.0f;
        for (int s = 0; s < NV; ++s) acc += wv[s] * val[vrow_of(s) * D + d];
        if (zprev) acc += zprev_scale * zprev[((long)sc * R + r) * D + d];
        for (int k = 0; k < reps; ++k) z_out[((long)sc * R + r) * ld_z + (long)k * D + d] = acc;
    }
    // 4. depth read-out and per-view argmax (wave 0 / wave 1)
    if (pt && wave == 0) {
        float acc[3] = {0.0f, 0.0f, 0.0f};
        for (int s = lane; s < S; s += 64) {
            const float* q = pt + row_of(s) * 3;
            for (int k = 0; k < 3; ++k) acc[k] += s_w[s] * fminf(fmaxf(q[k], -100.0f), 100.0f);
        }
        for (int k = 0; k < 3; ++k) acc[k] = wave_sum(acc[k]);
        if (lane == 0) {
            const float* Mi = poses[sc * V].inv_q;
            const float zc = ((acc[0] * Mi[8] + acc[1] * Mi[9]) + acc[2] * Mi[10]) + Mi[11];
            depth[(long)sc * R + r] = fminf(fmaxf(zc, 0.0f), 10.0f);
        }
    }
    if (w_argmax && wave == 1) {
        for (int v = 0; v < V; ++v) {
            float best = -1.0f;
            int bi = 0x7fffffff;
            for (int p = lane; p < P; p += 64) { const float w = s_w[v * P + p]; if (w > best) { best = w; bi = p; } }
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            // all-NaN weights leave the sentinel: torch.argmax returns the first NaN's index then
            if (lane == 0) w_argmax[(long)(sc * V + v) * R + r] = bi == 0x7fffffff ? 0 : bi;
        }
    }
}

__global__ void add_ray_bias_relu_kernel(float* __restrict__ r, const float* __restrict__ u, int b, int V, int R, int P,
                                         int C, long total4) {
    const int c4 = C / 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (long)gridDim.x * blockDim.x) {
        const int q = (int)(idx % c4);
        const long row = idx / c4;                  // ((sc*V+v)*R + ray)*P + p
        const long nr = row / P;
        const int ray = (int)(nr % R);
        const int sc = (int)(nr / R) / V;
        float4 x = *reinterpret_cast<float4*>(r + row * C + 4 * q);
        const float4 y = *reinterpret_cast<const float4*>(u + ((long)sc * R + ray) * C + 4 * q);
        x.x = fmaxf(x.x + y.x, 0.0f); x.y = fmaxf(x.y + y.y, 0.0f);
        x.z = fmaxf(x.z + y.z, 0.0f); x.w = fmaxf(x.w + y.w, 0.0f);
        *reinterpret_cast<float4*>(r + row * C + 4 * q) = x;
    }
}

}  // namespace

extern "C" int car_attend(const float* qa, const float* qb, int dq, const float* val, int D, int b, int V, int R,
                          int P, const float* zprev, float zprev_scale, float* w_out, float* z_out, int ld_z, int reps,
                          const float* pt, const float* poses, float* depth, int32_t* w_argmax, void* stream) {
    CAR_REQUIRE(qa && val && w_out && z_out, "car_attend: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && V <= CAR_MAX_VIEWS && R > 0 && P > 0 && V * P <= kMaxSamples, "car_attend: bad sizes");
    CAR_REQUIRE((!qb || (dq > 0 && dq % 4 == 0)) && D > 0 && reps >= 1 && ld_z >= reps * D, "car_attend: bad widths dq=%d D=%d", dq, D);
    CAR_REQUIRE(!pt || (poses && depth), "car_attend: pt needs poses and depth");
    (void)hipGetLastError();
    auto kern = D <= 64 * kNarrowSeg ? attend_kernel<false, kNarrowSeg> : attend_kernel<false, kWideSeg>;
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)b * R)), dim3(256), 0, (hipStream_t)stream, qa, qb, dq, val,
                       D, b, V, R, P, zprev, zprev_scale, w_out, z_out, ld_z, reps, pt, (const CarPose*)poses, depth,
                       w_argmax, 0);
    CAR_CHECK_LAUNCH("car_attend");
    return CAR_OK;
}

extern "C" int car_attend_parts(const float* logit, const float* part, int tile_steps, int D, int b, int V, int R, int P, float* w_out,
                                float* z_out, int ld_z, int reps, const float* pt, const float* poses, float* depth, int32_t* w_argmax,
                                void* stream) {
    CAR_REQUIRE(logit && part && w_out && z_out, "car_attend_parts: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && V <= CAR_MAX_VIEWS && R > 0 && P > 0 && V * P <= kMaxSamples, "car_attend_parts: bad sizes");
    CAR_REQUIRE(tile_steps == car_fused_tile_steps(), "car_attend_parts: tile_steps = %d, but `part` is made in groups of car_fused_tile_steps() = %d steps",
                tile_steps, car_fused_tile_steps());
    CAR_REQUIRE(tile_steps >= 4 && D > 0 && reps >= 1 && ld_z >= reps * D, "car_attend_parts: bad widths D=%d tile_steps=%d", D, tile_steps);
    CAR_REQUIRE(!pt || (poses && depth), "car_attend_parts: pt needs poses and depth");
    (void)hipGetLastError();
    auto kern = D <= 64 * kNarrowSeg ? attend_kernel<true, kNarrowSeg> : attend_kernel<true, kWideSeg>;
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)b * R)), dim3(256), 0, (hipStream_t)stream, logit, (const float*)nullptr, 0,
                       part, D, b, V, R, P, (const float*)nullptr, 0.0f, w_out, z_out, ld_z, reps, pt, (const CarPose*)poses, depth, w_argmax,
                       tile_steps);
    CAR_CHECK_LAUNCH("car_attend_parts");
    return CAR_OK;
}

extern "C" int car_add_ray_bias_relu(float* r, const float* u, int b, int V, int R, int P, int C, void* stream) {
    CAR_REQUIRE(r && u, "car_add_ray_bias_relu: null pointer");
    CAR_REQUIRE(b > 0 && V > 0 && R > 0 && P > 0 && C > 0 && C % 4 == 0, "car_add_ray_bias_relu: bad sizes");
    const long total4 = (long)b * V * R * P * (C / 4);
    const unsigned blocks = (unsigned)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
    (void)hipGetLastError();
    hipLaunchKernelGGL(add_ray_bias_relu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, r, u, b, V, R, P, C, total4);
    CAR_CHECK_LAUNCH("car_add_ray_bias_relu");
    return CAR_OK;
}
