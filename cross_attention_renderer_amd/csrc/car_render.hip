// car_render.hip — the one-call forward of the C ABI (include/car_hip.h: car_plan_*, car_project_maps, car_render_forward).
//
// Host-side C++ only (plus three trivial re-layout kernels): it carves the caller's plan / workspace buffers and issues the
// same launch sequence as cross_attention_renderer_amd/engine.py::RenderEngine._render_fused + _finish for the reference's
// default configuration, so that a host without Python gets `CrossAttentionRenderer.forward(input, z=z)`
// (reference models.py:190-626) from plain pointers.  tests/test_hip_parity.py checks it bit for bit against the Python engine.
#include "car_common.h"
#include <math.h>
#include <string.h>

extern "C" size_t car_fused2_blob_floats(void);
extern "C" size_t car_fused_bias_floats(void);

namespace {

constexpr int kC = 576, kE = 288, kD = 128, kPhiIn = 18, kPhiLd = 20, kBlocks = 3;
constexpr int kWShift = 8;                         // the split-fp16 weights carry 2^8 (car_fused.hip)
constexpr int kTile16 = 512;                       // floats per (K step, 16-channel tile) of the car_fused2.hip blob
// tile offsets of the layers inside that blob (car_fused16.h): W2 | Q1 | Q2 | UG | K1 (source 0, source 1) | K2
constexpr int kOffW2 = 0, kOffQ1 = 324, kOffQ2 = 332, kOffUG = 364, kOffK1 = 372, kOffK2 = 516;

inline size_t up64(size_t x) { return (x + 63) & ~(size_t)63; }

// ---- re-layout kernels ---------------------------------------------------------------------------------------------
// A-operand tiles of v_mfma_f32_16x16x32_f16 with fp16 hi/lo halves (engine._pack_tiles16_f16_split): per (K step, tile)
// [hi|lo][lane][8 halves]; lane l carries output 16 t + l % 16 and k = 32 ks + 8 (l >> 4) + e (mode 0) or the accumulator
// order base + 16 (2 ks + e / 4) + 4 (l >> 4) + e % 4 (mode 1); k == K selects the bias, k > K a zero.
__global__ void pack16_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ bias, int N, int K, int n_tiles,
                              int ksteps, int mode, int base, _Float16* __restrict__ out) {
    const long total = (long)ksteps * n_tiles * 512;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long tile = idx >> 9;
        const int t = (int)(tile % n_tiles), ks = (int)(tile / n_tiles);
        const int n = 16 * t + (lane & 15), q = lane >> 4;
        const int k = mode == 0 ? 32 * ks + 8 * q + e : base + 16 * (2 * ks + e / 4) + 4 * q + e % 4;
        float w = 0.0f;
        if (n < N) {
            if (k < K) w = W[(long)n * ldw + k] * (float)(1 << kWShift);
            else if (k == K && bias) w = bias[n] * (float)(1 << kWShift);
        }
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        _Float16* o = out + tile * 1024 + lane * 8 + e;
        o[0] = hi;
        o[512] = lo;
    }
}
// query_repeat_embed_2 for car_round2.hip (engine.pack_round2_weights): [chunk 4][tile 4][K group 2][hi|lo][lane][8 halves],
// output 32 t + l % 32, k = 32 c + 16 (l >> 5) + 8 kg + e
__global__ void pack32_kernel(const float* __restrict__ W, _Float16* __restrict__ out) {
    const int total = 4 * 4 * 2 * 2 * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, hl = (idx >> 9) & 1, kg = (idx >> 10) & 1, t = (idx >> 11) & 3, c = idx >> 13;
        const int n = 32 * t + (lane & 31), k = 32 * c + 16 * (lane >> 5) + 8 * kg + e;
        const float w = W[n * kD + k] * (float)(1 << kWShift);
        const _Float16 hi = (_Float16)w;
        out[idx] = hl == 0 ? hi : (_Float16)(w - (float)hi);
    }
}
// [C][4] table (W1[:, C:C+3], b1) of the per-texel first layer
__global__ void wpt_kernel(const float* __restrict__ w1, const float* __restrict__ b1, float* __restrict__ wpt) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < kC) {
        wpt[4 * ch + 0] = w1[(long)ch * (kC + 3) + kC + 0];
        wpt[4 * ch + 1] = w1[(long)ch * (kC + 3) + kC + 1];
        wpt[4 * ch + 2] = w1[(long)ch * (kC + 3) + kC + 2];
        wpt[4 * ch + 3] = b1[ch];
    }
}
// dst[r, 0:D) = scale * src[r, 0:D)   (z_local term of models.py:561-565 before the value projection is accumulated onto it)
__global__ void scale_rows_kernel(const float* __restrict__ src, int D, float scale, float* __restrict__ dst, int ld, long rows) {
    const long total = rows * D;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x)
        dst[(idx / D) * ld + idx % D] = src[idx] * scale;
}
// z[r, v*D : (v+1)*D) = z[r, 0:D) for v = 1..V-1   (per-view replication, models.py:541, 565, 605-606)
__global__ void replicate_views_kernel(float* __restrict__ z, int D, int V, long rows) {
    const long total = rows * D * (V - 1);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / (D * (V - 1));
        const int c = (int)(idx % (D * (V - 1)));
        z[r * D * V + D + c] = z[r * D * V + c % D];
    }
}

// ---- plan layout ---------------------------------------------------------------------------------------------------
struct Plan {
    size_t steps, blob, fbias, wpt, r2w, r2b, proj[CAR_MAX_LEVELS], latent_value, encode_latent, qre_h, lin_in, lin_out,
        lin_z[kBlocks], fc0[kBlocks], fc1[kBlocks], total;         // offsets in floats
};
Plan plan_layout(const car_dims& d) {
    Plan p;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += up64(n); return at; };
    p.steps = take((size_t)d.P);
    p.blob = take(car_fused2_blob_floats());
    p.fbias = take(car_fused_bias_floats());
    p.wpt = take((size_t)kC * 4);
    p.r2w = take(4 * 4 * 1024);
    p.r2b = take(kD);
    for (int l = 0; l < CAR_MAX_LEVELS; ++l) p.proj[l] = l < d.n_levels ? take(car_linear_packed_floats(d.level_c[l], kC)) : 0;
    p.latent_value = take(car_linear_packed_floats(kC, kE));
    p.encode_latent = take(car_linear_packed_floats(kE, kD));
    p.qre_h = take(car_linear_packed_floats(kD, kD));
    p.lin_in = take(car_linear_packed_floats(kPhiIn, kD));
    p.lin_out = take(car_linear_packed_floats(kD, 3));
    for (int i = 0; i < kBlocks; ++i) {
        p.lin_z[i] = take(car_linear_packed_floats(2 * kE, kD));
        p.fc0[i] = take(car_linear_packed_floats(kD, kD));
        p.fc1[i] = take(car_linear_packed_floats(kD, kD));
    }
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
    return CAR_OK;
}

// ---- workspace layout ----------------------------------------------------------------------------------------------
struct Work {
    size_t rays, phi_x, e, q, ug, logit, logit2, pt, pixel_val, coords, at_wt, at_wt2, amax, depth, ebar, z1, hb, uh, zrep, x, net,
        out3, valid, total;                                        // offsets in floats
};
Work work_layout(const car_dims& d) {
    Work w;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += up64(n); return at; };
    const size_t n = (size_t)d.b * d.V, S = n * d.R * d.P, BR = (size_t)d.b * d.R;
    w.rays = take(n * d.R * CAR_RAY_FLOATS);
    w.phi_x = take(BR * kPhiLd);
    w.e = take(S * kC);
    w.q = take(S * kD);
    w.ug = take(S * kD);
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
    w.hb = take(BR * kD);
    w.uh = take(BR * kD);
    w.zrep = take(BR * 2 * kE);
    w.x = take(BR * kD);
    w.net = take(BR * kD);
    w.out3 = take(BR * 4);
    w.valid = take(BR);
    w.total = o;
    return w;
}

#define CAR_TRY(call)                 \
    do {                              \
        const int rc_ = (call);       \
        if (rc_ != CAR_OK) return rc_; \
    } while (0)

}  // namespace

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
extern "C" size_t car_gmaps_floats(const car_dims* dims) {
    if (check_dims(dims, "car_gmaps_floats") != CAR_OK) return 0;
    size_t n = 0;
    for (int l = 0; l < dims->n_levels; ++l) n += (size_t)dims->b * dims->V * dims->level_h[l] * dims->level_w[l] * kC;
    return n;
}

extern "C" int car_plan_build(const car_dims* dims, const car_weights* w, void* plan, void* stream) {
    CAR_TRY(check_dims(dims, "car_plan_build"));
    CAR_REQUIRE(w && plan, "car_plan_build: null pointer");
    const float* const* all = reinterpret_cast<const float* const*>(w);
    for (size_t k = 0; k < sizeof(car_weights) / sizeof(const float*); ++k)
        CAR_REQUIRE(all[k], "car_plan_build: weight pointer %zu of car_weights is null", k);
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
    // split-fp16 operand tiles of the fused per-sample kernel, in its consumption layout
    _Float16* blob = reinterpret_cast<_Float16*>(base + p.blob);
    auto pack16 = [&](const float* W, int ldw, const float* bias, int N, int K, int n_tiles, int ksteps, int mode, int kbase, int tile_off) {
        hipLaunchKernelGGL(pack16_kernel, dim3(256), dim3(256), 0, st, W, ldw, bias, N, K, n_tiles, ksteps, mode, kbase,
                           blob + (size_t)tile_off * kTile16 * 2);
    };
    (void)hipGetLastError();
    pack16(w->query_encode_latent_2_w, kC, nullptr, kE, kC, kE / 16, kC / 32, 0, 0, kOffW2);
    pack16(w->query_embed_w, 16, w->query_embed_b, kD, 16, kD / 16, 1, 0, 0, kOffQ1);
    pack16(w->query_embed_2_w, kD, nullptr, kD, kD, kD / 16, 4, 1, 0, kOffQ2);
    pack16(w->query_repeat_embed_w + kD, kD + 16, w->query_repeat_embed_b, kD, 16, kD / 16, 1, 0, 0, kOffUG);
    pack16(w->key_map_w, kC, nullptr, kD, kC, kD / 16, 9, 1, 0, kOffK1);
    pack16(w->key_map_w, kC, nullptr, kD, kC, kD / 16, 9, 1, kE, kOffK1 + 9 * (kD / 16));
    pack16(w->key_map_2_w, kD, nullptr, kD, kD, kD / 16, 4, 1, 0, kOffK2);
    hipLaunchKernelGGL(pack32_kernel, dim3(128), dim3(256), 0, st, w->query_repeat_embed_2_w, reinterpret_cast<_Float16*>(base + p.r2w));
    hipLaunchKernelGGL(wpt_kernel, dim3((kC + 255) / 256), dim3(256), 0, st, w->query_encode_latent_w, w->query_encode_latent_b, base + p.wpt);
    CAR_CHECK_LAUNCH("car_plan_build");
    const float* fb[4] = {w->query_encode_latent_2_b, w->query_embed_2_b, w->key_map_b, w->key_map_2_b};
    const int fbn[4] = {kE, kD, kD, kD};
    size_t at = p.fbias;
    for (int k = 0; k < 4; ++k) {
        if (hipMemcpyAsync(base + at, fb[k], sizeof(float) * fbn[k], hipMemcpyDeviceToDevice, st) != hipSuccess) { car_set_error("car_plan_build: bias copy failed"); return CAR_E_LAUNCH; }
        at += fbn[k];
    }
    if (hipMemcpyAsync(base + p.r2b, w->query_repeat_embed_2_b, sizeof(float) * kD, hipMemcpyDeviceToDevice, st) != hipSuccess) { car_set_error("car_plan_build: bias copy failed"); return CAR_E_LAUNCH; }
    // fp32 MFMA layers (car_linear.hip)
    int coff = 0;
    for (int l = 0; l < dims->n_levels; ++l) {
        CAR_TRY(car_linear_pack(w->query_encode_latent_w + coff, kC + 3, nullptr, dims->level_c[l], kC, base + p.proj[l], stream));
        coff += dims->level_c[l];
    }
    CAR_TRY(car_linear_pack(w->latent_value_w, kC, w->latent_value_b, kC, kE, base + p.latent_value, stream));
    CAR_TRY(car_linear_pack(w->encode_latent_w, kE, w->encode_latent_b, kE, kD, base + p.encode_latent, stream));
    CAR_TRY(car_linear_pack(w->query_repeat_embed_w, kD + 16, nullptr, kD, kD, base + p.qre_h, stream));
    CAR_TRY(car_linear_pack(w->phi_lin_in_w, kPhiIn, w->phi_lin_in_b, kPhiIn, kD, base + p.lin_in, stream));
    CAR_TRY(car_linear_pack(w->phi_lin_out_w, kD, w->phi_lin_out_b, kD, 3, base + p.lin_out, stream));
    for (int i = 0; i < kBlocks; ++i) {
        CAR_TRY(car_linear_pack(w->phi_lin_z_w[i], 2 * kE, w->phi_lin_z_b[i], 2 * kE, kD, base + p.lin_z[i], stream));
        CAR_TRY(car_linear_pack(w->phi_fc_0_w[i], kD, w->phi_fc_0_b[i], kD, kD, base + p.fc0[i], stream));
        CAR_TRY(car_linear_pack(w->phi_fc_1_w[i], kD, w->phi_fc_1_b[i], kD, kD, base + p.fc1[i], stream));
    }
    return CAR_OK;
}

extern "C" int car_project_maps(const car_dims* dims, const void* plan, const float* const* maps, float* gmaps, void* stream) {
    CAR_TRY(check_dims(dims, "car_project_maps"));
    CAR_REQUIRE(plan && maps && gmaps, "car_project_maps: null pointer");
    const Plan p = plan_layout(*dims);
    const float* base = static_cast<const float*>(plan);
    size_t at = 0;
    for (int l = 0; l < dims->n_levels; ++l) {
        CAR_REQUIRE(maps[l], "car_project_maps: level %d is null", l);
        const long M = (long)dims->b * dims->V * dims->level_h[l] * dims->level_w[l];
        CAR_TRY(car_linear(maps[l], dims->level_c[l], base + p.proj[l], dims->level_c[l], kC, gmaps + at, kC, M, 0, stream));
        at += (size_t)M * kC;
    }
    return CAR_OK;
}

extern "C" int car_render_forward(const car_dims* dims, const void* plan, const car_inputs* in, const car_outputs* out,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    CAR_TRY(check_dims(dims, "car_render_forward"));
    CAR_REQUIRE(plan && in && out && workspace, "car_render_forward: null pointer");
    CAR_REQUIRE(in->poses && in->uv && in->gmaps && out->rgb, "car_render_forward: poses, uv, gmaps and rgb are required");
    const car_dims& d = *dims;
    const Plan p = plan_layout(d);
    const Work w = work_layout(d);
    CAR_REQUIRE(workspace_bytes >= w.total * sizeof(float), "car_render_forward: workspace of %zu bytes, %zu needed", workspace_bytes,
                w.total * sizeof(float));
    const float* pl = static_cast<const float*>(plan);
    float* ws = static_cast<float*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    const int b = d.b, V = d.V, R = d.R, P = d.P;
    const float* steps = in->steps ? in->steps : pl + p.steps;
    const long BR = (long)b * R, S = (long)b * V * R * P;
    float* coords = out->coords ? out->coords : ws + w.coords;
    float* pixel_val = out->pixel_val ? out->pixel_val : ws + w.pixel_val;
    float* at_wt = out->at_wt ? out->at_wt : ws + w.at_wt;
    float* depth = out->depth_ray ? out->depth_ray : ws + w.depth;
    float* valid = out->valid_mask ? out->valid_mask : ws + w.valid;
    int32_t* amax = out->at_wt_max ? out->at_wt_max : reinterpret_cast<int32_t*>(ws + w.amax);

    // a4-a6: rays, their epipolar segments, the decoder's ray input (columns 18, 19 of phi_x stay zero)
    if (hipMemsetAsync(ws + w.phi_x, 0, sizeof(float) * BR * kPhiLd, st) != hipSuccess) { car_set_error("car_render_forward: memset failed"); return CAR_E_LAUNCH; }
    CAR_TRY(car_ray_setup(in->poses, in->uv, b, V, R, d.H, d.W, P, 0, steps, ws + w.rays, coords, ws + w.phi_x, kPhiLd, stream));
    // a6-a13 + round-1 logits: the fused per-sample kernel
    const float* gm[3];
    size_t at = 0;
    for (int l = 0; l < 3; ++l) { gm[l] = in->gmaps + at; at += (size_t)b * V * d.level_h[l] * d.level_w[l] * kC; }
    CAR_TRY(car_fused_samples_v4(in->poses, ws + w.rays, steps, gm, d.level_h, d.level_w, 3, kC, pl + p.wpt, pl + p.blob, pl + p.fbias,
                                 b, V, R, P, d.H, d.W, ws + w.e, ws + w.q, ws + w.ug, ws + w.logit, ws + w.pt, pixel_val, stream));
    // a14 + a16: attention round 1, depth read-out, argmax
    CAR_TRY(car_attend(ws + w.logit, nullptr, kD, ws + w.e, kC, b, V, R, P, nullptr, 0.0f, at_wt, ws + w.ebar, kC, 1, ws + w.pt, in->poses,
                       depth, amax, stream));
    float* zrep = ws + w.zrep;
    if (d.repeat_attention) {
        // a15: z1 = Wv ebar + bv; second-round query; logits; attention; z = (Wv ebar2 + bv) + V z1
        CAR_TRY(car_linear(ws + w.ebar, kC, pl + p.latent_value, kC, kE, ws + w.z1, kE, BR, 0, stream));
        CAR_TRY(car_linear(ws + w.z1, kE, pl + p.encode_latent, kE, kD, ws + w.hb, kD, BR, 0, stream));
        CAR_TRY(car_linear(ws + w.hb, kD, pl + p.qre_h, kD, kD, ws + w.uh, kD, BR, 0, stream));
        CAR_TRY(car_round2_logits(ws + w.ug, ws + w.uh, ws + w.q, pl + p.r2w, pl + p.r2b, b, V, R, P, ws + w.logit2, stream));
        CAR_TRY(car_attend(ws + w.logit2, nullptr, kD, ws + w.e, kC, b, V, R, P, nullptr, 0.0f, ws + w.at_wt2, ws + w.ebar, kC, 1, nullptr,
                           nullptr, nullptr, nullptr, stream));
        (void)hipGetLastError();
        hipLaunchKernelGGL(scale_rows_kernel, dim3(1024), dim3(256), 0, st, ws + w.z1, kE, (float)V, zrep, V * kE, BR);
        CAR_CHECK_LAUNCH("car_render_forward (scale)");
        CAR_TRY(car_linear(ws + w.ebar, kC, pl + p.latent_value, kC, kE, zrep, V * kE, BR, CAR_LIN_ACCUM, stream));
    } else {
        CAR_TRY(car_linear(ws + w.ebar, kC, pl + p.latent_value, kC, kE, zrep, V * kE, BR, 0, stream));
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(replicate_views_kernel, dim3(1024), dim3(256), 0, st, zrep, kE, V, BR);
    CAR_CHECK_LAUNCH("car_render_forward (replicate)");
    // a17: light-field decoder (resnet_block_fc.py:132-168)
    CAR_TRY(car_linear(ws + w.phi_x, kPhiLd, pl + p.lin_in, kPhiIn, kD, ws + w.x, kD, BR, 0, stream));
    for (int i = 0; i < kBlocks; ++i) {
        CAR_TRY(car_linear(zrep, V * kE, pl + p.lin_z[i], V * kE, kD, ws + w.x, kD, BR, CAR_LIN_ACCUM, stream));
        CAR_TRY(car_linear(ws + w.x, kD, pl + p.fc0[i], kD, kD, ws + w.net, kD, BR, CAR_LIN_RELU_IN, stream));
        CAR_TRY(car_linear(ws + w.net, kD, pl + p.fc1[i], kD, kD, ws + w.x, kD, BR, CAR_LIN_RELU_IN | CAR_LIN_ACCUM, stream));
    }
    CAR_TRY(car_linear(ws + w.x, kD, pl + p.lin_out, kD, 3, ws + w.out3, 4, BR, CAR_LIN_RELU_IN, stream));
    // a18: valid mask, white background
    CAR_TRY(car_finalize(ws + w.rays, ws + w.out3, 4, b, V, R, out->rgb, valid, stream));
    (void)S;
    return CAR_OK;
}
