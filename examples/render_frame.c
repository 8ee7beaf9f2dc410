/* render_frame.c — a C host that renders one query view through libcar_hip.so, no Python anywhere.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/render_frame.c -I/opt/rocm/include -Iinclude \
 *       -Lcross_attention_renderer_amd -L/opt/rocm/lib -lcar_hip -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/cross_attention_renderer_amd -Wl,-rpath,/opt/rocm/lib -o render_frame && ./render_frame [H] [P]
 *
 * It stands where the reference's render script calls `model(model_input, z=z)` (render_realestate10k_traj.py:128-130): weights in
 * the reference's state_dict layout, the encoder's feature pyramid, camera matrices and pixel coordinates go in as device
 * pointers, `rgb / depth_ray / valid_mask` come out.  Weights and features are synthetic here (a small LCG); a real host uploads
 * the checkpoint tensors instead.  `./render_frame --fixture DIR` renders a reference fixture exported as raw files instead (below);
 * tests/test_c_host.py builds and runs both and compares the fixture render with the reference's committed outputs. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "car_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_CAR(x) do { int r_ = (x); if (r_ != CAR_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, car_last_error()); return 3; } } while (0)

static unsigned g_seed = 12345u;
static float frand(void) { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xffff) / 32768.0f - 1.0f; }   /* [-1, 1) */

/* uploads n floats drawn as scale * U(-1, 1) */
static float* upload_random(size_t n, float scale) {
    float* h = (float*)malloc(n * sizeof(float));
    float* d = NULL;
    for (size_t i = 0; i < n; ++i) h[i] = scale * frand();
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) { free(h); return NULL; }
    hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
    free(h);
    return d;
}
static float* upload(const float* h, size_t n) {
    float* d = NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
    return d;
}

/* ---- fixture mode: ./render_frame --fixture DIR [--host-poses] -------------------------------------------------------------
 * DIR holds raw little-endian float32 files (tests/test_c_host.py writes them from a reference fixture): dims.txt, one file per
 * car_weights field, map0..2.bin (channel-last pyramid levels), c2w_ctx / c2w_q / K_ctx / K_q / uv / steps .bin and, with
 * --host-poses, poses.bin (the reference's own pose matrices as CarPose records).  Writes rgb / depth / valid .bin back. */
static float* read_floats(const char* dir, const char* name, size_t n, int to_device) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return NULL; }
    float* h = (float*)malloc(n * sizeof(float));
    const size_t got = fread(h, sizeof(float), n, f);
    fclose(f);
    if (got != n) { fprintf(stderr, "%s: %zu floats, %zu expected\n", path, got, n); free(h); return NULL; }
    if (!to_device) return h;
    float* d = upload(h, n);
    free(h);
    return d;
}
static int write_floats(const char* dir, const char* name, const float* dev, size_t n) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    float* h = (float*)malloc(n * sizeof(float));
    if (hipMemcpy(h, dev, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { free(h); return 1; }
    FILE* f = fopen(path, "wb");
    if (!f) { free(h); return 1; }
    fwrite(h, sizeof(float), n, f);
    fclose(f);
    free(h);
    return 0;
}
static int run_fixture(const char* dir, int host_poses) {
    car_dims d;
    memset(&d, 0, sizeof d);
    char path[1024];
    snprintf(path, sizeof path, "%s/dims.txt", dir);
    FILE* f = fopen(path, "r");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return 2; }
    if (fscanf(f, "%d %d %d %d %d %d %d %d", &d.b, &d.V, &d.R, &d.P, &d.H, &d.W, &d.n_levels, &d.repeat_attention) != 8) { fclose(f); return 2; }
    for (int l = 0; l < d.n_levels && l < CAR_MAX_LEVELS; ++l)
        if (fscanf(f, "%d %d %d", &d.level_h[l], &d.level_w[l], &d.level_c[l]) != 3) { fclose(f); return 2; }
    fclose(f);
    car_weights w;
#define RD(field, n) if (!(w.field = read_floats(dir, #field, (n), 1))) return 2
#define RLAYER(name, N, K) RD(name##_w, (size_t)(N) * (K)); RD(name##_b, (N))
    RLAYER(query_encode_latent, 576, 579); RLAYER(query_encode_latent_2, 288, 576); RLAYER(latent_value, 288, 576);
    RLAYER(key_map, 128, 576); RLAYER(key_map_2, 128, 128); RLAYER(query_embed, 128, 16); RLAYER(query_embed_2, 128, 128);
    RLAYER(query_repeat_embed, 128, 144); RLAYER(query_repeat_embed_2, 128, 128); RLAYER(encode_latent, 128, 288);
    RLAYER(phi_lin_in, 128, 18); RLAYER(phi_lin_out, 3, 128);
    for (int i = 0; i < 3; ++i) {
        char nm[64];
#define RDI(field, n) snprintf(nm, sizeof nm, #field "%d", i); if (!(w.field[i] = read_floats(dir, nm, (n), 1))) return 2
        RDI(phi_lin_z_w, 128 * 576); RDI(phi_lin_z_b, 128); RDI(phi_fc_0_w, 128 * 128); RDI(phi_fc_0_b, 128);
        RDI(phi_fc_1_w, 128 * 128); RDI(phi_fc_1_b, 128);
    }
    void* plan = NULL;
    CHECK_HIP(hipMalloc(&plan, car_plan_bytes(&d)));
    CHECK_CAR(car_plan_build(&d, &w, plan, NULL));
    const float* maps[CAR_MAX_LEVELS];
    for (int l = 0; l < d.n_levels; ++l) {
        char nm[16];
        snprintf(nm, sizeof nm, "map%d", l);
        if (!(maps[l] = read_floats(dir, nm, (size_t)d.b * d.V * d.level_h[l] * d.level_w[l] * d.level_c[l], 1))) return 2;
    }
    float* gmaps = NULL;
    CHECK_HIP(hipMalloc((void**)&gmaps, car_gmaps_floats(&d) * sizeof(float)));
    CHECK_CAR(car_project_maps(&d, plan, maps, gmaps, NULL));
    float* poses = NULL;
    if (host_poses) {
        if (!(poses = read_floats(dir, "poses", (size_t)d.b * d.V * CAR_POSE_FLOATS, 1))) return 2;
    } else {
        float *c2w_ctx = read_floats(dir, "c2w_ctx", (size_t)d.b * d.V * 16, 1), *c2w_q = read_floats(dir, "c2w_q", (size_t)d.b * 16, 1),
              *K_ctx = read_floats(dir, "K_ctx", (size_t)d.b * d.V * 16, 1), *K_q = read_floats(dir, "K_q", (size_t)d.b * 16, 1);
        if (!c2w_ctx || !c2w_q || !K_ctx || !K_q) return 2;
        CHECK_HIP(hipMalloc((void**)&poses, (size_t)d.b * d.V * CAR_POSE_FLOATS * sizeof(float)));
        CHECK_CAR(car_pose_setup(c2w_ctx, c2w_q, K_ctx, K_q, d.b, d.V, d.H, poses, NULL));
    }
    car_inputs in;
    memset(&in, 0, sizeof in);
    in.poses = poses;
    if (!(in.uv = read_floats(dir, "uv", (size_t)d.b * d.R * 2, 1))) return 2;
    if (!(in.steps = read_floats(dir, "steps", (size_t)d.P, 1))) return 2;            /* torch.linspace of the fixture's host */
    in.lattice = gmaps; in.gmeta = gmaps + car_gmeta_offset(&d);
    car_outputs out;
    memset(&out, 0, sizeof out);
    const size_t BR = (size_t)d.b * d.R;
    CHECK_HIP(hipMalloc((void**)&out.rgb, BR * 3 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&out.valid_mask, BR * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&out.depth_ray, BR * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&out.at_wt, BR * d.V * d.P * sizeof(float)));
    void* ws = NULL;
    const size_t ws_bytes = car_workspace_bytes(&d);
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    CHECK_CAR(car_render_forward(&d, plan, &in, &out, ws, ws_bytes, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    if (write_floats(dir, "rgb", out.rgb, BR * 3) || write_floats(dir, "valid", out.valid_mask, BR) || write_floats(dir, "depth", out.depth_ray, BR) ||
        write_floats(dir, "at_wt", out.at_wt, BR * d.V * d.P)) return 4;
    printf("fixture %s rendered: %d scene(s) x %d rays, %s poses\n", dir, d.b, d.R, host_poses ? "host" : "device");
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 2 && strcmp(argv[1], "--fixture") == 0) return run_fixture(argv[2], argc > 3 && strcmp(argv[3], "--host-poses") == 0);
    const int H = argc > 1 ? atoi(argv[1]) : 64, P = argc > 2 ? atoi(argv[2]) : 32;
    car_dims d;
    memset(&d, 0, sizeof d);
    d.b = 1; d.V = 2; d.R = H * H; d.P = P; d.H = H; d.W = H; d.n_levels = 3; d.repeat_attention = 1;
    d.level_h[0] = d.level_w[0] = H / 4; d.level_c[0] = 256;
    d.level_h[1] = d.level_w[1] = H / 2; d.level_c[1] = 256;
    d.level_h[2] = d.level_w[2] = H;     d.level_c[2] = 64;
    printf("libcar_hip %d, %dx%d frame, %d samples per view, %d compute units\n", car_version(), H, H, P, car_device_cu_count());

    /* ---- weights, reference layout [out][in] (models.py:96-144); scale ~ 1/sqrt(fan_in) like the default init ---- */
    car_weights w;
#define LAYER(name, N, K) w.name##_w = upload_random((size_t)(N) * (K), 1.0f / sqrtf((float)(K))); w.name##_b = upload_random((N), 0.05f)
    LAYER(query_encode_latent, 576, 579); LAYER(query_encode_latent_2, 288, 576); LAYER(latent_value, 288, 576);
    LAYER(key_map, 128, 576); LAYER(key_map_2, 128, 128); LAYER(query_embed, 128, 16); LAYER(query_embed_2, 128, 128);
    LAYER(query_repeat_embed, 128, 144); LAYER(query_repeat_embed_2, 128, 128); LAYER(encode_latent, 128, 288);
    LAYER(phi_lin_in, 128, 18); LAYER(phi_lin_out, 3, 128);
    for (int i = 0; i < 3; ++i) {
        w.phi_lin_z_w[i] = upload_random(128 * 576, 1.0f / 24.0f); w.phi_lin_z_b[i] = upload_random(128, 0.05f);
        w.phi_fc_0_w[i] = upload_random(128 * 128, 0.09f); w.phi_fc_0_b[i] = upload_random(128, 0.05f);
        w.phi_fc_1_w[i] = upload_random(128 * 128, 0.09f); w.phi_fc_1_b[i] = upload_random(128, 0.05f);
    }
    void* plan = NULL;
    CHECK_HIP(hipMalloc(&plan, car_plan_bytes(&d)));
    CHECK_CAR(car_plan_build(&d, &w, plan, NULL));

    /* ---- the stereo pair: feature pyramid (channel-last) -> projected maps, once per pair ---- */
    const float* maps[3];
    for (int l = 0; l < 3; ++l) maps[l] = upload_random((size_t)d.V * d.level_h[l] * d.level_w[l] * d.level_c[l], 1.0f);
    float* gmaps = NULL;
    CHECK_HIP(hipMalloc((void**)&gmaps, car_gmaps_floats(&d) * sizeof(float)));
    CHECK_CAR(car_project_maps(&d, plan, maps, gmaps, NULL));

    /* ---- cameras: two context views 0.6 apart with 12 degrees of yaw, query half-way ---- */
    const float f = 0.879f * H, c = 0.5f * H, yaw = -12.0f * 3.14159265f / 180.0f;
    const float K[16] = {f, 0, c, 0,  0, f, c, 0,  0, 0, 1, 0,  0, 0, 0, 1};
    float c2w_ctx[32] = {1, 0, 0, 0,  0, 1, 0, 0,  0, 0, 1, 0,  0, 0, 0, 1,
                         cosf(yaw), 0, sinf(yaw), 0.6f,  0, 1, 0, 0,  -sinf(yaw), 0, cosf(yaw), 0,  0, 0, 0, 1};
    const float hy = 0.5f * yaw;
    const float c2w_q[16] = {cosf(hy), 0, sinf(hy), 0.3f,  0, 1, 0, 0,  -sinf(hy), 0, cosf(hy), 0,  0, 0, 0, 1};
    float K_ctx[32];
    memcpy(K_ctx, K, sizeof K); memcpy(K_ctx + 16, K, sizeof K);
    float *d_c2w_ctx = upload(c2w_ctx, 32), *d_c2w_q = upload(c2w_q, 16), *d_K_ctx = upload(K_ctx, 32), *d_K_q = upload(K, 16), *poses = NULL;
    CHECK_HIP(hipMalloc((void**)&poses, (size_t)d.V * CAR_POSE_FLOATS * sizeof(float)));
    CHECK_CAR(car_pose_setup(d_c2w_ctx, d_c2w_q, d_K_ctx, d_K_q, d.b, d.V, H, poses, NULL));
    float* uv_h = (float*)malloc((size_t)d.R * 2 * sizeof(float));
    for (int r = 0; r < d.R; ++r) { uv_h[2 * r] = (float)(r % H); uv_h[2 * r + 1] = (float)(r / H); }
    float* uv = upload(uv_h, (size_t)d.R * 2);

    /* ---- render ---- */
    void* ws = NULL;
    const size_t ws_bytes = car_workspace_bytes(&d);
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    car_inputs in;
    memset(&in, 0, sizeof in);
    in.poses = poses; in.uv = uv;
    in.lattice = gmaps;
    in.gmeta = gmaps + car_gmeta_offset(&d);
    car_outputs out;
    memset(&out, 0, sizeof out);
    CHECK_HIP(hipMalloc((void**)&out.rgb, (size_t)d.R * 3 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&out.valid_mask, (size_t)d.R * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&out.depth_ray, (size_t)d.R * sizeof(float)));
    hipEvent_t t0, t1;
    CHECK_HIP(hipEventCreate(&t0)); CHECK_HIP(hipEventCreate(&t1));
    CHECK_CAR(car_render_forward(&d, plan, &in, &out, ws, ws_bytes, NULL));           /* warm-up */
    CHECK_HIP(hipEventRecord(t0, NULL));
    CHECK_CAR(car_render_forward(&d, plan, &in, &out, ws, ws_bytes, NULL));
    CHECK_HIP(hipEventRecord(t1, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float ms = 0.0f;
    CHECK_HIP(hipEventElapsedTime(&ms, t0, t1));

    float* rgb = (float*)malloc((size_t)d.R * 3 * sizeof(float));
    float* valid = (float*)malloc((size_t)d.R * sizeof(float));
    float* depth = (float*)malloc((size_t)d.R * sizeof(float));
    CHECK_HIP(hipMemcpy(rgb, out.rgb, (size_t)d.R * 3 * sizeof(float), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(valid, out.valid_mask, (size_t)d.R * sizeof(float), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(depth, out.depth_ray, (size_t)d.R * sizeof(float), hipMemcpyDeviceToHost));
    double sum = 0.0, vsum = 0.0, dsum = 0.0;
    int bad = 0;
    for (int i = 0; i < d.R * 3; ++i) { if (!isfinite(rgb[i])) ++bad; sum += rgb[i]; }
    for (int i = 0; i < d.R; ++i) { vsum += valid[i]; dsum += depth[i]; if (!(depth[i] >= 0.0f && depth[i] <= 10.0f)) ++bad; }
    printf("frame: %.3f ms, %.0f rays/s | mean rgb %.6f | valid %.4f | mean depth %.4f | bad values %d\n", ms, d.R / (ms * 1e-3),
           sum / (d.R * 3), vsum / d.R, dsum / d.R, bad);
    return bad ? 1 : 0;
}
