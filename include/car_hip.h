/* car_hip.h — C ABI of libcar_hip.so, the MI355X (gfx950) implementation of the epipolar cross-attention
 * render forward of yilundu/cross_attention_renderer.
 *
 * What it replaces.  The reference has NO native/FFI boundary on this path: the whole of
 * `CrossAttentionRenderer.forward(input, z=z)` (reference models.py:190-626) is Python calling stock torch
 * ops.  This header therefore does not mirror an existing FFI; it is the boundary this repo introduces
 * underneath the Python class.  Each entry point names the reference code it stands for.
 *
 * Conventions
 *  - Every pointer is a caller-owned DEVICE pointer (e.g. torch.Tensor.data_ptr()) unless marked host;
 *    fp32, contiguous, 16-byte aligned.  The library allocates nothing, frees nothing and keeps no
 *    reference after the call returns.
 *  - All work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream).  No call synchronises the device.
 *  - Return value: 0 on success, a negative CAR_E_* code on failure; `car_last_error()` then returns a
 *    thread-local human-readable message.  Nothing throws or aborts.
 *  - Index conventions: b scenes, V context views, n = b*V + v "scene-views", R rays per scene,
 *    P samples per ray and view.  Feature maps are channel-last (NHWC) per pyramid level.
 */
#ifndef CAR_HIP_H
#define CAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAR_VERSION 300           /* 0.3.0 */

#define CAR_OK            0
#define CAR_E_ARG        (-1)     /* invalid argument (null pointer, bad size, unsupported shape) */
#define CAR_E_LAUNCH     (-2)     /* HIP launch / runtime error */
#define CAR_E_NODEVICE   (-3)     /* no gfx950 device visible */

#define CAR_POSE_FLOATS   96      /* sizeof(struct CarPose)/4, see csrc/car_geom.h */
#define CAR_RAY_FLOATS    12      /* sizeof(struct CarRay)/4 */
#define CAR_G_DIM         16      /* channels of the geometric query (models.py:528) */
#define CAR_MAX_VIEWS     3
#define CAR_MAX_LEVELS    4

/* flags of car_linear */
#define CAR_LIN_RELU_IN   1       /* apply ReLU to the input rows on load            (F.relu(x) feeding a conv) */
#define CAR_LIN_RELU_OUT  2       /* apply ReLU to the result                                                     */
#define CAR_LIN_ACCUM     4       /* Y += result instead of Y = result                 (x = x + lin_z(z), x + dx) */
#define CAR_LIN_NO_GLDS   8       /* stage weights through registers instead of global_load_lds (debug/A-B)       */
#define CAR_WGRAD_FP32    16      /* car_linear_wgrad: keep a wide layer on the fp32 matrix pipe (A/B, tests)       */

/* gather placement patterns (which output row a sampled point lands in), see car_gather_bilinear */
#define CAR_PLACE_PLAIN   0
#define CAR_PLACE_OWN     1
#define CAR_PLACE_OTHER2  2

int car_version(void);
const char* car_last_error(void);

/* Number of compute units of the current device (host query; used by tests/bench to size launches). */
int car_device_cu_count(void);

/* ---- a3: pose algebra (models.py:207-211, 226-228, 285-286; geometry.py:404) ---------------------------
 * c2w_ctx [b,V,4,4], c2w_q [b,4,4], K_ctx [b,V,4,4], K_q [b,4,4]  ->  poses [b*V, CAR_POSE_FLOATS].
 * Inverses are computed by partial-pivot Gauss-Jordan in fp64 (the reference uses LAPACK in fp32; results
 * agree to a few ulp).  Hosts that want the reference's exact matrices may fill `poses` themselves. */
int car_pose_setup(const float* c2w_ctx, const float* c2w_q, const float* K_ctx, const float* K_q,
                   int b, int V, int H, float* poses, void* stream);

/* ---- a4-a6: query rays and their epipolar segments (geometry.py:236-245; epipolar.py:175-253; models.py:213-258)
 * uv [b,R,2] pixel coords (x=col, y=row).  rays [b*V,R,CAR_RAY_FLOATS] (struct CarRay).
 * no_sample != 0 selects the uniform-depth variant (geometry.py:165-187): `depth_steps` [P] = linspace(0.1,10,P).
 * Optional outputs (may be NULL): coords9 [b*V,R,9] = [d, o x d, o]; phi_x [b,R,ld_phi] whose first V*9 columns are
 * the decoder's ray input (models.py:597-602), ld_phi >= 9*V. */
int car_ray_setup(const float* poses, const float* uv, int b, int V, int R, int H, int W, int P,
                  int no_sample, const float* depth_steps, float* rays, float* coords9, float* phi_x, int ld_phi,
                  void* stream);

/* ---- a6, a8, a9, a13: per-sample geometry (models.py:261-331, 494-528; geometry.py:98-162, 313-324, 374-393)
 * steps [P]: linspace(0,1,P) (or the depth steps when no_sample).  Outputs, each optional (NULL to skip):
 *   pixel_val [b*V,R,P,2]   grid_sample coordinates of the sample in its own view
 *   pt        [b*V,R,P,3]   closest point on the query ray (fp64 Pluecker intersection, stored fp32)
 *   g         [b*V,R,P,16]  local_coords
 *   grid_in   [b*V,R,P,V,2] where the point lands in every context view s
 *   xenc/ld_xenc/col_xenc   tanh(nan_to_num(T_s pt)/5) written to xenc[((n*R+r)*P+p)*V + s][col..col+3)
 *                           (the 3 point channels of models.py:330-338); with V == 1: tanh(pt/5), tanh(pt/100)
 *                           at [col..col+6) (models.py:483)
 *   pt_in     [b*V,R,P,V,3] nan_to_num(T_s pt): the point in every context frame s (models.py:288-289, 322-325) */
int car_sample_setup(const float* poses, const float* rays, const float* steps, int b, int V, int R, int P,
                     int H, int W, int no_sample, float* pixel_val, float* pt, float* g, float* grid_in,
                     float* xenc, int ld_xenc, int col_xenc, float* pt_in, void* stream);

/* geometry.project + normalize_for_grid_sample on explicit points (models.py:390-397, the three-view exchange):
 * pts [n_scenes, npts, 3] camera-frame points -> grid [n_scenes, npts, 2], using the intrinsics of context view `view`
 * of each scene (poses[(scene*V + view)].kc). */
int car_project_points(const float* poses, const float* pts, int n_scenes, long npts, int V, int view, int H, int W,
                       float* grid, void* stream);

/* ---- a7/a10: F.grid_sample(bilinear, align_corners=False) over a channel-last pyramid (models.py:278, 317)
 * maps[l] : device pointer to level l, [n_maps, Hl, Wl, Cl] (NHWC); level_c/h/w host arrays of n_levels ints.
 * grid [n_maps, pts, 2].  mode 0 = border, 1 = zeros.  Point i of map m writes sum(Cl) channels to row
 *   PLAIN : m*pts + i
 *   OWN   : (m*pts + i)*V + (m % V)                              (own-view features, source slot = own view)
 *   OTHER2: ((b*2 + (1-s))*pts + i)*2 + s  with m = b*2+s         (V == 2: features of view s for the other line)
 * of `out` (row stride ld_out floats) starting at column col_out.
 * run: when the points of a map are rays x `run` samples (point = ray*run + sample), a work group gathers the same sample of 16
 * neighbouring rays — their taps share cache lines; 1 = no such structure (any other value that does not divide pts is treated as 1).
 * The result does not depend on it. */
int car_gather_bilinear(const float* const* maps, const int* level_c, const int* level_h, const int* level_w,
                        int n_levels, int n_maps, const float* grid, long pts, int run, int mode, int place, int V,
                        float* out, int ld_out, int col_out, void* stream);

/* ---- a7 + a10 + first layer of a11, fused through linearity (models.py:278, 317, 330-341).
 * gmaps[l]: [n_maps, Hl, Wl, C] = the pyramid level l pushed once per texel through its slice of the first point-MLP
 * layer (G_l = query_encode_latent.weight[:, ch_l] F_l, computed with car_linear).  For sample i = (n, j), j < pts, and
 * source view s the kernel writes row i*V + s of `out`:
 *     relu( sum_l bilinear(G_l[map], grid) + wpt[:, 0:3] ptenc[row, 0:3] + wpt[:, 3] )
 * where (map, grid, padding) = (n, pixel_val[i], border) if s is the sample's own view, else
 * ((scene, s), grid_in[i, s], zeros).  ptenc [n_maps*pts*V, 4] = tanh(nan_to_num(T_s pt)/5) (car_sample_setup's xenc
 * with ld 4); wpt [C, 4] = (W1[:, C:C+3], b1).  V must be 2. */
int car_gather_encode(const float* const* gmaps, const int* level_h, const int* level_w, int n_levels, int C,
                      const float* pixel_val, const float* grid_in, const float* ptenc, const float* wpt,
                      int n_maps, int V, long pts, float* out, int ld_out, void* stream);
/* The same layer for an explicit list of rows (the three-view exchange, models.py:345-475): row i gathers map (row_src[i] & 0x3fffffff)
 * at row_grid[i] (2 floats) with padding mode (row_src[i] >> 30) & 1 and adds the point term of row_pe[i] (4 floats, the last unused). */
int car_gather_encode_rows(const float* const* gmaps, const int* level_h, const int* level_w, int n_levels, int Cg,
                           const int* row_src, const float* row_grid, const float* row_pe, const float* wpt, int n_maps,
                           long rows, float* out, int ld_out, void* stream);
/* The same rows from ONE merged lattice per (map, padding mode) instead of the levels: car_merge_lattice sums the projected
 * levels [n_maps, Hl, Wl, 576] on their common integer lattice — [n_maps][2][lat_h][lat_w][576]; with lattice == NULL it only returns the
 * shape — and car_lattice_encode_rows reads four taps of it per row (row_src / row_grid / row_pe as above). */
int car_merge_lattice(const float* const* levels, const int* level_h, const int* level_w, int n_levels, int n_maps, float* lattice,
                      int* lat_h, int* lat_w, int* lat_pad, void* stream);
/* The same merge, which also leaves the lattice's largest magnitude in gmax [1] (taken in the merge's own pass: one atomic per workgroup) —
 * the `gmeta` car_fused_rows wants. */
int car_merge_lattice_max(const float* const* levels, const int* level_h, const int* level_w, int n_levels, int n_maps, float* lattice,
                          float* gmax, void* stream);
int car_lattice_encode_rows(const float* lattice, int lat_h, int lat_w, int lat_pad, int Cg, const int* row_src, const float* row_grid,
                            const float* row_pe, const float* wpt, int n_maps, long rows, float* out, int ld_out, void* stream);
/* car_lattice_encode_rows followed by car_linear_x3 (below) in ONE kernel, for the three-view exchange's second layer (models.py:333-341 on
 * the rows of 345-475): Y[rows, N] = act(relu(lattice rows + point term) W^T + bias) — the 576-wide first-layer rows are gathered 32
 * channels at a time straight into the matrix pipe's operands and never written (2.3 KB per row saved twice).  Same arithmetic, operation
 * for operation, as the two entries it replaces (outputs bit-identical).  packed / bias / flags as car_linear_x3 (K = 576, N % 32 == 0);
 * row_src / row_grid / row_pe / wpt as car_lattice_encode_rows. */
int car_lattice_encode_linear(const float* lattice, int lat_h, int lat_w, int lat_pad, const int* row_src, const float* row_grid,
                              const float* row_pe, const float* wpt, int n_maps, long rows, const float* packed, const float* bias,
                              int K, int N, float* Y, int ldy, int flags, void* stream);

/* ---- a6-a13 + logits of a14 in ONE kernel for the default configuration (V = 2, C = 576, hidden 128): geometry, the
 * per-texel-projected encode (car_gather_encode's arithmetic, with ALL pyramid levels summed once per stereo pair on their common
 * lattice: four taps per sample and source instead of four per level), 576->288 per source, key_map, query_embed and the round-1
 * logits, every layer chained through the MFMA accumulator registers (csrc/car_fused.hip; models.py:261-344, 487-532).
 * Every layer runs on the f16 matrix pipe with both operands split into fp16 high/low halves (three products per term,
 * fp32-class accuracy).  `blob` / `bias` are the layer weights in the kernel's operand order: car_fused_blob_floats() /
 * car_fused_bias_floats() floats, written by car_plan_build (layout: csrc/car_fused_mma.h; the fp16 halves of each layer carry
 * a power of two chosen from the layer's largest weight, recorded in `bias`).
 * `lattice` [b*V][2][lat_h][lat_w][576]: for every context view and padding mode (0 border, 1 zeros) the sum over the pyramid
 * levels of grid_sample(G_l) (G_l as in car_gather_encode), evaluated on the integer lattice u = (x + 1) W_m - 1 in
 * [-lat_pad, 2 W_m - 1 + lat_pad] that the levels' texel centres share (W_m: the widest level; car_project_maps builds it,
 * car_lattice_shape gives its size; csrc/car_geom.h car_lattice_taps).  `gmeta` [1]: max |lattice| (car_project_maps writes it),
 * from which the kernel derives the power of two that keeps the activations of the first layer inside fp16's range.
 * `no_sample` = 1: `rays` come from car_ray_setup(no_sample = 1) and `steps` holds the P depths (models.py:221-222): a sample is the
 * projection of the query ray's point at that depth instead of a point of the clipped epipolar segment; everything after it is the same.
 * Outputs: e [S,576], g [S,16] (the geometric query local_coords, models.py:528), logit [S], pt [S,3],
 * pixel_val [S,2] with S = b*V*R*P.  Neither key nor qry (models.py:491, 529) is formed: logit = <key, qry> / 16 is evaluated as the bilinear
 * form r^T (M x + v) + u^T x + c of r = relu(key_map(e)) and x = relu(query_embed(g)) with M = Wk2^T Wq2 folded once per checkpoint by
 * car_fused_pack (csrc/car_fused_layout.h) — one 128 x 128 layer per sample instead of key_map_2 and query_embed_2.  The lattice of one (view, padding mode) must stay below 2 GiB (nodes are addressed by 32-bit
 * byte offsets with the upper range reserved for samples that read zeros): a finest level up to ~470 pixels wide; wider pyramids take the
 * stage entries (the Python engine falls back by itself). */
size_t car_fused_blob_floats(void);
size_t car_fused_bias_floats(void);
int car_fused_samples(const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h, int lat_w,
                      int lat_pad, const float* gmeta, const float* wpt, const float* blob, const float* bias, int b, int V, int R, int P,
                      int H, int W, int no_sample, float* e, float* g, float* logit, float* pt, float* pixel_val, void* stream);
/* The same launch, which also leaves the FIRST attention round's value sum of every (view, ray, group of car_fused_tile_steps()
 * consecutive steps) in `part` [b*V][R][ceil(P / tile_steps)][576]: sum_j exp(logit_j - m) e_j over the group's samples, m their largest
 * logit — each workgroup reads the rows of e it has just written back from L2, an eighth of the bytes the attention launch would
 * otherwise stream from HBM.  car_attend_parts (below) folds a ray's groups together; models.py:533-541. */
int car_fused_tile_steps(void);
int car_fused_samples_parts(const float* poses, const float* rays, const float* steps, const float* lattice, int lat_h, int lat_w,
                            int lat_pad, const float* gmeta, const float* wpt, const float* blob, const float* bias, int b, int V, int R,
                            int P, int H, int W, int no_sample, float* e, float* g, float* logit, float* pt, float* pixel_val,
                            float* part, void* stream);
/* The exchange's row lists on the device: for samples [n_scenes*V][pts] with car_sample_setup's pixel_val [.,2], pt_in [.,V,3] (the point in every
 * context frame) and ptenc [.,V,4] (its tanh encoding), row (sample, k) = component k of sample: k = 0 own view (border padding, own grid point),
 * k >= 1 the other views o in ascending order (zeros padding) at the projection of CONTEXT o's sample of the same index moved into the sample's frame. */
int car_exchange_rows(const float* poses, const float* pixel_val, const float* pt_in, const float* ptenc, int n_scenes, int V, long pts, int H,
                      int W, int* row_src, float* row_grid, float* row_pe, void* stream);
/* The kernel's source pass alone, over explicit rows (the three-view exchange, models.py:345-475): row = sample * ncomp + comp gathers the
 * merged lattice of map (row_src & 0x3fffffff), padding mode (row_src >> 30) & 1 at row_grid [2], adds the point term of row_pe [4] (as
 * car_lattice_encode_rows) and runs the second point-MLP layer: e [rows][288].  Samples are [n_sets][R][P]; every 24-ray x 8-step tile of one
 * (set, comp) must share its (map, padding mode).  lattice: car_merge_lattice; gmeta [1]: its largest magnitude; blob / bias / wpt:
 * car_fused_pack_rows(W1 [576][579], b1, W2 [288][576], b2, ...) with car_fused_blob_floats() / car_fused_bias_floats() / 576 * 4 floats. */
int car_fused_pack_rows(const float* w1, const float* b1, const float* w2, const float* b2, float* blob, float* bias, float* wpt, void* stream);
int car_fused_rows(const float* lattice, int lat_h, int lat_w, int lat_pad, const float* gmeta, const float* wpt, const float* blob,
                   const float* bias, const int* row_src, const float* row_grid, const float* row_pe, int n_sets, int R, int P, int ncomp,
                   float* e, void* stream);

/* ---- 1x1 convolutions / linear layers on channel-last rows, fp32 MFMA (models.py:333-341, 487-491, 529, 548, 553;
 *      resnet_block_fc.py:53-62, 132-168).  Y[M,N] = act(X[M,K] W^T + bias).
 * Weights are re-laid out once into the MFMA operand order by car_linear_pack (bias folded in as column K). */
size_t car_linear_packed_floats(int K, int N);
int car_linear_pack(const float* W, int ldw, const float* bias, int K, int N, float* packed, void* stream);
int car_linear(const float* X, int ldx, const float* packed, int K, int N, float* Y, int ldy, long M, int flags,
               void* stream);

/* The same layer on the f16 matrix pipe (csrc/car_linear16.hip), for the stage entries' wide layers: fp16 hi / lo halves of both
 * operands, three products per term, fp32 accumulation — the fused kernel's arithmetic (fp32-class accuracy; car_linear's fp32 pipe
 * peaks at 157 TFLOP/s, this path at 2500 / 3).  The weights carry a power of two chosen by car_linear_x3_pack; every row of X carries
 * one that follows the largest magnitude seen SO FAR along the row (32 columns at a time, after act_in): when a later chunk outgrows it
 * the row's accumulators are multiplied by the ratio of the two powers — exact — so X is read once; a row whose leading chunks are all
 * zero starts from the clamp (the powers of two live in [2^-90, 2^43], so bias * 2^43 * 2^shift stays inside fp32) and is moved down
 * by the first non-zero chunk.  Y[M, N] = act(act_in(X[M, K]) W^T + bias (+ Y)), flags as car_linear; bias: N
 * floats or NULL (not folded into the pack).  N % 32 == 0, ldx % 4 == 0 (and >= K rounded up to 4), ldy % 4 == 0, X / Y / bias 16-byte
 * aligned; car_linear serves every other shape. */
size_t car_linear_x3_packed_floats(int K, int N);
int car_linear_x3_pack(const float* W, int ldw, int K, int N, float* packed, void* stream);
int car_linear_x3(const float* X, int ldx, const float* packed, const float* bias, int K, int N, float* Y, int ldy, long M, int flags,
                  void* stream);
/* car_linear_x3 with car_relu_mask (below) applied while the result is stored: Y[m][n] = act[m][n] > 0 ? (X W^T + bias (+ Y))[m][n] : 0.
 * The backward's data gradient dX = dY W of a layer whose input went through a ReLU (what torch autograd does for F.relu in front of the
 * convolutions of models.py:333-341, 487-491, 529) without a second pass over dX and the activation; bit-identical to the two calls.
 * act [M, N] with row stride lda (a multiple of 4, >= N), 16-byte aligned. */
int car_linear_x3_masked(const float* X, int ldx, const float* packed, const float* bias, int K, int N, float* Y, int ldy, long M, int flags,
                         const float* act, int lda, void* stream);

/* ---- a12 + a13 + the logits of a14 for the stage route's variants (n_view 1 / 3, no_latent_concat): key = key_map_2(relu(key_map(e))),
 * qry = query_embed_2(relu(query_embed(g))), logit = <key, qry> / 16 (models.py:487-491, 529, 533) in ONE kernel — the gather-free instance of
 * csrc/car_linear16.hip's layer (key_map over the rows of e, split fp16 x 3) whose 128 outputs stay in the accumulators and run through the
 * fused per-sample kernel's closing layers; the 128-wide k1, key and q1 rows are never written.  e [M, Ce] (row stride lde, a multiple of 4),
 * g [M, 16]; packed_k1 / bias_k1: key_map laid out by car_linear_x3_pack (K = Ce, N = 128) and its bias; tail [car_kq_tail_floats()] /
 * tail_bias [car_kq_bias_floats()]: key_map_2, query_embed, query_embed_2 packed by car_kq_pack.  Out: qry [M, 128], logit [M]. */
size_t car_kq_tail_floats(void);
size_t car_kq_bias_floats(void);
int car_kq_pack(const float* k2w, const float* k2b, const float* q1w, const float* q1b, const float* q2w, const float* q2b, float* tail,
                float* bias, void* stream);
int car_key_query_logits(const float* e, int lde, const float* packed_k1, const float* bias_k1, int Ce, const float* g, const float* tail,
                         const float* tail_bias, long M, float* qry, float* logit, void* stream);

/* ---- a14-a16: per-ray softmax attention over the V*P samples (models.py:532-594)
 * qa, qb [b*V,R,P,dq] (row stride dq): logit = <qa,qb>/16; with qb == NULL, qa is the precomputed logit [b*V,R,P]
 * (dq ignored).  val [b*V,R,P,D].
 * w_out [b*V,R,P] softmax weights (ordered [view 1's P, view 2's P] per ray).
 * z_out [b,R,ld_z]: sum_s w_s val_s (+ zprev_scale * zprev[b,R,D] if zprev != NULL), written `reps` times
 *   side by side (the per-view replication of models.py:541, 565, 605-606).
 * If pt != NULL also: depth [b,R] = clamp((inv_q . sum_s w_s clamp(pt_s,+-100)).z, 0, 10) and
 *   w_argmax [b*V,R] (int32) = argmax_p w. */
int car_attend(const float* qa, const float* qb, int dq, const float* val, int D, int b, int V, int R, int P,
               const float* zprev, float zprev_scale, float* w_out, float* z_out, int ld_z, int reps,
               const float* pt, const float* poses, float* depth, int32_t* w_argmax, void* stream);
/* The same round from precomputed logits [b*V,R,P] and per-step-group partial value sums `part` [b*V][R][ceil(P / tile_steps)][D]
 * (car_fused_samples_parts): z_out = sum_g exp(m_g - M) / L part_g, with m_g the group's largest logit, M the ray's, L the softmax
 * denominator — the same sum as car_attend's over the sample rows; w_out, depth and w_argmax exactly as car_attend computes them.
 * `tile_steps` must be car_fused_tile_steps() (the group size `part` was made with; checked). */
int car_attend_parts(const float* logit, const float* part, int tile_steps, int D, int b, int V, int R, int P, float* w_out,
                     float* z_out, int ld_z, int reps, const float* pt, const float* poses, float* depth, int32_t* w_argmax,
                     void* stream);

/* ---- a15, per-sample half in one kernel (models.py:549-555):
 *     logit[row] = < Wr2 relu(Wr1[:,128:] g[row] + br1 + uh[ray(row)]) + br2 , qry[row] > / 16
 * g [b*V,R,P,16] (local_coords), qry [b*V,R,P,128]; uh [b,R,128] = Wr1[:,:128] encode_latent(z1) (no bias).
 * wpacked: car_round2_packed_floats() floats = query_repeat_embed_2.weight (128x128) then query_repeat_embed.weight[:,128:]
 * (128x16) in MFMA operand order with fp16 hi/lo halves; bias: car_round2_bias_floats() floats = br1 [128], br2 [128], the two
 * layers' 2^-shift, 0, 0 (both written by car_plan_build).  Neither the 128-wide local query half nor q2 is ever written. */
size_t car_round2_packed_floats(void);
size_t car_round2_bias_floats(void);
int car_round2_logits(const float* g, const float* uh, const float* qry, const float* wpacked, const float* bias,
                      int b, int V, int R, int P, float* logit, void* stream);
/* The same logits WITHOUT any 128-wide rows (the one-call forward's form since round 6).  With y = relu(Wr1[:,128:] g + br1 + uh) and
 * x = relu(Wq1 g + bq1) (query_embed, models.py:126, 529) the product <Wr2 y + br2, Wq2 x + bq2> is the bilinear form
 *     y^T (M x + v) + u^T x + c,   M = Wr2^T Wq2,  v = Wr2^T bq2,  u = Wq2^T br2,  c = <br2, bq2>
 * folded once per checkpoint by car_round2q_pack (fp64 accumulation, rounded once to fp32): ONE 128 x 128 layer per sample — no more matrix
 * work than car_round2_logits spends on q2 alone — and the first round's query rows are neither written (car_fused_samples folds its own
 * pair of closing layers the same way and forms no qry) nor read: 2 x 4.3 GB per 65 536-ray frame.  Same arithmetic otherwise
 * (v_mfma_f32_32x32x16_f16, fp16 hi / lo halves, per-sample powers of two); agrees with car_round2_logits on stored rows to fp32 rounding.
 * wpacked [car_round2q_packed_floats()], bias [car_round2q_bias_floats()]: car_round2q_pack. */
size_t car_round2q_packed_floats(void);
size_t car_round2q_bias_floats(void);
int car_round2_logits_from_g(const float* g, const float* uh, const float* wpacked, const float* bias, int b, int V, int R, int P,
                             float* logit, void* stream);

/* r[row, c] = relu(r[row, c] + u[ray(row), c]) with ray(row) = scene b, ray r of the sample row (models.py:549-553:
 * the z_embed half of query_repeat_embed is constant along the samples of a ray). r [b*V,R,P,C], u [b,R,C]. */
int car_add_ray_bias_relu(float* r, const float* u, int b, int V, int R, int P, int C, void* stream);

/* ---- the per-ray layers as two kernels (csrc/car_raychain.hip): a layer's outputs stay in the MFMA accumulators of the wave that owns
 * the 32 rays and are the next layer's B operands; f16 matrix pipe with fp16 hi/lo operand halves (three products per term, fp32
 * accumulate), weights scaled by a power of two per layer, activations by one per ray and layer, both undone exactly.
 *   car_chain_pack : split-fp16 tiles of one layer, car_chain_packed_floats(K, N) floats.  chained != 0: the layer is fed by another
 *                    layer's accumulators (their K order); 0: by rows in memory.  W2 (optional, same shape and row stride) is added
 *                    element-wise.  scale: device array of 32 floats shared by the layers of a plan; the layer's 2^shift goes to
 *                    scale[slot], 2^-shift to scale[16 + slot] (slot < 16).
 *   car_ray_mid    : z1 = latent_value(ebar) [M,288];  uh = query_repeat_embed[:, :128] encode_latent(z1) [M,128]   (models.py:487, 548, 552)
 *   car_ray_tail   : z = latent_value(ebar) + V z1;  rgb = phi([z, z], coords) valid + (1 - valid), valid [b R]   (models.py:561-565, 597-617;
 *                    resnet_block_fc.py:132-168)
 * `arena` + host arrays offs / nts (n_chunks entries): where every K = 32 weight chunk lives (float offset from `arena`) and how many
 * 32-output tiles it has, in consumption order — mid: latent_value (18 x 9), encode_latent (9 x 4), query_repeat_embed[:, :128]
 * (4 x 4); tail: latent_value, lin_in (1 x 4), 3 x {lin_z halves added (9 x 4), fc_0 (4 x 4), fc_1 (4 x 4)}, lin_out (4 x 1).
 * bias — mid: latent_value.bias (288), encode_latent.bias (128); tail: latent_value.bias (288), lin_in.bias (128), 3 x {lin_z, fc_0,
 * fc_1 biases}, lin_out.bias padded to 32.  layers (host array, n_layers = 3 / 12): the scale slot of every layer in that order.
 * car_plan_build / car_render_forward set all of this up. */
size_t car_chain_packed_floats(int K, int N);
int car_chain_pack(const float* W, int ldw, const float* W2, int K, int N, int chained, float* packed, float* scale, int slot, void* stream);
int car_ray_mid(const float* arena, const unsigned* offs, const int* nts, int n_chunks, const float* bias, const float* scale,
                const int* layers, int n_layers, const float* ebar, int ld_ebar, float* z1, float* uh, long M, void* stream);
int car_ray_tail(const float* arena, const unsigned* offs, const int* nts, int n_chunks, const float* bias, const float* scale,
                 const int* layers, int n_layers, const float* ebar, int ld_ebar, const float* phi_x, int ld_phi, const float* z1,
                 const float* rays, int b, int V, int R, float* rgb, float* valid, void* stream);

/* ---- a18: valid mask and white background (models.py:614-617).
 * rays [b*V,R,CAR_RAY_FLOATS]; rgb_in [b,R,ld_in] (first 3 columns used) -> rgb [b,R,3], valid [b,R]. */
int car_finalize(const float* rays, const float* rgb_in, int ld_in, int b, int V, int R, float* rgb, float* valid,
                 void* stream);

/* =====================================================================================================================
 * Backward of the staged route.  The reference trains by torch autograd over models.py:190-626
 * (training.py:92-136: loss on `rgb` and `depth_ray`, loss_functions.py:74-132); these are the kernels a host needs to push
 * the gradient of (rgb, depth_ray) back to every renderer parameter and to the feature pyramid z.  The data gradient of a
 * linear layer, dX = dY W, is car_linear with the transposed weight packed by car_linear_pack.
 *
 *  car_linear_wgrad   dW[N, K] += dY[M, N]^T X[M, K]  and, with db != NULL, db[N] += column sums of dY (fp32 atomics into dW / db:
 *                     zero them first).  Layers over >= 2048 rows with N K >= 1024 run on the bf16 matrix pipe as three products of
 *                     bf16 hi / lo halves of both operands (fp32 accumulate; bf16 keeps fp32's exponent, so no scale is involved;
 *                     error ~2^-17 per term, rounding-like); the rest — and everything with CAR_WGRAD_FP32 — on the fp32 matrix
 *                     pipe.  flags: CAR_LIN_RELU_IN = X is used as relu(X).  Columns of dY / X beyond N / K (row padding) may hold
 *                     anything, NaN included: they only meet entries of dW that are never written.
 *  car_attend_backward   one attention round (models.py:532-541 / 555-565, depth read-out 577-590).  w [b*V,R,P] the round's
 *                     softmax weights, val [b*V,R,P,D]; dz [b,R,ld_dz]: gradient of sum_s w_s val_s; ddepth [b,R] (optional, with
 *                     pt [b*V,R,P,3] and poses): gradient of depth_ray.  dval [b*V,R,P,D] (+)= w_s dz, dlogit [b*V,R,P].
 *  car_gather_bilinear_backward   grid_sample backward w.r.t. the maps: dmaps[l] [n_maps,Hl,Wl,Cl] += tap weight x the gradient of
 *                     the gathered row (same arguments and row placement as car_gather_bilinear; fp32 atomics).
 *  car_gather_bilinear_backward_binned   the same gradient without floating-point atomics (csrc/car_scatter.hip): the taps of n_gathers
 *                     gathers (grids[j] [n_maps, pts, 2], padding modes[j], placements[j]; all reading dout) are binned by texel — a counting
 *                     sort on (row, weight) records in `workspace` (car_scatter_workspace_bytes) — and every texel of every level is then
 *                     WRITTEN once (dmaps = the gradient, not +=; no zero fill beforehand).  The sum runs over the same terms as the atomic
 *                     form's, in the order the records landed.
 *  car_relu_mask      grad[m][n] = act[m][n] > 0 ? grad[m][n] : 0
 *  car_scale_rows     out[m][:] (+)= scale * s[m / group] * x[m][:]        (logit gradients to keys / queries, valid mask)
 *  car_add            out = alpha a + beta b (b may be NULL)
 *  car_reduce_samples du[b,R,C] = sum over the V*P samples of a ray of d[b*V,R,P,C]   (backward of car_add_ray_bias_relu's broadcast) */
int car_linear_wgrad(const float* dY, int ldy, const float* X, int ldx, long M, int N, int K, int flags, float* dW, int lddw,
                     float* db, void* stream);
int car_attend_backward(const float* w, const float* val, int D, int b, int V, int R, int P, const float* dz, int ld_dz,
                        const float* ddepth, const float* pt, const float* poses, float* dval, int accumulate, float* dlogit,
                        void* stream);
int car_gather_bilinear_backward(float* const* dmaps, const int* level_c, const int* level_h, const int* level_w, int n_levels,
                                 int n_maps, const float* grid, long pts, int mode, int place, int V, const float* dout, int ld_out,
                                 int col_out, void* stream);
size_t car_scatter_workspace_bytes(const int* level_h, const int* level_w, int n_levels, int n_maps, long pts, int n_gathers);
int car_gather_bilinear_backward_binned(float* const* dmaps, const int* level_c, const int* level_h, const int* level_w, int n_levels,
                                        int n_maps, const float* const* grids, const int* modes, const int* places, int n_gathers,
                                        long pts, int V, const float* dout, int ld_out, int col_out, void* workspace,
                                        size_t workspace_bytes, void* stream);
int car_relu_mask(float* grad, int ldg, const float* act, int lda, long M, int N, void* stream);
int car_scale_rows(float* out, int ldo, const float* x, int ldx, const float* s, long group, float scale, long M, int N, int accumulate,
                   void* stream);
int car_add(float* out, int ldo, const float* a, int lda, float alpha, const float* b, int ldb, float beta, long M, int N, void* stream);
int car_reduce_samples(const float* d, int b, int V, int R, int P, int C, float* du, void* stream);

/* =====================================================================================================================
 * One-call forward for hosts without the Python engine: the launch sequence of
 * cross_attention_renderer_amd/engine.py::RenderEngine for the reference's default configuration
 * (CrossAttentionRenderer(model="midas_vit", n_view=2), models.py:43-145, forward models.py:190-626).
 *
 *   car_plan_bytes / car_plan_build      once per set of weights: re-lays every layer out for the kernels ("plan", device memory)
 *   car_project_maps                     once per stereo pair: first point-MLP layer applied per texel of the pyramid and the
 *                                        levels summed on their common lattice (§4.3 DESIGN.md)
 *   car_workspace_bytes / car_render_forward   once per batch of rays
 *
 * All pointers are device pointers unless marked host; nothing is allocated, nothing is kept.  car_plan_build synchronises
 * the stream once (it uploads a small host table); the other calls are asynchronous.  Unsupported configurations
 * (n_view != 2, other widths, no_latent_concat) return CAR_E_ARG: they run through the stage entries above. */
typedef struct car_dims {
    int b, V, R, P, H, W;          /* scenes, context views (2), rays per scene, samples per ray and view, image size            */
    int n_levels;                  /* pyramid levels (3)                                                                      */
    int level_h[CAR_MAX_LEVELS], level_w[CAR_MAX_LEVELS], level_c[CAR_MAX_LEVELS];   /* e.g. 64x64x256, 128x128x256, 256x256x64 */
    int repeat_attention;          /* second attention round (models.py:547), the reference's default: 1                          */
    int no_sample;                 /* 1: samples at uniform depths on the query ray (`steps` = the depths; geometry.py:165-187,     */
                                   /*    models.py:221-222, 265-266) instead of uniformly along the clipped epipolar segment; default 0 */
} car_dims;

/* Parameters in the reference's state_dict layout: row-major [out][in] fp32, 1x1 convolutions flattened (models.py:96-144). */
typedef struct car_weights {
    const float *query_encode_latent_w, *query_encode_latent_b;         /* (576, 579), (576)  models.py:102 */
    const float *query_encode_latent_2_w, *query_encode_latent_2_b;     /* (288, 576)         models.py:103 */
    const float *latent_value_w, *latent_value_b;                       /* (288, 576)         models.py:117 */
    const float *key_map_w, *key_map_b, *key_map_2_w, *key_map_2_b;     /* (128, 576), (128, 128)   models.py:118-119 */
    const float *query_embed_w, *query_embed_b, *query_embed_2_w, *query_embed_2_b;                 /* (128, 16), (128, 128)   :126-127 */
    const float *query_repeat_embed_w, *query_repeat_embed_b, *query_repeat_embed_2_w, *query_repeat_embed_2_b;   /* (128, 144), (128, 128)  :136-137 */
    const float *encode_latent_w, *encode_latent_b;                     /* (128, 288)         models.py:142 */
    const float *phi_lin_in_w, *phi_lin_in_b, *phi_lin_out_w, *phi_lin_out_b;                       /* (128, 18), (3, 128)   resnet_block_fc.py:88-99 */
    const float *phi_lin_z_w[3], *phi_lin_z_b[3];                       /* (128, 576)         resnet_block_fc.py:108-113 */
    const float *phi_fc_0_w[3], *phi_fc_0_b[3], *phi_fc_1_w[3], *phi_fc_1_b[3];                     /* (128, 128)            resnet_block_fc.py:29-33 */
} car_weights;

typedef struct car_inputs {
    const float* poses;            /* [b*V, CAR_POSE_FLOATS]  from car_pose_setup, or filled by the host (poses.py)        */
    const float* uv;               /* [b, R, 2] pixel coordinates (x = column, y = row)                                     */
    const float* lattice;          /* lattice of THESE b scenes, [b*V, 2, lat_h, lat_w, 576]: the start of car_project_maps' buffer (a host
                                      that renders a sub-range of scenes adds scene0 * V * 2 * lat_h * lat_w * 576 floats)      */
    const float* gmeta;            /* [CAR_MAX_LEVELS] max |lattice| over all scenes: car_gmeta_offset() floats into that buffer   */
    const float* steps;            /* optional [P]: sample positions along the epipolar segment.  NULL = the plan's linspace(0,1,P)
                                      (car_linspace).  torch.linspace itself differs in the last ulp between hosts (its vectorised
                                      kernel depends on the CPU's vector width), so a host that wants torch's exact values passes them. */
} car_inputs;

typedef struct car_outputs {       /* the tensors of the reference's output dict (models.py:597-626); any may be NULL except rgb */
    float* rgb;                    /* [b, R, 3]                                                                              */
    float* valid_mask;             /* [b, R]                                                                                 */
    float* depth_ray;              /* [b, R]                                                                                 */
    float* at_wt;                  /* [b*V, R, P]   first-round attention weights                                            */
    int32_t* at_wt_max;            /* [b*V, R]      argmax_p of at_wt                                                        */
    float* coords;                 /* [b*V, R, 9]                                                                            */
    float* pixel_val;              /* [b*V, R, P, 2]                                                                         */
} car_outputs;

/* The two split-fp16 packers car_plan_build is made of, exported for hosts that drive the stage entries themselves:
 *   car_fused_pack : blob [car_fused_blob_floats()], bias [car_fused_bias_floats()], wpt [576*4] for car_fused_samples / car_gather_encode
 *                    (only the first six layer pairs of car_weights are read);
 *   car_round2_pack: wr1 = query_repeat_embed.weight (128, 144), wr2 = query_repeat_embed_2.weight (128, 128) ->
 *                    wpacked [car_round2_packed_floats()], bias [car_round2_bias_floats()] for car_round2_logits. */
int car_fused_pack(const car_weights* weights, float* blob, float* bias, float* wpt, void* stream);
int car_round2_pack(const float* wr1, const float* br1, const float* wr2, const float* br2, float* wpacked, float* bias, void* stream);
/*   car_round2q_pack: wr1, wr2 as above, wq1 = query_embed.weight (128, 16), wq2 = query_embed_2.weight (128, 128) -> the folded layer
 *                    M = wr2^T wq2 and the two 16 -> 128 layers, wpacked [car_round2q_packed_floats()], bias [car_round2q_bias_floats()]
 *                    (the table and the packer's scratch behind it) for car_round2_logits_from_g. */
int car_round2q_pack(const float* wr1, const float* br1, const float* wr2, const float* br2, const float* wq1, const float* bq1,
                     const float* wq2, const float* bq2, float* wpacked, float* bias, void* stream);

size_t car_plan_bytes(const car_dims* dims);
int car_plan_build(const car_dims* dims, const car_weights* weights, void* plan, void* stream);
size_t car_gmaps_floats(const car_dims* dims);                 /* lattice + gmeta + the projected levels (scratch of the merge)    */
size_t car_gmeta_offset(const car_dims* dims);                 /* float offset of gmeta inside that buffer                        */
/* Lattice size for these levels.  Every level must be an integer factor r_l coarser than the widest one (same factor in both
 * directions): lat_w = 2 W_m + 2 r_max + 1, lat_pad = r_max + 1 (521 x 521, pad 5 for levels of 64 / 128 / 256: 2.5 GB per scene).
 * Other pyramids return CAR_E_ARG (stage entries). */
int car_lattice_shape(const car_dims* dims, int* lat_h, int* lat_w, int* lat_pad);
/* maps[l]: level l of the encoder's pyramid, channel-last [b*V, Hl, Wl, Cl] (`maps` is a host array of device pointers). */
int car_project_maps(const car_dims* dims, const void* plan, const float* const* maps, float* gmaps, void* stream);
size_t car_workspace_bytes(const car_dims* dims);
int car_render_forward(const car_dims* dims, const void* plan, const car_inputs* in, const car_outputs* out,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same launches in two phases, for hosts that overlap them across batches of rays on two streams (engine.py):
 * CAR_PHASE_SAMPLES = rays + the fused per-sample kernel (matrix-pipe / power bound; writes e, g, logit, pt into the workspace),
 * CAR_PHASE_RAYS = both attention rounds and the per-ray chains (HBM bound; reads them, writes the outputs).  The second phase of a
 * batch must be ordered after its first phase (an event) and use the same dims / inputs / outputs / workspace; batches with their own
 * workspaces are independent.  phases = both is car_render_forward.
 * CAR_PHASE_ROWS_FIRST_ROUND (a flag OR-ed onto both phases of a batch): the first attention round streams the rows of e
 * (car_fused_samples + car_attend, the form of rounds 1-4) instead of folding the fused kernel's per-step-group partial sums
 * (car_fused_samples_parts + car_attend_parts, the default) — kept for A/B measurements and tests; same results to fp32 rounding. */
#define CAR_PHASE_SAMPLES 1
#define CAR_PHASE_RAYS 2
#define CAR_PHASE_ROWS_FIRST_ROUND 4
int car_render_forward_phase(const car_dims* dims, const void* plan, const car_inputs* in, const car_outputs* out,
                             void* workspace, size_t workspace_bytes, int phases, void* stream);
/* Where a named intermediate lives inside the workspace after car_render_forward (tests, debugging, profiling): one of
 * "rays" "e" "g" "logit" "logit2" "pt" "at_wt2" "ebar" "z1" "uh" "part".  Returns 0 and the float offset / count. */
int car_workspace_find(const car_dims* dims, const char* name, size_t* offset_floats, size_t* n_floats);

/* ---- stage timing (the reference's only hooks are record_function labels, resnet_block_fc.py:54, 139, and one time.time() pair,
 * eval_realestate10k.py:151-164).  With profiling on, car_render_forward brackets every stage it launches with a pair of HIP
 * events on the caller's stream (no synchronisation, a few microseconds of host time per stage); car_profile_read synchronises on
 * the recorded events and returns the stages' times.  Per host thread. */
void car_profile_enable(int on);                               /* also drops the recorded stages                                 */
int car_profile_count(void);                                   /* stages recorded since enable / the last read                   */
int car_profile_read(int i, const char** name, float* ms);     /* stage i; CAR_E_ARG when out of range                            */
void car_profile_reset(void);

/* host helper: linspace(a, b, n) the way torch's scalar CPU kernel computes it (models.py:261): step = (b-a)/(n-1), first half
 * a + step*i, second half b - step*(n-1-i); `out` is a HOST array.  Equal to torch.linspace for n < 16, within 1 ulp otherwise. */
void car_linspace(float a, float b, int n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* CAR_HIP_H */
