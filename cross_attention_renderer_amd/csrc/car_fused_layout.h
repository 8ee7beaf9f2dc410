// car_fused_layout.h — sizes and the packed-weight layout of the fused per-sample kernel, shared by the kernel (car_fused.hip via
// car_fused_mma.h) and by the host code that packs the weights (car_render.hip).  Plain constants only.
#pragma once

constexpr int kC = 576;            // feature channels = width of h
constexpr int kE = 288;            // per-source width of e
constexpr int kD = 128;            // hidden width of the key / query MLPs
constexpr int kKS = kC / 32;       // 18 K steps (= weight chunks) of the 576 -> 288 layer
constexpr int kTE = kE / 16;       // 18 output tiles of 16 channels
constexpr int kTD = kD / 16;       // 8 output tiles
constexpr int kTile = 512;         // packed floats per (K step, tile): [hi|lo][64 lanes][8 halves] = 2 KB

// ---- packed-weight blob: offsets in tiles, layer by layer, [K step][tile] inside a layer ---------------------------
// The first round's logit <key, qry> / 16 with key = Wk2 relu(k1) + bk2 and qry = Wq2 relu(q1) + bq2 (models.py:491, 529, 533) is a
// bilinear form of the two hidden vectors r = relu(k1), x = relu(q1):
//     <key, qry> = r^T (M x + v) + u^T x + c,    M = Wk2^T Wq2,  v = Wk2^T bq2,  u = Wq2^T bk2,  c = <bk2, bq2>
// so ONE 128 x 128 layer (M, folded once per checkpoint by car_fused_pack) replaces the two closing layers key_map_2 and query_embed_2;
// neither key nor qry is ever formed (the second round folds its own pair of layers the same way: car_round2.hip).
constexpr int kOffW2 = 0;                          // 18 steps x 18 tiles, standard K mapping
constexpr int kOffQ1 = kOffW2 + kKS * kTE;         // 1 x 8, standard, bias folded at k = 16
constexpr int kOffM = kOffQ1 + kTD;                // 4 x 8, chained over x = relu(q1): M = Wk2^T Wq2 (output index: r's channel)
constexpr int kOffK1 = kOffM + 4 * kTD;            // 18 x 8, chained over [e_0 ; e_1] (9 steps per source)
constexpr int kBlobTiles = kOffK1 + 18 * kTD;
constexpr int kNumChunks = 2 * kKS + 2 * 5 + 1 + 2;                   // 49 weight chunks per pass
constexpr int kChunkTiles = kTE;                   // largest chunk: 18 tiles = 36 KB

// bias table: b2 (288) | v (128) | bk1 (128) | u (128) | scales (16): [0..3] 2^-shift of the packed layers W2, Q1, M, K1, [5] the largest row
// sum of |(W1 point columns, b1)|, [6] c, [8..11] 2^shift (scratch of car_fused_pack).  Behind the table (not loaded by the kernel):
// kD x kD floats of pack-time scratch (M in fp32).
constexpr int kBiasE = 0, kBiasV = kE, kBiasK1 = kE + kD, kBiasU = kE + 2 * kD, kBiasScale = kE + 3 * kD, kBiasFloats = kBiasScale + 16;
constexpr int kBiasConst = kBiasScale + 6;
constexpr int kBiasScratch = kD * kD;
enum { kLayerW2 = 0, kLayerQ1, kLayerM, kLayerK1 };
