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
constexpr int kOffW2 = 0;                          // 18 steps x 18 tiles, standard K mapping
constexpr int kOffQ1 = kOffW2 + kKS * kTE;         // 1 x 8, standard, bias folded at k = 16
constexpr int kOffQ2 = kOffQ1 + kTD;               // 4 x 8, chained
constexpr int kOffK1 = kOffQ2 + 4 * kTD;           // 18 x 8, chained over [e_0 ; e_1] (9 steps per source)
constexpr int kOffK2 = kOffK1 + 18 * kTD;          // 4 x 8, chained
constexpr int kBlobTiles = kOffK2 + 4 * kTD;
constexpr int kNumChunks = 2 * kKS + 2 * 5 + 2 + 1 + 2;               // 51 weight chunks per pass
constexpr int kChunkTiles = kTE;                   // largest chunk: 18 tiles = 36 KB

// bias table: b2 (288) | bq2 (128) | bk1 (128) | bk2 (128) | scales (16): [0..4] 2^-shift of the packed layers W2, Q1, Q2, K1, K2,
// [5] the largest row sum of |(W1 point columns, b1)|, [8..12] 2^shift (scratch of car_fused_pack)
constexpr int kBiasE = 0, kBiasQ2 = kE, kBiasK1 = kE + kD, kBiasK2 = kE + 2 * kD, kBiasScale = kE + 3 * kD, kBiasFloats = kBiasScale + 16;
enum { kLayerW2 = 0, kLayerQ1, kLayerQ2, kLayerK1, kLayerK2 };

