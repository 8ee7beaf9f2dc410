"""Reference (pure torch, CPU) statement of the operand packing the device packers ``car_fused_pack`` / ``car_round2_pack``
(csrc/car_render.hip) must produce for csrc/car_fused.hip and csrc/car_round2.hip.  Test infrastructure: the product packs on the
device; tests/test_fused_pack.py checks (on the CPU) that this layout reproduces the layers when walked the way the MFMA
contracts it, and (on the GPU) that the device packers emit exactly these bytes."""
from __future__ import annotations

import math
from typing import Optional

import torch

Tensor = torch.Tensor


def pow2_scale(m: float) -> float:
    """2^k with m * 2^k in [2^13, 2^14): the window the split-fp16 operands are moved into (car_fused_mma.h pow2_scale)."""
    _, ex = math.frexp(max(m, 1e-30))          # m = f * 2^ex, f in [0.5, 1)
    return 2.0 ** (14 - ex)


def std16_k(ksteps: int) -> Tensor:
    """16x16x32 tiles, standard mapping: K step m, lane l, element e -> k = 32 m + 8 (l >> 4) + e"""
    m = torch.arange(ksteps)[:, None, None]
    lane = torch.arange(64)[None, :, None]
    e = torch.arange(8)[None, None, :]
    return 32 * m + 8 * (lane // 16) + e


def chained16_k(ksteps: int, base: int = 0) -> Tensor:
    """16x16x32 tiles chained over the accumulators of 16-row source tiles (channel 16 T + 4 (l >> 4) + r): K step m takes
    source tiles 2m and 2m+1, so  k = base + 16 (2 m + e // 4) + 4 (l >> 4) + e % 4"""
    m = torch.arange(ksteps)[:, None, None]
    lane = torch.arange(64)[None, :, None]
    e = torch.arange(8)[None, None, :]
    return base + 16 * (2 * m + e // 4) + 4 * (lane // 16) + e % 4


def pack_tiles16(W: Tensor, bias: Optional[Tensor], n_tiles: int, kmap: Tensor, p: float) -> Tensor:
    """Split-fp16 A-operand tiles of v_mfma_f32_16x16x32_f16: per (K step, tile) [hi | lo][lane (64)][8 halves] = 512 float32
    words; lane l carries output channel 16 tile + l % 16 and the eight k of ``kmap[step, l]`` (k == K: the bias, k > K: zero).
    Every value is multiplied by the power of two p first."""
    N, K = W.shape
    Wext = torch.zeros(16 * n_tiles, K + 2, dtype=torch.float32)
    Wext[:N, :K] = W * p
    if bias is not None:
        Wext[:N, K] = bias * p
    ks = kmap.shape[0]
    k = kmap.clamp(max=K + 1)                                                # (ks, 64, 8)
    lane = torch.arange(64)
    n = 16 * torch.arange(n_tiles)[:, None] + (lane % 16)[None, :]           # (tiles, 64)
    w = Wext[n[None, :, :, None].expand(ks, -1, -1, 8), k[:, None, :, :].expand(-1, n_tiles, -1, -1)]      # (ks, tiles, 64, 8)
    hi = w.half()
    lo = (w - hi.float()).half()
    both = torch.stack([hi, lo], dim=2).contiguous()                         # (ks, tiles, hl, 64, 8)
    return both.view(torch.float32).reshape(ks, n_tiles, 512)


LAYERS = ("W2", "Q1", "M", "K1")


def bilinear_fold(wa: Tensor, ba: Tensor, wb: Tensor, bb: Tensor):
    """<Wa r + ba, Wb x + bb> = r^T (M x + v) + u^T x + c: M = Wa^T Wb, v = Wa^T bb, u = Wb^T ba, c = <ba, bb>, accumulated in fp64 in
    ascending k exactly as bilinear_fold_kernel (csrc/car_render.hip) does, rounded once to fp32."""
    wa, wb, ba, bb = wa.double(), wb.double(), ba.double(), bb.double()
    D = wa.shape[0]
    M = torch.zeros(D, D, dtype=torch.float64)
    v = torch.zeros(D, dtype=torch.float64)
    u = torch.zeros(D, dtype=torch.float64)
    c = torch.zeros((), dtype=torch.float64)
    for k in range(D):
        M += wa[k][:, None] * wb[k][None, :]
        v += wa[k] * bb[k]
        u += wb[k] * ba[k]
        c += ba[k] * bb[k]
    return M.float(), v.float(), u.float(), c.float()


def pack_fused(m):
    """(blob, bias table + scratch, wpt) of csrc/car_fused.hip for the module's parameters (layout: csrc/car_fused_layout.h): key_map_2 and
    query_embed_2 enter only through their bilinear fold M, v, u, c."""
    f = lambda t: t.detach().float().cpu().reshape(t.shape[0], -1)
    v = lambda t: t.detach().float().cpu()
    C = m.query_encode_latent.weight.shape[0]
    E2 = C // 2
    w2, q1, k1 = f(m.query_encode_latent_2.weight), f(m.query_embed.weight), f(m.key_map.weight)
    M, fv, fu, fc = bilinear_fold(f(m.key_map_2.weight), v(m.key_map_2.bias), f(m.query_embed_2.weight), v(m.query_embed_2.bias))
    p = {"W2": pow2_scale(w2.abs().max().item()), "Q1": pow2_scale(max(q1.abs().max().item(), v(m.query_embed.bias).abs().max().item())),
         "M": pow2_scale(M.abs().max().item()), "K1": pow2_scale(k1.abs().max().item())}
    parts = [
        pack_tiles16(w2, None, E2 // 16, std16_k(C // 32), p["W2"]),
        pack_tiles16(q1, v(m.query_embed.bias), 8, std16_k(1), p["Q1"]),
        pack_tiles16(M, None, 8, chained16_k(4), p["M"]),
        torch.cat([pack_tiles16(k1, None, 8, chained16_k(E2 // 32, base=E2 * sv), p["K1"]) for sv in range(2)]),
    ]
    blob = torch.cat([x.reshape(-1) for x in parts])
    w1 = f(m.query_encode_latent.weight)
    wpt = torch.cat([w1[:, C:C + 3], v(m.query_encode_latent.bias)[:, None]], dim=1).contiguous()
    scales = torch.zeros(16)
    for i, n in enumerate(LAYERS):
        scales[i] = 1.0 / p[n]
        scales[8 + i] = p[n]
    a = wpt.abs()
    scales[5] = (((a[:, 0] + a[:, 1]) + a[:, 2]) + a[:, 3]).max()
    scales[6] = fc
    bias = torch.cat([v(m.query_encode_latent_2.bias), fv, v(m.key_map.bias), fu, scales, M.reshape(-1)])      # M: the packer's scratch
    return blob, bias, wpt


def pack_tiles32(W: Tensor, chunks: int, kgs: int, chained: bool, p: float) -> Tensor:
    """A-operand tiles of v_mfma_f32_32x32x16_f16 for csrc/car_round2.hip: [chunk][tile 4][K group][hi|lo][lane][8 halves];
    lane l carries output 32 t + l % 32; chained: k = 32 c + (e & 3) + 8 (2 kg + (e >> 2)) + 4 (l >> 5), else k = 16 c + 8 (l >> 5) + e."""
    c = torch.arange(chunks)[:, None, None, None, None]
    t = torch.arange(4)[None, :, None, None, None]
    kg = torch.arange(kgs)[None, None, :, None, None]
    lane = torch.arange(64)[None, None, None, :, None]
    e = torch.arange(8)[None, None, None, None, :]
    n = (32 * t + lane % 32).expand(chunks, 4, kgs, 64, 8)
    k = (32 * c + (e & 3) + 8 * (2 * kg + (e >> 2)) + 4 * (lane // 32)) if chained else (16 * c + 8 * (lane // 32) + e + 0 * kg)
    k = k.expand(chunks, 4, kgs, 64, 8)
    w = (W * p)[n, k]
    hi = w.half()
    lo = (w - hi.float()).half()
    return torch.stack([hi, lo], dim=3).contiguous().view(torch.float32).reshape(-1)      # (c, t, kg, hl, lane, 8 halves)


def pack_round2(m):
    """(packed weights, bias table) of csrc/car_round2.hip: query_repeat_embed_2 in the chained K order, then the local_coords
    half of query_repeat_embed; bias = br1 | br2 | 2^-shift of (Wr1g, Wr2) | their 2^shift."""
    wr1 = m.query_repeat_embed.weight.detach().float().cpu().reshape(128, 144)[:, 128:].contiguous()
    wr2 = m.query_repeat_embed_2.weight.detach().float().cpu().reshape(128, 128)
    p1, p2 = pow2_scale(wr1.abs().max().item()), pow2_scale(wr2.abs().max().item())
    packed = torch.cat([pack_tiles32(wr2, 4, 2, True, p2), pack_tiles32(wr1, 1, 1, False, p1)])
    bias = torch.cat([m.query_repeat_embed.bias.detach().float().cpu(), m.query_repeat_embed_2.bias.detach().float().cpu(),
                      torch.tensor([1.0 / p1, 1.0 / p2, p1, p2])])
    return packed, bias


def pack_round2q(m):
    """(packed weights, bias table + scratch) of car_round2_logits_from_g: M = Wr2^T Wq2 in the chained K order, then Wr1[:, 128:] and Wq1;
    bias = br1 | v | bq1 | u | 2^-shift of (Wr1g, M, Wq1) | c | their 2^shift | 0 | M in fp32 (scratch)."""
    f = lambda t: t.detach().float().cpu().reshape(t.shape[0], -1)
    v = lambda t: t.detach().float().cpu()
    wr1 = f(m.query_repeat_embed.weight)[:, 128:].contiguous()
    wq1 = f(m.query_embed.weight).contiguous()
    M, fv, fu, fc = bilinear_fold(f(m.query_repeat_embed_2.weight), v(m.query_repeat_embed_2.bias), f(m.query_embed_2.weight), v(m.query_embed_2.bias))
    p1, pm, pq = pow2_scale(wr1.abs().max().item()), pow2_scale(M.abs().max().item()), pow2_scale(wq1.abs().max().item())
    packed = torch.cat([pack_tiles32(M, 4, 2, True, pm), pack_tiles32(wr1, 1, 1, False, p1), pack_tiles32(wq1, 1, 1, False, pq)])
    bias = torch.cat([v(m.query_repeat_embed.bias), fv, v(m.query_embed.bias), fu, torch.tensor([1.0 / p1, 1.0 / pm, 1.0 / pq]), fc.reshape(1),
                      torch.tensor([p1, pm, pq, 0.0]), M.reshape(-1)])
    return packed, bias
