"""Operand packing consumed by csrc/car_fused.hip and csrc/car_round2.hip.

CPU: evaluating a layer exactly the way the MFMA contracts the packed tiles (K step, tile, lane group, element) must reproduce
W x for the standard, bias-folded and chained K mappings (tests/pack_reference.py is the layout's reference statement).
GPU: the device packers of the C ABI (car_fused_pack / car_round2_pack, csrc/car_render.hip) must emit exactly those bytes."""
import ctypes

import pytest
import torch

import pack_reference as PR
from cross_attention_renderer_amd import synthetic as S
from cross_attention_renderer_amd.models import CrossAttentionRenderer


def _module(seed):
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=8, with_encoder=False).eval()
    S.perturb_parameters(m, seed=seed)
    return m


def _walk16(tiles, x_of):
    """tiles (ksteps, n_tiles, 2, 64, 8) fp16 halves as floats; x_of(step, lane group q, e) -> B value of that K slot.  Evaluates the
    layer the way v_mfma_f32_16x16x32_f16 contracts it: lane l = 16 q + i holds A[i][8 q + e] and B[8 q + e][.]."""
    ks, nt = tiles.shape[:2]
    y = torch.zeros(16 * nt, dtype=torch.float64)
    w = (tiles[:, :, 0].double() + tiles[:, :, 1].double())                   # hi + lo: (ks, nt, 64, 8)
    for m in range(ks):
        for q in range(4):
            for e in range(8):
                y += w[m, :, 16 * q:16 * q + 16, e].reshape(-1) * x_of(m, q, e)
    return y


def test_fused_blob_layout():
    """Standard, bias-folded and chained K mappings of the 16x16x32 tiles, every layer with its own power of two."""
    from cross_attention_renderer_amd import _lib
    m = _module(1)
    blob, bias, wpt = PR.pack_fused(m)
    lib = _lib.load()
    assert blob.numel() == lib.car_fused_blob_floats() and bias.numel() == lib.car_fused_bias_floats()
    T = 512
    down = {n: bias[672 + i].double().item() for i, n in enumerate(PR.LAYERS)}
    g = torch.Generator().manual_seed(0)
    halves = lambda a, b_, ks, nt: blob[a * T:b_ * T].view(torch.float16).reshape(ks, nt, 2, 64, 8).float()
    chained = lambda x, base=0: (lambda m_, q, e: x[base + 16 * (2 * m_ + e // 4) + 4 * q + e % 4])
    # W2: 18 K steps x 18 tiles, standard
    x = torch.randn(576, generator=g).double()
    y = _walk16(halves(0, 324, 18, 18), lambda m_, q, e: x[32 * m_ + 8 * q + e]) * down["W2"]
    want = m.query_encode_latent_2.weight.detach().reshape(288, 576).double() @ x
    assert (y - want).abs().max() < 1e-5
    # Q1: one K step, lane group 2 element 0 carries the constant 1 of the folded bias
    gq = torch.randn(16, generator=g).double()
    gin = lambda m_, q, e: gq[8 * q + e] if q < 2 else (1.0 if (q == 2 and e == 0) else 0.0)
    y = _walk16(halves(324, 332, 1, 8), gin) * down["Q1"]
    want = m.query_embed.weight.detach().reshape(128, 16).double() @ gq + m.query_embed.bias.detach().double()
    assert (y - want).abs().max() < 1e-5
    # M = Wk2^T Wq2: chained over a 128-wide accumulator set (8 source tiles, 4 K steps); with v, u, c it reproduces <key, qry> of the two
    # closing layers for any pair of hidden vectors (models.py:491, 529, 533)
    x, r = torch.randn(128, generator=g).double(), torch.randn(128, generator=g).double()
    wk2, bk2 = m.key_map_2.weight.detach().reshape(128, 128).double(), m.key_map_2.bias.detach().double()
    wq2, bq2 = m.query_embed_2.weight.detach().reshape(128, 128).double(), m.query_embed_2.bias.detach().double()
    y = _walk16(halves(332, 364, 4, 8), chained(x)) * down["M"]
    assert (y - wk2.T @ wq2 @ x).abs().max() < 1e-5
    folded = r @ (y + bias[288:416].double()) + bias[544:672].double() @ x + bias[678].double()
    assert abs(folded - (wk2 @ r + bk2) @ (wq2 @ x + bq2)) < 1e-4
    # K1: chained over [e_0 ; e_1], 9 K steps per source
    x = torch.randn(576, generator=g).double()
    y = sum(_walk16(halves(364 + 72 * sv, 364 + 72 * (sv + 1), 9, 8), chained(x, 288 * sv)) for sv in range(2)) * down["K1"]
    assert (y - m.key_map.weight.detach().reshape(128, 576).double() @ x).abs().max() < 1e-5
    # bias table order, the point / bias bound of h
    assert torch.equal(bias[:288], m.query_encode_latent_2.bias.detach()) and torch.equal(bias[416:544], m.key_map.bias.detach())
    w1 = m.query_encode_latent.weight.detach().reshape(576, 579)
    assert bias[677] >= (w1[:, 576:].abs().sum(1) + m.query_encode_latent.bias.detach().abs()).max() * (1 - 1e-6)


@pytest.mark.parametrize("scale", [1.0, 1e-4, 3e3])
def test_layer_scale_keeps_the_fp16_halves_normal(scale):
    """Whatever the magnitude of a layer's weights, its packed high halves sit below 2^14 and hi + lo reproduces W 2^shift to
    fp32 accuracy (absolute error below 2^-22 of the largest weight)."""
    m = _module(2)
    with torch.no_grad():
        m.query_encode_latent_2.weight.mul_(scale)
    W = m.query_encode_latent_2.weight.detach().reshape(288, 576)
    p = PR.pow2_scale(W.abs().max().item())
    assert 2 ** 13 <= W.abs().max().item() * p < 2 ** 14
    tiles = PR.pack_tiles16(W, None, 18, PR.std16_k(18), p).view(torch.float16).reshape(18, 18, 2, 64, 8).float()
    assert torch.isfinite(tiles).all() and tiles[:, :, 0].abs().max() < 2 ** 14
    rec = (tiles[:, :, 0] + tiles[:, :, 1]) / p                               # (K step, tile, lane, e)
    n_idx = 16 * torch.arange(18)[:, None] + (torch.arange(64) % 16)[None, :]
    for ks in (0, 7, 17):
        k_idx = 32 * ks + 8 * (torch.arange(64) // 16)[:, None] + torch.arange(8)[None, :]
        want = W[n_idx[:, :, None].expand(18, 64, 8), k_idx[None].expand(18, 64, 8)]
        assert (rec[ks] - want).abs().max() <= 2 ** -22 * W.abs().max()


def test_round2_packing_layout():
    """csrc/car_round2.hip: Wr2 in the accumulator ("chained") K order of the layer before it, then Wr1[:, 128:]; 32x32x16 tiles:
    lane l = 32 h + i holds A[i][8 h + e] and B[8 h + e][.] of a K step (chunk, group kg)."""
    m = _module(3)
    packed, bias = PR.pack_round2(m)
    from cross_attention_renderer_amd import _lib
    lib = _lib.load()
    assert packed.numel() == lib.car_round2_packed_floats() and bias.numel() == lib.car_round2_bias_floats()
    g = torch.Generator().manual_seed(1)
    w2 = packed[:16384].view(torch.float16).reshape(4, 4, 2, 2, 64, 8).double()      # (c, t, kg, hl, lane, e)
    w2 = w2[:, :, :, 0] + w2[:, :, :, 1]
    x = torch.randn(128, generator=g).double()
    y = torch.zeros(128, dtype=torch.float64)
    for c in range(4):
        for kg in range(2):
            for h in range(2):
                for e in range(8):
                    # B element e of lane half h in K step (c, kg) = accumulator register 8 kg + e of source tile c
                    k = 32 * c + (e & 3) + 8 * (2 * kg + (e >> 2)) + 4 * h
                    y += w2[c, :, kg, 32 * h:32 * h + 32, e].reshape(-1) * x[k]
    want = m.query_repeat_embed_2.weight.detach().reshape(128, 128).double() @ x
    assert (y * bias[257].double() - want).abs().max() < 1e-5
    w1 = packed[16384:].view(torch.float16).reshape(4, 2, 64, 8).double()              # (t, hl, lane, e)
    w1 = w1[:, 0] + w1[:, 1]
    gq = torch.randn(16, generator=g).double()
    y = torch.zeros(128, dtype=torch.float64)
    for h in range(2):
        for e in range(8):
            y += w1[:, 32 * h:32 * h + 32, e].reshape(-1) * gq[8 * h + e]
    want = m.query_repeat_embed.weight.detach().reshape(128, 144)[:, 128:].double() @ gq
    assert (y * bias[256].double() - want).abs().max() < 1e-5


def test_round2q_fold_reproduces_the_two_layers():
    """car_round2_logits_from_g's operands: M = Wr2^T Wq2 walked the way the 32x32x16 MFMA contracts the chained tiles, with v, u, c, gives
    <query_repeat_embed_2(y), query_embed_2(x)> for any pair of hidden vectors; the two 16 -> 128 layers sit behind it."""
    m = _module(5)
    packed, bias = PR.pack_round2q(m)
    from cross_attention_renderer_amd import _lib
    lib = _lib.load()
    assert packed.numel() == lib.car_round2q_packed_floats() and bias.numel() == lib.car_round2q_bias_floats()
    g = torch.Generator().manual_seed(2)
    wm = packed[:16384].view(torch.float16).reshape(4, 4, 2, 2, 64, 8).double()      # (c, t, kg, hl, lane, e)
    wm = wm[:, :, :, 0] + wm[:, :, :, 1]
    x, y = torch.randn(128, generator=g).double().abs(), torch.randn(128, generator=g).double().abs()
    t = torch.zeros(128, dtype=torch.float64)
    for c in range(4):
        for kg in range(2):
            for h in range(2):
                for e in range(8):
                    k = 32 * c + (e & 3) + 8 * (2 * kg + (e >> 2)) + 4 * h
                    t += wm[c, :, kg, 32 * h:32 * h + 32, e].reshape(-1) * x[k]
    t = t * bias[513].double() + bias[128:256].double()                                  # M x + v
    folded = y @ t + bias[384:512].double() @ x + bias[515].double()
    wr2, br2 = m.query_repeat_embed_2.weight.detach().reshape(128, 128).double(), m.query_repeat_embed_2.bias.detach().double()
    wq2, bq2 = m.query_embed_2.weight.detach().reshape(128, 128).double(), m.query_embed_2.bias.detach().double()
    assert abs(folded - (wr2 @ y + br2) @ (wq2 @ x + bq2)) < 1e-4
    gq = torch.randn(16, generator=g).double()
    for off, slot, W in ((16384, 512, m.query_repeat_embed.weight.detach().reshape(128, 144)[:, 128:]), (16384 + 2048, 514, m.query_embed.weight.detach().reshape(128, 16))):
        w1 = packed[off:off + 2048].view(torch.float16).reshape(4, 2, 64, 8).double()  # (t, hl, lane, e)
        w1 = w1[:, 0] + w1[:, 1]
        yy = torch.zeros(128, dtype=torch.float64)
        for h in range(2):
            for e in range(8):
                yy += w1[:, 32 * h:32 * h + 32, e].reshape(-1) * gq[8 * h + e]
        assert (yy * bias[slot].double() - W.double() @ gq).abs().max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1.0, 2e-4, 5e3])
def test_device_packers_emit_the_reference_bytes(scale):
    from cross_attention_renderer_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    m = _module(4)
    with torch.no_grad():
        for p_ in (m.query_encode_latent_2.weight, m.key_map.weight, m.query_embed.bias, m.query_repeat_embed_2.weight):
            p_.mul_(scale)
    blob, bias, wpt = PR.pack_fused(m)
    r2w, r2b = PR.pack_round2(m)
    keep = []

    def d(t):
        t = t.detach().float().reshape(t.shape[0], -1).contiguous().to(dev) if t.dim() > 1 else t.detach().float().contiguous().to(dev)
        keep.append(t)
        return t.data_ptr()
    w = _lib.CarWeights()
    for n in _lib.WEIGHT_FIELDS[0]:
        mod = m
        for part in n.split("."):
            mod = getattr(mod, part)
        setattr(w, f"{n.replace('.', '_')}_w", d(mod.weight))
        setattr(w, f"{n.replace('.', '_')}_b", d(mod.bias))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dblob, dbias, dwpt = torch.empty_like(blob, device=dev), torch.empty_like(bias, device=dev), torch.empty(576 * 4, device=dev)
    rc = lib.car_fused_pack(ctypes.byref(w), dblob.data_ptr(), dbias.data_ptr(), dwpt.data_ptr(), st)
    assert rc == 0, lib.car_last_error()
    dr2w, dr2b = torch.empty_like(r2w, device=dev), torch.empty_like(r2b, device=dev)
    rc = lib.car_round2_pack(w.query_repeat_embed_w, w.query_repeat_embed_b, w.query_repeat_embed_2_w, w.query_repeat_embed_2_b,
                             dr2w.data_ptr(), dr2b.data_ptr(), st)
    assert rc == 0, lib.car_last_error()
    torch.cuda.synchronize()
    assert torch.equal(dblob.cpu().view(torch.int32), blob.view(torch.int32))
    assert torch.equal(dbias.cpu(), bias)
    assert torch.equal(dwpt.cpu().view(576, 4), wpt)
    assert torch.equal(dr2w.cpu().view(torch.int32), r2w.view(torch.int32))
    assert torch.equal(dr2b.cpu(), r2b)
    # the folded layer and the two 16 -> 128 layers car_round2_logits_from_g reads
    r2qw, r2qb = PR.pack_round2q(m)
    assert r2qw.numel() == lib.car_round2q_packed_floats() and r2qb.numel() == lib.car_round2q_bias_floats()
    dqw, dqb = torch.empty_like(r2qw, device=dev), torch.empty_like(r2qb, device=dev)
    rc = lib.car_round2q_pack(w.query_repeat_embed_w, w.query_repeat_embed_b, w.query_repeat_embed_2_w, w.query_repeat_embed_2_b,
                              w.query_embed_w, w.query_embed_b, w.query_embed_2_w, w.query_embed_2_b, dqw.data_ptr(), dqb.data_ptr(), st)
    assert rc == 0, lib.car_last_error()
    torch.cuda.synchronize()
    assert torch.equal(dqw.cpu().view(torch.int32), r2qw.view(torch.int32))
    assert torch.equal(dqb.cpu(), r2qb)
