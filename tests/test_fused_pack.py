"""CPU check of the operand packing consumed by csrc/car_fused.hip: evaluating a layer exactly the way the kernel walks
the packed tiles (chunk, tile, MFMA step, lane half) must reproduce W x for the standard and the chained K mappings."""
import torch

from cross_attention_renderer_amd import synthetic as S
from cross_attention_renderer_amd.engine import pack_fused_weights
from cross_attention_renderer_amd.models import CrossAttentionRenderer


def _walk(tiles, x_of):
    """tiles (chunks, n_tiles, 4, 64, 4); x_of(chunk, r, h) -> scalar B operand of lane half h at MFMA step r."""
    chunks, nt = tiles.shape[:2]
    y = torch.zeros(32 * nt, dtype=torch.float64)
    for c in range(chunks):
        for j4 in range(4):
            for e in range(4):
                r = 4 * j4 + e
                for h in range(2):
                    a = tiles[c, :, j4, 32 * h:32 * h + 32, e].double().reshape(-1)      # outputs 0..32*nt-1
                    y += a * x_of(c, r, h)
    return y


def test_fused_blob_layout():
    from cross_attention_renderer_amd import _lib
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=8).eval()
    S.perturb_parameters(m, seed=1)
    blob, bias = pack_fused_weights(m, "cpu")
    lib = _lib.load()
    assert blob.numel() == lib.car_fused_blob_floats() and bias.numel() == lib.car_fused_bias_floats()
    g = torch.Generator().manual_seed(0)
    T = 1024
    perm = lambda r, h: (r & 3) + 8 * (r >> 2) + 4 * h
    # W2: standard mapping, 18 chunks x 9 tiles
    x = torch.randn(576, generator=g).double()
    y = _walk(blob[:162 * T].reshape(18, 9, 4, 64, 4), lambda c, r, h: x[32 * c + 16 * h + r])
    want = m.query_encode_latent_2.weight.detach().reshape(288, 576).double() @ x
    assert (y - want).abs().max() < 1e-5
    # Q1: standard, one chunk, bias folded at k = 16 (upper lane half, step 0 carries the constant 1)
    gq = torch.randn(16, generator=g).double()
    y = _walk(blob[162 * T:166 * T].reshape(1, 4, 4, 64, 4), lambda c, r, h: gq[r] if h == 0 else (1.0 if r == 0 else 0.0))
    want = m.query_embed.weight.detach().reshape(128, 16).double() @ gq + m.query_embed.bias.detach().double()
    assert (y - want).abs().max() < 1e-5
    # Q2 / K2: chained over a 128-wide accumulator
    for off, layer in ((166, m.query_embed_2), (258, m.key_map_2)):
        x = torch.randn(128, generator=g).double()
        y = _walk(blob[off * T:(off + 16) * T].reshape(4, 4, 4, 64, 4), lambda c, r, h: x[32 * c + perm(r, h)])
        assert (y - layer.weight.detach().reshape(128, 128).double() @ x).abs().max() < 1e-5
    # UG: standard with folded bias, the local_coords half of query_repeat_embed
    y = _walk(blob[182 * T:186 * T].reshape(1, 4, 4, 64, 4), lambda c, r, h: gq[r] if h == 0 else (1.0 if r == 0 else 0.0))
    wr = m.query_repeat_embed.weight.detach().reshape(128, 144).double()
    assert (y - (wr[:, 128:] @ gq + m.query_repeat_embed.bias.detach().double())).abs().max() < 1e-5
    # K1: chained over [e_0 ; e_1], 18 chunks x 4 tiles
    x = torch.randn(576, generator=g).double()
    y = _walk(blob[186 * T:258 * T].reshape(18, 4, 4, 64, 4), lambda c, r, h: x[288 * (c // 9) + 32 * (c % 9) + perm(r, h)])
    assert (y - m.key_map.weight.detach().reshape(128, 576).double() @ x).abs().max() < 1e-5
    # bias table order
    assert torch.equal(bias[:288], m.query_encode_latent_2.bias.detach()) and torch.equal(bias[416:544], m.key_map.bias.detach())


def test_split_fp16_packing_reconstructs_the_weights():
    """hi + lo of the packed fp16 halves reproduces W * 2^8 to ~2^-21 relative, in the [K group][hi|lo][lane][8] order."""
    from cross_attention_renderer_amd.engine import W_SHIFT, _pack_tiles_f16_split, _std_k
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=8).eval()
    S.perturb_parameters(m, seed=2)
    W = m.query_encode_latent_2.weight.detach().reshape(288, 576)
    tiles = _pack_tiles_f16_split(W, 9, _std_k(18)).view(torch.float16).reshape(18, 9, 2, 2, 64, 8).float()
    rec = (tiles[:, :, :, 0] + tiles[:, :, :, 1]) / float(1 << W_SHIFT)        # (chunk, tile, kg, lane, e)
    for c, t, kg, lane, e in [(0, 0, 0, 0, 0), (3, 2, 1, 37, 5), (17, 8, 1, 63, 7), (9, 4, 0, 31, 3)]:
        n, k = 32 * t + lane % 32, 32 * c + 16 * (lane // 32) + 8 * kg + e
        assert abs(rec[c, t, kg, lane, e].item() - W[n, k].item()) <= 2e-6 * abs(W[n, k].item()) + 1e-9
    # thanks to the 2^8 scale most low halves are normal fp16 numbers; the subnormal ones belong to residuals that are tiny
    # anyway (absolute error <= 2^-25 in scaled units)
    lo = tiles[:, :, :, 1].abs()
    assert (lo[lo > 0] >= 6.1e-5).float().mean() > 0.9
    n_idx = 32 * torch.arange(9)[:, None] + (torch.arange(64) % 32)[None, :]
    for c in (0, 7, 17):
        for kg in (0, 1):
            k_idx = 32 * c + 16 * (torch.arange(64) // 32)[:, None] + 8 * kg + torch.arange(8)[None, :]      # (lane, e)
            want = W[n_idx[:, :, None].expand(9, 64, 8), k_idx[None].expand(9, 64, 8)]
            assert (rec[c, :, kg] - want).abs().max() <= 4e-7 * W.abs().max()


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


def test_fused2_blob_layout():
    """Operand packing of csrc/car_fused2.hip: standard, bias-folded and chained K mappings of the 16x16x32 tiles."""
    from cross_attention_renderer_amd import _lib
    from cross_attention_renderer_amd.engine import W_SHIFT, pack_fused2_weights
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=8).eval()
    S.perturb_parameters(m, seed=1)
    blob, bias = pack_fused2_weights(m, "cpu")
    lib = _lib.load()
    assert blob.numel() == lib.car_fused2_blob_floats() and bias.numel() == lib.car_fused_bias_floats()
    T = 512
    sc = float(1 << W_SHIFT)
    g = torch.Generator().manual_seed(0)
    halves = lambda a, b_, ks, nt: blob[a * T:b_ * T].view(torch.float16).reshape(ks, nt, 2, 64, 8).float()
    chained = lambda x, base=0: (lambda m_, q, e: x[base + 16 * (2 * m_ + e // 4) + 4 * q + e % 4])
    # W2: 18 K steps x 18 tiles, standard
    x = torch.randn(576, generator=g).double()
    y = _walk16(halves(0, 324, 18, 18), lambda m_, q, e: x[32 * m_ + 8 * q + e]) / sc
    want = m.query_encode_latent_2.weight.detach().reshape(288, 576).double() @ x
    assert (y - want).abs().max() < 1e-5
    # Q1 / UG: one K step, lane group 2 element 0 carries the constant 1 of the folded bias
    gq = torch.randn(16, generator=g).double()
    gin = lambda m_, q, e: gq[8 * q + e] if q < 2 else (1.0 if (q == 2 and e == 0) else 0.0)
    y = _walk16(halves(324, 332, 1, 8), gin) / sc
    want = m.query_embed.weight.detach().reshape(128, 16).double() @ gq + m.query_embed.bias.detach().double()
    assert (y - want).abs().max() < 1e-5
    wr = m.query_repeat_embed.weight.detach().reshape(128, 144).double()
    y = _walk16(halves(364, 372, 1, 8), gin) / sc
    assert (y - (wr[:, 128:] @ gq + m.query_repeat_embed.bias.detach().double())).abs().max() < 1e-5
    # Q2 / K2: chained over a 128-wide accumulator set (8 source tiles, 4 K steps)
    for off, layer in ((332, m.query_embed_2), (516, m.key_map_2)):
        x = torch.randn(128, generator=g).double()
        y = _walk16(halves(off, off + 32, 4, 8), chained(x)) / sc
        assert (y - layer.weight.detach().reshape(128, 128).double() @ x).abs().max() < 1e-5
    # K1: chained over [e_0 ; e_1], 9 K steps per source
    x = torch.randn(576, generator=g).double()
    y = sum(_walk16(halves(372 + 72 * sv, 372 + 72 * (sv + 1), 9, 8), chained(x, 288 * sv)) for sv in range(2)) / sc
    assert (y - m.key_map.weight.detach().reshape(128, 576).double() @ x).abs().max() < 1e-5
