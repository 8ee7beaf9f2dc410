"""Independent checks of the timm-defined layers of the image encoder.

The reference takes ``vit_base_resnet50_384`` — ResNetV2 stem / stages, the ViT blocks, the hybrid patch embedding — from ``timm==0.5.4``,
which is neither under the reference tree nor installable here, so ``encoder.py`` restates those layers and the ``getz_*`` fixtures were
produced by the reference's code running on a second restatement (``tests/golden/timm_stub.py``): circular for exactly these layers.
What CAN be pinned without timm is pinned here, each against something that is not a copy of the restatement:
  * the state_dict key / shape table against a manifest written out from the architecture's published numbers and the reference's constructor
    arguments (``tests/golden/dpt_hybrid_manifest.py``), and its size against the published 123 M parameters of DPT-Hybrid;
  * ``StdConv2dSame``: the statistics of the weights it convolves with, and its output against a float64 convolution over an explicitly
    padded image, the TF "SAME" padding arithmetic worked out by hand at the sizes the encoder sees (256, 384 and the odd sizes in between);
  * ``GroupNormAct`` against a float64 restatement of the group-norm formula; ``MaxPool2dSame`` against a window maximum on the padded image;
  * the transformer ``Block`` against the attention / GELU / LayerNorm formulas written out in float64.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

import dpt_hybrid_manifest as M                                    # noqa: E402
from cross_attention_renderer_amd import encoder as E              # noqa: E402


def test_state_dict_matches_the_independent_manifest():
    want = M.manifest("")
    got = {k: tuple(v.shape) for k, v in E.MultiViewDPTEncoder().state_dict().items()}
    assert sorted(got) == sorted(want), (sorted(set(want) - set(got))[:5], sorted(set(got) - set(want))[:5])
    assert got == want, [k for k in want if got[k] != want[k]][:5]
    # DPT-Hybrid is published as a 123 M-parameter model; the reference adds pos_embed_second, pose_embed (vit_models.py) and keeps
    # timm's 1000-class head in the state_dict
    extra = sum(math.prod(s) for k, s in want.items() if "pos_embed_second" in k or "pose_embed" in k or ".head." in k)
    assert abs((M.n_params(want) - extra) / 1e6 - 123.0) < 1.5


def test_manifest_matches_the_reference_side_fixture():
    """The names / shapes the REFERENCE's constructor produced (on the timm stub) are stored in the getz fixtures: the manifest must
    describe the same table — three descriptions (manifest, reference-on-stub, encoder.py) of one checkpoint layout."""
    import encoder_cases as EC
    fx = np.load(EC.fixture_path("default"))
    ref = {n: s for n, s in zip(fx["names"], fx["shapes"]) if str(n).startswith("encoder.")}
    want = M.manifest("encoder.")
    assert sorted(ref) == sorted(want)
    assert all(str(want[k]) == ref[k] for k in want)


# TF "SAME": out = ceil(in / stride); total padding = max((out - 1) * stride + k - in, 0); the smaller half goes in front.
# Worked by hand for the windows the trunk uses (7x7 / 2 stem, 3x3 / 2 pool and stage convolutions, 1x1 / 2 shortcut):
SAME_CASES = [  # (in, k, s, pad_before, pad_after, out)
    (256, 7, 2, 2, 3, 128), (384, 7, 2, 2, 3, 192), (128, 3, 2, 0, 1, 64), (192, 3, 2, 0, 1, 96), (64, 3, 2, 0, 1, 32), (96, 3, 2, 0, 1, 48),
    (32, 3, 2, 0, 1, 16), (64, 1, 2, 0, 0, 32), (255, 7, 2, 3, 3, 128), (127, 3, 2, 1, 1, 64), (33, 3, 2, 1, 1, 17), (5, 3, 1, 1, 1, 5),
]


@pytest.mark.parametrize("n,k,s,before,after,out", SAME_CASES)
def test_same_padding_arithmetic(n, k, s, before, after, out):
    x = torch.zeros(1, 1, n, n)
    x[0, 0, 0, 0] = 1.0
    x[0, 0, n - 1, n - 1] = 2.0
    p = E._same_pad(x, k, s)
    assert p.shape[-1] == n + before + after == p.shape[-2]
    assert p[0, 0, before, before] == 1.0 and p[0, 0, before + n - 1, before + n - 1] == 2.0      # where the image sits inside the padding
    assert (p.shape[-1] - k) // s + 1 == out == math.ceil(n / s)


@pytest.mark.parametrize("cin,cout,k,s,eps,n", [(3, 64, 7, 2, 1e-6, 256), (3, 64, 7, 2, 1e-6, 384), (16, 8, 3, 2, 1e-8, 33), (16, 8, 3, 1, 1e-8, 20),
                                                (16, 32, 1, 2, 1e-8, 24), (16, 32, 1, 1, 1e-8, 9)])
def test_std_conv_same_against_an_explicit_convolution(cin, cout, k, s, eps, n):
    g = torch.Generator().manual_seed(cin * 100 + k * 10 + s)
    conv = E.StdConv2dSame(cin, cout, k, stride=s, eps=eps).double()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g, dtype=torch.float64) * 0.3 + 0.1)
    x = torch.randn(2, cin, n, n, generator=g, dtype=torch.float64)
    got = conv(x)
    # the weights it must have convolved with: zero mean and unit BIASED variance per output filter (up to eps), numpy float64
    w = conv.weight.detach().numpy().reshape(cout, -1)
    ws = (w - w.mean(axis=1, keepdims=True)) / np.sqrt(w.var(axis=1, keepdims=True) + eps)
    assert np.abs(ws.mean(axis=1)).max() < 1e-12 and np.abs(ws.var(axis=1) - 1.0).max() < 10 * eps / w.var(axis=1).min() + 1e-12
    # explicit TF-SAME padding, then a "valid" convolution
    out = math.ceil(n / s)
    tot = max((out - 1) * s + k - n, 0)
    xp = torch.zeros(2, cin, n + tot, n + tot, dtype=torch.float64)
    xp[:, :, tot // 2: tot // 2 + n, tot // 2: tot // 2 + n] = x
    want = torch.nn.functional.conv2d(xp, torch.from_numpy(ws.reshape(conv.weight.shape)), None, stride=s)
    assert got.shape == want.shape == (2, cout, out, out)
    assert (got - want).abs().max() < 1e-10


def test_group_norm_act_against_the_formula():
    g = torch.Generator().manual_seed(4)
    gn = E.GroupNormAct(64).double()
    with torch.no_grad():
        gn.weight.copy_(torch.randn(64, generator=g, dtype=torch.float64))
        gn.bias.copy_(torch.randn(64, generator=g, dtype=torch.float64))
    x = torch.randn(3, 64, 7, 5, generator=g, dtype=torch.float64) * 3 + 1
    xs = x.numpy().reshape(3, 32, 2 * 7 * 5)                                           # 32 groups of 2 channels
    norm = ((xs - xs.mean(axis=2, keepdims=True)) / np.sqrt(xs.var(axis=2, keepdims=True) + 1e-5)).reshape(3, 64, 7, 5)
    want = norm * gn.weight.detach().numpy()[None, :, None, None] + gn.bias.detach().numpy()[None, :, None, None]
    assert np.abs(gn(x).detach().numpy() - np.maximum(want, 0.0)).max() < 1e-12
    lin = E.GroupNormAct(64, apply_act=False).double()
    lin.load_state_dict(gn.state_dict())
    assert np.abs(lin(x).detach().numpy() - want).max() < 1e-12                                # the third norm of a block and the shortcut's: no ReLU


@pytest.mark.parametrize("n", [128, 192, 33])
def test_max_pool_same(n):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(1, 2, n, n, generator=g) - 5.0                                      # all negative: zero padding would win, -inf must not
    got = E.MaxPool2dSame()(x)
    out = math.ceil(n / 2)
    tot = max((out - 1) * 2 + 3 - n, 0)
    xp = torch.full((1, 2, n + tot, n + tot), -float("inf"))
    xp[:, :, tot // 2: tot // 2 + n, tot // 2: tot // 2 + n] = x
    want = xp.unfold(2, 3, 2).unfold(3, 3, 2).amax(dim=(-1, -2))
    assert got.shape == (1, 2, out, out) and torch.equal(got, want)


def test_bottleneck_is_the_non_preactivation_block():
    """conv -> GroupNorm -> ReLU twice, conv -> GroupNorm, add the (projected) input, ReLU — checked by composing the block's own layers by
    hand, with and without the projecting shortcut, at stride 2 and 1."""
    g = torch.Generator().manual_seed(9)
    for cin, cout, stride, project in ((64, 128, 2, True), (128, 128, 1, False)):
        blk = E.Bottleneck(cin, cout, stride, project).double()
        with torch.no_grad():
            for p in blk.parameters():
                p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) * 0.2)
        x = torch.randn(1, cin, 12, 12, generator=g, dtype=torch.float64)
        h = torch.relu(torch.nn.functional.group_norm(blk.conv1(x), 32, blk.norm1.weight, blk.norm1.bias, 1e-5))
        h = torch.relu(torch.nn.functional.group_norm(blk.conv2(h), 32, blk.norm2.weight, blk.norm2.bias, 1e-5))
        h = torch.nn.functional.group_norm(blk.conv3(h), 32, blk.norm3.weight, blk.norm3.bias, 1e-5)
        sc = x if not project else torch.nn.functional.group_norm(blk.downsample.conv(x), 32, blk.downsample.norm.weight, blk.downsample.norm.bias, 1e-5)
        assert (blk(x) - torch.relu(h + sc)).abs().max() < 1e-12
        assert blk(x).shape == (1, cout, 12 // stride, 12 // stride)


def test_transformer_block_against_the_formulas():
    """Pre-norm block: x + proj(softmax(q k^T / sqrt(d_head)) v) over 12 heads, then x + fc2(gelu_erf(fc1(LN(x)))); LayerNorm eps 1e-6."""
    g = torch.Generator().manual_seed(11)
    D, heads, N = 48, 12, 7
    blk = E.Block(D, heads).double()
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) * 0.3)
    x = torch.randn(2, N, D, generator=g, dtype=torch.float64)

    def ln(t, w, b):
        a = t.numpy()
        return (a - a.mean(-1, keepdims=True)) / np.sqrt(a.var(-1, keepdims=True) + 1e-6) * w.detach().numpy() + b.detach().numpy()

    def lin(a, layer):
        return a @ layer.weight.detach().numpy().T + layer.bias.detach().numpy()

    a = ln(x, blk.norm1.weight, blk.norm1.bias)
    qkv = lin(a, blk.attn.qkv).reshape(2, N, 3, heads, D // heads)
    q, k, v = [qkv[:, :, i].transpose(0, 2, 1, 3) for i in range(3)]                    # (B, heads, N, d)
    s = q @ k.transpose(0, 1, 3, 2) / math.sqrt(D // heads)
    s = np.exp(s - s.max(-1, keepdims=True))
    att = (s / s.sum(-1, keepdims=True)) @ v
    y = x.numpy() + lin(att.transpose(0, 2, 1, 3).reshape(2, N, D), blk.attn.proj)
    h = lin(ln(torch.from_numpy(y), blk.norm2.weight, blk.norm2.bias), blk.mlp.fc1)
    erf = np.vectorize(math.erf)
    h = 0.5 * h * (1.0 + erf(h / math.sqrt(2.0)))                                       # exact GELU
    want = y + lin(h, blk.mlp.fc2)
    assert np.abs(blk(x).detach().numpy() - want).max() < 1e-10
