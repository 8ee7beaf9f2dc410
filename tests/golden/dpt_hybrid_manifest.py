"""Name -> shape manifest of the reference's image encoder (``CrossAttentionRenderer.encoder`` for ``model="midas_vit"``), written out
from the architecture's published numbers and the reference's constructor arguments — NOT by instantiating ``encoder.py`` or the timm
restatement of ``timm_stub.py``, so that a key or a shape that both of those got wrong the same way still shows up.

Sources, each a constructor call with literal arguments:
  * reference models.py:82-94: ``DPTDepthModel(path=None, backbone="vitb_rn50_384", non_negative=True)`` with the stem convolution replaced by
    ``StdConv2dSame(3, 64, kernel_size=7, stride=2, bias=False)``;
  * reference midas/dpt_depth.py:26-66, 94-117: ``features=256``, ``use_bn=False``, hooks (0, 1, 8, 11), ``readout="project"``, the
    four ``_make_fusion_block`` RefineNets and the ``output_conv`` depth head (256 -> 128 -> 32 -> 1);
  * reference midas/vit.py:392-541 (``_make_vit_b_rn50_backbone``: ``features=[256, 512, 768, 768]``, ``size=[384, 384]``,
    ``vit_features=768``, ``use_vit_only=False``: taps 1 and 2 are the ResNet stages, act_postprocess1/2 hold no parameters;
    act_postprocess3 = project readout + 1x1 conv to 768; act_postprocess4 = project readout + 1x1 conv + 3x3 stride-2 conv) and
    midas/blocks.py:11-75 (``_make_scratch([256, 512, 768, 768], 256, groups=1, expand=False)``: four bias-free 3x3 convolutions);
  * reference vit_models.py:10-97 (``VisionTransformerMultiView``: ``pos_embed_second`` and ``pose_embed = Linear(16, embed_dim)`` beside
    the ViT's own parameters);
  * timm 0.5.4 ``vit_base_resnet50_384`` (vision_transformer_hybrid.py): ViT-B/16 — embed_dim 768, depth 12, 12 heads, mlp_ratio 4,
    qkv_bias=True, num_classes 1000, img_size 384 -> a 24 x 24 token grid + class token = 577 position embeddings — on
    ``_resnetv2(layers=(3, 4, 9))``: ResNetV2(preact=False, stem_type="same", conv_layer=StdConv2dSame, num_classes=0, global_pool=""):
    stem 64 channels, stages of 256 / 512 / 1024 channels with bottleneck width out / 4, a projecting ``downsample`` (1x1 conv +
    GroupNorm) in the first block of every stage, GroupNorm(32) with affine parameters after every convolution, no biases in the
    convolutions, no final norm (preact=False) and no classifier parameters (num_classes=0); ``HybridEmbed.proj`` = Conv2d(1024, 768, 1).
The published size of DPT-Hybrid, 123 M parameters (Ranftl et al. 2021, table 1 of the DPT paper), is the anchor test_encoder_layers.py
checks the sum against.
"""
from __future__ import annotations

from typing import Dict, Tuple


def manifest(prefix: str = "encoder.") -> Dict[str, Tuple[int, ...]]:
    m: Dict[str, Tuple[int, ...]] = {}

    def put(name, *shape):
        m[prefix + name] = tuple(shape)

    def affine(name, n):
        put(name + ".weight", n)
        put(name + ".bias", n)

    def conv(name, cout, cin, k, bias=True):
        put(name + ".weight", cout, cin, k, k)
        if bias:
            put(name + ".bias", cout)

    def linear(name, nout, nin):
        put(name + ".weight", nout, nin)
        put(name + ".bias", nout)

    vit = "pretrained.model."
    D, depth, grid = 768, 12, 384 // 16
    put(vit + "cls_token", 1, 1, D)
    put(vit + "pos_embed", 1, grid * grid + 1, D)
    put(vit + "pos_embed_second", 1, grid * grid + 1, D)
    linear(vit + "pose_embed", D, 16)
    bb = vit + "patch_embed.backbone."
    conv(bb + "stem.conv", 64, 3, 7, bias=False)
    affine(bb + "stem.norm", 64)
    cin = 64
    for s, (cout, n_blocks) in enumerate(((256, 3), (512, 4), (1024, 9))):
        mid = cout // 4
        for b in range(n_blocks):
            blk = f"{bb}stages.{s}.blocks.{b}."
            if b == 0:
                conv(blk + "downsample.conv", cout, cin, 1, bias=False)
                affine(blk + "downsample.norm", cout)
            conv(blk + "conv1", mid, cin if b == 0 else cout, 1, bias=False)
            affine(blk + "norm1", mid)
            conv(blk + "conv2", mid, mid, 3, bias=False)
            affine(blk + "norm2", mid)
            conv(blk + "conv3", cout, mid, 1, bias=False)
            affine(blk + "norm3", cout)
        cin = cout
    conv(vit + "patch_embed.proj", D, 1024, 1)
    for i in range(depth):
        blk = f"{vit}blocks.{i}."
        affine(blk + "norm1", D)
        linear(blk + "attn.qkv", 3 * D, D)
        linear(blk + "attn.proj", D, D)
        affine(blk + "norm2", D)
        linear(blk + "mlp.fc1", 4 * D, D)
        linear(blk + "mlp.fc2", D, 4 * D)
    affine(vit + "norm", D)
    linear(vit + "head", 1000, D)
    # re-assembly of taps 3 and 4 (Sequential indices as in the reference: 0 readout, 1 transpose, 2 unflatten, 3 conv, 4 conv)
    for tap in (3, 4):
        linear(f"pretrained.act_postprocess{tap}.0.project.0", D, 2 * D)
        conv(f"pretrained.act_postprocess{tap}.3", 768, D, 1)
    conv("pretrained.act_postprocess4.4", 768, 768, 3)
    F = 256
    for i, c in enumerate((256, 512, 768, 768), start=1):
        conv(f"scratch.layer{i}_rn", F, c, 3, bias=False)
    for i in range(1, 5):
        conv(f"scratch.refinenet{i}.out_conv", F, F, 1)
        for u in (1, 2):
            for k in (1, 2):
                conv(f"scratch.refinenet{i}.resConfUnit{u}.conv{k}", F, F, 3)
    conv("scratch.output_conv.0", F // 2, F, 3)
    conv("scratch.output_conv.2", 32, F // 2, 3)
    conv("scratch.output_conv.4", 1, 32, 1)
    return m


def n_params(m: Dict[str, Tuple[int, ...]]) -> int:
    total = 0
    for shape in m.values():
        n = 1
        for d in shape:
            n *= d
        total += n
    return total
