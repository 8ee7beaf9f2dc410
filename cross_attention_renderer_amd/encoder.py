"""The multi-view DPT-hybrid image encoder behind ``CrossAttentionRenderer.get_z``.

Runs once per stereo pair, outside the per-ray hot loop, so it is plain PyTorch-ROCm (MIOpen convolutions, rocBLAS /
``scaled_dot_product_attention`` for the transformer) — no hand-written kernels here.  What matters is that a reference checkpoint
loads unchanged (``encoder.*`` keys and shapes are the reference's) and that ``get_z`` produces the reference's feature pyramid.

What it computes (reference call sites):
  * ``DPTDepthModel.forward(rgb, rel_pose16, nviews)`` (midas/dpt_depth.py:67-89, 94-117): hybrid backbone -> four reassembled
    feature maps -> RefineNet fusion -> ``[path_2 (256 @ H/4), path_1 (256 @ H/2)]``;
  * the backbone (midas/vit.py:124-202, 392-541 with vit_models.py:10-97): ResNetV2-50 stem + stages (3, 4, 9) at strides 4 / 8 /
    16, a 1x1 projection to 768-wide tokens, a class token, the bilinearly resized position embedding and a ``Linear(16, 768)``
    embedding of the view's relative pose added to every token; then the tokens of ALL views of a scene are concatenated into one
    sequence (midas/vit.py:183-186) and run through 12 pre-norm transformer blocks; blocks 8 and 11 are tapped, split back per view
    (midas/vit.py:65-69), the class token is folded in by ``ProjectReadout`` (midas/vit.py:31-42) and the maps are re-assembled;
  * the fusion blocks (midas/blocks.py:231-341).

Third-party arithmetic.  The reference builds the ResNetV2 trunk, the transformer ``Block`` and ``HybridEmbed`` from
``timm==0.5.4`` (requirements.txt; vit_models.py:1-4), which is not part of the reference tree and not installed here.  Those
pieces are restated below from timm 0.5.4's published definitions (``resnetv2.py``: non-pre-activation ``Bottleneck``,
``StdConv2dSame`` weight standardisation with TF "SAME" padding, ``GroupNormAct`` with 32 groups, ``MaxPool2dSame``;
``vision_transformer.py``: ``Block`` / ``Attention`` / ``Mlp`` with LayerNorm eps 1e-6 and exact GELU).  Parity of everything the
reference itself defines is pinned by tests/golden/make_encoder_golden.py, which runs the reference's own modules; the timm-derived
layers cannot be pinned against timm in this image ("parity unpinned" for them).
"""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------------------------------
# timm 0.5.4 layers (restated): weight-standardised convolution with TF "SAME" padding, GroupNorm + ReLU, SAME max-pool
# ----------------------------------------------------------------------------------------------------------------------
def _same_pad(x: Tensor, k: int, s: int, value: float = 0.0) -> Tensor:
    """TF-style SAME padding for a stride-s window of size k: total max((ceil(i/s) - 1) s + k - i, 0), the extra pixel on the
    bottom / right."""
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)
    return x


class StdConv2dSame(nn.Conv2d):
    """Conv2d whose weight is standardised per output channel (zero mean, unit biased variance over in x kh x kw, ``eps`` inside the
    square root) on every forward; SAME padding: static (k - 1) / 2 at stride 1, dynamic (input-size dependent) otherwise."""

    def __init__(self, in_ch: int, out_ch: int, kernel_size: int, stride: int = 1, eps: float = 1e-6):
        static = stride == 1
        super().__init__(in_ch, out_ch, kernel_size, stride=stride, padding=(kernel_size - 1) // 2 if static else 0, bias=False)
        self.dynamic_pad = not static
        self.eps = eps

    def forward(self, x: Tensor) -> Tensor:
        if self.dynamic_pad:
            x = _same_pad(x, self.kernel_size[0], self.stride[0])
        w = self.weight.reshape(self.out_channels, -1)
        var, mean = torch.var_mean(w, dim=1, unbiased=False, keepdim=True)
        w = ((w - mean) * torch.rsqrt(var + self.eps)).reshape_as(self.weight)
        return F.conv2d(x, w, None, self.stride, self.padding)


class GroupNormAct(nn.GroupNorm):
    def __init__(self, channels: int, apply_act: bool = True):
        super().__init__(32, channels, eps=1e-5)
        self.apply_act = apply_act

    def forward(self, x: Tensor) -> Tensor:
        x = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        return F.relu(x) if self.apply_act else x


class MaxPool2dSame(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return F.max_pool2d(_same_pad(x, 3, 2, value=-float("inf")), 3, 2)


class _Downsample(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, stride: int):
        super().__init__()
        self.conv = StdConv2dSame(in_ch, out_ch, 1, stride=stride, eps=1e-8)
        self.norm = GroupNormAct(out_ch, apply_act=False)

    def forward(self, x: Tensor) -> Tensor:
        return self.norm(self.conv(x))


class Bottleneck(nn.Module):
    """timm resnetv2 ``Bottleneck`` (the non-pre-activation block used for ViT hybrids): 1x1 -> 3x3 (stride) -> 1x1, GroupNorm after
    every convolution, ReLU after the first two and after the residual sum."""

    def __init__(self, in_ch: int, out_ch: int, stride: int, project: bool):
        super().__init__()
        mid = out_ch // 4
        self.downsample = _Downsample(in_ch, out_ch, stride) if project else None
        self.conv1 = StdConv2dSame(in_ch, mid, 1, eps=1e-8)
        self.norm1 = GroupNormAct(mid)
        self.conv2 = StdConv2dSame(mid, mid, 3, stride=stride, eps=1e-8)
        self.norm2 = GroupNormAct(mid)
        self.conv3 = StdConv2dSame(mid, out_ch, 1, eps=1e-8)
        self.norm3 = GroupNormAct(out_ch, apply_act=False)

    def forward(self, x: Tensor) -> Tensor:
        shortcut = x if self.downsample is None else self.downsample(x)
        x = self.norm1(self.conv1(x))
        x = self.norm2(self.conv2(x))
        x = self.norm3(self.conv3(x))
        return F.relu(x + shortcut)


class _Stage(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, stride: int, depth: int):
        super().__init__()
        self.blocks = nn.Sequential(*[Bottleneck(in_ch if i == 0 else out_ch, out_ch, stride if i == 0 else 1, project=(i == 0))
                                      for i in range(depth)])

    def forward(self, x: Tensor) -> Tensor:
        return self.blocks(x)


class ResNetV2Trunk(nn.Module):
    """``timm.models.vision_transformer_hybrid._resnetv2((3, 4, 9))``: stem (7x7 stride 2, GroupNorm+ReLU, SAME max-pool) and three
    stages of 256 / 512 / 1024 channels at strides 4 / 8 / 16.  The reference replaces the stem convolution by a fresh
    ``StdConv2dSame(3, 64, 7, stride 2)`` with the class default eps 1e-6 (models.py:93)."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential()
        self.stem.add_module("conv", StdConv2dSame(3, 64, 7, stride=2, eps=1e-6))
        self.stem.add_module("norm", GroupNormAct(64))
        self.stem.add_module("pool", MaxPool2dSame())
        self.stages = nn.Sequential(_Stage(64, 256, 1, 3), _Stage(256, 512, 2, 4), _Stage(512, 1024, 2, 9))

    def forward(self, x: Tensor) -> List[Tensor]:
        x = self.stem(x)
        feats = []
        for stage in self.stages:
            x = stage(x)
            feats.append(x)
        return feats


class HybridEmbed(nn.Module):
    def __init__(self, embed_dim: int = 768):
        super().__init__()
        self.backbone = ResNetV2Trunk()
        self.proj = nn.Conv2d(1024, embed_dim, kernel_size=1, stride=1)


# ----------------------------------------------------------------------------------------------------------------------
# timm 0.5.4 transformer block (restated)
# ----------------------------------------------------------------------------------------------------------------------
class Attention(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x: Tensor) -> Tensor:
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4).unbind(0)
        x = F.scaled_dot_product_attention(q, k, v)            # softmax(q k^T / sqrt(head_dim)) v
        return self.proj(x.transpose(1, 2).reshape(B, N, C))


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x: Tensor) -> Tensor:
        return self.fc2(F.gelu(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, 4 * dim)

    def forward(self, x: Tensor) -> Tensor:
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class MultiViewViT(nn.Module):
    """``VisionTransformerMultiView`` (vit_models.py:10-97) with the reference's ``forward_flex`` (midas/vit.py:124-202): parameter
    holder for ``encoder.pretrained.model.*``.  ``head`` (768 -> 1000) and ``pos_embed_second`` exist in the reference's state_dict
    and have no effect on the features."""

    def __init__(self, embed_dim: int = 768, depth: int = 12, heads: int = 12, grid: int = 24):
        super().__init__()
        self.patch_embed = HybridEmbed(embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, grid * grid + 1, embed_dim))
        self.pos_embed_second = nn.Parameter(torch.zeros(1, grid * grid + 1, embed_dim))
        self.blocks = nn.Sequential(*[Block(embed_dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.pose_embed = nn.Linear(16, embed_dim)
        self.head = nn.Linear(embed_dim, 1000)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.pos_embed_second, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)

    def resized_pos_embed(self, gh: int, gw: int) -> Tensor:
        """Class-token entry kept, grid entries bilinearly resized to gh x gw (midas/vit.py:107-121)."""
        tok, grid = self.pos_embed[:, :1], self.pos_embed[0, 1:]
        g = int(math.sqrt(grid.shape[0]))
        grid = F.interpolate(grid.reshape(1, g, g, -1).permute(0, 3, 1, 2), size=(gh, gw), mode="bilinear")
        return torch.cat([tok, grid.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)


class ProjectReadout(nn.Module):
    """Concatenates the class token to every patch token and projects back to the token width (midas/vit.py:31-42)."""

    def __init__(self, dim: int):
        super().__init__()
        self.project = nn.Sequential(nn.Linear(2 * dim, dim), nn.GELU())

    def forward(self, x: Tensor) -> Tensor:
        readout = x[:, :1].expand(-1, x.shape[1] - 1, -1)
        return self.project(torch.cat((x[:, 1:], readout), -1))


class _Pretrained(nn.Module):
    """``encoder.pretrained``: the transformer plus the re-assembly heads of taps 3 and 4 (midas/vit.py:482-513); taps 1 and 2 are the
    ResNet stages themselves (identity post-processing, midas/vit.py:474-480).  Sequential indices follow the reference so that the
    checkpoint keys match: [0] readout, [1] transpose, [2] unflatten, [3] 1x1 conv, ([4] 3x3 stride-2 conv)."""

    def __init__(self, dim: int = 768):
        super().__init__()
        self.model = MultiViewViT(dim)
        self.act_postprocess3 = nn.Sequential(ProjectReadout(dim), nn.Identity(), nn.Identity(), nn.Conv2d(dim, 768, 1))
        self.act_postprocess4 = nn.Sequential(ProjectReadout(dim), nn.Identity(), nn.Identity(), nn.Conv2d(dim, 768, 1),
                                              nn.Conv2d(768, 768, 3, stride=2, padding=1))


# ----------------------------------------------------------------------------------------------------------------------
# RefineNet fusion (midas/blocks.py:231-341) and the DPT wrapper
# ----------------------------------------------------------------------------------------------------------------------
class ResidualConvUnit(nn.Module):
    def __init__(self, ch: int):
        super().__init__()
        self.conv1 = nn.Conv2d(ch, ch, 3, padding=1)
        self.conv2 = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x: Tensor) -> Tensor:
        return self.conv2(F.relu(self.conv1(F.relu(x)))) + x


class FeatureFusionBlock(nn.Module):
    def __init__(self, ch: int):
        super().__init__()
        self.out_conv = nn.Conv2d(ch, ch, 1)
        self.resConfUnit1 = ResidualConvUnit(ch)
        self.resConfUnit2 = ResidualConvUnit(ch)

    def forward(self, x: Tensor, skip: Tensor = None) -> Tensor:
        if skip is not None:
            x = x + self.resConfUnit1(skip)
        x = self.resConfUnit2(x)
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        return self.out_conv(x)


class _Scratch(nn.Module):
    def __init__(self, ch: int = 256):
        super().__init__()
        for i, c in enumerate((256, 512, 768, 768), start=1):
            setattr(self, f"layer{i}_rn", nn.Conv2d(c, ch, 3, padding=1, bias=False))
        for i in range(1, 5):
            setattr(self, f"refinenet{i}", FeatureFusionBlock(ch))
        # the monocular-depth head of DPTDepthModel (midas/dpt_depth.py:98-106): in the checkpoint, never evaluated by get_z
        self.output_conv = nn.Sequential(nn.Conv2d(ch, ch // 2, 3, padding=1), nn.Identity(), nn.Conv2d(ch // 2, 32, 3, padding=1),
                                         nn.Identity(), nn.Conv2d(32, 1, 1), nn.Identity(), nn.Identity())


class MultiViewDPTEncoder(nn.Module):
    """Drop-in for the reference's ``self.encoder`` when ``model == "midas_vit"`` (models.py:82-94): same parameter names, same
    ``forward(rgb, rel_pose16, nviews) -> [path_2, path_1]``."""

    TOKENS_PER_VIEW = 257          # `os = 257` in the reference's forward_flex (midas/vit.py:183): 16 x 16 patches + class token

    def __init__(self):
        super().__init__()
        self.pretrained = _Pretrained()
        self.scratch = _Scratch()

    def forward(self, x: Tensor, rel_pose16: Tensor, nviews: int) -> List[Tensor]:
        vit = self.pretrained.model
        BV, _, H, W = x.shape
        gh, gw = H // 16, W // 16
        if gh * gw + 1 != self.TOKENS_PER_VIEW:
            raise ValueError(f"the multi-view encoder splits the token sequence at {self.TOKENS_PER_VIEW} tokens per view "
                             f"(midas/vit.py:183, 199): only 256x256 inputs work, got {H}x{W}")
        if BV % nviews:
            raise ValueError("batch of views is not a multiple of nviews")
        layer_1, layer_2, feat = vit.patch_embed.backbone(x)                   # 256 @ H/4, 512 @ H/8, 1024 @ H/16
        tok = vit.patch_embed.proj(feat).flatten(2).transpose(1, 2)            # (BV, gh*gw, 768)
        tok = torch.cat((vit.cls_token.expand(BV, -1, -1), tok), dim=1)
        tok = tok + vit.resized_pos_embed(gh, gw) + vit.pose_embed(rel_pose16)[:, None, :]
        T = tok.shape[1]
        tok = tok.view(BV // nviews, nviews * T, -1)                           # all views of a scene in one sequence
        taps = {}
        for i, blk in enumerate(vit.blocks):
            tok = blk(tok)
            if i in (8, 11):
                taps[i] = tok.reshape(BV, T, -1)                               # back to one row per view (midas/vit.py:65-69)

        def reassemble(t: Tensor, post: nn.Sequential) -> Tensor:
            t = post[0](t).transpose(1, 2).unflatten(2, (gh, gw))
            for layer in post[3:]:
                t = layer(t)
            return t
        layer_3 = reassemble(taps[8], self.pretrained.act_postprocess3)        # 768 @ H/16
        layer_4 = reassemble(taps[11], self.pretrained.act_postprocess4)       # 768 @ H/32
        s = self.scratch
        path_4 = s.refinenet4(s.layer4_rn(layer_4))
        path_3 = s.refinenet3(path_4, s.layer3_rn(layer_3))
        path_2 = s.refinenet2(path_3, s.layer2_rn(layer_2))
        path_1 = s.refinenet1(path_2, s.layer1_rn(layer_1))
        return [path_2, path_1]
