"""A minimal stand-in for ``timm==0.5.4`` (the reference's pinned version, requirements.txt), enough to import and RUN the
reference's own encoder code (vit_models.py, midas/vit.py, midas/dpt_depth.py, midas/blocks.py) in this container, where timm is not
installed.  Test infrastructure only: used by ``make_encoder_golden.py`` to produce the encoder fixtures.

The classes restate timm 0.5.4's published definitions the reference builds on — ``resnetv2.py`` (``ResNetV2`` with the
non-pre-activation ``Bottleneck``, ``StdConv2dSame``, ``GroupNormAct``, ``MaxPool2dSame``), ``vision_transformer.py`` (``Block``,
``Attention``, ``Mlp``) and ``vision_transformer_hybrid.py`` (``HybridEmbed``, ``_resnetv2``) — written independently of
``cross_attention_renderer_amd/encoder.py`` (batch_norm-based weight standardisation, explicit softmax attention, padding helpers of
their own), so that the fixtures also cross-check two formulations of the same layers.  Everything the REFERENCE defines (token
concatenation across views, pose embedding, position-embedding resize, readout projection, re-assembly, RefineNet fusion) runs as
the reference's own code.  What this cannot do is pin those layers against timm's actual source: "parity unpinned" for them.
"""
from __future__ import annotations

import math
import sys
import types
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- timm.models.layers ----------------------------------------------------------------------------------------------
def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def get_same_padding(x: int, k: int, s: int, d: int) -> int:
    return max((math.ceil(x / s) - 1) * s + (k - 1) * d + 1 - x, 0)


def pad_same(x, k, s, d=(1, 1), value: float = 0.0):
    ih, iw = x.size()[-2:]
    pad_h, pad_w = get_same_padding(ih, k[0], s[0], d[0]), get_same_padding(iw, k[1], s[1], d[1])
    if pad_h > 0 or pad_w > 0:
        x = F.pad(x, [pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2], value=value)
    return x


class StdConv2dSame(nn.Conv2d):
    """Conv2d with weight standardisation and TF-compatible SAME padding (static when stride 1 and (k - 1) even, else dynamic)."""

    def __init__(self, in_channel, out_channels, kernel_size, stride=1, padding="SAME", dilation=1, groups=1, bias=False, eps=1e-6):
        k, s, d = _pair(kernel_size), _pair(stride), _pair(dilation)
        static = all(si == 1 and (di * (ki - 1)) % 2 == 0 for ki, si, di in zip(k, s, d))
        pad = tuple(((si - 1) + di * (ki - 1)) // 2 for ki, si, di in zip(k, s, d)) if static else 0
        super().__init__(in_channel, out_channels, k, stride=s, padding=pad, dilation=d, groups=groups, bias=bias)
        self.same_pad = not static
        self.eps = eps

    def forward(self, x):
        if self.same_pad:
            x = pad_same(x, self.kernel_size, self.stride, self.dilation)
        weight = F.batch_norm(self.weight.reshape(1, self.out_channels, -1), None, None, training=True, momentum=0.0,
                              eps=self.eps).reshape_as(self.weight)
        return F.conv2d(x, weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class GroupNormAct(nn.GroupNorm):
    def __init__(self, num_channels, num_groups=32, eps=1e-5, affine=True, apply_act=True, act_layer=nn.ReLU, inplace=True, drop_block=None):
        super().__init__(num_groups, num_channels, eps=eps, affine=affine)
        self.act = act_layer(inplace=inplace) if apply_act else nn.Identity()

    def forward(self, x):
        return self.act(F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps))


class MaxPool2dSame(nn.MaxPool2d):
    def __init__(self, kernel_size, stride=None):
        super().__init__(_pair(kernel_size), _pair(stride), (0, 0), (1, 1), False)

    def forward(self, x):
        x = pad_same(x, self.kernel_size, self.stride, value=-float("inf"))
        return F.max_pool2d(x, self.kernel_size, self.stride, (0, 0), self.dilation, self.ceil_mode)


def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)


class PatchEmbed(nn.Module):           # imported by vit_models.py as a default argument; the hybrid model never builds it
    def __init__(self, *a, **k):
        raise RuntimeError("PatchEmbed is not used by the hybrid encoder")


# ---- timm.models.resnetv2 (ViT-hybrid flavour) -------------------------------------------------------------------------
class DownsampleConv(nn.Module):
    def __init__(self, in_chs, out_chs, stride, conv_layer, norm_layer):
        super().__init__()
        self.conv = conv_layer(in_chs, out_chs, 1, stride=stride)
        self.norm = norm_layer(out_chs, apply_act=False)

    def forward(self, x):
        return self.norm(self.conv(x))


class Bottleneck(nn.Module):
    """Non pre-activation bottleneck (V1.5-style), the block of the ViT-hybrid ResNetV2."""

    def __init__(self, in_chs, out_chs, stride, conv_layer, norm_layer, proj):
        super().__init__()
        mid_chs = int(out_chs * 0.25)
        self.downsample = DownsampleConv(in_chs, out_chs, stride, conv_layer, norm_layer) if proj else None
        self.conv1 = conv_layer(in_chs, mid_chs, 1)
        self.norm1 = norm_layer(mid_chs)
        self.conv2 = conv_layer(mid_chs, mid_chs, 3, stride=stride)
        self.norm2 = norm_layer(mid_chs)
        self.conv3 = conv_layer(mid_chs, out_chs, 1)
        self.norm3 = norm_layer(out_chs, apply_act=False)
        self.act3 = nn.ReLU(inplace=True)

    def forward(self, x):
        shortcut = x
        if self.downsample is not None:
            shortcut = self.downsample(x)
        x = self.norm1(self.conv1(x))
        x = self.norm2(self.conv2(x))
        x = self.norm3(self.conv3(x))
        return self.act3(x + shortcut)


class ResNetStage(nn.Module):
    def __init__(self, in_chs, out_chs, stride, depth, conv_layer, norm_layer):
        super().__init__()
        self.blocks = nn.Sequential()
        prev = in_chs
        for i in range(depth):
            self.blocks.add_module(str(i), Bottleneck(prev, out_chs, stride if i == 0 else 1, conv_layer, norm_layer, proj=(i == 0)))
            prev = out_chs

    def forward(self, x):
        return self.blocks(x)


class ResNetV2(nn.Module):
    def __init__(self, layers, channels=(256, 512, 1024, 2048), in_chans=3, stem_chs=64, conv_layer=None, norm_layer=None):
        super().__init__()
        self.stem = nn.Sequential()
        self.stem.add_module("conv", conv_layer(in_chans, stem_chs, kernel_size=7, stride=2))
        self.stem.add_module("norm", norm_layer(stem_chs))
        self.stem.add_module("pool", MaxPool2dSame(3, 2))
        self.stages = nn.Sequential()
        prev = stem_chs
        for idx, (d, c) in enumerate(zip(layers, channels)):
            self.stages.add_module(str(idx), ResNetStage(prev, c, 1 if idx == 0 else 2, d, conv_layer, norm_layer))
            prev = c
        self.num_features = prev
        self.norm = nn.Identity()            # preact=False
        self.head = nn.Identity()            # num_classes=0, global_pool=''

    def forward(self, x):
        return self.head(self.norm(self.stages(self.stem(x))))


def _resnetv2(layers=(3, 4, 9), **kwargs):
    conv_layer = partial(StdConv2dSame, eps=1e-8)
    return ResNetV2(layers=layers, in_chans=kwargs.get("in_chans", 3), conv_layer=conv_layer,
                    norm_layer=partial(GroupNormAct, num_groups=32))


class HybridEmbed(nn.Module):
    def __init__(self, backbone, img_size=224, patch_size=1, feature_size=None, in_chans=3, embed_dim=768):
        super().__init__()
        img_size, patch_size = _pair(img_size), _pair(patch_size)
        self.img_size, self.patch_size, self.backbone = img_size, patch_size, backbone
        feature_size = (img_size[0] // 16, img_size[1] // 16)           # stride of the (3, 4, 9) trunk
        self.grid_size = (feature_size[0] // patch_size[0], feature_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(backbone.num_features, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        x = self.backbone(x)
        return self.proj(x).flatten(2).transpose(1, 2)


# ---- timm.models.vision_transformer ----------------------------------------------------------------------------------
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = self.attn_drop(attn.softmax(dim=-1))
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


def _init_vit_weights(module, name="", head_bias=0.0, jax_impl=False):
    if isinstance(module, nn.Linear):
        trunc_normal_(module.weight, std=0.02)
        if module.bias is not None:
            nn.init.zeros_(module.bias)
    elif isinstance(module, (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm2d)):
        nn.init.zeros_(module.bias)
        nn.init.ones_(module.weight)


def checkpoint_filter_fn(state_dict, model):
    return state_dict


default_cfgs = {"vit_base_r50_s16_384": {"url": "", "num_classes": 1000, "input_size": (3, 384, 384), "fixed_input_size": True,
                                         "first_conv": "patch_embed.backbone.stem.conv", "classifier": "head"}}


def build_model_with_cfg(model_cls, variant, pretrained, default_cfg=None, representation_size=None, pretrained_filter_fn=None,
                         pretrained_custom_load=False, **kwargs):
    assert not pretrained
    kwargs.setdefault("num_classes", default_cfg["num_classes"])
    if default_cfg.get("fixed_input_size", False):
        kwargs.setdefault("img_size", default_cfg["input_size"][-2:])
    return model_cls(representation_size=representation_size, **kwargs)


def install() -> None:
    """Registers the stand-in under timm's module names (must run before the reference's modules are imported)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    std_conv = mod("timm.models.layers.std_conv", StdConv2dSame=StdConv2dSame)
    layers = mod("timm.models.layers", PatchEmbed=PatchEmbed, trunc_normal_=trunc_normal_, std_conv=std_conv, StdConv2dSame=StdConv2dSame)
    vt = mod("timm.models.vision_transformer", Block=Block, _init_vit_weights=_init_vit_weights, checkpoint_filter_fn=checkpoint_filter_fn,
             _create_vision_transformer=None)
    vth = mod("timm.models.vision_transformer_hybrid", _resnetv2=_resnetv2, HybridEmbed=HybridEmbed, default_cfgs=default_cfgs)
    helpers = mod("timm.models.helpers", build_model_with_cfg=build_model_with_cfg)
    models = mod("timm.models", layers=layers, vision_transformer=vt, vision_transformer_hybrid=vth, helpers=helpers)
    mod("timm", models=models)
