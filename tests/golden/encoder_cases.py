"""Inputs of the ``get_z`` / encoder fixtures, shared by ``make_encoder_golden.py`` (writer, build container only) and
``tests/test_encoder.py`` (reader): seeded weights for a name -> shape table, a seeded context pair, and which samples of the
feature pyramid are stored."""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import numpy as np
import torch

from cross_attention_renderer_amd import synthetic as S

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
H = 256                        # the multi-view encoder only works at 256x256 (midas/vit.py:183: 257 tokens per view)
VARIANTS = {"default": {}, "no_multiview": {"no_multiview": True}, "no_high_freq": {"no_high_freq": True}}
STRIDES = ((8, 4), (8, 8), (4, 16))          # (channel, pixel) stride of the stored samples of [path_2, path_1, conv_map]


def seeded_weights(shapes: Dict[str, Sequence[int]], seed: int = 7) -> Dict[str, torch.Tensor]:
    """Values for every entry of a state_dict shape table, drawn in sorted-name order from one generator: matrices and kernels
    ~ N(0, 1/fan_in) (x2 for the weight-standardised convolutions, whose scale is normalised away), norm scales 1 + 0.2 N, every
    bias / embedding 0.1-0.2 N — nothing is left at an initial value that would hide a wiring mistake."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        t = torch.randn(shp, generator=g)
        if len(shp) >= 2 and "embed" not in name.split(".")[-1] and "token" not in name:
            fan_in = 1
            for s_ in shp[1:]:
                fan_in *= s_
            t = t * (2.0 / max(fan_in, 1)) ** 0.5
        elif name.endswith("weight") and len(shp) == 1:
            t = 1.0 + 0.2 * t
        elif len(shp) >= 2:
            t = 0.2 * t                          # cls_token, pos_embed, pos_embed_second
        else:
            t = 0.1 * t
        out[name] = t
    return out


def context_pair(seed: int = 3):
    """Input dict with two seeded context images in [-1, 1] and the cameras of the synthetic stereo rig."""
    inp = S.stereo_scene(H, b=1, seed=5, uv=S.pixel_grid(H, H)[:16].contiguous())
    g = torch.Generator().manual_seed(seed)
    # smooth-ish images: low-frequency pattern + noise, so that every stage sees structure
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, H), indexing="ij")
    base = torch.stack([torch.sin(3 * xs + ys), torch.cos(2 * ys - xs), xs * ys], dim=-1)
    rgb = (0.6 * base[None, None] + 0.4 * (torch.rand(1, 2, H, H, 3, generator=g) * 2 - 1)).clamp(-1, 1)
    rgb[:, 1] = rgb[:, 1].flip(2)
    inp["context"]["rgb"] = rgb.contiguous()
    return inp


def sample(z: List[torch.Tensor]) -> List[np.ndarray]:
    return [t[:, ::cs, ::ps, ::ps].contiguous().numpy() for t, (cs, ps) in zip(z, STRIDES)]


def stats(z: List[torch.Tensor]) -> np.ndarray:
    return np.asarray([[t.double().mean().item(), t.double().abs().mean().item(), t.double().abs().max().item()] for t in z])


def fixture_path(variant: str) -> str:
    return os.path.join(GOLDEN_DIR, f"getz_{variant}.npz")
