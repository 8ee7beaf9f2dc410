"""Shared by ``make_grad_golden.py`` (writer, build container only) and ``tests/test_grad_*.py`` (readers): the seeded cotangents,
the storage format of a gradient fixture and the oracle's autograd gradients.  Test infrastructure."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
GRAD_CASES = ("t0_default", "t0_no_repeat", "t1_c1", "t0_no_sample", "t0_no_latent_concat", "t0_nview1", "t1_no_sample", "t1_no_latent_concat", "t1_nview1", "t0_nview3", "t1_nview3")
WHOLE = 8192            # tensors up to this many entries are stored whole
SAMPLE = 4096           # entries sampled from a larger one
COT_SEED, IDX_SEED = 77, 78


def grad_fixture_path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, f"grad_{name}.npz")


def cotangents(rgb_shape, depth_shape):
    """d L / d rgb and d L / d depth_ray of L = sum(rgb * c_rgb) + sum(depth_ray * c_depth)."""
    g = torch.Generator().manual_seed(COT_SEED)
    return torch.randn(tuple(rgb_shape), generator=g), torch.randn(tuple(depth_shape), generator=g)


def sample_index(key: str, numel: int) -> torch.Tensor:
    seed = IDX_SEED + sum(ord(ch) * (i + 1) for i, ch in enumerate(key)) % 100000
    return torch.randperm(numel, generator=torch.Generator().manual_seed(seed))[:SAMPLE]


def pack(grads: Dict[str, torch.Tensor]) -> Dict[str, np.ndarray]:
    out = {}
    for k, g in grads.items():
        g = g.detach().float().contiguous()
        out[k + "|shape"] = np.array(g.shape, dtype=np.int64)
        out[k + "|sum"] = np.float64(g.double().sum().item())
        out[k + "|sqnorm"] = np.float64((g.double() ** 2).sum().item())
        out[k + "|absmax"] = np.float64(g.double().abs().max().item())
        if g.numel() <= WHOLE:
            out[k + "|whole"] = g.numpy()
        else:
            out[k + "|sample"] = g.view(-1)[sample_index(k, g.numel())].numpy()
    return out


def keys(fx) -> list:
    return sorted({k.split("|")[0] for k in fx.files if "|" in k})


FLIP_FRACTION, FLIP_COUNT, FLIP_WORST = 2e-3, 2, 2e-2


def deviation(got: np.ndarray, want: np.ndarray, scale: float, tol: float):
    """(largest deviation, fraction of entries beyond tol), both relative to the stored tensor's largest entry."""
    e = np.abs(got.astype(np.float64) - want.astype(np.float64)) / scale
    return float(e.max()), float((e > tol).mean())


def within_flip_budget(frac: float, n: int) -> bool:
    """At most FLIP_FRACTION of the entries, but never fewer than FLIP_COUNT of them (a 128-entry bias has room for one flipped row)."""
    return frac * n <= max(FLIP_COUNT, FLIP_FRACTION * n) + 1e-9


def compare(fx, key: str, got: torch.Tensor, tol: float = 1e-3):
    """``got`` against the stored gradient ``key``: every entry within ``tol`` of the stored tensor's largest entry, except for a
    ReLU-flip budget.  The forward has 2e7 pre-activations per case; one that lands within rounding distance of zero takes the other
    branch under any other fp32 summation order, which moves the affected row of the first layer's weight gradient (and the texels
    under that sample) by one sample's contribution.  The oracle — plain torch fp32, same formulas — differs from the reference by
    1.4e-3 on one such row of ``t1_c1`` and by 4e-6 everywhere else (make_grad_golden.py prints it).  Budget: at most FLIP_FRACTION
    of the entries (or FLIP_COUNT of them, for small tensors) beyond ``tol``, none beyond FLIP_WORST; shape and norm are asserted too
    (the norm sees every entry of a sampled tensor).  Returns (max deviation, fraction beyond tol)."""
    got = got.detach().float().cpu().contiguous()
    assert tuple(got.shape) == tuple(fx[key + "|shape"]), (key, tuple(got.shape), tuple(fx[key + "|shape"]))
    scale = max(float(fx[key + "|absmax"]), 1e-12)
    if key + "|whole" in fx.files:
        worst, frac = deviation(got.numpy(), fx[key + "|whole"], scale, tol)
        n = got.numel()
    else:
        idx = sample_index(key, got.numel())
        worst, frac = deviation(got.view(-1)[idx].numpy(), fx[key + "|sample"], scale, tol)
        n = idx.numel()
    norm = float(fx[key + "|sqnorm"]) ** 0.5
    assert abs(float((got.double() ** 2).sum().item()) ** 0.5 - norm) <= 10 * tol * max(norm, 1e-12), (key, "norm")
    assert within_flip_budget(frac, n) and worst <= FLIP_WORST, (key, worst, frac)
    return worst, frac


def oracle_gradients(sd, inp, z, cfg) -> Dict[str, torch.Tensor]:
    """Autograd through the CPU oracle: gradients of the same scalar with respect to every parameter it reads and z."""
    from oracle import car_oracle as O
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    zr = [t.clone().requires_grad_(True) for t in z]
    out = O.render_forward(sdr, inp, zr, cfg)
    c_rgb, c_depth = cotangents(out["rgb"].shape, out["depth_ray"].shape)
    loss = (out["rgb"] * c_rgb).sum() + (out["depth_ray"] * c_depth).sum()
    loss.backward()
    grads = {"param." + k: v.grad for k, v in sdr.items() if v.grad is not None}
    grads.update({f"z.{l}": t.grad for l, t in enumerate(zr)})
    return grads
