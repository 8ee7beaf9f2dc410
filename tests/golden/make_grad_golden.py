"""Gradient golden vectors: the REFERENCE's autograd gradients, captured in this container.

    python tests/golden/make_grad_golden.py [case ...]          (default: t0_default t1_c1)

For each case the reference ``CrossAttentionRenderer.forward(input, z=z)`` (imported read-only from /root/reference through
``ref_import.py``) is run with autograd on, the scalar ``L = sum(rgb * c_rgb) + sum(depth_ray * c_depth)`` with seeded cotangents
is back-propagated — the two outputs the reference's loss reads (loss_functions.py:74-132) — and the gradients with respect to
every renderer parameter and every level of the feature pyramid ``z`` are written to ``tests/golden/grad_<case>.npz``.
Small tensors are stored whole; of a large one the file keeps its sum, its squared norm and a seeded sample of 4096 entries (the
index set is regenerated from the seed by the reader).  Data only: the reference never travels.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cases as C                                      # noqa: E402
import grad_cases as G                                 # noqa: E402
import ref_import                                      # noqa: E402
from cross_attention_renderer_amd import synthetic as S   # noqa: E402
from oracle import car_oracle as O                     # noqa: E402


def run_case(name: str) -> None:
    c = C.case_config(name)
    torch.manual_seed(0)
    inp, z = C.build_inputs(c)
    sd = S.seeded_state_dict(C.param_shapes(c), seed=c["w_seed"])
    model = ref_import.build_reference_model(
        n_view=c["n_view"], npoints=c["P"], model=c["model"], H=c["H"], no_sample=c["no_sample"],
        no_latent_concat=c["no_latent_concat"], repeat_attention=c["repeat_attention"])
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("encoder.") for k in missing.missing_keys), missing
    zr = [t.clone().requires_grad_(True) for t in z]
    out = model(inp, z=zr)
    c_rgb, c_depth = G.cotangents(out["rgb"].shape, out["depth_ray"].shape)
    loss = (out["rgb"] * c_rgb).sum() + (out["depth_ray"] * c_depth).sum()
    loss.backward()
    grads = {}
    for k, p in model.named_parameters():
        if k.startswith("encoder.") or p.grad is None:
            continue
        grads["param." + k] = p.grad.detach()
    for l, t in enumerate(zr):
        grads[f"z.{l}"] = t.grad.detach()
    unused = sorted(k for k, p in model.named_parameters() if not k.startswith("encoder.") and p.grad is None)

    # the oracle's autograd against the reference's, right here where the reference exists
    cfg = O.RenderConfig(n_view=c["n_view"], npoints=c["P"], no_sample=c["no_sample"], no_latent_concat=c["no_latent_concat"],
                         repeat_attention=c["repeat_attention"], H=c["H"], W=c["H"])
    og = G.oracle_gradients(sd, inp, z, cfg)
    worst, flips = 0.0, []
    for k, g in grads.items():
        assert k in og, f"oracle has no gradient for {k}"
        w, frac = G.deviation(og[k].numpy(), g.numpy(), max(g.double().abs().max().item(), 1e-12), 1e-4)
        assert G.within_flip_budget(frac, g.numel()) and w <= G.FLIP_WORST, (k, w, frac)
        if w > 1e-4:
            flips.append(f"{k} {w:.1e} ({frac:.1e} of its entries)")
        else:
            worst = max(worst, w)
    assert not (set(og) - set(grads)), f"oracle produced gradients the reference does not: {set(og) - set(grads)}"
    payload = {"loss": np.float64(loss.item()), "unused": np.array(unused)}
    payload.update(G.pack(grads))
    path = G.grad_fixture_path(name)
    np.savez_compressed(path, **payload)
    print(f"{name:14s} loss {loss.item():+.6e}  {len(grads)} gradient tensors, {len(unused)} parameters without gradient; "
          f"oracle-vs-reference worst {worst:.1e} of each tensor's largest entry" + (f", ReLU flips: {'; '.join(flips)}" if flips else "") + f"; wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    if not ref_import.reference_available():
        sys.exit("the reference tree is not present: golden vectors can only be generated in the build container")
    for n in sys.argv[1:] or list(G.GRAD_CASES):
        run_case(n)
