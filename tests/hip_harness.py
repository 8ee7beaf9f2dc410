"""Shared helpers of the ``-m gpu`` parity tests: run the HIP path (through the module / C ABI) and the CPU oracle on
the same seeded inputs and return both.  Test infrastructure; the oracle is only ever the checker."""
from __future__ import annotations

import ctypes
from typing import Dict

import numpy as np
import torch

import cases as C
from golden_util import load_case, rel_err
from oracle import car_oracle as O


def oracle_cfg(c) -> O.RenderConfig:
    return O.RenderConfig(n_view=c["n_view"], npoints=c["P"], no_sample=c["no_sample"],
                          no_latent_concat=c["no_latent_concat"], repeat_attention=c["repeat_attention"],
                          H=c["H"], W=c["H"])


def build_module(c, sd, device):
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    m = CrossAttentionRenderer(model=c["model"], n_view=c["n_view"], npoints=c["P"], no_sample=c["no_sample"],
                               no_latent_concat=c["no_latent_concat"], repeat_attention=c["repeat_attention"], with_encoder=False).eval()
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    m.H = m.W = c["H"]
    return m.to(device)


CAMERA_KEYS = ("cam2world", "intrinsics")


def to_device(inp, device, cameras_on_host=False):
    """Moves the input dict to the device; with cameras_on_host the 4x4 camera matrices stay CPU tensors.  Either way the engine runs
    the reference's own torch.inverse / matmul on the host (engine._poses; cameras on the GPU are downloaded first) unless the module's
    pose_route is "device" (car_pose_setup)."""
    return {k: {kk: (vv if (cameras_on_host and kk in CAMERA_KEYS) else vv.to(device)) for kk, vv in v.items()} for k, v in inp.items()}


def run_case(name: str, device="cuda:0", debug=True, linear_flags=0, fixture_poses=False, project_maps=True,
             fuse_samples=True, fuse_round2=True, engine_setup=None, sd_edit=None, z_edit=None, poses=None, cameras_on_host=True,
             pose_route="host"):
    """Returns (config, fixture, oracle output dict (CPU), HIP output dict (tensors moved to CPU)).

    fixture_poses=True: both sides use the relative-pose matrices stored in the fixture (the ones the reference
    computed in the build container) instead of running torch.inverse on this host; ``poses`` (b*V, 96): explicit records for
    both sides; otherwise the engine computes them with the same torch calls as the oracle (strict comparisons) — from CPU camera
    tensors (cameras_on_host=True) or, as the reference's scripts hand them over, from the whole input dict on the GPU (False: the
    engine downloads the matrices).  pose_route="device" with the cameras on the GPU selects car_pose_setup (budgeted comparisons).
    sd_edit / z_edit: functions applied to the case's state_dict / feature pyramid before either side sees them;
    engine_setup(engine): last-minute engine knobs."""
    from cross_attention_renderer_amd.engine import RenderEngine
    c, inp, z, sd, fx = load_case(name)
    if sd_edit is not None:
        sd = sd_edit(dict(sd))
    if z_edit is not None:
        z = z_edit(z)
    if poses is None:
        poses = torch.as_tensor(fx["poses"]) if fixture_poses else None
    with torch.no_grad():
        ora = O.render_forward(sd, inp, z, oracle_cfg(c), debug=True, poses96=poses)
    m = build_module(c, sd, device)
    m._engine = RenderEngine(m)
    m._engine.linear_flags = linear_flags
    m._engine.pose_records = poses
    m.pose_route = pose_route
    m._engine.project_maps = project_maps
    m._engine.fuse_samples = fuse_samples
    m._engine.fuse_round2 = fuse_round2
    if engine_setup is not None:
        engine_setup(m._engine)
    with torch.no_grad():
        out = m(to_device(inp, device, cameras_on_host), z=[t.to(device) for t in z], debug=debug)
    torch.cuda.synchronize()

    def cpu(v):
        if isinstance(v, torch.Tensor):
            return v.detach().cpu()
        if isinstance(v, dict):
            return {k: cpu(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [cpu(x) for x in v]
        return v
    return c, fx, ora, cpu(out)


def err_stats(a, b) -> Dict[str, float]:
    """Relative error |a-b|/max(1,|b|): max, and the fraction of elements above 1e-4 / 1e-3."""
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    e = (a - b).abs() / b.abs().clamp_min(1.0)
    e = torch.where(torch.isfinite(e), e, torch.full_like(e, float("inf")))
    return {"max": e.max().item(), "f1e-4": (e > 1e-4).double().mean().item(), "f1e-3": (e > 1e-3).double().mean().item()}


def ray_err(a, b, ray_dim: int):
    """Per-ray max relative error (reduces every dim except ``ray_dim`` and the leading batch dims before it)."""
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    e = (a - b).abs() / b.abs().clamp_min(1.0)
    dims = [d for d in range(e.dim()) if d > ray_dim]
    return e.amax(dim=dims) if dims else e


def argmax_exact_where_decided(got_idx, want_w, margin=1e-6):
    """the per-view argmax of the attention weights must be exact on every ray whose decision margin (largest minus
    second-largest reference weight of that view) exceeds ``margin``; closer calls are ties the last ulp may flip.  Returns
    (number of decided rays, number of disagreements among them)."""
    w = torch.as_tensor(np.asarray(want_w)).double()
    top2 = w.topk(2, dim=-1).values
    decided = (top2[..., 0] - top2[..., 1]) > margin
    got = torch.as_tensor(np.asarray(got_idx)).reshape(w.shape[:-1])
    want = w.argmax(dim=-1)
    return int(decided.sum()), int(((got != want) & decided).sum())
