"""Host-side pose algebra of the render forward (reference models.py:207-211, 226-228, 285-286; geometry.py:404).

The reference computes these tiny 4x4 products with ``torch.inverse`` (LAPACK) and ``torch.matmul`` in fp32.
The per-sample fp64 Pluecker intersection downstream amplifies last-ulp differences of these matrices on
ill-conditioned samples, so the product does *literally the same torch calls on the CPU* and ships the result
to the device as ``CarPose`` records (``csrc/car_geom.h``) — 96 floats per (scene, context view).  This costs
one small D2H/H2D round trip per ``forward`` (the reference itself syncs once per call, models.py:570).
"""
from __future__ import annotations

import torch

POSE_FLOATS = 96
MAX_VIEWS = 3


def pack_poses(inp, H: int) -> torch.Tensor:
    """input dict -> (b*V, 96) float32 CPU tensor laid out as ``struct CarPose``."""
    c2w = inp["context"]["cam2world"].detach().float().cpu()
    Kc = inp["context"]["intrinsics"].detach().float().cpu()
    c2w_q = inp["query"]["cam2world"].detach().float().cpu()
    Kq = inp["query"]["intrinsics"].detach().float().cpu()
    b, V = c2w.shape[:2]
    assert V <= MAX_VIEWS
    inv_ctx = torch.inverse(c2w)
    q_rel = torch.matmul(inv_ctx, c2w_q)                       # (b,V,4,4)
    c_rel = torch.matmul(inv_ctx, c2w)
    out = torch.zeros(b, V, POSE_FLOATS)
    out[..., 0:12] = q_rel[..., :3, :].reshape(b, V, 12)
    out[..., 12:24] = c_rel[..., :3, :].reshape(b, V, 12)
    for s in range(V):
        Ts = torch.matmul(torch.inverse(c2w[:, s:s + 1]), c2w)
        out[..., 24 + 12 * s:36 + 12 * s] = Ts[..., :3, :].reshape(b, V, 12)
    out[..., 60] = Kc[..., 0, 0]; out[..., 61] = Kc[..., 1, 1]; out[..., 62] = Kc[..., 0, 2]; out[..., 63] = Kc[..., 1, 2]
    K01 = Kc[..., :3, :3].clone()
    K01[..., :2, :] = Kc[..., :2, :3] / H
    out[..., 64:73] = K01.reshape(b, V, 9)
    out[..., 73] = Kq[:, :, 0, 0]; out[..., 74] = Kq[:, :, 1, 1]; out[..., 75] = Kq[:, :, 0, 2]; out[..., 76] = Kq[:, :, 1, 2]
    inv_q = torch.inverse(c2w_q[:, 0])
    out[..., 77:89] = inv_q[:, None, :3, :].reshape(b, 1, 12)
    return out.reshape(b * V, POSE_FLOATS).contiguous()
