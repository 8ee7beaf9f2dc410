"""Seeded synthetic inputs for tests and benchmarks (no dataset, no checkpoint).

Everything is generated on the CPU with ``torch.Generator`` so that the same seed gives the same
tensors in the build container and on the GPU box (same image, same torch build).  The shapes follow
the reference input dict (realestate10k_dataio.py:180-188, 456-466) and ``get_z`` output
(models.py:178-186): ``z = [(b*V,256,H/4,W/4), (b*V,256,H/2,W/2), (b*V,64,H,W)]``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

Tensor = torch.Tensor


def _rot_y(deg: float) -> Tensor:
    a = math.radians(deg)
    return torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]],
                        dtype=torch.float64)


def _rot_x(deg: float) -> Tensor:
    a = math.radians(deg)
    return torch.tensor([[1.0, 0.0, 0.0], [0.0, math.cos(a), -math.sin(a)], [0.0, math.sin(a), math.cos(a)]],
                        dtype=torch.float64)


def _pose(Rm: Tensor, t: Sequence[float]) -> Tensor:
    T = torch.eye(4, dtype=torch.float64)
    T[:3, :3] = Rm
    T[:3, 3] = torch.tensor(list(t), dtype=torch.float64)
    return T


def _so3_log(Rm: Tensor) -> Tensor:
    cos = ((torch.trace(Rm) - 1) / 2).clamp(-1, 1)
    ang = torch.acos(cos)
    if ang.abs() < 1e-12:
        return torch.zeros(3, dtype=torch.float64)
    w = torch.stack([Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1]]) / (2 * torch.sin(ang))
    return w * ang


def _so3_exp(w: Tensor) -> Tensor:
    ang = w.norm()
    if ang < 1e-12:
        return torch.eye(3, dtype=torch.float64)
    k = w / ang
    Kx = torch.tensor([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=torch.float64)
    return torch.eye(3, dtype=torch.float64) + torch.sin(ang) * Kx + (1 - torch.cos(ang)) * (Kx @ Kx)


def interpolate_pose(A: Tensor, B: Tensor, alpha: float) -> Tensor:
    """Geodesic rotation interpolation + linear translation between two cam2world matrices
    (the trajectory construction of load_video_superglue.py:83-111, without ``roma``)."""
    Rrel = A[:3, :3].T @ B[:3, :3]
    Rm = A[:3, :3] @ _so3_exp(alpha * _so3_log(Rrel))
    t = (1 - alpha) * A[:3, 3] + alpha * B[:3, 3]
    return _pose(Rm, t.tolist())


def pinhole(H: int, focal_scale: float = 0.879) -> Tensor:
    """RealEstate10K-like intrinsics in pixel units: f = 0.879*H (=225 at 256), c = H/2
    (load_video_superglue.py:465; realestate10k_dataio.py:142-145)."""
    K = torch.eye(4, dtype=torch.float64)
    K[0, 0] = K[1, 1] = focal_scale * H
    K[0, 2] = K[1, 2] = H / 2.0
    return K


def pixel_grid(H: int, W: int) -> Tensor:
    """(H*W, 2) pixel coordinates, x (column) fastest: ray index = row*W + col
    (realestate10k_dataio.py:238-245)."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing="ij")
    return torch.stack([xs, ys], dim=-1).reshape(H * W, 2)


def stereo_scene(H: int, b: int = 1, alpha: float = 0.5, baseline: float = 0.6, yaw_deg: float = -12.0,
                 uv: Optional[Tensor] = None, seed: int = 0, n_view: int = 2,
                 query_at_context: Optional[int] = None) -> Dict[str, Dict[str, Tensor]]:
    """A wide-baseline stereo pair plus one query camera on the path between them.

    Scene ``i`` of the batch perturbs the rig slightly (seeded) so batched scenes are not identical.
    ``query_at_context=k`` puts the query camera exactly on context camera ``k`` (the degenerate
    "origin at the camera" branch of epipolar.py:212-221).
    """
    g = torch.Generator().manual_seed(seed)
    K = pinhole(H)
    if uv is None:
        uv = pixel_grid(H, H)
    ctx_c2w, q_c2w = [], []
    for i in range(b):
        jit = (torch.rand(6, generator=g, dtype=torch.float64) - 0.5)
        world = _pose(_rot_y(20.0 + 10 * jit[0].item()) @ _rot_x(-8.0 + 6 * jit[1].item()),
                      [0.3 + jit[2].item(), -0.2, 1.5])
        rel = _pose(_rot_y(yaw_deg + 4 * jit[3].item()) @ _rot_x(2.0 * jit[4].item()),
                    [baseline, 0.03 + 0.05 * jit[5].item(), 0.05])
        cams = [world, world @ rel]
        if n_view == 3:
            rel3 = _pose(_rot_y(-yaw_deg * 0.6) @ _rot_x(3.0), [-0.5 * baseline, -0.04, 0.08])
            cams.append(world @ rel3)
        cams = cams[:n_view]
        if query_at_context is not None:
            q = cams[query_at_context].clone()
        elif n_view == 1:
            q = world @ _pose(_rot_y(yaw_deg * alpha), [baseline * alpha, 0.02, 0.03])
        else:
            q = interpolate_pose(cams[0], cams[1], alpha) @ _pose(_rot_x(1.5), [0.0, 0.02, -0.03])
        ctx_c2w.append(torch.stack(cams))
        q_c2w.append(q[None])
    ctx_c2w = torch.stack(ctx_c2w).float()
    q_c2w = torch.stack(q_c2w).float()
    Kf = K.float()
    R = uv.shape[0]
    return {
        "context": {
            "rgb": torch.zeros(b, n_view, H, H, 3),
            "cam2world": ctx_c2w,
            "intrinsics": Kf[None, None].expand(b, n_view, 4, 4).contiguous(),
        },
        "query": {
            "cam2world": q_c2w,
            "intrinsics": Kf[None, None].expand(b, 1, 4, 4).contiguous(),
            "uv": uv[None, None].expand(b, 1, R, 2).contiguous(),
        },
    }


def feature_maps(b: int, n_view: int, H: int, seed: int = 1,
                 channels: Sequence[int] = (256, 256, 64), strides: Sequence[int] = (4, 2, 1)) -> List[Tensor]:
    """N(0,1) stand-ins for ``get_z``'s output (NCHW), level order = [path_2, path_1, conv_map]."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(b * n_view, c, H // s, H // s, generator=g) for c, s in zip(channels, strides)]


def perturb_parameters(module: torch.nn.Module, seed: int = 0, scale: float = 0.02) -> None:
    """Adds ``scale*N(0,1)`` to every parameter in ``named_parameters`` order.

    Needed because ``phi.blocks.*.fc_1.weight`` is zero-initialised (resnet_block_fc.py:39): a fresh
    model has dead residual branches that a parity test would never exercise.
    """
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in module.named_parameters():
            p.add_(scale * torch.randn(p.shape, generator=g))


def seeded_state_dict(shapes: Dict[str, Sequence[int]], seed: int = 0) -> Dict[str, Tensor]:
    """Fan-in scaled N(0,1) weights for a name->shape table (independent of nn.Module init order).

    ``w ~ N(0, 1/fan_in)``, ``b ~ 0.1*N(0,1)``, drawn in sorted-name order from one generator, so the
    reference model and ours can be loaded with identical values via ``load_state_dict``.
    """
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        t = torch.randn(shp, generator=g)
        if name.endswith("bias"):
            t = 0.1 * t
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            t = t / math.sqrt(max(fan_in, 1))
        out[name] = t
    return out


def unposed_scene(H: int, frame: int = 38, n_poses: int = 80, uv: Optional[Tensor] = None, seed: int = 0, sf: float = 1.2):
    """Config 5's synthetic input (load_video_superglue.py:465-483): two views of UNKNOWN pose, related by the (R, t) an
    essential-matrix decomposition returns — a seeded rotation (about 14 degrees about a near-vertical axis) and a UNIT translation, the
    scale being unobservable — placed as the unposed demo places them: the first camera is the world frame, the second sits at
    inv([R | t]) with its position divided by ``sf``; the query is pose ``frame`` of ``trajectory.rotate_interpolate`` between them
    (``n_poses`` - 4 of them); intrinsics = ``pinhole(H)`` for all three.  Returns an input dict shaped like ``stereo_scene``'s (b = 1)."""
    import numpy as np
    from . import trajectory as T
    g = torch.Generator().manual_seed(seed)
    r = torch.rand(6, generator=g, dtype=torch.float64)
    axis = np.array([0.15 * (r[0].item() - 0.5), 1.0, 0.15 * (r[1].item() - 0.5)])
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(12.0 + 4.0 * r[2].item())
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * (Kx @ Kx)                 # Rodrigues
    t = np.array([-1.0, 0.08 * (r[3].item() - 0.5), 0.25 * (r[4].item() - 0.5)])
    t /= np.linalg.norm(t)                                                          # recoverPose: unit translation
    pose2 = np.eye(4)
    pose2[:3, :3], pose2[:3, 3] = R, t
    pose2 = np.linalg.inv(pose2)
    pose2[:3, 3] /= sf
    ctx = np.stack([np.eye(4), pose2])
    path = T.rotate_interpolate(ctx, n_poses)
    q = path[min(max(frame, 0), path.shape[0] - 1)]
    if uv is None:
        uv = pixel_grid(H, H)
    Kf = pinhole(H).float()
    Rn = uv.shape[0]
    return {
        "context": {"rgb": torch.zeros(1, 2, H, H, 3), "cam2world": torch.from_numpy(ctx).float()[None],
                    "intrinsics": Kf[None, None].expand(1, 2, 4, 4).contiguous()},
        "query": {"cam2world": torch.from_numpy(q).float()[None, None], "intrinsics": Kf[None, None].expand(1, 1, 4, 4).contiguous(),
                  "uv": uv[None, None].expand(1, 1, Rn, 2).contiguous()},
    }

