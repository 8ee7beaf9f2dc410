"""Camera trajectories and the unposed-pair model input of the render scripts.

Behaviour of the reference's ``dataset/load_video_superglue.py``: ``linear_interpolate`` (:33-54), ``make_circle`` (:57-82),
``rotate_interpolate`` (:85-111) and the input dict assembled by its ``get_camera_pose`` (:462-506) once the relative pose (R, t)
of the second image is known.  The reference gets (R, t) from a SuperPoint + SuperGlue matcher followed by
``cv2.findEssentialMat`` / ``recoverPose`` (:114-139, 421-460); that matcher's weights are not part of the reference tree and the
estimate is made once per image pair, so here (R, t) is an input.  Rotations are interpolated with ``roma.rotmat_slerp`` in the
reference (roma is not installed here): unit-quaternion spherical interpolation along the shorter arc, restated below.
tests/test_trajectory.py replays vectors produced by the reference's own functions (tests/golden/make_trajectory_golden.py)."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch


def _quat_from_rotmat(R: np.ndarray) -> np.ndarray:
    """Unit quaternion (x, y, z, w) of a rotation matrix (Shepperd's method: the largest of the four candidates as pivot)."""
    m00, m11, m22 = R[0, 0], R[1, 1], R[2, 2]
    cand = np.array([1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22, 1 + m00 + m11 + m22])
    i = int(np.argmax(cand))
    if i == 3:
        q = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], cand[3]])
    elif i == 0:
        q = np.array([cand[0], R[0, 1] + R[1, 0], R[0, 2] + R[2, 0], R[2, 1] - R[1, 2]])
    elif i == 1:
        q = np.array([R[0, 1] + R[1, 0], cand[1], R[1, 2] + R[2, 1], R[0, 2] - R[2, 0]])
    else:
        q = np.array([R[0, 2] + R[2, 0], R[1, 2] + R[2, 1], cand[2], R[1, 0] - R[0, 1]])
    return q / np.linalg.norm(q)


def _rotmat_from_quat(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rotmat_slerp(R0: np.ndarray, R1: np.ndarray, steps: np.ndarray) -> np.ndarray:
    """Spherical linear interpolation between two rotations at the parameters ``steps`` (0 -> R0, 1 -> R1), shorter arc.  The
    reference hands float32 copies of the matrices to roma and takes float32 rotations back (load_video_superglue.py:44-48): the
    inputs are rounded to float32 here too, the arithmetic is float64, the result is rounded to float32 precision."""
    R0, R1 = np.asarray(R0, dtype=np.float32).astype(np.float64), np.asarray(R1, dtype=np.float32).astype(np.float64)
    q0, q1 = _quat_from_rotmat(R0), _quat_from_rotmat(R1)
    d = float(np.dot(q0, q1))
    if d < 0:
        q1, d = -q1, -d
    omega = np.arccos(min(d, 1.0))
    out = []
    for s in np.asarray(steps, dtype=np.float64):
        if omega < 1e-8:
            q = (1 - s) * q0 + s * q1
        else:
            q = (np.sin((1 - s) * omega) * q0 + np.sin(s * omega) * q1) / np.sin(omega)
        out.append(_rotmat_from_quat(q / np.linalg.norm(q)))
    return np.stack(out).astype(np.float32).astype(np.float64)


def _steps(n: int) -> np.ndarray:
    return torch.linspace(0, 1, n).numpy()               # the reference's interval (float32 values)


def linear_interpolate(poses: np.ndarray, n: int) -> np.ndarray:
    """n camera-to-world matrices from poses[0] to poses[1]: rotation by slerp, position on the straight segment."""
    t0, t1 = poses[0][:3, -1], poses[1][:3, -1]
    s = _steps(n)
    out = np.tile(np.eye(4)[None], (n, 1, 1))
    out[:, :3, :3] = rotmat_slerp(poses[0][:3, :3], poses[1][:3, :3], s)
    out[:, :3, -1] = t0[None, :] + (t1[None, :] - t0[None, :]) * s[:, None]
    return out


def make_circle(direction: np.ndarray, n: int, radius: float = 0.03) -> np.ndarray:
    """n points spiralling twice around the unit segment along ``direction`` (a helix of the given radius whose axis runs from the
    origin to ``direction``), in a frame whose first axis is the x axis made orthogonal to ``direction``."""
    ang = np.linspace(0, 4 * np.pi, n)
    coord = np.stack([np.cos(ang) * radius, np.sin(ang) * radius, np.linspace(0, 1, n)], axis=-1)
    a1 = np.array([1.0, 0.0, 0.0])
    a1 = a1 - (direction * a1).sum() * direction
    a1 = a1 / np.linalg.norm(a1)
    a2 = np.cross(a1, direction)
    rot = np.stack([a1, a2, direction], axis=1)
    return (rot @ coord[:, :, None])[:, :, 0]


def rotate_interpolate(poses: np.ndarray, n: int) -> np.ndarray:
    """The unposed demo's trajectory: rotations by slerp between the two context cameras, positions on a helix (radius 0.05 of the
    baseline) around the line from the FIRST camera's frame origin along the baseline direction — the start position itself is not
    added, as in the reference — and the first and last two poses dropped: n - 4 matrices."""
    t0, t1 = poses[0][:3, -1], poses[1][:3, -1]
    d = t1 - t0
    norm = np.linalg.norm(d)
    out = np.tile(np.eye(4)[None], (n, 1, 1))
    out[:, :3, :3] = rotmat_slerp(poses[0][:3, :3], poses[1][:3, :3], _steps(n))
    out[:, :3, -1] = make_circle(d / norm, n, radius=0.05) * norm
    return out[2:-2]


UNPOSED_K = np.array([[225.0, 0.0, 128.0, 0.0], [0.0, 225.0, 128.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])


def center_crop_square(im: np.ndarray) -> np.ndarray:
    """Columns (w - h) // 2 ... of a landscape frame, as the reference slices them (im[:, offset:-offset])."""
    h, w = im.shape[:2]
    off = (w - h) // 2
    return im[:, off:-off] if off > 0 else im


def unposed_pair_input(image0: np.ndarray, image1: np.ndarray, R: np.ndarray, t: np.ndarray, uv: torch.Tensor, sf: float = 1.2,
                       n_poses: int = 80) -> Dict[str, Dict[str, torch.Tensor]]:
    """Model input for two images of unknown pose, given the relative pose (R, t) of the second camera (x_2 = R x_1 + t, as
    ``cv2.recoverPose`` returns it).  image0 / image1: (256, 256, 3) floats in [0, 1], already cropped and resized.  The first
    camera is the world frame, the second sits at inv([R | t]) with its position divided by ``sf`` (the essential matrix fixes the
    translation only up to scale), both with the fixed intrinsics of a 256 x 256 RealEstate10K crop; the queries are
    ``rotate_interpolate`` of the two, each carrying the first image as its (placeholder) target."""
    ctx_rgb = np.stack([(image0 - 0.5) * 2, (image1 - 0.5) * 2], axis=0)
    pose2 = np.eye(4)
    pose2[:3, :3] = R
    pose2[:3, -1] = np.asarray(t).reshape(3)
    pose2 = np.linalg.inv(pose2)
    pose2[:3, -1] = pose2[:3, -1] / sf
    ctx_c2w = np.stack([np.eye(4), pose2], axis=0)
    q_c2w = rotate_interpolate(ctx_c2w, n_poses)
    nq = q_c2w.shape[0]
    f32 = lambda a: torch.Tensor(a)[None].float()
    query = {"rgb": f32(np.tile(ctx_rgb[:1], (nq, 1, 1, 1))), "cam2world": f32(q_c2w), "intrinsics": f32(np.tile(UNPOSED_K[None], (nq, 1, 1))),
             "uv": uv.view(-1, 2)[None, None].expand(1, nq, -1, -1)}
    ctx = {"rgb": f32(ctx_rgb), "cam2world": f32(ctx_c2w), "intrinsics": f32(np.tile(UNPOSED_K[None], (2, 1, 1)))}
    return {"query": query, "context": ctx}
