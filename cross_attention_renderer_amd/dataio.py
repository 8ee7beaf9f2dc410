"""RealEstate10K / ACID scene reader for the render scripts.

Behaviour of the reference's ``dataset/realestate10k_dataio.py`` (``Camera`` :61-73, ``unnormalize_intrinsics`` :75-79,
``parse_pose_file`` :82-91, ``get_camera_pose`` :104-188), written against its observable results: the committed fixture
``tests/golden/dataio_scene`` holds a small scene and what the reference returns for it (``tests/golden/make_dataio_golden.py``).

On disk a scene is a directory with one ``*.npz`` whose keys are ``<timestamp>.<ext>`` -> decoded uint8 frame, and a camera
file ``<pose_dir>/<scene name>.txt``: one header line, then per frame
``timestamp fx fy cx cy _ _ r00 r01 r02 t0 r10 r11 r12 t1 r20 r21 r22 t2`` with intrinsics normalised by the image size
and a 3x4 world-to-camera matrix."""
from __future__ import annotations

from pathlib import Path
from typing import Dict

import numpy as np
import torch

# The reference hard-codes the frame geometry its intrinsics refer to (realestate10k_dataio.py:124-128): focal lengths are
# scaled by the un-cropped 256 x 456 frame, principal points by the square 256 x 256 crop.
FRAME_H, FRAME_W = 256, 456
MAX_FRAMES = 128


class Camera:
    """One line of a camera file: 4x4 normalised intrinsics, world-to-camera and camera-to-world matrices (float64)."""

    def __init__(self, entry):
        fx, fy, cx, cy = entry[1:5]
        self.intrinsics = np.array([[fx, 0.0, cx, 0.0], [0.0, fy, cy, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
        self.w2c_mat = np.eye(4)
        self.w2c_mat[:3, :] = np.asarray(entry[7:19], dtype=np.float64).reshape(3, 4)
        self.c2w_mat = np.linalg.inv(self.w2c_mat)


def unnormalize_intrinsics(intrinsics: np.ndarray, h: int, w: int) -> np.ndarray:
    out = intrinsics.copy()
    out[0] *= w
    out[1] *= h
    return out


def parse_pose_file(path) -> Dict[int, Camera]:
    """timestamp -> Camera; the first line of the file is a header (the video URL)."""
    cams: Dict[int, Camera] = {}
    with open(path, "r") as f:
        for i, line in enumerate(f):
            if i == 0:
                continue
            entry = [float(x) for x in line.split()]
            if entry:
                cams[int(entry[0])] = Camera(entry)
    return cams


def _pixel_intrinsics(cam: Camera) -> np.ndarray:
    K = unnormalize_intrinsics(cam.intrinsics, FRAME_H, FRAME_W)
    short = min(FRAME_H, FRAME_W)
    K[0, 2] = K[0, 2] / (FRAME_W / short)
    K[1, 2] = K[1, 2] / (FRAME_H / short)
    return K


def get_camera_pose(scene_path, all_pose_dir, uv: torch.Tensor, views: int = 1) -> Dict[str, Dict[str, torch.Tensor]]:
    """The model input of the trajectory render script: context = first / middle / last of the first 129 frames (by ``views``),
    queries = frames 1 .. 127 with their ground-truth images; rgb in [-1, 1]; all tensors float32 with a leading batch of 1."""
    scene_path = Path(scene_path)
    data = np.load(sorted(scene_path.glob("*.npz"))[0])
    names = list(data.keys())
    stamps = np.array([int(n.split(".")[0]) for n in names])
    order = np.argsort(stamps)
    names = [names[i] for i in order]
    stamps = stamps[order]
    cams = parse_pose_file(Path(all_pose_dir) / (scene_path.name + ".txt"))

    def frame(i):
        cam = cams[int(stamps[i])]
        return data[names[i]].astype(np.float32) / 127.5 - 1, cam.c2w_mat, _pixel_intrinsics(cam)

    n = len(names)
    n_render = min(MAX_FRAMES, n)
    last = min(n - 1, MAX_FRAMES)
    if views == 1:
        ctx_ids = [0]
    elif views == 2:
        ctx_ids = [0, last]
    elif views == 3:
        ctx_ids = [0, last // 2, last]
    else:
        raise ValueError(f"views must be 1, 2 or 3 (got {views})")

    def pack(ids):
        rgb, c2w, K = zip(*(frame(i) for i in ids)) if ids else ((), (), ())
        as_t = lambda arrs: torch.from_numpy(np.stack(arrs).astype(np.float32))[None] if arrs else torch.zeros(1, 0)
        return {"rgb": as_t(rgb), "cam2world": as_t(c2w), "intrinsics": as_t(K)}

    query = pack(list(range(1, n_render)))
    query["uv"] = uv.view(-1, 2)[None, None].expand(1, n_render - 1, -1, -1)
    return {"query": query, "context": pack(ctx_ids)}
