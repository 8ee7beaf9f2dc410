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


# ----------------------------------------------------------------------------------------------------------------------
# Evaluation items (reference RealEstate10kVis / ACIDVis, realestate10k_dataio.py:469-719, acid_dataio.py:504-): one scene ->
# (model_input, query) with the first / last (/ middle) of the first 129 frames as context and one random in-between frame as query.
# ----------------------------------------------------------------------------------------------------------------------
def parse_pose(pose_rows: np.ndarray, timestep: int) -> Camera:
    """The camera of frame ``timestep`` from the rows of a ``.mat`` pose table (one row per frame, same 19 numbers as a line of a
    camera file; timestamps are matched after rounding, realestate10k_dataio.py:95-101)."""
    mask = (np.around(pose_rows[:, :1]) == timestep)[:, 0]
    return Camera(pose_rows[mask][0])


def square_crop_img(img: np.ndarray) -> np.ndarray:
    """Centre crop to the shorter side (utils/data_util.py:116-121; for an odd difference the crop is one pixel short, as there)."""
    m = int(np.amin(img.shape[:2]))
    c = np.array(img.shape[:2]) // 2
    return img[c[0] - m // 2:c[0] + m // 2, c[1] - m // 2:c[1] + m // 2]


def _linear_coefs(n_dst: int, n_src: int):
    """Source index and the two 11-bit fixed-point weights of every destination index, as OpenCV's ``resize`` computes them for
    INTER_LINEAR on 8-bit images: centre-aligned coordinate ``(d + 0.5) * scale - 0.5`` in float32, clamped at both ends,
    weights rounded to 1/2048."""
    scale = 1.0 / (float(n_dst) / float(n_src))
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    i0 = np.floor(f).astype(np.int64)
    f = f - i0.astype(np.float32)
    lo, hi = i0 < 0, i0 >= n_src - 1
    f = np.where(lo | hi, np.float32(0), f)
    i0 = np.where(lo, 0, np.where(hi, n_src - 1, i0))
    i1 = np.minimum(i0 + 1, n_src - 1)
    w1 = np.rint(f.astype(np.float32) * np.float32(2048)).astype(np.int64)
    w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    return i0, i1, w0, w1


def resize_linear_u8(img: np.ndarray, width: int, height: int) -> np.ndarray:
    """``cv2.resize(img, (width, height))`` (INTER_LINEAR) for a uint8 H x W x C image, restated from OpenCV's fixed-point path
    (imgproc/resize.cpp: 11-bit coefficients, horizontal pass in int32, vertical pass ``((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4))
    >> 16) + 2) >> 2``).  The reference resizes its 360-line frames this way when an item is read (realestate10k_dataio.py:606-607).
    cv2 is not installed in this image, so the restatement is unpinned against cv2 itself (tests check it against float bilinear
    interpolation to one grey level)."""
    if img.dtype != np.uint8 or img.ndim != 3:
        raise ValueError("resize_linear_u8 expects a uint8 H x W x C image")
    x0, x1, a0, a1 = _linear_coefs(width, img.shape[1])
    y0, y1, b0, b1 = _linear_coefs(height, img.shape[0])
    src = img.astype(np.int64)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]               # horizontal pass: values x 2048
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


class RealEstate10kVis:
    """Evaluation dataset of the reference's eval scripts (eval_realestate10k.py:101-105, eval_acid.py): ``img_root`` holds one
    directory per scene with a ``*.npz`` of frames, ``pose_root`` is a ``.mat`` file mapping scene name -> pose table.  The
    download scripts store raw 360-line frames; like the reference (realestate10k_dataio.py:606-607) the reader resizes such a frame
    to the working size 256 x 455 (``resize_linear_u8``: OpenCV's INTER_LINEAR restated) before the centre crop to 256 x 256."""

    H, W = 256, 455

    def __init__(self, img_root, pose_root, num_ctxt_views: int, num_query_views: int = 1, query_sparsity=None, max_num_scenes=None,
                 square_crop: bool = True, augment: bool = False, lpips: bool = False):
        from scipy.io import loadmat
        if augment or query_sparsity is not None or lpips:
            raise ValueError("RealEstate10kVis here is the evaluation reader: augment / query_sparsity / lpips are training-time options")
        if num_ctxt_views not in (1, 2, 3):
            raise ValueError("More than 3 context views not supported")
        self.num_ctxt_views = num_ctxt_views
        self.all_pose = loadmat(str(pose_root))
        self.all_scenes = sorted(p for p in Path(img_root).glob("*/") if p.is_dir())
        if max_num_scenes:
            self.all_scenes = self.all_scenes[:max_num_scenes]
        self.square_crop = square_crop
        short = min(self.H, self.W)
        self.xscale, self.yscale = self.W / short, self.H / short
        if square_crop:
            ys, xs = torch.meshgrid(torch.arange(0, short), torch.arange(0, short), indexing="ij")
        else:
            ys, xs = torch.meshgrid(torch.arange(0, self.H), torch.arange(0, self.W), indexing="ij")
        self.uv = torch.stack([xs.float(), ys.float()], dim=-1).reshape(-1, 2)          # (x = column, y = row), row-major

    def __len__(self) -> int:
        return len(self.all_scenes)

    def _frame(self, data, name, pose, stamp):
        rgb = data[name]
        if rgb.shape[0] == 360:
            rgb = resize_linear_u8(np.ascontiguousarray(rgb), self.W, self.H)
        if self.square_crop:
            rgb = square_crop_img(rgb)
        cam = parse_pose(pose, stamp)
        K = unnormalize_intrinsics(cam.intrinsics, self.H, self.W)
        if self.square_crop:
            K[0, 2] = K[0, 2] / self.xscale
            K[1, 2] = K[1, 2] / self.yscale
        return rgb.astype(np.float32) / 127.5 - 1, K, cam.c2w_mat

    def __getitem__(self, idx):
        import random
        retry = lambda: self.__getitem__(random.randint(0, len(self.all_scenes) - 1))       # the reference's answer to a bad scene
        scene = self.all_scenes[idx]
        files = sorted(scene.glob("*.npz"))
        if scene.name not in self.all_pose or not files:
            return retry()
        pose = self.all_pose[scene.name]
        try:
            data = np.load(files[0])
        except Exception:
            return retry()
        names = list(data.keys())
        if len(names) <= 10:
            return retry()
        stamps = np.array([int(n.split(".")[0]) for n in names])
        order = np.argsort(stamps)
        names, stamps = np.array(names)[order], stamps[order]
        end = min(len(names) - 1, MAX_FRAMES)
        id_feat = {1: [0], 2: [0, end], 3: [0, end // 2, end]}[self.num_ctxt_views]
        candidates = [i for i in range(0, end) if np.abs(np.array(id_feat) - i).min() > 10]
        if not candidates:
            return retry()
        q = random.choice(candidates)
        rgb, K, c2w = self._frame(data, names[q], pose, stamps[q])
        query = {"rgb": torch.from_numpy(rgb.reshape(-1, 3)[None]).float(), "cam2world": torch.from_numpy(c2w[None]).float(),
                 "intrinsics": torch.from_numpy(K[None]).float(), "uv": self.uv[None].float(), "mask": 0.0}
        ctx = [self._frame(data, names[i], pose, stamps[i]) for i in id_feat]
        context = {"rgb": torch.from_numpy(np.stack([c[0] for c in ctx])).float(),
                   "cam2world": torch.from_numpy(np.stack([c[2] for c in ctx])).float(),
                   "intrinsics": torch.from_numpy(np.stack([c[1] for c in ctx])).float()}
        return {"query": query, "context": context}, query


ACIDVis = RealEstate10kVis          # acid_dataio.py:504- is the same reader over the ACID download (eval_acid.py)
