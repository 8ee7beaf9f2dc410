"""Render / eval plumbing shared by ``experiment_scripts/`` (reference render_realestate10k_traj.py:84-183,
eval_realestate10k.py:123-199): chunked rendering of a frame, trajectory construction, PSNR, frame output.

Everything numerically interesting happens inside ``CrossAttentionRenderer.forward``; this module only slices rays
into chunks, shards them over ranks and puts the tiles back together.
"""
from __future__ import annotations

import math
import os
import zlib
import struct
from typing import Dict, List, Optional

import torch

from . import sharding, synthetic

CHUNK_RAYS = 8192          # render_realestate10k_traj.py:96


CAMERA_KEYS = ("cam2world", "intrinsics")


def to_device(inp, device, cameras: str = "host"):
    """Moves an input dict to the device.  All three choices but the last give the strict-parity pose algebra (the reference's own
    ``torch.inverse`` / ``matmul`` on the host CPU, poses.pack_poses; the reference's fixtures reproduce to 1e-4):
      ``"host"`` (default) leaves the 4x4 camera matrices on the CPU — the engine uploads 768 bytes of records per new pose and never
                 waits for the device;
      ``"gpu"``  moves them too, as the reference's ``dict_to_gpu`` does (render_realestate10k_traj.py:85): the engine copies them back
                 for the host algebra, one small download + stream synchronisation per new pose (engine._poses);
      ``"device"`` moves them and is meant for a module with ``pose_route = "device"``: ``car_pose_setup`` on the GPU, no host work per
                 frame, equal to the host algebra to a few ulp — which the fp64 Pluecker intersection amplifies on near-parallel
                 samples."""
    if cameras not in ("host", "gpu", "device"):
        raise ValueError("cameras must be 'host', 'gpu' or 'device'")
    keep = CAMERA_KEYS if cameras == "host" else ()
    return {k: {kk: (vv.to(device) if torch.is_tensor(vv) and kk not in keep else vv) for kk, vv in v.items()} for k, v in inp.items()}


@torch.no_grad()
def render_frame(model, model_input, z, chunk_rays: int = CHUNK_RAYS, rank: int = 0, world: int = 1) -> torch.Tensor:
    """Renders every query ray of ``model_input`` in chunks (the reference's loop, render_realestate10k_traj.py:118-145).
    With world > 1 this rank renders its ray band and the tiles are all-gathered.  Returns (b, R, 5) = rgb, depth, valid."""
    n_rays = model_input["query"]["uv"].shape[2]
    shard, (s, e) = sharding.shard_query(model_input, rank, world) if world > 1 else (model_input, (0, n_rays))
    uv = shard["query"]["uv"]
    tiles = []
    for c0 in range(0, uv.shape[2], chunk_rays):
        chunk = {"context": shard["context"], "query": dict(shard["query"], uv=uv[:, :, c0:c0 + chunk_rays])}
        tiles.append(sharding.pack_tile(model(chunk, z=z)))
    tile = torch.cat(tiles, dim=1)
    return sharding.gather_rays(tile, n_rays) if world > 1 else tile


def trajectory(inp, n_frames: int) -> List[Dict]:
    """One input dict per frame of a camera path between the first and the last context camera of every scene:
    ``trajectory.linear_interpolate`` (the reference's load_video_superglue.linear_interpolate: rotation by slerp, position on the
    segment), ``n_frames`` poses including both ends."""
    from . import trajectory as T
    c2w = inp["context"]["cam2world"]
    b = c2w.shape[0]
    paths = [T.linear_interpolate(c2w[s, [0, -1]].double().cpu().numpy(), max(n_frames, 2)) for s in range(b)]
    frames = []
    for i in range(n_frames):
        q = torch.stack([torch.from_numpy(paths[s][i]).float() for s in range(b)])[:, None]
        frames.append({"context": inp["context"], "query": dict(inp["query"], cam2world=q.to(inp["query"]["cam2world"].device))})
    return frames


def psnr(img: torch.Tensor, ref: torch.Tensor) -> float:
    """mse2psnr of the reference scripts (render_realestate10k_traj.py:34-35), images in [0, 1]."""
    mse = torch.mean((img - ref) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)


def write_png(path: str, rgb: torch.Tensor) -> None:
    """(H, W, 3) float image in [-1, 1] -> 8-bit PNG (imageio is not available in this image)."""
    img = ((rgb.clamp(-1, 1) + 1) * 127.5).round().to(torch.uint8).cpu().numpy()
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def synthetic_pair(H: int, n_view: int, seed: int = 5):
    """A seeded stereo pair + feature pyramid standing in for a dataset item and ``get_z`` (no dataset / encoder here)."""
    inp = synthetic.stereo_scene(H, b=1, seed=seed, n_view=n_view)
    z = synthetic.feature_maps(1, n_view, H, seed=1)
    return inp, z
