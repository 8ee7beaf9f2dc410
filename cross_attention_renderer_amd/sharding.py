"""Multi-GPU ray sharding: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" on CPU for the tests).

Rays are independent (the only coupling in the forward is the per-ray softmax), so the renderer partitions over
pixels: rank g renders a contiguous band of rays of every scene, with the feature pyramid and the weights replicated.
The only exchange is an all-gather of the rendered tile ``[rgb(3), depth_ray, valid]`` — 20 bytes per ray, ~1.3 MB per
256x256 frame — which is latency-bound on xGMI (7 links x ~153 GB/s per GPU), so it is issued on a side stream and
overlapped with the next frame's kernels.  The reference never shards rays: its ``--gpus N`` eval/render spawns N
identical replicas (eval_realestate10k.py:95-99, 211-214).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

TILE_CHANNELS = 5          # rgb(3), depth_ray, valid_mask


def ray_band(n_rays: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of the rays rank ``rank`` renders: contiguous, balanced to within one ray."""
    base, extra = divmod(n_rays, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_query(inp, rank: int, world: int):
    """Input dict restricted to this rank's ray band (shares every other tensor; the caller's dict is not mutated)."""
    uv = inp["query"]["uv"]
    s, e = ray_band(uv.shape[2], rank, world)
    q = dict(inp["query"], uv=uv[:, :, s:e].contiguous())
    return {"context": inp["context"], "query": q}, (s, e)


def pack_tile(out) -> torch.Tensor:
    """forward() output dict -> (b, R_local, 5) tile [rgb, depth_ray, valid_mask]."""
    return torch.cat([out["rgb"][:, 0], out["depth_ray"], out["valid_mask"]], dim=-1).contiguous()


def gather_rays(tile: torch.Tensor, n_rays: int, group=None) -> torch.Tensor:
    """All-gather the per-rank tiles (b, R_g, 5) of a ray-sharded render into the full (b, n_rays, 5) on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, _, c = tile.shape
    # bands differ by at most one ray: pad to the largest so one fixed-size all-gather suffices
    rmax = -(-n_rays // world)
    padded = tile.new_zeros(b, rmax, c)
    padded[:, : tile.shape[1]] = tile
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    out = tile.new_empty(b, n_rays, c)
    for g, p in enumerate(parts):
        s, e = ray_band(n_rays, g, world)
        out[:, s:e] = p[:, : e - s]
    return out


def assemble_tiles(gathered: torch.Tensor, b: int, n_rays: int) -> torch.Tensor:
    """TileGather's buffer (world, b * R_g, C) — every rank's (b, R_g, C) tile of a batch of b scenes, flattened to rows — as the whole
    (b, n_rays, C) frames: rank g holds rays [g R_g, (g + 1) R_g) of EVERY scene (config 3: twelve scenes, rays banded over the ranks;
    the bands must be equal: n_rays % world == 0)."""
    world = gathered.shape[0]
    if n_rays % world or gathered.shape[1] != b * (n_rays // world):
        raise ValueError(f"assemble_tiles: {tuple(gathered.shape)} is not {world} equal bands of {b} x {n_rays} rays")
    return gathered.view(world, b, n_rays // world, -1).permute(1, 0, 2, 3).reshape(b, n_rays, -1)


class TileGather:
    """Overlapped all-gather of equally sized per-rank tiles (R, C) -> (world, R, C).

    ``__call__(tile)`` snapshots the tile and starts the collective on a side stream; compute on the caller's stream
    continues immediately.  ``wait()`` (or the next call) joins it.  With RCCL the message is tiny (latency-bound), so
    hiding it under the next frame's kernels is what matters, not bandwidth.
    """

    def __init__(self, world: int, rows: int, channels: int, device, group=None):
        self.group = group
        self.out = torch.empty(world, rows, channels, device=device)
        self.stage = torch.empty(rows, channels, device=device)
        self.cuda = torch.device(device).type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if self.cuda else None
        self.work = None
        self.copied = None

    def __call__(self, tile: torch.Tensor) -> None:
        self.wait()
        parts = list(self.out.unbind(0))
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stage.copy_(tile)
                self.copied = torch.cuda.Event()
                self.copied.record(self.stream)
                if dist.get_backend(self.group) == "nccl":
                    # one RCCL all-gather straight into the (world, rows, channels) buffer: no per-rank output copies
                    self.work = dist.all_gather_into_tensor(self.out, self.stage, group=self.group, async_op=True)
                else:
                    self.work = dist.all_gather(parts, self.stage, group=self.group, async_op=True)
            # the caller may overwrite `tile` as soon as the snapshot is taken
            torch.cuda.current_stream().wait_event(self.copied)
        else:
            self.stage.copy_(tile)
            self.work = dist.all_gather(parts, self.stage, group=self.group, async_op=True)

    def wait(self) -> Optional[torch.Tensor]:
        if self.work is not None:
            self.work.wait()
            self.work = None
            if self.cuda:
                torch.cuda.current_stream().wait_stream(self.stream)
        return self.out
