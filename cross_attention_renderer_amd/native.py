"""Python view of the one-call C ABI (include/car_hip.h: car_plan_build, car_project_maps, car_render_forward).

This is what a host WITHOUT the Python engine binds: plain device pointers in, the reference's output tensors out.  It is
used by the tests (bit-for-bit against engine.RenderEngine) and by ``CrossAttentionRenderer(..., native=True)``-style
experiments; the engine remains the default because it also covers the constructor variants."""
from __future__ import annotations

import ctypes
from typing import Dict, List

import torch
from torch import Tensor

from . import _lib
from .poses import pack_poses


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class NativeRenderer:
    """Owns the plan (packed weights), the projected maps of the current stereo pair and a workspace; all torch tensors."""

    def __init__(self, module, device):
        self.m, self.dev, self.lib = module, torch.device(device), _lib.load()
        self._keep: List[Tensor] = []
        self.plan = None
        self.gmaps = None
        self.work = None
        self._dims_key = None

    def dims(self, b: int, R: int, z: List[Tensor]) -> _lib.CarDims:
        d = _lib.CarDims()
        d.b, d.V, d.R, d.P, d.H, d.W = b, self.m.n_view, R, self.m.npoints, self.m.H, self.m.W
        d.n_levels = len(z)
        for l, t in enumerate(z):
            d.level_c[l], d.level_h[l], d.level_w[l] = t.shape[1], t.shape[2], t.shape[3]
        d.repeat_attention = int(self.m.repeat_attention)
        return d

    def _weights(self) -> _lib.CarWeights:
        w = _lib.CarWeights()
        sd = dict(self.m.named_parameters())
        self._keep = []

        def dev(name):
            t = sd[name].detach().to(device=self.dev, dtype=torch.float32)
            t = t.reshape(t.shape[0], -1).contiguous() if t.dim() > 1 else t.contiguous()
            self._keep.append(t)
            return t.data_ptr()

        for n in _lib.WEIGHT_FIELDS[0]:
            setattr(w, f"{n.replace('.', '_')}_w", dev(n + ".weight"))
            setattr(w, f"{n.replace('.', '_')}_b", dev(n + ".bias"))
        for i in range(3):
            w.phi_lin_z_w[i], w.phi_lin_z_b[i] = dev(f"phi.lin_z.{i}.weight"), dev(f"phi.lin_z.{i}.bias")
            w.phi_fc_0_w[i], w.phi_fc_0_b[i] = dev(f"phi.blocks.{i}.fc_0.weight"), dev(f"phi.blocks.{i}.fc_0.bias")
            w.phi_fc_1_w[i], w.phi_fc_1_b[i] = dev(f"phi.blocks.{i}.fc_1.weight"), dev(f"phi.blocks.{i}.fc_1.bias")
        return w

    def prepare(self, d: _lib.CarDims, z: List[Tensor]) -> None:
        """Plan (once per weights) and projected maps (once per stereo pair)."""
        lib = self.lib
        n = lib.car_plan_bytes(ctypes.byref(d))
        if n == 0:
            _lib.check(-1, "car_plan_bytes")
        self.plan = torch.empty(n // 4, device=self.dev, dtype=torch.float32)
        w = self._weights()
        _lib.check(lib.car_plan_build(ctypes.byref(d), ctypes.byref(w), self.plan.data_ptr(), _stream()), "car_plan_build")
        maps = [t.detach().to(self.dev).float().permute(0, 2, 3, 1).contiguous() for t in z]      # channel-last levels
        self.gmaps = torch.empty(lib.car_gmaps_floats(ctypes.byref(d)), device=self.dev, dtype=torch.float32)
        ptrs = (ctypes.c_void_p * len(maps))(*[t.data_ptr() for t in maps])
        _lib.check(lib.car_project_maps(ctypes.byref(d), self.plan.data_ptr(), ptrs, self.gmaps.data_ptr(), _stream()), "car_project_maps")
        torch.cuda.current_stream().synchronize()       # `maps` may be freed now

    def forward(self, inp, z: List[Tensor], poses96: Tensor = None) -> Dict[str, Tensor]:
        m, lib, dev = self.m, self.lib, self.dev
        uv = inp["query"]["uv"]
        b, R = uv.shape[0], uv.shape[2]
        d = self.dims(b, R, z)
        key = (b, tuple(t.data_ptr() for t in z))
        if self.plan is None or key != self._dims_key:
            self.prepare(d, z)
            self._dims_key = key
        nbytes = lib.car_workspace_bytes(ctypes.byref(d))
        if self.work is None or self.work.numel() * 4 < nbytes:
            self.work = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        poses = (pack_poses(inp, m.H) if poses96 is None else poses96).to(dev).float().contiguous()
        n, P = b * m.n_view, m.npoints
        f32 = dict(device=dev, dtype=torch.float32)
        out = {"rgb": torch.empty(b, 1, R, 3, **f32), "valid_mask": torch.empty(b, R, 1, **f32), "depth_ray": torch.empty(b, R, 1, **f32),
               "at_wt": torch.empty(n, R, P, **f32), "at_wt_max": torch.empty(n, R, 1, device=dev, dtype=torch.int32),
               "coords": torch.empty(n, R, 9, **f32), "pixel_val": torch.empty(n, R, P, 2, **f32)}
        uvd = uv.reshape(b, R, 2).to(dev).float().contiguous()
        steps = torch.linspace(0.0, 1.0, P).to(dev)                        # torch's own values, like the engine and the reference
        ci = _lib.CarInputs(poses.data_ptr(), uvd.data_ptr(), self.gmaps.data_ptr(), steps.data_ptr())
        co = _lib.CarOutputs(*[out[k].data_ptr() for k in ("rgb", "valid_mask", "depth_ray", "at_wt", "at_wt_max", "coords", "pixel_val")])
        _lib.check(lib.car_render_forward(ctypes.byref(d), self.plan.data_ptr(), ctypes.byref(ci), ctypes.byref(co),
                                          self.work.data_ptr(), self.work.numel() * 4, _stream()), "car_render_forward")
        out["at_wt_max"] = out["at_wt_max"].long()
        out["at_wts"] = [out["at_wt"]]
        out["uv"] = uv
        out["z"] = z
        return out
