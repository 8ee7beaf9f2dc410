"""Host side of the render forward: what ``CrossAttentionRenderer.forward`` runs.

Every arithmetic step of the reference forward (models.py:206-621) is a HIP kernel of ``libcar_hip.so`` reached through the
C ABI (``include/car_hip.h``); PyTorch is used for device memory, the stream handle and the (reference-identical) 4x4 pose
algebra on the host.  There is no CPU or eager-PyTorch fallback: tensors must live on a ROCm device and the library must load.

Two routes:
  * the reference's default configuration (n_view = 2, three pyramid levels, 576 channels; epipolar sampling or ``no_sample``'s depth
    sampling, with or without the second attention round) goes through the
    ONE-CALL C ABI — ``car_plan_build`` once per set of weights, ``car_project_maps`` once per stereo pair,
    ``car_render_forward`` per batch of rays (csrc/car_render.hip: weight packing and the launch sequence are C++).  This
    module only sizes the calls: rays (and, if need be, scenes) are chunked so that the per-call workspace fits the free device
    memory (rays and scenes are independent, so this changes nothing);
  * the other constructor variants (n_view 1 / 3, no_latent_concat, other widths) are sequenced here stage by stage —
    also the A/B partner of the first route in the tests.

Stage map:
  a3 poses.pack_poses (host, torch.inverse like the reference)          a11-a13, a15, a17  car_linear (fp32 MFMA)
  a4-a6 car_ray_setup / car_sample_setup                                  a14-a16            car_attend
  a7/a10 car_gather_bilinear                                              a18                car_finalize
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import torch

from . import _lib
from .poses import pack_poses

Tensor = torch.Tensor

RELU_IN, RELU_OUT, ACCUM, NO_GLDS = 1, 2, 4, 8
PLACE_PLAIN, PLACE_OWN, PLACE_OTHER2 = 0, 1, 2


def _ptr(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class PackedLinear:
    """A linear / 1x1-conv layer re-laid out for the MFMA kernel (``car_linear_pack``)."""

    def __init__(self, weight: Tensor, bias: Optional[Tensor], device, name: str = ""):
        self.name = name
        lib = _lib.load()
        w = weight.detach().reshape(weight.shape[0], -1).to(device=device, dtype=torch.float32).contiguous()
        self.N, self.K = int(w.shape[0]), int(w.shape[1])
        bdev = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        n = lib.car_linear_packed_floats(self.K, self.N)
        self.packed = torch.empty(n, device=device, dtype=torch.float32)
        _lib.check(lib.car_linear_pack(_ptr(w), self.K, _ptr(bdev), self.K, self.N, _ptr(self.packed), _stream()),
                   "car_linear_pack")
        # the same layer for the split-fp16 kernel (car_linear_x3: 5x the fp32 pipe's ceiling): layers wide enough to matter, whose
        # outputs fill whole 32-channel tiles; the narrow ones (point embeddings, the 16-wide query input, rgb) stay on car_linear
        self.x3 = None
        if self.N % 32 == 0 and self.K >= 64:
            tiles = torch.empty(lib.car_linear_x3_packed_floats(self.K, self.N), device=device, dtype=torch.float32)
            _lib.check(lib.car_linear_x3_pack(_ptr(w), self.K, self.K, self.N, _ptr(tiles), _stream()), "car_linear_x3_pack")
            self.x3 = (tiles, bdev)


class RenderEngine:
    """Per-module state of the HIP path: packed weights (re-packed when the parameters change) and the
    channel-last copies of the last feature pyramid."""

    def __init__(self, module):
        self.m = module
        self.lib = _lib.load()
        self._packed: Dict[str, PackedLinear] = {}
        self._packed_key = None
        self._maps_key = None
        self._maps: List[Tensor] = []
        self._maps_src = None          # the z tensors the channel-last copies were made from (kept alive, see _channel_last)
        self._steps: Dict[tuple, Tensor] = {}
        self.linear_flags = 0          # tests may set NO_GLDS for A/B
        self.linear_x3 = True          # stage entries: wide layers on the split-fp16 path (car_linear_x3); False = all on the fp32 pipe
        self.linear_x3_min_rows = 2048 # below this many rows a launch is latency bound either way (a training step has 2304 rays: 16.2 -> 15.8 ms)
        self.pose_records = None       # tests: (b*V, 96) CarPose records to use instead of the host pose algebra
        # "host": the reference's torch.inverse on the CPU wherever the cameras live (strict parity; cameras on the GPU cost one small
        # download per new pose); "device": car_pose_setup when the cameras are on the GPU (no host round trip, budgeted parity)
        # -> the module's ``pose_route`` attribute (scripts set it before the first forward creates the engine)
        self.last_pose_sync_ms = 0.0
        # first point-MLP layer as a gather over per-texel pre-projected maps (csrc/car_encode.hip) instead of a
        # K=579 GEMM per sample; False selects the literal gather -> GEMM pipeline (A/B and stage tests)
        self.project_maps = True
        # the one-call C ABI with the fused per-sample kernel (needs V == 2, three pyramid levels, C == 576); False keeps the
        # stage-by-stage pipeline of this module (A/B and stage tests)
        self.fuse_samples = True
        self.fuse_round2 = True        # staged route: round-2 per-sample layer + logits in one kernel (csrc/car_round2.hip)
        # one-call route: the first attention round folds the fused kernel's per-step-group partial sums (default); False = it streams
        # the rows of e as in rounds 1-4 (CAR_PHASE_ROWS_FIRST_ROUND: A/B measurements and tests)
        self.first_round_parts = True
        self.wgrad_fp32 = False        # training: True keeps the wide layers' weight gradients on the fp32 matrix pipe (CAR_WGRAD_FP32) instead of bf16 x 3
        self.fuse_kq = True            # staged route: key / query chains and the first round's logits in one kernel (car_key_query_logits); False = five launches (A/B)
        self._kq = None
        self._kq_key = None
        # three-view exchange, first + second layer: "rows" = the fused per-sample kernel's source pass over the rows (car_fused_rows, default);
        # True = the gather-fed linear kernel (car_lattice_encode_linear); False = two launches (A/B partners)
        self.fuse_exchange = "rows"
        self._xpack = None
        self._xpack_key = None
        # sizing of the one-call route (tests shrink them to force several calls)
        self.max_workspace_bytes: Optional[int] = None     # None: 85 % of the free device memory
        self.max_level_bytes: Optional[int] = None         # tests: lattice bytes of one call (forces scene groups); None = no limit
        self.max_pair_bytes: Optional[int] = None          # tests: bytes of one lattice buffer (forces lattice groups, each projected on its own)
        self.last_pair_groups = 1      # number of lattice buffers (car_project_maps calls' scene groups) of the last forward
        self.last_calls = 0            # number of car_render_forward calls the last forward was split into
        self._round2_key = None
        self._round2 = None
        self._pf = None                # prefetch(): announced stereo pairs (key -> channel-last pyramid + lattice), projected on a side stream
        self._pf_stream = None
        self.prefetch_side_stream = True   # False (A/B): prefetch() projects on the launch stream itself
        self._pose_key = None
        self._pose_dev = None
        self._pose_src = None
        self._gmaps_key = None
        self._gmaps: List[Tensor] = []
        self._wpt: Optional[Tensor] = None
        self._plan_key = None
        self._plan: Optional[Tensor] = None
        self._plan_keep: List[Tensor] = []
        self._pair_key = None
        self._xlat = None              # merged lattice of the three-view exchange (car_merge_lattice), cached like the projected maps
        self._xlat_key = None
        self._pair: Optional[Tensor] = None
        self._work: Optional[Tensor] = None

    @property
    def pose_route(self) -> str:
        return getattr(self.m, "pose_route", "host")

    @pose_route.setter
    def pose_route(self, route: str) -> None:
        if route not in ("host", "device"):
            raise ValueError("pose_route must be 'host' or 'device'")
        self.m.pose_route = route

    # ------------------------------------------------------------------ weights
    def _weights(self, device) -> Dict[str, PackedLinear]:
        m = self.m
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in m.parameters())
        if key == self._packed_key:
            return self._packed
        sd = {k: v for k, v in m.named_parameters()}
        pk: Dict[str, PackedLinear] = {}

        def add(name, w=None, b=None, use_bias=True):
            W = sd[name + ".weight"] if w is None else w
            B = (sd[name + ".bias"] if b is None else b) if use_bias else None
            pk[name] = PackedLinear(W, B, device, name)

        if m.n_view > 1 and not m.no_latent_concat:
            add("query_encode_latent")
            add("query_encode_latent_2")
        elif not m.no_latent_concat:
            add("update_val_merge")
        for n in ("latent_value", "key_map", "key_map_2", "query_embed", "query_embed_2", "encode_latent",
                  "query_repeat_embed_2", "phi.lin_in", "phi.lin_out"):
            add(n)
        if m.n_view == 3 and not m.no_latent_concat:
            # inference keeps e in component-major order [S, 3, C/2] (what query_encode_latent_2 writes) instead of the reference's channel-
            # major interleave e[s, 3 ch + k] (models.py:446): the layers that read e get their input columns permuted once instead
            for n in ("latent_value", "key_map"):
                w = sd[n + ".weight"].reshape(sd[n + ".weight"].shape[0], -1)
                pk[n + ".kmajor"] = PackedLinear(w.view(w.shape[0], -1, 3).permute(0, 2, 1).reshape(w.shape[0], -1), sd[n + ".bias"], device, n + ".kmajor")
        wr = sd["query_repeat_embed.weight"].reshape(128, -1)
        pk["query_repeat_embed.h"] = PackedLinear(wr[:, :128], None, device, "query_repeat_embed.h")           # z_embed half, per ray
        pk["query_repeat_embed.g"] = PackedLinear(wr[:, 128:], sd["query_repeat_embed.bias"], device, "query_repeat_embed.g")  # local_coords half
        for i in range(m.phi.n_blocks):
            add(f"phi.lin_z.{i}")
            add(f"phi.blocks.{i}.fc_0")
            add(f"phi.blocks.{i}.fc_1")
        self._packed, self._packed_key = pk, key
        return pk

    # ------------------------------------------------------------------ feature maps
    def _channel_last(self, z: List[Tensor]) -> List[Tensor]:
        """Channel-last copies of the pyramid, cached on the identity/version of the z tensors.  The tensors themselves are kept
        next to the key: a freed z would hand its address (and _version 0) to the next scene's pyramid of the same shape, and
        the stale maps would be rendered."""
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in z)
        if key != self._maps_key or self._maps_src is None or any(a is not b_ for a, b_ in zip(self._maps_src, z)):
            pf = self._pf.get(key) if self._pf else None
            if pf is not None and all(a is b_ for a, b_ in zip(pf["src"], z)):
                # this pyramid was announced by prefetch(): its channel-last copies and its lattice are (being) made on the side stream.
                # The launch stream waits for them here — at the latest possible point — and takes them over
                del self._pf[key]
                torch.cuda.current_stream().wait_event(pf["done"])
                self._maps, self._maps_key, self._maps_src = pf["maps"], key, list(z)
                self._pair = None
                self._pair, self._pair_key = pf["pair"], (key, pf["plan_key"], 0, pf["b"])
                return self._maps
            self._maps = [self._as_channel_last(t) for t in z]
            self._maps_key = key
            self._maps_src = list(z)
        return self._maps

    @staticmethod
    def _as_channel_last(t: Tensor) -> Tensor:
        """[n, C, H, W] -> [n, H, W, C] rows for the kernels: a VIEW when the level already lies channel-last in memory (torch.channels_last:
        what an encoder run in that memory format returns, and what the training scripts allocate) — no copy; a channel-last copy
        otherwise (the reference's NCHW pyramid)."""
        tt = t.detach()
        if tt.dtype == torch.float32:
            v = tt.permute(0, 2, 3, 1)
            if v.is_contiguous():
                return v
        return tt.float().permute(0, 2, 3, 1).contiguous()

    def prefetch(self, z: List[Tensor]) -> bool:
        """Announces the NEXT stereo pair's pyramid while the current frame is still to be rendered (the eval loop knows its next batch:
        eval_realestate10k.py:142-161): the channel-last copies and car_project_maps of ``z`` run on a side stream, ordered behind
        everything queued so far (so behind the ``get_z`` that made ``z``) and beside the render that follows on the launch stream; the
        forward that later receives these very tensors waits for the side stream's event and finds its lattice ready.  Needs a plan (one
        forward with the current weights) and room for a second lattice; returns False — and does nothing — otherwise, or when ``z`` is the
        pyramid already in place.  Same kernels, same arguments, same results as the projection inside forward."""
        m = self.m
        if (self._plan is None or not self.fuse_samples or not self.project_maps or m.n_view != 2 or len(z) != 3 or z[0].device.type != "cuda"
                or sum(t.shape[1] for t in z) != 576 or not self._common_lattice(z)):
            return False
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in z)
        if key == self._maps_key and self._pair is not None:
            return False
        if self._pf is None:
            self._pf = {}
        if key in self._pf and all(a is b_ for a, b_ in zip(self._pf[key]["src"], z)):
            return True
        dev = z[0].device
        V = m.n_view
        b = z[0].shape[0] // V
        d = self._dims(b, 48, z)
        if d.P != self._plan_key[1] or tuple(d.level_c[:3]) != self._plan_key[2]:
            return False
        need = 4 * self.lib.car_gmaps_floats(ctypes.byref(d))
        if need == 0 or not self._lattice_fits(b, 48, z) or need > self._free_budget(dev) // 3:
            return False                                             # another lattice must leave the workspace its room
        with torch.cuda.device(dev):
            if self._pf_stream is None:
                self._pf_stream = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream()
            side = self._pf_stream if self.prefetch_side_stream else main
            # the loop announces pair i + 1 BEFORE it renders pair i (announced one step earlier and not taken over yet): two announced pairs
            # may wait; an older one nobody came for is dropped — once the side stream is done with it (its buffers go back to the launch
            # stream's pool)
            while len(self._pf) >= 2:
                old = self._pf.pop(next(iter(self._pf)))
                main.wait_event(old["done"])
            # buffers come from the launch stream's pool; the side stream starts behind everything queued there so far (the allocator's
            # reuse of a freed block is ordered on the launch stream) and the tensors are handed back to it through the event
            maps = [torch.empty(t.shape[0], t.shape[2], t.shape[3], t.shape[1], device=dev, dtype=torch.float32) for t in z]
            pair = torch.empty(need // 4, device=dev, dtype=torch.float32)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for dst, t in zip(maps, z):
                    dst.copy_(t.detach().permute(0, 2, 3, 1))
                ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in maps])
                _lib.check(self.lib.car_project_maps(ctypes.byref(d), _ptr(self._plan), ptrs, _ptr(pair), ctypes.c_void_p(side.cuda_stream)),
                           "car_project_maps")
                done = torch.cuda.Event()
                done.record(side)
            self._pf[key] = {"key": key, "src": list(z), "maps": maps, "pair": pair, "done": done, "plan_key": self._plan_key, "b": b}
        return True

    def drop_prefetched(self) -> None:
        """Forgets every announced pair (after the side stream is done with them)."""
        if self._pf:
            for old in self._pf.values():
                torch.cuda.current_stream().wait_event(old["done"])
        self._pf = None

    def _projected_maps(self, maps: List[Tensor], device):
        """G_l = query_encode_latent.weight[:, ch_l] F_l per pyramid level (channel-last, C wide), plus the [C,4]
        table (W1[:, C:C+3], b1).  Recomputed only when the pyramid or the layer's parameters change."""
        m = self.m
        w1, b1 = m.query_encode_latent.weight, m.query_encode_latent.bias
        key = (self._maps_key, w1.data_ptr(), w1._version, b1.data_ptr(), b1._version, str(device))
        if key == self._gmaps_key:
            return self._gmaps, self._wpt
        C = w1.shape[0]
        w = w1.detach().reshape(C, -1).to(device=device, dtype=torch.float32)
        gm, off = [], 0
        for t in maps:
            n, Hl, Wl, Cl = t.shape
            layer = PackedLinear(w[:, off:off + Cl].contiguous(), None, device, "project_maps")
            g = torch.empty(n, Hl, Wl, C, device=device, dtype=torch.float32)
            self.linear(t, Cl, layer, g, C, n * Hl * Wl)
            gm.append(g)
            off += Cl
        wpt = torch.cat([w[:, off:off + 3], b1.detach().to(device=device, dtype=torch.float32)[:, None]], dim=1).contiguous()
        self._gmaps, self._wpt, self._gmaps_key = gm, wpt, key
        return gm, wpt

    def gather_encode(self, gmaps: List[Tensor], wpt: Tensor, pixel_val: Tensor, grid_in: Tensor, ptenc: Tensor,
                      V: int, pts: int, out: Tensor, ld_out: int):
        L = len(gmaps)
        ptrs = (ctypes.c_void_p * L)(*[g.data_ptr() for g in gmaps])
        hs = (ctypes.c_int * L)(*[g.shape[1] for g in gmaps])
        ws = (ctypes.c_int * L)(*[g.shape[2] for g in gmaps])
        _lib.check(self.lib.car_gather_encode(ptrs, hs, ws, L, gmaps[0].shape[3], _ptr(pixel_val), _ptr(grid_in),
                                              _ptr(ptenc), _ptr(wpt), gmaps[0].shape[0], V, pts, _ptr(out), ld_out,
                                              _stream()), "car_gather_encode")

    def _poses(self, inp, H: int, n: int, dev) -> Tensor:
        """Device pose records (``struct CarPose``) for this input (models.py:207-211, 285-286).

        ``pose_route == "host"`` (the default, the strict-parity route): the reference's own ``torch.inverse`` / ``matmul`` calls on
        the host CPU (poses.pack_poses) wherever the four camera tensors live.  Cameras that arrive on the GPU — the reference's
        scripts move the whole input dict there (render_realestate10k_traj.py:85) — are copied to the host first: ONE pinned
        download of the 4x4 matrices (a few hundred bytes; the stream is synchronised once, which the reference does on every
        call anyway, models.py:570), then one pinned upload of the records.  Cached on the identity/version of the four tensors, so a
        frame rendered in chunks pays once.  ``last_pose_sync_ms`` holds the host time of the last such round trip.
        ``pose_route == "device"`` (opt-in; ``--cameras device`` of the scripts): ``car_pose_setup`` on the device when the cameras
        are there — no host round trip, but fp64 Gauss-Jordan instead of LAPACK's fp32: the matrices differ in the last ulp, which
        the fp64 Pluecker intersection amplifies on the few samples whose pixel ray is nearly parallel to the query ray
         — as it does between two LAPACK builds."""
        if self.pose_records is not None:
            poses = self.pose_records.float().contiguous()
            if tuple(poses.shape) != (n, 96):
                raise ValueError(f"pose records must have shape ({n}, 96)")
            return poses.to(dev, non_blocking=True)
        if self.pose_route not in ("host", "device"):
            raise ValueError("RenderEngine.pose_route must be 'host' or 'device'")
        ts = (inp["context"]["cam2world"], inp["context"]["intrinsics"], inp["query"]["cam2world"], inp["query"]["intrinsics"])
        on_gpu = all(t.is_cuda for t in ts)
        if self.pose_route == "device" and on_gpu:
            b, V = ts[0].shape[:2]
            c2w, Kc, c2w_q, Kq = [t.detach().to(device=dev, dtype=torch.float32).contiguous() for t in ts]
            poses = torch.empty(n, 96, device=dev, dtype=torch.float32)
            _lib.check(self.lib.car_pose_setup(_ptr(c2w), _ptr(c2w_q), _ptr(Kc), _ptr(Kq), b, V, H, _ptr(poses), _stream()),
                       "car_pose_setup")
            return poses
        key = tuple((t.data_ptr(), t._version, tuple(t.shape), str(t.device)) for t in ts) + (H, str(dev))
        if key != self._pose_key:
            src = inp
            if any(t.is_cuda for t in ts):
                import time
                t0 = time.perf_counter()
                src = self._cameras_to_host(ts)
                self.last_pose_sync_ms = (time.perf_counter() - t0) * 1e3
            # pinned staging: the 768-byte upload is queued behind the previous frame's kernels instead of waiting for them
            self._pose_dev = pack_poses(src, H).pin_memory().to(dev, non_blocking=True)
            self._pose_key = key
            self._pose_src = ts                      # keep the tensors alive so data_ptr cannot be recycled
        return self._pose_dev

    @staticmethod
    def _cameras_to_host(ts):
        """The four camera tensors as CPU tensors in the input dict's shape, through ONE device-to-host copy (tensors already on the
        host are passed through)."""
        gpu = [t for t in ts if t.is_cuda]
        flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in gpu])
        host = torch.empty(flat.numel(), dtype=torch.float32, pin_memory=True)
        host.copy_(flat, non_blocking=True)
        torch.cuda.current_stream(flat.device).synchronize()
        out, off = [], 0
        for t in ts:
            if t.is_cuda:
                out.append(host[off:off + t.numel()].view(t.shape).clone())
                off += t.numel()
            else:
                out.append(t)
        return {"context": {"cam2world": out[0], "intrinsics": out[1]}, "query": {"cam2world": out[2], "intrinsics": out[3]}}

    def _linspace(self, a: float, b_: float, P: int, device) -> Tensor:
        k = (a, b_, P, str(device))
        if k not in self._steps:
            self._steps[k] = torch.linspace(a, b_, P).to(device)      # CPU linspace, like the reference's values
        return self._steps[k]

    def _round2_weights(self, device):
        """query_repeat_embed[:, 128:] and query_repeat_embed_2 packed for csrc/car_round2.hip (device-side packer of the C ABI)."""
        m = self.m
        ps = (m.query_repeat_embed.weight, m.query_repeat_embed.bias, m.query_repeat_embed_2.weight, m.query_repeat_embed_2.bias)
        key = tuple((t.data_ptr(), t._version) for t in ps) + (str(device),)
        if key != self._round2_key:
            lib = self.lib
            f = [t.detach().to(device=device, dtype=torch.float32).reshape(t.shape[0], -1).contiguous() for t in ps]
            w = torch.empty(lib.car_round2_packed_floats(), device=device, dtype=torch.float32)
            bz = torch.empty(lib.car_round2_bias_floats(), device=device, dtype=torch.float32)
            _lib.check(lib.car_round2_pack(_ptr(f[0]), _ptr(f[1]), _ptr(f[2]), _ptr(f[3]), _ptr(w), _ptr(bz), _stream()), "car_round2_pack")
            self._round2, self._round2_key = (w, bz), key
        return self._round2

    def _exchange_pack(self, device):
        """query_encode_latent (its point columns and bias) and query_encode_latent_2 packed for car_fused_rows (car_fused_pack_rows)."""
        m = self.m
        ps = (m.query_encode_latent.weight, m.query_encode_latent.bias, m.query_encode_latent_2.weight, m.query_encode_latent_2.bias)
        key = tuple((t.data_ptr(), t._version) for t in ps) + (str(device),)
        if key != self._xpack_key:
            lib = self.lib
            f = [t.detach().to(device=device, dtype=torch.float32).reshape(t.shape[0], -1).contiguous() if t.dim() > 1
                 else t.detach().to(device=device, dtype=torch.float32).contiguous() for t in ps]
            blob = torch.zeros(lib.car_fused_blob_floats(), device=device, dtype=torch.float32)
            bias = torch.empty(lib.car_fused_bias_floats(), device=device, dtype=torch.float32)
            wpt = torch.empty(576 * 4, device=device, dtype=torch.float32)
            _lib.check(lib.car_fused_pack_rows(*[_ptr(t) for t in f], _ptr(blob), _ptr(bias), _ptr(wpt), _stream()), "car_fused_pack_rows")
            self._xpack, self._xpack_key = (blob, bias, wpt), key
        return self._xpack

    def _kq_weights(self, device):
        """key_map_2, query_embed and query_embed_2 packed for the key / query chain kernel (car_kq_pack)."""
        m = self.m
        ps = (m.key_map_2.weight, m.key_map_2.bias, m.query_embed.weight, m.query_embed.bias, m.query_embed_2.weight, m.query_embed_2.bias)
        key = tuple((t.data_ptr(), t._version) for t in ps) + (str(device),)
        if key != self._kq_key:
            lib = self.lib
            f = [t.detach().to(device=device, dtype=torch.float32).reshape(t.shape[0], -1).contiguous() if t.dim() > 1
                 else t.detach().to(device=device, dtype=torch.float32).contiguous() for t in ps]
            tail = torch.empty(lib.car_kq_tail_floats(), device=device, dtype=torch.float32)
            bias = torch.empty(lib.car_kq_bias_floats(), device=device, dtype=torch.float32)
            _lib.check(lib.car_kq_pack(*[_ptr(t) for t in f], _ptr(tail), _ptr(bias), _stream()), "car_kq_pack")
            self._kq, self._kq_key = (tail, bias), key
        return self._kq

    # ------------------------------------------------------------------ the one-call route (default configuration)
    def _dims(self, b: int, R: int, z: List[Tensor]) -> "_lib.CarDims":
        m = self.m
        d = _lib.CarDims()
        d.b, d.V, d.R, d.P, d.H, d.W = b, m.n_view, R, m.npoints, m.H, m.W
        d.n_levels = len(z)
        for l, t in enumerate(z):
            d.level_c[l], d.level_h[l], d.level_w[l] = t.shape[1], t.shape[2], t.shape[3]
        d.repeat_attention = int(m.repeat_attention)
        d.no_sample = int(m.no_sample)
        return d

    def _plan_for(self, d, device) -> Tensor:
        """car_plan_build: every layer packed for the kernels, once per set of parameter values."""
        m, lib = self.m, self.lib
        names = list(_lib.WEIGHT_FIELDS[0]) + [f"phi.lin_z.{i}" for i in range(3)] + [f"phi.blocks.{i}.fc_0" for i in range(3)] \
            + [f"phi.blocks.{i}.fc_1" for i in range(3)]
        sd = dict(m.named_parameters())
        key = (str(device), d.P, tuple(d.level_c[:d.n_levels])) + tuple(
            (sd[n + k].data_ptr(), sd[n + k]._version) for n in names for k in (".weight", ".bias"))
        if key == self._plan_key:
            return self._plan
        keep: List[Tensor] = []

        def dev(name):
            t = sd[name].detach().to(device=device, dtype=torch.float32)
            t = t.reshape(t.shape[0], -1).contiguous() if t.dim() > 1 else t.contiguous()
            keep.append(t)
            return t.data_ptr()

        w = _lib.CarWeights()
        for n in _lib.WEIGHT_FIELDS[0]:
            setattr(w, f"{n.replace('.', '_')}_w", dev(n + ".weight"))
            setattr(w, f"{n.replace('.', '_')}_b", dev(n + ".bias"))
        for i in range(3):
            w.phi_lin_z_w[i], w.phi_lin_z_b[i] = dev(f"phi.lin_z.{i}.weight"), dev(f"phi.lin_z.{i}.bias")
            w.phi_fc_0_w[i], w.phi_fc_0_b[i] = dev(f"phi.blocks.{i}.fc_0.weight"), dev(f"phi.blocks.{i}.fc_0.bias")
            w.phi_fc_1_w[i], w.phi_fc_1_b[i] = dev(f"phi.blocks.{i}.fc_1.weight"), dev(f"phi.blocks.{i}.fc_1.bias")
        nbytes = lib.car_plan_bytes(ctypes.byref(d))
        if nbytes == 0:
            _lib.check(-1, "car_plan_bytes")
        plan = torch.empty(nbytes // 4, device=device, dtype=torch.float32)
        _lib.check(lib.car_plan_build(ctypes.byref(d), ctypes.byref(w), _ptr(plan), _stream()), "car_plan_build")
        self._plan, self._plan_key, self._plan_keep = plan, key, keep
        self._pair_key = None
        return plan

    def _pair_for(self, plan: Tensor, z: List[Tensor], device, s0: int, s1: int, R: int):
        """car_project_maps for scenes [s0, s1): the first point-MLP layer applied per texel of the pyramid and every level summed on the
        common lattice, once per stereo pair (and weights).  Returns (buffer, its car_dims).  One buffer is cached: the whole batch
        normally; when the lattices of all scenes do not fit the device memory (_render_one_call) the last group's."""
        maps = self._channel_last(z)
        V = self.m.n_view
        d = self._dims(s1 - s0, R, z)
        key = (self._maps_key, self._plan_key, s0, s1)
        if key != self._pair_key or self._pair is None:
            lib = self.lib
            self._pair = None                                # release the previous pair's maps before allocating
            pair = torch.empty(lib.car_gmaps_floats(ctypes.byref(d)), device=device, dtype=torch.float32)
            ptrs = (ctypes.c_void_p * len(maps))(*[t[s0 * V:s1 * V].data_ptr() for t in maps])
            _lib.check(lib.car_project_maps(ctypes.byref(d), _ptr(plan), ptrs, _ptr(pair), _stream()), "car_project_maps")
            self._pair, self._pair_key = pair, key
        return self._pair, d

    @staticmethod
    def _common_lattice(z: List[Tensor]) -> bool:
        """The fused kernel gathers every level from their common lattice (car_lattice_shape): each must be an integer factor coarser
        than the widest one, the same factor in both directions.  Other pyramids take the stage route."""
        sizes = [(t.shape[2], t.shape[3]) for t in z]
        hm, wm = max(h for h, _ in sizes), max(w for _, w in sizes)
        return all(hm % h == 0 and wm % w == 0 and hm // h == wm // w for h, w in sizes)

    def _lattice_fits(self, b: int, R: int, z: List[Tensor]) -> bool:
        """car_fused_samples addresses the lattice of one (view, padding mode) with 32-bit byte offsets below 2 GiB (a finest level
        up to ~470 pixels wide at 576 channels); wider pyramids take the stage route, which has no such limit."""
        lh, lw, lpad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self.lib.car_lattice_shape(ctypes.byref(self._dims(b, R, z)), ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(lpad)), "car_lattice_shape")
        return lh.value * lw.value * 576 * 4 < 2**31

    def _free_budget(self, device) -> int:
        """85 % of what this engine could allocate now: free device memory, the caching allocator's idle blocks, and its own workspace
        (which every caller either re-uses in place or releases before it allocates against this figure)."""
        free, _ = torch.cuda.mem_get_info(device)
        cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        mine = self._work.numel() * 4 if self._work is not None else 0
        return int(0.85 * (free + cached + mine))

    def _workspace_budget(self, device) -> int:
        return int(self.max_workspace_bytes) if self.max_workspace_bytes is not None else self._free_budget(device)

    def _render_one_call(self, inp, z, poses, uv, steps, b, V, R, P, H, W, debug) -> Dict[str, Tensor]:
        m, lib = self.m, self.lib
        dev = uv.device
        f32 = dict(device=dev, dtype=torch.float32)
        n = b * V
        d_all = self._dims(b, R, z)
        plan = self._plan_for(d_all, dev)
        lh, lw, lpad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.car_lattice_shape(ctypes.byref(d_all), ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(lpad)), "car_lattice_shape")
        lattice_scene = V * 2 * lh.value * lw.value * 576                                        # floats per scene

        phases = 1 | 2 | (0 if self.first_round_parts else 4)

        def ws_bytes(nb, nr):
            return lib.car_workspace_bytes(ctypes.byref(self._dims(nb, nr, z)))

        def pair_bytes(nb):
            return 4 * lib.car_gmaps_floats(ctypes.byref(self._dims(nb, R, z)))

        # Scenes per lattice buffer: all of them (2.5 GB per scene at 256 x 256, 5.6 GB at 384 x 384) unless that would leave less than
        # half of the usable device memory for the workspace; then the scenes are rendered in groups, each with its own
        # car_project_maps (re-done on every forward: the one cached buffer holds the last group — a fallback, not a fast path).
        self._channel_last(z)
        whole_cached = self._pair is not None and self._pair_key == (self._maps_key, self._plan_key, 0, b)
        pg = b
        if not whole_cached:
            # what is live while a group renders: ONE lattice buffer (the previous group's is dropped before the next is allocated, below)
            # and the workspace.  The cached buffer and workspace of an earlier forward count as available because both are released
            # before this forward allocates (a stale workspace sized for another split would otherwise sit beside the new pair)
            self._pair = None
            self._pair_key = None
            if pair_bytes(b) > (self._free_budget(dev)) // 2:
                self._work = None                                      # grouped fallback: start from everything this engine can free
            usable = self._free_budget(dev)
            if self.max_pair_bytes is not None:
                usable = min(usable, 2 * int(self.max_pair_bytes))
            while pg > 1 and pair_bytes(pg) > usable // 2:
                pg = (pg + 1) // 2

        # inside a lattice group — scenes per call: any number (the fused kernel addresses the lattice of ONE view and padding mode with
        # 32-bit offsets); max_level_bytes lets tests force smaller calls.  Rays per call: the workspace (~0.44 MB per ray at 64 samples)
        # must fit the free memory.  Rays and scenes are independent, so every split is exact.
        gs_cap = pg if self.max_level_bytes is None else max(1, min(pg, self.max_level_bytes // (4 * lattice_scene)))

        out = {"rgb": torch.empty(b, 1, R, 3, **f32), "valid_mask": torch.empty(b, R, 1, **f32), "depth_ray": torch.empty(b, R, 1, **f32),
               "at_wt": torch.empty(n, R, P, **f32), "at_wt_max": torch.empty(n, R, 1, device=dev, dtype=torch.int32),
               "coords": torch.empty(n, R, 9, **f32), "pixel_val": torch.empty(n, R, P, 2, **f32)}
        lead = {"rgb": 1, "valid_mask": 1, "depth_ray": 1, "at_wt": V, "at_wt_max": V, "coords": V, "pixel_val": V}   # rows per scene
        order = ("rgb", "valid_mask", "depth_ray", "at_wt", "at_wt_max", "coords", "pixel_val")
        st = _stream()
        calls = 0
        d_last = None
        self.last_pair_groups = -(-b // pg)
        pair = work = None
        for p0 in range(0, b, pg):
            p1 = min(b, p0 + pg)
            if pg < b:
                # several lattice groups (memory pressure): the previous group's buffer must be gone before the next one is allocated —
                # _pair_for drops the engine's reference, this drops the loop's — and the workspace is re-sized against what is then free
                pair = work = None
                self._work = None
            pair, d_pair = self._pair_for(plan, z, dev, p0, p1, R)
            gmeta_ptr = pair.data_ptr() + 4 * lib.car_gmeta_offset(ctypes.byref(d_pair))
            budget = self._workspace_budget(dev)
            gs = min(gs_cap, p1 - p0)
            while gs > 1 and ws_bytes(gs, R) > budget:
                gs = (gs + 1) // 2
            rc = R
            if ws_bytes(gs, R) > budget:
                per_ray = ws_bytes(gs, 4800) / 4800.0
                rc = int(budget / per_ray) // 48 * 48            # whole sample tiles of the fused kernel (24 rays each)
                if rc < 48:
                    raise RuntimeError(f"forward: {budget / 2**20:.0f} MiB of workspace cannot hold even 48 rays "
                                       f"({ws_bytes(gs, 48) / 2**20:.0f} MiB needed): free device memory")
            need = ws_bytes(gs, min(rc, R))
            if self._work is None or self._work.numel() * 4 < need:
                self._work = None
                self._work = torch.empty(need // 4, **f32)
            work = self._work
            for s0 in range(p0, p1, gs):
                s1 = min(p1, s0 + gs)
                for r0 in range(0, R, rc):
                    r1 = min(R, r0 + rc)
                    d = self._dims(s1 - s0, r1 - r0, z)
                    whole = (r0 == 0 and r1 == R)
                    if whole:
                        tgt = {k: out[k][s0 * lead[k]:s1 * lead[k]] for k in order}            # contiguous scene slices: written in place
                        uv_c = uv[s0:s1]
                    else:
                        tgt = {k: torch.empty((s1 - s0) * lead[k], r1 - r0, *out[k].shape[2 if k != "rgb" else 3:],
                                              device=dev, dtype=out[k].dtype) for k in order}
                        uv_c = uv[s0:s1, r0:r1].contiguous()
                    ci = _lib.CarInputs()
                    ci.poses = poses.data_ptr() + 4 * 96 * s0 * V
                    ci.uv = uv_c.data_ptr()
                    ci.lattice = pair.data_ptr() + 4 * (s0 - p0) * lattice_scene
                    ci.gmeta = gmeta_ptr
                    ci.steps = steps.data_ptr()
                    co = _lib.CarOutputs(*[tgt[k].data_ptr() for k in order])
                    if phases == 3:
                        _lib.check(lib.car_render_forward(ctypes.byref(d), _ptr(plan), ctypes.byref(ci), ctypes.byref(co),
                                                          _ptr(work), work.numel() * 4, st), "car_render_forward")
                    else:
                        _lib.check(lib.car_render_forward_phase(ctypes.byref(d), _ptr(plan), ctypes.byref(ci), ctypes.byref(co),
                                                                _ptr(work), work.numel() * 4, phases, st), "car_render_forward_phase")
                    if not whole:
                        for k in order:
                            dst = out[k][:, 0] if k == "rgb" else out[k]
                            dst[s0 * lead[k]:s1 * lead[k], r0:r1] = tgt[k]
                    calls += 1
                    d_last = d
        self.last_calls = calls
        res = {
            "rgb": out["rgb"], "valid_mask": out["valid_mask"], "depth_ray": out["depth_ray"], "at_wt": out["at_wt"],
            "at_wts": [out["at_wt"]], "at_wt_max": out["at_wt_max"].long(), "coords": out["coords"], "uv": inp["query"]["uv"],
            # the reference returns pixel_val on the CPU (models.py:570), forcing a device sync on every call; here it stays on the
            # device unless debug is set
            "pixel_val": out["pixel_val"].cpu() if debug else out["pixel_val"], "z": z,
        }
        if debug:
            if calls != 1:
                raise RuntimeError("debug=True needs the forward to fit one car_render_forward call (intermediates live in its workspace)")

            def ws(name, *shape):
                off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
                _lib.check(lib.car_workspace_find(ctypes.byref(d_last), name.encode(), ctypes.byref(off), ctypes.byref(cnt)), "car_workspace_find")
                return work[off.value:off.value + cnt.value].view(*shape).clone()
            res["stages"] = {"rays": ws("rays", n, R, 12), "pt": ws("pt", n, R, P, 3), "local_coords": None,
                             "g": ws("g", n, R, P, 16), "interp_val": ws("e", n, R, P, 576),
                             "z1": ws("z1", b, R, 288),
                             "at_wt2": ws("at_wt2", n, R, P) if m.repeat_attention else None, "poses": poses}
        return res

    # ------------------------------------------------------------------ stage timing (csrc/car_render.hip)
    def profile(self, on: bool) -> None:
        """Brackets every stage of the one-call route with HIP events on the launch stream (no synchronisation)."""
        self.lib.car_profile_enable(1 if on else 0)

    def stage_times(self) -> List[tuple]:
        """[(stage name, milliseconds)] of everything recorded since profile(True) / the last call; synchronises on the events."""
        lib = self.lib
        rows = []
        for i in range(lib.car_profile_count()):
            name, ms = ctypes.c_char_p(), ctypes.c_float()
            _lib.check(lib.car_profile_read(i, ctypes.byref(name), ctypes.byref(ms)), "car_profile_read")
            rows.append((name.value.decode(), ms.value))
        lib.car_profile_reset()
        return rows

    # ------------------------------------------------------------------ kernels
    def _lds_refused(self, rc: int, what: str) -> bool:
        """True when ``rc`` says the device refused the kernel's LDS reservation (CAR_E_LAUNCH with the "cannot reserve" text: a part with less
        than gfx950's 160 KB per compute unit) — the ONE condition under which a route steps down to the next HIP kernel of the same library.
        Anything else (an argument error, a sticky fault of an earlier launch) raises: it must not silently change the arithmetic."""
        if rc == 0:
            return False
        msg = (self.lib.car_last_error() or b"").decode()
        if rc == -2 and "cannot reserve" in msg:
            import warnings
            warnings.warn(f"{what}: {msg}: taking the next kernel of the library")
            return True
        raise RuntimeError(f"{what} failed ({rc}): {msg}")

    def linear(self, x: Tensor, ldx: int, layer: PackedLinear, y: Tensor, ldy: int, M: int, flags: int = 0, mask=None):
        """``mask = (act, lda)``: the result is zeroed where ``act <= 0`` (the backward of a ReLU in front of the layer whose data gradient this
        is) — in the split-fp16 kernel's store (car_linear_x3_masked), or by car_relu_mask behind the fp32-pipe kernel."""
        # linear_flags (the NO_GLDS A/B knob of car_linear) selects the fp32-pipe kernel for every layer: the split-fp16 kernel has no such
        # variant, and an A/B run must not change arithmetic on some layers only
        if (self.linear_x3 and not self.linear_flags and layer.x3 is not None and ldx % 4 == 0 and ldy % 4 == 0 and x.data_ptr() % 16 == 0
                and y.data_ptr() % 16 == 0 and M >= self.linear_x3_min_rows):
            tiles, bias = layer.x3
            if mask is not None and mask[1] % 4 == 0 and mask[0].data_ptr() % 16 == 0:
                rc = self.lib.car_linear_x3_masked(_ptr(x), ldx, _ptr(tiles), _ptr(bias), layer.K, layer.N, _ptr(y), ldy, M, flags, _ptr(mask[0]),
                                                   mask[1], _stream())
                mask = None if rc == 0 else mask
            else:
                rc = self.lib.car_linear_x3(_ptr(x), ldx, _ptr(tiles), _ptr(bias), layer.K, layer.N, _ptr(y), ldy, M, flags, _stream())
            if not self._lds_refused(rc, "car_linear_x3"):
                if mask is not None:
                    _lib.check(self.lib.car_relu_mask(_ptr(y), ldy, _ptr(mask[0]), mask[1], M, layer.N, _stream()), "car_relu_mask")
                return
            # the split-fp16 kernel's 72 KB weight double buffer was refused: the same layer on the fp32 matrix pipe from here on — another
            # HIP kernel of the same library, never a host path
            self.linear_x3 = False
        _lib.check(self.lib.car_linear(_ptr(x), ldx, _ptr(layer.packed), layer.K, layer.N, _ptr(y), ldy, M,
                                       flags | self.linear_flags, _stream()), "car_linear")
        if mask is not None:
            _lib.check(self.lib.car_relu_mask(_ptr(y), ldy, _ptr(mask[0]), mask[1], M, layer.N, _stream()), "car_relu_mask")

    def gather(self, maps: List[Tensor], grid: Tensor, pts: int, mode: int, place: int, V: int, out: Tensor,
               ld_out: int, col_out: int, run: int = 1):
        L = len(maps)
        ptrs = (ctypes.c_void_p * L)(*[m.data_ptr() for m in maps])
        cs = (ctypes.c_int * L)(*[m.shape[3] for m in maps])
        hs = (ctypes.c_int * L)(*[m.shape[1] for m in maps])
        ws = (ctypes.c_int * L)(*[m.shape[2] for m in maps])
        _lib.check(self.lib.car_gather_bilinear(ptrs, cs, hs, ws, L, maps[0].shape[0], _ptr(grid), pts, run, mode, place, V,
                                                _ptr(out), ld_out, col_out, _stream()), "car_gather_bilinear")

    # ------------------------------------------------------------------ the forward pass
    @torch.no_grad()
    def render(self, inp, z: List[Tensor], debug: bool = False) -> Dict[str, Tensor]:
        dev = inp["query"]["uv"].device
        if dev.type != "cuda":
            raise RuntimeError("CrossAttentionRenderer.forward runs on the HIP engine only: move the model, the input "
                               "dict and z to a ROCm device (there is no CPU fallback)")
        # kernels are launched on the current device's current stream: make the tensors' device current for the call
        with torch.cuda.device(dev):
            return self._render(inp, z, debug)

    def _render(self, inp, z: List[Tensor], debug: bool) -> Dict[str, Tensor]:
        m, lib = self.m, self.lib
        ctx, qry = inp["context"], inp["query"]
        uv_in = qry["uv"]
        dev = uv_in.device
        b, V = ctx["rgb"].shape[:2]
        n_qry, R = uv_in.shape[1:3]
        if n_qry != 1:
            raise ValueError("forward supports one query view per scene (reference models.py:213, 619)")
        if V != m.n_view:
            raise ValueError(f"input has {V} context views, module was built with n_view={m.n_view}")
        P, H, W = m.npoints, m.H, m.W
        n, S = b * V, b * V * R * P
        st = _stream()
        f32 = dict(device=dev, dtype=torch.float32)
        # a3: pose algebra on the host, exactly the reference's torch calls
        poses = self._poses(inp, H, n, dev)
        uv = uv_in.detach().reshape(b, R, 2).float().contiguous()
        steps = self._linspace(0.1, 10.0, P, dev) if m.no_sample else self._linspace(0.0, 1.0, P, dev)

        concat2 = (V == 2 and not m.no_latent_concat)
        if (self.fuse_samples and self.project_maps and concat2 and len(z) == 3
                and sum(t.shape[1] for t in z) == 576 and m.hidden_dim == 128 and m.phi.n_blocks == 3 and m.phi.d_hidden == 128
                and self._common_lattice(z) and self._lattice_fits(b, R, z)):
            return self._render_one_call(inp, z, poses, uv, steps, b, V, R, P, H, W, debug)

        pk = self._weights(dev)
        maps = self._channel_last(z)
        C = sum(t.shape[3] for t in maps)
        Dl = m.latent_dim

        # a4-a6: rays
        rays = torch.empty(n, R, 12, **f32)
        coords9 = torch.empty(n, R, 9, **f32)
        ld_phi = _round_up(9 * V, 4)
        phi_x = torch.zeros(b * R, ld_phi, **f32)
        _lib.check(lib.car_ray_setup(_ptr(poses), _ptr(uv), b, V, R, H, W, P, int(m.no_sample), _ptr(steps),
                                     _ptr(rays), _ptr(coords9), _ptr(phi_x), ld_phi, st), "car_ray_setup")

        # a6, a8, a9, a13: samples
        pixel_val = torch.empty(n, R, P, 2, **f32)
        pt = torch.empty(n, R, P, 3, **f32)
        g = torch.empty(S, 16, **f32)
        single = (V == 1 and not m.no_latent_concat)
        concat3 = (V == 3 and not m.no_latent_concat)
        kmajor = False
        grid_in = torch.empty(n, R, P, V, 2, **f32) if concat2 else None
        pt_in = torch.empty(n, R, P, V, 3, **f32) if concat3 else None
        x1 = None
        proj = concat2 and self.project_maps
        if proj:
            ld1 = 4                                   # only the 3 point channels: [S*V, 4]
            x1 = torch.zeros(S * V, ld1, **f32)
        elif concat3:
            ld1 = 4                                   # point channels only; the 579-wide rows are assembled below
            x1 = torch.zeros(S * V, ld1, **f32)
        elif concat2:
            ld1 = _round_up(C + 3, 32)
            x1 = torch.empty(S * V, ld1, **f32)
        elif single:
            ld1 = _round_up(C + 6, 32)
            x1 = torch.empty(S, ld1, **f32)
        _lib.check(lib.car_sample_setup(_ptr(poses), _ptr(rays), _ptr(steps), b, V, R, P, H, W, int(m.no_sample),
                                        _ptr(pixel_val), _ptr(pt), _ptr(g), _ptr(grid_in), _ptr(x1),
                                        ld1 if x1 is not None else 0, 0 if (proj or concat3) else C, _ptr(pt_in), st),
                   "car_sample_setup")

        # a7, a9-a11: per-sample features e
        if proj:
            gmaps, wpt = self._projected_maps(maps, dev)
            h1 = torch.empty(S * V, C, **f32)
            self.gather_encode(gmaps, wpt, pixel_val, grid_in, x1, V, R * P, h1, C)
            e = torch.empty(S, V * (C // 2), **f32)
            self.linear(h1, C, pk["query_encode_latent_2"], e, C // 2, S * V)
            del h1
            Ce = V * (C // 2)
        elif concat2:
            self.gather(maps, pixel_val, R * P, 0, PLACE_OWN, V, x1, ld1, 0, run=P)
            gi = grid_in.view(b, V, R, P, V, 2)
            # pixel_val_stack (models.py:316): map (b, s) is sampled where the *other* line's points land in view s
            grid_other = torch.stack([gi[:, 1, :, :, 0], gi[:, 0, :, :, 1]], dim=1).contiguous()
            self.gather(maps, grid_other, R * P, 1, PLACE_OTHER2, V, x1, ld1, 0, run=P)
            h1 = torch.empty(S * V, C, **f32)
            self.linear(x1, ld1, pk["query_encode_latent"], h1, C, S * V, RELU_OUT)
            e = torch.empty(S, V * (C // 2), **f32)
            self.linear(h1, C, pk["query_encode_latent_2"], e, C // 2, S * V)
            del h1
            Ce = V * (C // 2)
        elif concat3:
            e = self._encode_three_views(maps, poses, pixel_val, x1, pt_in, b, R, P, H, W, C, pk)
            Ce = 3 * (C // 2)
            kmajor = self.project_maps                              # the inference path of _encode_three_views leaves e component-major
        elif single:
            self.gather(maps, pixel_val, R * P, 0, PLACE_PLAIN, V, x1, ld1, 0, run=P)
            e = torch.empty(S, C, **f32)
            self.linear(x1, ld1, pk["update_val_merge"], e, C, S)
            Ce = C
        else:
            e = torch.empty(S, C, **f32)
            self.gather(maps, pixel_val, R * P, 0, PLACE_PLAIN, V, e, C, 0, run=P)
            Ce = C
        del x1

        # a12: keys;  a13: geometric query.  The value projection (latent_value, no nonlinearity before the weighted sum)
        # commutes with the attention average: sum_s w_s (Wv e_s + bv) = Wv (sum_s w_s e_s) + bv because the softmax
        # weights of a ray sum to 1, so it is applied once per ray after the reduction instead of once per sample.
        kmap = pk["key_map.kmajor" if kmajor else "key_map"]
        q = torch.empty(S, 128, **f32)
        if (self.fuse_kq and self.linear_x3 and not self.linear_flags and kmap.x3 is not None and kmap.N == 128 and m.hidden_dim == 128
                and Ce % 4 == 0 and e.data_ptr() % 16 == 0 and S >= self.linear_x3_min_rows):
            # key_map -> relu -> key_map_2, query_embed -> relu -> query_embed_2 and the first round's logits in one kernel: the 128-wide
            # k1 / key / q1 rows are never written (csrc/car_linear16.hip, KQ instance)
            tiles, bias_k1 = kmap.x3
            tail, tail_bias = self._kq_weights(dev)
            logit1 = torch.empty(S, **f32)
            rc = lib.car_key_query_logits(_ptr(e), Ce, _ptr(tiles), _ptr(bias_k1), Ce, _ptr(g), _ptr(tail), _ptr(tail_bias), S, _ptr(q), _ptr(logit1), st)
            if not self._lds_refused(rc, "car_key_query_logits"):
                return self._finish(inp, z, b, V, R, P, Ce, Dl, e, None, q, g, pt, pixel_val, poses, rays, coords9, phi_x, ld_phi, debug, kmajor,
                                    logit1=logit1)
            self.fuse_kq = False                                   # LDS reservation refused: the five separate launches below
        k1 = torch.empty(S, 128, **f32)
        self.linear(e, Ce, kmap, k1, 128, S, RELU_OUT)
        key = torch.empty(S, 128, **f32)
        self.linear(k1, 128, pk["key_map_2"], key, 128, S)
        self.linear(g, 16, pk["query_embed"], k1, 128, S, RELU_OUT)
        self.linear(k1, 128, pk["query_embed_2"], q, 128, S)

        return self._finish(inp, z, b, V, R, P, Ce, Dl, e, key, q, g, pt, pixel_val, poses, rays, coords9, phi_x, ld_phi, debug, kmajor)

    def _exchange_lattice(self, gmaps, ptrs, hs, ws, n_maps, C, dev):
        """The projected levels of the three-view exchange summed on their common lattice (car_merge_lattice), cached with the projected
        maps; None when the levels have no common lattice, the channel count is not the lattice kernels' 576, or it would not fit."""
        if C != 576:
            return None
        key = self._gmaps_key                                      # identifies the projected levels' CONTENT (pyramid + layer versions)
        if self._xlat is not None and self._xlat_key == key:
            return self._xlat
        L = len(gmaps)
        lh, lw, lpad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if self.lib.car_merge_lattice(ptrs, hs, ws, L, n_maps, None, ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(lpad), _stream()) != 0:
            return None
        floats = n_maps * 2 * lh.value * lw.value * C
        if n_maps * 2 * lh.value * lw.value >= 2 ** 31 - 1 or floats * 4 > self._free_budget(dev) // 2:
            return None
        self._xlat = None                                           # the previous lattice goes before the next one is allocated
        lattice = torch.empty(floats, device=dev, dtype=torch.float32)
        gmax = torch.empty(1, device=dev, dtype=torch.float32)     # largest |lattice value|: taken in the merge's own pass
        _lib.check(self.lib.car_merge_lattice_max(ptrs, hs, ws, L, n_maps, _ptr(lattice), _ptr(gmax), _stream()), "car_merge_lattice_max")
        self._xlat, self._xlat_key = (lattice, lh.value, lw.value, lpad.value, gmax), key
        return self._xlat

    def _encode_three_views(self, maps, poses, pixel_val, ptenc, pt_in, b, R, P, H, W, C, pk, keep=None):
        """Cross-view exchange for three context views (models.py:345-475), restated literally: for the samples of context c
        the row is the channel-interleaved concatenation of enc(own features, point in frame c) and, for every other view o
        in ascending order, enc(features of view o where context o's points — moved into frame c, projected with view o's
        intrinsics — land in image o, those points in frame c).  All arithmetic is in HIP kernels (gather, projection, the two
        1x1 layers); torch only moves rows into place.  ``keep`` (a dict): the training forward's saves — the 579-wide rows, the first
        layer's activations and every cross-view gather's grid (training.render_train)."""
        V, pts = 3, R * P
        S = b * V * pts
        dev = pixel_val.device
        f32 = dict(device=dev, dtype=torch.float32)
        if keep is None and self.project_maps:
            # inference: the first layer applied per texel (car_project_maps' arithmetic, as on the two-view route): the 579 -> 576 GEMM over
            # 3 S rows — two thirds of this variant's FLOPs — becomes one gather over the projected pyramid with an explicit row list
            gmaps, wpt = self._projected_maps(maps, dev)
            n = b * V
            # the row lists (map | padding mode, grid point, point encoding per (sample, component)) in one launch (car_exchange_rows)
            src = torch.empty(b, V, pts, 3, dtype=torch.int32, device=dev)
            rgrid = torch.empty(b, V, pts, 3, 2, **f32)
            rpe = torch.empty(b, V, pts, 3, 4, **f32)
            _lib.check(self.lib.car_exchange_rows(_ptr(poses), _ptr(pixel_val), _ptr(pt_in), _ptr(ptenc), b, V, pts, H, W, _ptr(src), _ptr(rgrid),
                                                  _ptr(rpe), _stream()), "car_exchange_rows")
            L = len(gmaps)
            ptrs = (ctypes.c_void_p * L)(*[g.data_ptr() for g in gmaps])
            hs = (ctypes.c_int * L)(*[g.shape[1] for g in gmaps])
            ws = (ctypes.c_int * L)(*[g.shape[2] for g in gmaps])
            lat = self._exchange_lattice(gmaps, ptrs, hs, ws, n, C, dev)
            layer2 = pk["query_encode_latent_2"]
            if lat is not None and self.fuse_exchange == "rows" and not self.linear_flags and C == 576:
                # the fused per-sample kernel's source pass over the exchange's rows (car_fused_rows): its gather machinery and W2 in one
                # launch, e written [S, 3, 288]; the lattice's largest magnitude (the first layer's fp16 scale) is taken once per lattice
                lattice, lh, lw, lpad, gmax = lat
                blob, fbias, fwpt = self._exchange_pack(dev)
                enc = torch.empty(S * 3, C // 2, **f32)
                rc = self.lib.car_fused_rows(_ptr(lattice), lh, lw, lpad, _ptr(gmax), _ptr(fwpt), _ptr(blob), _ptr(fbias), _ptr(src), _ptr(rgrid), _ptr(rpe),
                                             n, R, P, 3, _ptr(enc), _stream())
                if not self._lds_refused(rc, "car_fused_rows"):
                    return enc.view(S, 3 * (C // 2))
                self.fuse_exchange = True                          # LDS reservation refused: the gather-fed linear kernel, then two launches
            if lat is not None and self.fuse_exchange and self.linear_x3 and not self.linear_flags and layer2.x3 is not None:
                # first AND second exchange layer in one kernel: the lattice rows are gathered 32 channels at a time into the matrix pipe's
                # operands, the 576-wide rows (2.3 KB x 3 S) are never written (csrc/car_linear16.hip, GATHER instance; bit-identical
                # to the two launches below)
                lattice, lh, lw, lpad, _ = lat
                tiles, bias2 = layer2.x3
                enc = torch.empty(S * 3, C // 2, **f32)
                rc = self.lib.car_lattice_encode_linear(_ptr(lattice), lh, lw, lpad, _ptr(src), _ptr(rgrid), _ptr(rpe), _ptr(wpt), n, S * 3,
                                                        _ptr(tiles), _ptr(bias2), C, C // 2, _ptr(enc), C // 2, 0, _stream())
                if not self._lds_refused(rc, "car_lattice_encode_linear"):
                    return enc.view(S, 3 * (C // 2))
                self.fuse_exchange = False
            h1 = torch.empty(S * 3, C, **f32)
            if lat is not None:                                     # four taps of the merged lattice per row instead of twelve
                lattice, lh, lw, lpad, _ = lat
                _lib.check(self.lib.car_lattice_encode_rows(_ptr(lattice), lh, lw, lpad, C, _ptr(src), _ptr(rgrid), _ptr(rpe), _ptr(wpt), n, S * 3,
                                                            _ptr(h1), C, _stream()), "car_lattice_encode_rows")
            else:
                _lib.check(self.lib.car_gather_encode_rows(ptrs, hs, ws, L, C, _ptr(src), _ptr(rgrid), _ptr(rpe), _ptr(wpt), n, S * 3, _ptr(h1), C,
                                                           _stream()), "car_gather_encode_rows")
            enc = torch.empty(S * 3, C // 2, **f32)
            self.linear(h1, C, pk["query_encode_latent_2"], enc, C // 2, S * 3)
            return enc.view(S, 3 * (C // 2))                          # component-major: the consumers' weights are permuted (_weights)
        ld = _round_up(C + 3, 32)
        x3 = torch.empty(S * 3, ld, **f32)
        x3v = x3.view(b, V, pts, 3, ld)                     # [scene, context c, point, component k, channel]
        pe = ptenc.view(b, V, pts, V, 4)                    # tanh(nan_to_num(T_s pt)/5): [scene, context, point, frame s]
        pin = pt_in.view(b, V, pts, V, 3)
        tmp = torch.empty(S, C, **f32)
        self.gather(maps, pixel_val, pts, 0, PLACE_PLAIN, V, tmp, C, 0, run=P)
        x3v[:, :, :, 0, :C] = tmp.view(b, V, pts, C)
        per_view = [[t.view(b, V, *t.shape[1:])[:, o].contiguous() for t in maps] for o in range(V)]
        tmp2 = torch.empty(b * pts, C, **f32)
        grid = torch.empty(b, pts, 2, **f32)
        for c in range(V):
            x3v[:, c, :, 0, C:C + 3] = pe[:, c, :, c, :3]
            k = 1
            for o in range(V):
                if o == c:
                    continue
                q = pin[:, o, :, c, :].contiguous()         # context o's points expressed in frame c
                _lib.check(self.lib.car_project_points(_ptr(poses), _ptr(q), b, pts, V, o, H, W, _ptr(grid), _stream()),
                           "car_project_points")
                self.gather(per_view[o], grid, pts, 1, PLACE_PLAIN, 1, tmp2, C, 0, run=P)
                if keep is not None:
                    keep.setdefault("cross", []).append((c, o, k, grid.clone()))
                x3v[:, c, :, k, :C] = tmp2.view(b, pts, C)
                x3v[:, c, :, k, C:C + 3] = pe[:, o, :, c, :3]
                k += 1
        h1 = torch.empty(S * 3, C, **f32)
        self.linear(x3, ld, pk["query_encode_latent"], h1, C, S * 3, RELU_OUT)
        enc = torch.empty(S * 3, C // 2, **f32)
        self.linear(h1, C, pk["query_encode_latent_2"], enc, C // 2, S * 3)
        if keep is not None:
            keep.update(x3=x3, h1=h1, ld=ld)
        # channel index = ch*3 + k (torch.cat on dim 2 then flatten(1, 2), models.py:446)
        return enc.view(S, 3, C // 2).permute(0, 2, 1).contiguous().view(S, 3 * (C // 2))

    def _finish(self, inp, z, b, V, R, P, Ce, Dl, e, key, q, g, pt, pixel_val, poses, rays, coords9, phi_x, ld_phi, debug, kmajor=False, logit1=None):
        """Attention rounds, decoder and output dict of the staged route; ``g`` is the geometric
        query local_coords [S,16]."""
        m, lib = self.m, self.lib
        dev = e.device
        f32 = dict(device=dev, dtype=torch.float32)
        st = _stream()
        pk = self._packed
        lv = pk["latent_value.kmajor" if kmajor else "latent_value"]
        n, S = b * V, b * V * R * P
        n_qry = 1
        qry = inp["query"]
        # a14 + a16: attention round 1, depth read-out
        at_wt = torch.empty(n, R, P, **f32)
        depth = torch.empty(b, R, **f32)
        amax = torch.empty(n, R, dtype=torch.int32, device=dev)
        rep = m.repeat_attention
        ebar = torch.empty(b * R, Ce, **f32)
        if logit1 is not None:                             # the logits came out of car_key_query_logits
            _lib.check(lib.car_attend(_ptr(logit1), None, 128, _ptr(e), Ce, b, V, R, P, None, 0.0, _ptr(at_wt), _ptr(ebar),
                                      Ce, 1, _ptr(pt), _ptr(poses), _ptr(depth), _ptr(amax), st), "car_attend")
        else:
            _lib.check(lib.car_attend(_ptr(key), _ptr(q), 128, _ptr(e), Ce, b, V, R, P, None, 0.0, _ptr(at_wt), _ptr(ebar),
                                      Ce, 1, _ptr(pt), _ptr(poses), _ptr(depth), _ptr(amax), st), "car_attend")
        zrep = torch.empty(b * R, V * Dl, **f32)
        at_wt2 = None
        if rep:
            z1 = torch.empty(b * R, Dl, **f32)
            self.linear(ebar, Ce, lv, z1, Dl, b * R)
            # a15: second round; the z_embed half of query_repeat_embed is per ray, the local_coords half per sample
            hb = torch.empty(b * R, 128, **f32)
            self.linear(z1, Dl, pk["encode_latent"], hb, 128, b * R)
            uh = torch.empty(b * R, 128, **f32)
            self.linear(hb, 128, pk["query_repeat_embed.h"], uh, 128, b * R)
            at_wt2 = torch.empty(n, R, P, **f32)
            if self.fuse_round2:
                # ug = Wr1[:,128:] g + br1, q2 = Wr2 relu(ug + uh) + b and <q2, qry>/16 in one kernel: neither is written
                r2w, r2b = self._round2_weights(dev)
                logit2 = torch.empty(S, **f32)
                _lib.check(lib.car_round2_logits(_ptr(g), _ptr(uh), _ptr(q), _ptr(r2w), _ptr(r2b),
                                                 b, V, R, P, _ptr(logit2), st), "car_round2_logits")
                _lib.check(lib.car_attend(_ptr(logit2), None, 128, _ptr(e), Ce, b, V, R, P, None, 0.0, _ptr(at_wt2),
                                          _ptr(ebar), Ce, 1, None, None, None, None, st), "car_attend")
            else:
                k1 = torch.empty(S, 128, **f32)
                if key is None:
                    key = torch.empty(S, 128, **f32)
                self.linear(g, 16, pk["query_repeat_embed.g"], k1, 128, S)
                _lib.check(lib.car_add_ray_bias_relu(_ptr(k1), _ptr(uh), b, V, R, P, 128, st), "car_add_ray_bias_relu")
                self.linear(k1, 128, pk["query_repeat_embed_2"], key, 128, S)
                _lib.check(lib.car_attend(_ptr(key), _ptr(q), 128, _ptr(e), Ce, b, V, R, P, None, 0.0, _ptr(at_wt2),
                                          _ptr(ebar), Ce, 1, None, None, None, None, st), "car_attend")
            # z = (Wv ebar2 + bv) + V * z1   (models.py:561-565: "+ z_local" per view, then the view sum)
            zv = zrep.view(b * R, V, Dl)
            zv[:, 0] = z1 * float(V)
            self.linear(ebar, Ce, lv, zrep, V * Dl, b * R, ACCUM)
        else:
            self.linear(ebar, Ce, lv, zrep, V * Dl, b * R)
        if V > 1:                                         # the per-view replication of models.py:541, 565, 605-606
            zv = zrep.view(b * R, V, Dl)
            zv[:, 1:] = zv[:, :1]

        # a17: light-field decoder
        hid = m.phi.d_hidden
        x = torch.empty(b * R, hid, **f32)
        net = torch.empty(b * R, hid, **f32)
        self.linear(phi_x, ld_phi, pk["phi.lin_in"], x, hid, b * R)
        for i in range(m.phi.n_blocks):
            self.linear(zrep, V * Dl, pk[f"phi.lin_z.{i}"], x, hid, b * R, ACCUM)
            self.linear(x, hid, pk[f"phi.blocks.{i}.fc_0"], net, hid, b * R, RELU_IN)
            self.linear(net, hid, pk[f"phi.blocks.{i}.fc_1"], x, hid, b * R, RELU_IN | ACCUM)
        out3 = torch.empty(b * R, 4, **f32)
        self.linear(x, hid, pk["phi.lin_out"], out3, 4, b * R, RELU_IN)

        # a18: valid mask, white background, output dict
        rgb = torch.empty(b, R, 3, **f32)
        valid = torch.empty(b, R, **f32)
        _lib.check(lib.car_finalize(_ptr(rays), _ptr(out3), 4, b, V, R, _ptr(rgb), _ptr(valid), st), "car_finalize")

        out = {
            "rgb": rgb.view(b, n_qry, R, 3),
            "valid_mask": valid[..., None],
            "depth_ray": depth[..., None],
            "at_wt": at_wt,
            "at_wts": [at_wt],
            "at_wt_max": amax.long()[..., None],
            "coords": coords9,
            "uv": qry["uv"],
            # the reference returns pixel_val on the CPU (models.py:570), forcing a device sync on every call; here it
            # stays on the device unless debug is set
            "pixel_val": pixel_val.view(n, R, P, 2).cpu() if debug else pixel_val.view(n, R, P, 2),
            "z": z,
        }
        if debug:
            out["stages"] = {"rays": rays, "pt": pt.view(n, R, P, 3),
                             "local_coords": g.view(n, R, P, 16),
                             # the reference's channel order e[s, 3 ch + k] (models.py:446) when e was kept component-major
                             "interp_val": (e.view(S, 3, Ce // 3).permute(0, 2, 1).reshape(n, R, P, Ce) if kmajor else e.view(n, R, P, Ce)),
                             "z_final": zrep[:, :Dl].reshape(b, R, Dl), "at_wt2": at_wt2, "poses": poses}
        return out
