"""Host side of the render forward: sequences the HIP stage kernels of ``libcar_hip.so`` on the current stream.

``RenderEngine.render(input, z)`` is what ``CrossAttentionRenderer.forward`` runs.  Every arithmetic step of the
reference forward (models.py:206-621) is a HIP kernel launched through the C ABI (``include/car_hip.h``); PyTorch
is used for device memory, the stream handle and the (reference-identical) 4x4 pose algebra on the host.
There is no CPU or eager-PyTorch fallback: tensors must live on a ROCm device and the library must load.

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


def _pack_tiles(W: Tensor, bias: Optional[Tensor], n_tiles: int, chunk_k: Tensor) -> Tensor:
    """Packs weight rows into MFMA A-operand tiles.  ``chunk_k`` (chunks, 64 lanes, 16 steps) gives, for every chunk, lane
    and MFMA step, the input index k that lane supplies (k == K selects the bias, k > K a zero).  Result:
    (chunks, n_tiles, 4, 64, 4) floats = [chunk][tile][j4][lane][e] with step r = 4*j4 + e, output n = 32*tile + lane%32."""
    N, K = W.shape
    Wext = torch.zeros(32 * n_tiles, K + 2, dtype=torch.float32)
    Wext[:N, :K] = W
    if bias is not None:
        Wext[:N, K] = bias
    chunks = chunk_k.shape[0]
    k = chunk_k.clamp(max=K + 1)                                             # (chunks, 64, 16)
    lane = torch.arange(64)
    n = (32 * torch.arange(n_tiles)[:, None] + (lane % 32)[None, :])         # (tiles, 64)
    out = Wext[n[None, :, :, None].expand(chunks, -1, -1, 16), k[:, None, :, :].expand(-1, n_tiles, -1, -1)]   # (chunks,tiles,64,16)
    return out.reshape(chunks, n_tiles, 64, 4, 4).permute(0, 1, 3, 2, 4).contiguous()


def _std_k(chunks: int) -> Tensor:
    """standard mapping: chunk c, lane l, step r -> k = 32 c + 16 (l >> 5) + r"""
    c = torch.arange(chunks)[:, None, None]
    lane = torch.arange(64)[None, :, None]
    r = torch.arange(16)[None, None, :]
    return 32 * c + 16 * (lane // 32) + r


def _chained_k(n_src_tiles: int, base: int = 0) -> Tensor:
    """chained mapping over the accumulator layout of a 32-row tile T: k = base + 32 T + (r&3) + 8 (r>>2) + 4 (l >> 5)"""
    T = torch.arange(n_src_tiles)[:, None, None]
    lane = torch.arange(64)[None, :, None]
    r = torch.arange(16)[None, None, :]
    return base + 32 * T + (r % 4) + 8 * (r // 4) + 4 * (lane // 32)


W_SHIFT = 8          # kWShift of csrc/car_fused.hip: packed fp16 weights carry a factor 2^8


def _pack_tiles_f16_split(W: Tensor, n_tiles: int, chunk_k: Tensor) -> Tensor:
    """Split-fp16 operand tiles of the f16 matrix pipe: per (chunk, tile) [K group (2)][hi | lo][lane (64)][8 halves],
    returned as float32 words (same 1024 floats per tile as the fp32 packing).  w * 2^W_SHIFT = hi + lo with hi = fp16(.),
    lo = fp16(. - hi): the scale keeps the low halves out of the fp16 subnormal range; the kernel undoes it exactly."""
    N, K = W.shape
    Wext = torch.zeros(32 * n_tiles, K + 1, dtype=torch.float32)
    Wext[:N, :K] = W * float(1 << W_SHIFT)
    chunks = chunk_k.shape[0]
    k = chunk_k.clamp(max=K)                                                 # (chunks, 64, 16); k >= K -> zero column
    lane = torch.arange(64)
    n = 32 * torch.arange(n_tiles)[:, None] + (lane % 32)[None, :]           # (tiles, 64)
    w = Wext[n[None, :, :, None].expand(chunks, -1, -1, 16), k[:, None, :, :].expand(-1, n_tiles, -1, -1)]   # (chunks,tiles,64,16)
    hi = w.half()
    lo = (w - hi.float()).half()
    both = torch.stack([hi, lo], dim=2)                                      # (chunks, tiles, hl, 64, 16)
    both = both.reshape(chunks, n_tiles, 2, 64, 2, 8).permute(0, 1, 4, 2, 3, 5).contiguous()   # -> (.., kg, hl, lane, 8)
    return both.view(torch.float32).reshape(chunks, n_tiles, 1024)


def pack_fused_weights(m, device, split_fp16: bool = False):
    """Weights of the fused per-sample kernel (csrc/car_fused.hip) in its operand order: (blob, bias table)."""
    f = lambda t: t.detach().float().cpu().reshape(t.shape[0], -1)
    v = lambda t: t.detach().float().cpu()
    C = m.query_encode_latent.weight.shape[0]
    E2 = C // 2
    wr = f(m.query_repeat_embed.weight)
    w2 = f(m.query_encode_latent_2.weight)
    # 128-output chained layers: fp32 tiles, or fp16 hi/lo tiles when the f16 matrix pipe is used
    pk4 = (lambda W, ck: _pack_tiles_f16_split(W, 4, ck)) if split_fp16 else (lambda W, ck: _pack_tiles(W, None, 4, ck))
    parts = [
        _pack_tiles_f16_split(w2, E2 // 32, _std_k(C // 32)) if split_fp16 else _pack_tiles(w2, None, E2 // 32, _std_k(C // 32)),   # W2
        _pack_tiles(f(m.query_embed.weight), v(m.query_embed.bias), 4, _std_k(1)),                            # Q1 (bias folded)
        pk4(f(m.query_embed_2.weight), _chained_k(4)),                                                        # Q2
        _pack_tiles(wr[:, 128:].contiguous(), v(m.query_repeat_embed.bias), 4, _std_k(1)),                    # UG (bias folded)
        torch.cat([pk4(f(m.key_map.weight), _chained_k(E2 // 32, base=E2 * sv)) for sv in range(2)]),         # K1
        pk4(f(m.key_map_2.weight), _chained_k(4)),                                                            # K2
    ]
    blob = torch.cat([p_.reshape(-1) for p_ in parts]).to(device)
    bias = torch.cat([v(m.query_encode_latent_2.bias), v(m.query_embed_2.bias), v(m.key_map.bias), v(m.key_map_2.bias)]).to(device)
    return blob, bias


def _std16_k(ksteps: int) -> Tensor:
    """16x16x32 tiles, standard mapping: K step m, lane l, element e -> k = 32 m + 8 (l >> 4) + e"""
    m = torch.arange(ksteps)[:, None, None]
    lane = torch.arange(64)[None, :, None]
    e = torch.arange(8)[None, None, :]
    return 32 * m + 8 * (lane // 16) + e


def _chained16_k(ksteps: int, base: int = 0) -> Tensor:
    """16x16x32 tiles chained over the accumulators of 16-row source tiles (channel 16 T + 4 (l >> 4) + r): K step m takes
    source tiles 2m and 2m+1, so  k = base + 16 (2 m + e // 4) + 4 (l >> 4) + e % 4"""
    m = torch.arange(ksteps)[:, None, None]
    lane = torch.arange(64)[None, :, None]
    e = torch.arange(8)[None, None, :]
    return base + 16 * (2 * m + e // 4) + 4 * (lane // 16) + e % 4


def _pack_tiles16_f16_split(W: Tensor, bias: Optional[Tensor], n_tiles: int, kmap: Tensor) -> Tensor:
    """Split-fp16 A-operand tiles of v_mfma_f32_16x16x32_f16: per (K step, tile) [hi | lo][lane (64)][8 halves] = 512 float32
    words; lane l carries output channel 16 tile + l % 16 and the eight k of ``kmap[step, l]`` (k == K: the bias, k > K: zero).
    Weights carry 2^W_SHIFT like _pack_tiles_f16_split."""
    N, K = W.shape
    Wext = torch.zeros(16 * n_tiles, K + 2, dtype=torch.float32)
    Wext[:N, :K] = W * float(1 << W_SHIFT)
    if bias is not None:
        Wext[:N, K] = bias * float(1 << W_SHIFT)
    ks = kmap.shape[0]
    k = kmap.clamp(max=K + 1)                                                # (ks, 64, 8)
    lane = torch.arange(64)
    n = 16 * torch.arange(n_tiles)[:, None] + (lane % 16)[None, :]           # (tiles, 64)
    w = Wext[n[None, :, :, None].expand(ks, -1, -1, 8), k[:, None, :, :].expand(-1, n_tiles, -1, -1)]      # (ks, tiles, 64, 8)
    hi = w.half()
    lo = (w - hi.float()).half()
    both = torch.stack([hi, lo], dim=2).contiguous()                         # (ks, tiles, hl, 64, 8)
    return both.view(torch.float32).reshape(ks, n_tiles, 512)


def pack_fused2_weights(m, device):
    """Weights of csrc/car_fused2.hip (16x16x32 f16 tiles, every layer split-fp16) in its operand order: (blob, bias table)."""
    f = lambda t: t.detach().float().cpu().reshape(t.shape[0], -1)
    v = lambda t: t.detach().float().cpu()
    C = m.query_encode_latent.weight.shape[0]
    E2 = C // 2
    wr = f(m.query_repeat_embed.weight)
    pk = _pack_tiles16_f16_split
    parts = [
        pk(f(m.query_encode_latent_2.weight), None, E2 // 16, _std16_k(C // 32)),                             # W2
        pk(f(m.query_embed.weight), v(m.query_embed.bias), 8, _std16_k(1)),                                   # Q1 (bias folded)
        pk(f(m.query_embed_2.weight), None, 8, _chained16_k(4)),                                              # Q2
        pk(wr[:, 128:].contiguous(), v(m.query_repeat_embed.bias), 8, _std16_k(1)),                           # UG (bias folded)
        torch.cat([pk(f(m.key_map.weight), None, 8, _chained16_k(E2 // 32, base=E2 * sv)) for sv in range(2)]),   # K1
        pk(f(m.key_map_2.weight), None, 8, _chained16_k(4)),                                                  # K2
    ]
    blob = torch.cat([p_.reshape(-1) for p_ in parts]).to(device)
    bias = torch.cat([v(m.query_encode_latent_2.bias), v(m.query_embed_2.bias), v(m.key_map.bias), v(m.key_map_2.bias)]).to(device)
    return blob, bias


def pack_round2_weights(m, device):
    """query_repeat_embed_2 (128 -> 128) in the operand order of csrc/car_round2.hip: (packed [4,4,1024], bias [128])."""
    W = m.query_repeat_embed_2.weight.detach().float().cpu().reshape(128, 128)
    return (_pack_tiles_f16_split(W, 4, _std_k(4)).reshape(-1).to(device),
            m.query_repeat_embed_2.bias.detach().float().to(device).contiguous())


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
        self._steps: Dict[tuple, Tensor] = {}
        self.linear_flags = 0          # tests may set NO_GLDS for A/B
        self.timing = None             # bench: dict layer name -> [(start, end) HIP events on the launch stream]
        self.pose_records = None       # tests: (b*V, 96) CarPose records to use instead of the host pose algebra
        # first point-MLP layer as a gather over per-texel pre-projected maps (csrc/car_encode.hip) instead of a
        # K=579 GEMM per sample; False selects the literal gather -> GEMM pipeline (A/B and stage tests)
        self.project_maps = True
        # geometry + encode + 576->288 + key/query MLPs + logits as one kernel (csrc/car_fused.hip); needs V == 2, three
        # pyramid levels, C == 576.  False keeps the stage-by-stage pipeline (A/B and stage tests)
        self.fuse_samples = True
        # 576->288 layer of the fused kernel on the f16 matrix pipe with fp16 hi/lo operand splits (3 products per term,
        # fp32-class accuracy) instead of the fp32 pipe
        self.split_fp16 = True
        # 4: csrc/car_fused4.hip (as 2 with 12 waves x 16 samples, three waves per SIMD, same packed weights);
        # 2: csrc/car_fused2.hip (8 waves x 16 samples, 16x16x32 f16 tiles, two waves per SIMD; always split-fp16);
        # 1: csrc/car_fused.hip (4 waves x 32 samples, one wave per SIMD; fp32 or split-fp16 per ``split_fp16``)
        self.fused_version = 4
        self.fuse_round2 = True        # round-2 per-sample layer + logits in one kernel (csrc/car_round2.hip)
        self._round2_key = None
        self._round2 = None
        self._fused_key = None
        self._fused = None
        self._pose_key = None
        self._pose_dev = None
        self._pose_src = None
        self._gmaps_key = None
        self._gmaps: List[Tensor] = []
        self._wpt: Optional[Tensor] = None

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
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in z)
        if key != self._maps_key:
            self._maps = [t.detach().float().permute(0, 2, 3, 1).contiguous() for t in z]
            self._maps_key = key
        return self._maps

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
        """Device pose records for this input.  The host pose algebra needs the camera matrices on the CPU (one small D2H
        sync); a frame is rendered as several chunks with the SAME cameras (render_realestate10k_traj.py:118-137), so the
        records are cached on the identity/version of the four camera tensors and the sync happens once per frame."""
        if self.pose_records is not None:
            poses = self.pose_records.float().contiguous()
            if tuple(poses.shape) != (n, 96):
                raise ValueError(f"pose records must have shape ({n}, 96)")
            return poses.to(dev, non_blocking=True)
        ts = (inp["context"]["cam2world"], inp["context"]["intrinsics"], inp["query"]["cam2world"], inp["query"]["intrinsics"])
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in ts) + (H, str(dev))
        if key != self._pose_key:
            self._pose_dev = pack_poses(inp, H).to(dev, non_blocking=True)
            self._pose_key = key
            self._pose_src = ts                      # keep the tensors alive so data_ptr cannot be recycled
        return self._pose_dev

    def _linspace(self, a: float, b_: float, P: int, device) -> Tensor:
        k = (a, b_, P, str(device))
        if k not in self._steps:
            self._steps[k] = torch.linspace(a, b_, P).to(device)      # CPU linspace, like the reference's values
        return self._steps[k]

    # ------------------------------------------------------------------ kernels
    def linear(self, x: Tensor, ldx: int, layer: PackedLinear, y: Tensor, ldy: int, M: int, flags: int = 0):
        ev = None
        if self.timing is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        _lib.check(self.lib.car_linear(_ptr(x), ldx, _ptr(layer.packed), layer.K, layer.N, _ptr(y), ldy, M,
                                       flags | self.linear_flags, _stream()), "car_linear")
        if ev is not None:
            ev[1].record()
            self.timing.setdefault(layer.name, []).append(
                (ev[0], ev[1], 2.0 * M * (layer.K + 1) * layer.N, f"linear_kernel {layer.name}: {layer.K}->{layer.N} on {M} rows"))

    def gather(self, maps: List[Tensor], grid: Tensor, pts: int, mode: int, place: int, V: int, out: Tensor,
               ld_out: int, col_out: int):
        L = len(maps)
        ptrs = (ctypes.c_void_p * L)(*[m.data_ptr() for m in maps])
        cs = (ctypes.c_int * L)(*[m.shape[3] for m in maps])
        hs = (ctypes.c_int * L)(*[m.shape[1] for m in maps])
        ws = (ctypes.c_int * L)(*[m.shape[2] for m in maps])
        _lib.check(self.lib.car_gather_bilinear(ptrs, cs, hs, ws, L, maps[0].shape[0], _ptr(grid), pts, mode, place, V,
                                                _ptr(out), ld_out, col_out, _stream()), "car_gather_bilinear")

    # ------------------------------------------------------------------ the forward pass
    @torch.no_grad()
    def render(self, inp, z: List[Tensor], debug: bool = False) -> Dict[str, Tensor]:
        m, lib = self.m, self.lib
        ctx, qry = inp["context"], inp["query"]
        uv_in = qry["uv"]
        dev = uv_in.device
        if dev.type != "cuda":
            raise RuntimeError("CrossAttentionRenderer.forward runs on the HIP engine only: move the model, the input "
                               "dict and z to a ROCm device (there is no CPU fallback)")
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
        pk = self._weights(dev)
        maps = self._channel_last(z)
        C = sum(t.shape[3] for t in maps)
        Dl = m.latent_dim

        # a3: pose algebra on the host, exactly the reference's torch calls
        poses = self._poses(inp, H, n, dev)
        uv = uv_in.detach().reshape(b, R, 2).float().contiguous()
        steps = self._linspace(0.1, 10.0, P, dev) if m.no_sample else self._linspace(0.0, 1.0, P, dev)

        # a4-a6: rays
        rays = torch.empty(n, R, 12, **f32)
        coords9 = torch.empty(n, R, 9, **f32)
        ld_phi = _round_up(9 * V, 4)
        phi_x = torch.zeros(b * R, ld_phi, **f32)
        _lib.check(lib.car_ray_setup(_ptr(poses), _ptr(uv), b, V, R, H, W, P, int(m.no_sample), _ptr(steps),
                                     _ptr(rays), _ptr(coords9), _ptr(phi_x), ld_phi, st), "car_ray_setup")

        concat2 = (V == 2 and not m.no_latent_concat)
        fused = (self.fuse_samples and self.project_maps and concat2 and not m.no_sample and len(maps) == 3 and C == 576
                 and m.hidden_dim == 128)
        if fused:
            return self._render_fused(inp, z, maps, poses, rays, coords9, phi_x, ld_phi, steps, b, V, R, P, H, W, debug)

        # a6, a8, a9, a13: samples
        pixel_val = torch.empty(n, R, P, 2, **f32)
        pt = torch.empty(n, R, P, 3, **f32)
        g = torch.empty(S, 16, **f32)
        single = (V == 1 and not m.no_latent_concat)
        concat3 = (V == 3 and not m.no_latent_concat)
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
            self.gather(maps, pixel_val, R * P, 0, PLACE_OWN, V, x1, ld1, 0)
            gi = grid_in.view(b, V, R, P, V, 2)
            # pixel_val_stack (models.py:316): map (b, s) is sampled where the *other* line's points land in view s
            grid_other = torch.stack([gi[:, 1, :, :, 0], gi[:, 0, :, :, 1]], dim=1).contiguous()
            self.gather(maps, grid_other, R * P, 1, PLACE_OTHER2, V, x1, ld1, 0)
            h1 = torch.empty(S * V, C, **f32)
            self.linear(x1, ld1, pk["query_encode_latent"], h1, C, S * V, RELU_OUT)
            e = torch.empty(S, V * (C // 2), **f32)
            self.linear(h1, C, pk["query_encode_latent_2"], e, C // 2, S * V)
            del h1
            Ce = V * (C // 2)
        elif concat3:
            e = self._encode_three_views(maps, poses, pixel_val, x1, pt_in, b, R, P, H, W, C, pk)
            Ce = 3 * (C // 2)
        elif single:
            self.gather(maps, pixel_val, R * P, 0, PLACE_PLAIN, V, x1, ld1, 0)
            e = torch.empty(S, C, **f32)
            self.linear(x1, ld1, pk["update_val_merge"], e, C, S)
            Ce = C
        else:
            e = torch.empty(S, C, **f32)
            self.gather(maps, pixel_val, R * P, 0, PLACE_PLAIN, V, e, C, 0)
            Ce = C
        del x1

        # a12: keys;  a13: geometric query.  The value projection (latent_value, no nonlinearity before the weighted sum)
        # commutes with the attention average: sum_s w_s (Wv e_s + bv) = Wv (sum_s w_s e_s) + bv because the softmax
        # weights of a ray sum to 1, so it is applied once per ray after the reduction instead of once per sample.
        k1 = torch.empty(S, 128, **f32)
        self.linear(e, Ce, pk["key_map"], k1, 128, S, RELU_OUT)
        key = torch.empty(S, 128, **f32)
        self.linear(k1, 128, pk["key_map_2"], key, 128, S)
        self.linear(g, 16, pk["query_embed"], k1, 128, S, RELU_OUT)
        q = torch.empty(S, 128, **f32)
        self.linear(k1, 128, pk["query_embed_2"], q, 128, S)

        return self._finish(inp, z, b, V, R, P, Ce, Dl, e, key, q, None, g, pt, pixel_val, poses, rays, coords9, phi_x, ld_phi,
                            debug)

    def _encode_three_views(self, maps, poses, pixel_val, ptenc, pt_in, b, R, P, H, W, C, pk):
        """Cross-view exchange for three context views (models.py:345-475), restated literally: for the samples of context c
        the row is the channel-interleaved concatenation of enc(own features, point in frame c) and, for every other view o
        in ascending order, enc(features of view o where context o's points — moved into frame c, projected with view o's
        intrinsics — land in image o, those points in frame c).  All arithmetic is in HIP kernels (gather, projection, the two
        1x1 layers); torch only moves rows into place."""
        V, pts = 3, R * P
        S = b * V * pts
        dev = pixel_val.device
        f32 = dict(device=dev, dtype=torch.float32)
        ld = _round_up(C + 3, 32)
        x3 = torch.empty(S * 3, ld, **f32)
        x3v = x3.view(b, V, pts, 3, ld)                     # [scene, context c, point, component k, channel]
        pe = ptenc.view(b, V, pts, V, 4)                    # tanh(nan_to_num(T_s pt)/5): [scene, context, point, frame s]
        pin = pt_in.view(b, V, pts, V, 3)
        tmp = torch.empty(S, C, **f32)
        self.gather(maps, pixel_val, pts, 0, PLACE_PLAIN, V, tmp, C, 0)
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
                self.gather(per_view[o], grid, pts, 1, PLACE_PLAIN, 1, tmp2, C, 0)
                x3v[:, c, :, k, :C] = tmp2.view(b, pts, C)
                x3v[:, c, :, k, C:C + 3] = pe[:, o, :, c, :3]
                k += 1
        h1 = torch.empty(S * 3, C, **f32)
        self.linear(x3, ld, pk["query_encode_latent"], h1, C, S * 3, RELU_OUT)
        enc = torch.empty(S * 3, C // 2, **f32)
        self.linear(h1, C, pk["query_encode_latent_2"], enc, C // 2, S * 3)
        # channel index = ch*3 + k (torch.cat on dim 2 then flatten(1, 2), models.py:446)
        return enc.view(S, 3, C // 2).permute(0, 2, 1).contiguous().view(S, 3 * (C // 2))

    def _render_fused(self, inp, z, maps, poses, rays, coords9, phi_x, ld_phi, steps, b, V, R, P, H, W, debug):
        m, lib = self.m, self.lib
        dev = poses.device
        f32 = dict(device=dev, dtype=torch.float32)
        S = b * V * R * P
        gmaps, wpt = self._projected_maps(maps, dev)
        key = tuple((p_.data_ptr(), p_._version) for p_ in (
            m.query_encode_latent_2.weight, m.query_encode_latent_2.bias, m.query_embed.weight, m.query_embed.bias,
            m.query_embed_2.weight, m.query_embed_2.bias, m.query_repeat_embed.weight, m.query_repeat_embed.bias,
            m.key_map.weight, m.key_map.bias, m.key_map_2.weight, m.key_map_2.bias)) + (str(dev), self.split_fp16, self.fused_version)
        version = self.fused_version
        # the 16x16x32 kernels address texel rows by 32-bit byte offsets inside a level: a level of 4 GiB or more (more than 14 scenes
        # of two 256x256 views in one call) goes through the first-generation kernel, which uses 64-bit addresses
        if version in (2, 4) and max(t.numel() * 4 for t in gmaps) >= 1 << 32:
            version = 1
        key = key[:-1] + (version,)
        v4 = version == 4
        v2 = version == 2 or v4
        if key != self._fused_key:
            self._fused = pack_fused2_weights(m, dev) if v2 else pack_fused_weights(m, dev, self.split_fp16)
            assert self._fused[0].numel() == (lib.car_fused2_blob_floats() if v2 else lib.car_fused_blob_floats())
            assert self._fused[1].numel() == lib.car_fused_bias_floats()
            self._fused_key = key
        blob, bias = self._fused
        e = torch.empty(S, 576, **f32)
        q = torch.empty(S, 128, **f32)
        ug = torch.empty(S, 128, **f32)
        logit = torch.empty(S, **f32)
        pt = torch.empty(S, 3, **f32)
        pixel_val = torch.empty(S, 2, **f32)
        L = 3
        ptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in gmaps])
        hs = (ctypes.c_int * L)(*[t.shape[1] for t in gmaps])
        ws = (ctypes.c_int * L)(*[t.shape[2] for t in gmaps])
        ev = None
        if self.timing is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if v2:
            _lib.check((lib.car_fused_samples_v4 if v4 else lib.car_fused_samples_v2)(_ptr(poses), _ptr(rays), _ptr(steps), ptrs, hs, ws, L, 576, _ptr(wpt), _ptr(blob),
                                                _ptr(bias), b, V, R, P, H, W, _ptr(e), _ptr(q), _ptr(ug), _ptr(logit), _ptr(pt),
                                                _ptr(pixel_val), _stream()), "car_fused_samples_v4" if v4 else "car_fused_samples_v2")
        else:
            _lib.check(lib.car_fused_samples(_ptr(poses), _ptr(rays), _ptr(steps), ptrs, hs, ws, L, 576, _ptr(wpt), _ptr(blob),
                                             _ptr(bias), b, V, R, P, H, W, _ptr(e), _ptr(q), _ptr(ug), _ptr(logit), _ptr(pt),
                                             _ptr(pixel_val), int(self.split_fp16), _stream()), "car_fused_samples")
        if ev is not None:
            ev[1].record()
            # algorithmic MACs per sample on the matrix pipe: 2 x 576x288 (e), 576x128 + 128x128 (key), 16x128 + 128x128 (qry),
            # 16x128 (ug); the gather FMAs and the geometry are not counted
            macs = 2 * 576 * 288 + 576 * 128 + 128 * 128 + 16 * 128 + 128 * 128 + 16 * 128
            pipe = "f16 matrix pipe, fp16 hi/lo split x3" if (self.split_fp16 or v2) else "fp32 matrix pipe"
            kname = "fused4_kernel" if v4 else "fused2_kernel" if v2 else "fused_sample_kernel"
            self.timing.setdefault("fused_samples", []).append(
                (ev[0], ev[1], 2.0 * S * macs, f"{kname} on {S} samples (e, key, qry, ug, logits; {pipe})"))
        return self._finish(inp, z, b, V, R, P, 576, m.latent_dim, e, None, q, logit, ug, pt, pixel_val, poses, rays, coords9,
                            phi_x, ld_phi, debug, ug_ready=True)

    def _finish(self, inp, z, b, V, R, P, Ce, Dl, e, key, q, logit, g_or_ug, pt, pixel_val, poses, rays, coords9, phi_x, ld_phi,
                debug, ug_ready=False):
        """Attention rounds, decoder and output dict.  Either (key, q) or precomputed round-1
        logits are given; ``g_or_ug`` is the geometric query g [S,16] or, with ug_ready, Wr1[:,128:] g + br1 [S,128]."""
        m, lib = self.m, self.lib
        dev = e.device
        f32 = dict(device=dev, dtype=torch.float32)
        st = _stream()
        pk = self._packed
        n, S = b * V, b * V * R * P
        n_qry = 1
        qry = inp["query"]
        k1 = torch.empty(S, 128, **f32) if not ug_ready else g_or_ug
        # a14 + a16: attention round 1, depth read-out
        at_wt = torch.empty(n, R, P, **f32)
        depth = torch.empty(b, R, **f32)
        amax = torch.empty(n, R, dtype=torch.int32, device=dev)
        rep = m.repeat_attention
        ebar = torch.empty(b * R, Ce, **f32)
        _lib.check(lib.car_attend(_ptr(key) if logit is None else _ptr(logit), _ptr(q) if logit is None else None, 128,
                                  _ptr(e), Ce, b, V, R, P, None, 0.0, _ptr(at_wt), _ptr(ebar),
                                  Ce, 1, _ptr(pt), _ptr(poses), _ptr(depth), _ptr(amax), st), "car_attend")
        zrep = torch.empty(b * R, V * Dl, **f32)
        at_wt2 = None
        if rep:
            z1 = torch.empty(b * R, Dl, **f32)
            self.linear(ebar, Ce, pk["latent_value"], z1, Dl, b * R)
            # a15: second round; the z_embed half of query_repeat_embed is per ray, the local_coords half per sample
            hb = torch.empty(b * R, 128, **f32)
            self.linear(z1, Dl, pk["encode_latent"], hb, 128, b * R)
            uh = torch.empty(b * R, 128, **f32)
            self.linear(hb, 128, pk["query_repeat_embed.h"], uh, 128, b * R)
            if not ug_ready:
                self.linear(g_or_ug, 16, pk["query_repeat_embed.g"], k1, 128, S)
            at_wt2 = torch.empty(n, R, P, **f32)
            if self.fuse_round2:
                # q2 = Wr2 relu(ug + uh) + b and <q2, qry>/16 in one kernel: q2 is never written
                w = m.query_repeat_embed_2.weight
                rk = (w.data_ptr(), w._version, m.query_repeat_embed_2.bias._version, str(dev))
                if rk != self._round2_key:
                    self._round2 = pack_round2_weights(m, dev)
                    self._round2_key = rk
                logit2 = torch.empty(S, **f32)
                _lib.check(lib.car_round2_logits(_ptr(k1), _ptr(uh), _ptr(q), _ptr(self._round2[0]), _ptr(self._round2[1]),
                                                 b, V, R, P, _ptr(logit2), st), "car_round2_logits")
                _lib.check(lib.car_attend(_ptr(logit2), None, 128, _ptr(e), Ce, b, V, R, P, None, 0.0, _ptr(at_wt2),
                                          _ptr(ebar), Ce, 1, None, None, None, None, st), "car_attend")
            else:
                if key is None:
                    key = torch.empty(S, 128, **f32)
                _lib.check(lib.car_add_ray_bias_relu(_ptr(k1), _ptr(uh), b, V, R, P, 128, st), "car_add_ray_bias_relu")
                self.linear(k1, 128, pk["query_repeat_embed_2"], key, 128, S)
                _lib.check(lib.car_attend(_ptr(key), _ptr(q), 128, _ptr(e), Ce, b, V, R, P, None, 0.0, _ptr(at_wt2),
                                          _ptr(ebar), Ce, 1, None, None, None, None, st), "car_attend")
            # z = (Wv ebar2 + bv) + V * z1   (models.py:561-565: "+ z_local" per view, then the view sum)
            zv = zrep.view(b * R, V, Dl)
            zv[:, 0] = z1 * float(V)
            self.linear(ebar, Ce, pk["latent_value"], zrep, V * Dl, b * R, ACCUM)
        else:
            self.linear(ebar, Ce, pk["latent_value"], zrep, V * Dl, b * R)
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
                             "local_coords": None if ug_ready else g_or_ug.view(n, R, P, 16), "interp_val": e.view(n, R, P, Ce),
                             "z_final": zrep[:, :Dl].reshape(b, R, Dl), "at_wt2": at_wt2, "poses": poses}
        return out
