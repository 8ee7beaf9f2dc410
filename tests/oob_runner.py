"""Out-of-bounds harness (tests/test_oob_guard.py; tools/oob_selfcheck.sh): a stage entry of libcar_hip.so is called twice on the same
inputs — once on ordinary torch tensors, once with EVERY pointer argument inside its own arena whose margins (1 MiB on either side of
the payload) are filled with NaNs.  A read outside an argument whose value reaches the result — torch's caching allocator normally
serves it silently from a neighbouring tensor, where 0 x neighbour happens to be 0 — brings a NaN into the outputs, which must be
bit-identical between the two calls; a write outside an argument changes a margin, which is checked bit for bit afterwards.
(A first version put the arguments against UNMAPPED address space with HIP's virtual-memory API instead: on this pool's boxes such
accesses do not fault — the old car_linear16.hip ran to completion there — and re-used address ranges served stale data; dropped.)
Test infrastructure only.

Second instrument, for reads whose value nobody uses (no margin can show them): the library built with -DCAR_BOUNDS (tools/build_bounds.py),
in which every LDS-DMA / buffer-load / row-load helper compares its source range with the extent its launcher passed and TRAPS outside it —
the process then dies in the middle of a case (a "RUN name" line without its "OK").  CAR_OOB_FULL_LIB=<path> runs every case on such a build.

usage: python tests/oob_runner.py <family | case>
       CAR_OOB_LIB=<path>       another build of car_linear_x3 only (checking the harness against a known bug)
       CAR_OOB_FULL_LIB=<path>  another build of the whole library (the -DCAR_BOUNDS build)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

from cross_attention_renderer_amd import _lib as L  # noqa: E402

if os.environ.get("CAR_OOB_FULL_LIB"):               # every entry from another build of the library (the package's loader is handed it)
    _full = ctypes.CDLL(os.environ["CAR_OOB_FULL_LIB"])
    for _name, (_res, _args) in L.SIGNATURES.items():
        _fn = getattr(_full, _name)
        _fn.restype, _fn.argtypes = _res, _args
    L._lib = _full

dev = torch.device("cuda:0")
MARGIN = 256 * 1024                    # elements (4 bytes each) on either side of the payload
NAN_BITS = 0x7fc00000


class Guarded:
    """A copy of tensor `t` (4-byte elements) in the middle of an arena whose margins hold the quiet-NaN bit pattern."""

    def __init__(self, t: torch.Tensor):
        self.t = t.contiguous()
        assert self.t.element_size() == 4
        n = self.t.numel()
        self.n = n
        pad = (-n) % 4                                                  # the far margin starts on a 16-byte boundary like the payload
        self.arena = torch.full((MARGIN + n + pad + MARGIN,), NAN_BITS, dtype=torch.int32, device=dev)
        self.arena[MARGIN:MARGIN + n] = self.t.reshape(-1).view(torch.int32)
        self.ptr = self.arena.data_ptr() + 4 * MARGIN

    def read(self) -> torch.Tensor:
        return self.arena[MARGIN:MARGIN + self.n].clone().view(self.t.dtype).view(self.t.shape)

    def margins_intact(self) -> bool:
        return bool((self.arena[:MARGIN] == NAN_BITS).all() and (self.arena[MARGIN + self.n:] == NAN_BITS).all())


def run_both(call, tensors, outputs, tail=True):
    """call(ptr_of): invokes the entry, taking every pointer through ptr_of(name).  tensors: name -> torch tensor (inputs hold data,
    outputs their initial content).  Asserts that the call on NaN-margined arenas reproduces the plain one bit for bit, leaves its
    inputs alone and writes nothing outside its arguments."""
    plain = {k: v.clone() for k, v in tensors.items()}
    call(lambda k: ctypes.c_void_p(plain[k].data_ptr()))
    torch.cuda.synchronize()
    guarded = {k: Guarded(v) for k, v in tensors.items()}
    call(lambda k: ctypes.c_void_p(guarded[k].ptr))
    torch.cuda.synchronize()
    for k in outputs:
        a, b = plain[k], guarded[k].read()
        assert torch.isfinite(a.float()).all() if a.dtype.is_floating_point else True, f"{k}: the plain call already holds non-finite values"
        if not torch.equal(a, b):
            bad = (a != b) | (b != b) if a.dtype.is_floating_point else (a != b)
            raise AssertionError(f"{k}: {int(bad.sum())} of {a.numel()} elements differ when the neighbours of every argument are NaN "
                                 f"({int((b != b).sum()) if b.dtype.is_floating_point else 0} NaN): a value from outside an argument reaches the result")
    for k in tensors:
        assert guarded[k].margins_intact(), f"{k}: bytes outside the argument were written"
        if k not in outputs:
            assert torch.equal(tensors[k], guarded[k].read()), f"{k}: an input was written"


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def load_lib():
    alt = os.environ.get("CAR_OOB_LIB")
    if not alt:
        return L.load()
    L.load()                                       # torch's HIP runtime first, and the product library's error plumbing
    lib = ctypes.CDLL(alt)
    for name in ("car_linear_x3", "car_linear_x3_pack", "car_linear_x3_packed_floats"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = L.SIGNATURES[name]
    return lib


# ---------------------------------------------------------------------------------------------------------------------------------
def case_linear_x3(M, K, N, flags, tail):
    lib = load_lib()
    g = torch.Generator().manual_seed(M + K + N)
    ldx, ldy = (K + 3) // 4 * 4, N
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    packed = torch.empty(lib.car_linear_x3_packed_floats(K, N), device=dev)
    assert lib.car_linear_x3_pack(ctypes.c_void_p(W.data_ptr()), K, K, N, ctypes.c_void_p(packed.data_ptr()), stream()) == 0
    torch.cuda.synchronize()
    t = {"X": torch.randn(M, ldx, generator=g).to(dev), "packed": packed, "bias": torch.randn(N, generator=g).to(dev),
         "Y": torch.randn(M, ldy, generator=g).to(dev)}
    run_both(lambda p: L.check(lib.car_linear_x3(p("X"), ldx, p("packed"), p("bias"), K, N, p("Y"), ldy, M, flags, stream()), "car_linear_x3"),
             t, ["Y"], tail)


def case_linear(M, K, N, flags, tail):
    lib = L.load()
    g = torch.Generator().manual_seed(M + K + N)
    ldx, ldy = (K + 3) // 4 * 4, (N + 3) // 4 * 4
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    packed = torch.empty(lib.car_linear_packed_floats(K, N), device=dev)
    L.check(lib.car_linear_pack(ctypes.c_void_p(W.data_ptr()), K, ctypes.c_void_p(b.data_ptr()), K, N, ctypes.c_void_p(packed.data_ptr()), stream()), "pack")
    torch.cuda.synchronize()
    t = {"X": torch.randn(M, ldx, generator=g).to(dev), "packed": packed, "Y": torch.randn(M, ldy, generator=g).to(dev)}
    run_both(lambda p: L.check(lib.car_linear(p("X"), ldx, p("packed"), K, N, p("Y"), ldy, M, flags, stream()), "car_linear"), t, ["Y"], tail)


def case_gather(chans, pts, mode, tail):
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    n = 3
    sizes = [(5, 7), (9, 6), (16, 16)][:len(chans)]
    maps = {f"map{l}": torch.randn(n, *sizes[l], c, generator=g).to(dev) for l, c in enumerate(chans)}
    grid = torch.rand(n, pts, 2, generator=g) * 2.6 - 1.3
    grid[0, 0] = torch.tensor([1e10, 1e10]); grid[0, 1] = torch.tensor([-1.0, 1.0]); grid[n - 1, pts - 1] = torch.tensor([1.0, 1.0])
    Ct = sum(chans)
    t = dict(maps, grid=grid.to(dev), out=torch.full((n * pts, Ct), -7.0, device=dev))
    nl = len(chans)
    cs = (ctypes.c_int * nl)(*chans)
    hs = (ctypes.c_int * nl)(*[s[0] for s in sizes])
    ws = (ctypes.c_int * nl)(*[s[1] for s in sizes])

    def call(p):
        ptrs = (ctypes.c_void_p * nl)(*[p(f"map{l}").value for l in range(nl)])
        L.check(lib.car_gather_bilinear(ptrs, cs, hs, ws, nl, n, p("grid"), pts, 1, mode, 0, 1, p("out"), Ct, 0, stream()), "car_gather_bilinear")
    run_both(call, t, ["out"], tail)


def _forward_state(R, P, b):
    """One real forward of the default configuration at ragged sizes; returns what the per-kernel cases need from it."""
    import cases as C
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    H = 64
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).eval()
    S.perturb_parameters(m, seed=4)
    m.H = m.W = H
    inp = S.stereo_scene(H, b=b, uv=C.select_rays(H, R), seed=7, alpha=0.35)
    z = [t.to(dev) for t in S.feature_maps(b, 2, H, seed=2)]
    md = m.to(dev)
    dinp = {k: {kk: (vv if kk in ("cam2world", "intrinsics") else vv.to(dev)) for kk, vv in v.items()} for k, v in inp.items()}
    with torch.no_grad():
        out = md(dinp, z=z, debug=True)
    torch.cuda.synchronize()
    return md, md._engine, z, out, dinp, H


def _ws(eng, lib, d, name):
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
    L.check(lib.car_workspace_find(ctypes.byref(d), name.encode(), ctypes.byref(off), ctypes.byref(cnt)), "car_workspace_find")
    return eng._work[off.value:off.value + cnt.value].clone()


def case_fused(R, P, b, tail):
    lib = L.load()
    md, eng, z, out, dinp, H = _forward_state(R, P, b)
    d = eng._dims(b, R, z)
    V, n = 2, 2 * b
    S = n * R * P
    sd = dict(md.named_parameters())
    w = L.CarWeights()
    keep = []
    for nme in L.WEIGHT_FIELDS[0]:
        for k, suffix in (("w", ".weight"), ("b", ".bias")):
            tt = sd[nme + suffix].detach().float()
            tt = tt.reshape(tt.shape[0], -1).contiguous() if tt.dim() > 1 else tt.contiguous()
            keep.append(tt)
            setattr(w, f"{nme.replace('.', '_')}_{k}", tt.data_ptr())
    blob = torch.empty(lib.car_fused_blob_floats(), device=dev)
    bias = torch.empty(lib.car_fused_bias_floats(), device=dev)
    wpt = torch.empty(576 * 4, device=dev)
    P_ = lambda t: ctypes.c_void_p(t.data_ptr())
    L.check(lib.car_fused_pack(ctypes.byref(w), P_(blob), P_(bias), P_(wpt), stream()), "car_fused_pack")
    pair = eng._pair
    lh, lw, lpad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.check(lib.car_lattice_shape(ctypes.byref(d), ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(lpad)), "shape")
    nlat = n * 2 * lh.value * lw.value * 576
    goff = lib.car_gmeta_offset(ctypes.byref(d))
    ts = lib.car_fused_tile_steps()
    torch.cuda.synchronize()
    t = {"poses": out["stages"]["poses"].to(dev).contiguous(), "rays": out["stages"]["rays"].contiguous(), "steps": torch.linspace(0, 1, P).to(dev),
         "lattice": pair[:nlat].clone(), "gmeta": pair[goff:goff + 4].clone(), "wpt": wpt, "blob": blob, "bias": bias,
         "e": torch.zeros(S, 576, device=dev), "g": torch.zeros(S, 16, device=dev),
         "logit": torch.zeros(S, device=dev), "pt": torch.zeros(S, 3, device=dev), "pixel_val": torch.zeros(S, 2, device=dev),
         "part": torch.zeros(n * R * (-(-P // ts)), 576, device=dev)}
    outs = ["e", "g", "logit", "pt", "pixel_val", "part"]
    run_both(lambda p: L.check(lib.car_fused_samples_parts(p("poses"), p("rays"), p("steps"), p("lattice"), lh.value, lw.value, lpad.value, p("gmeta"),
                                                           p("wpt"), p("blob"), p("bias"), b, V, R, P, H, H, 0, p("e"), p("g"), p("logit"),
                                                           p("pt"), p("pixel_val"), p("part"), stream()), "car_fused_samples_parts"), t, outs, tail)


def case_tail_kernels(R, P, b, tail):
    """car_attend_parts, car_attend, car_round2_logits and car_round2_logits_from_g on the tensors a real forward left in its workspace."""
    lib = L.load()
    md, eng, z, out, dinp, H = _forward_state(R, P, b)
    d = eng._dims(b, R, z)
    V, n = 2, 2 * b
    S = n * R * P
    ts = lib.car_fused_tile_steps()
    base = {"logit": _ws(eng, lib, d, "logit"), "part": _ws(eng, lib, d, "part"), "e": _ws(eng, lib, d, "e"), "pt": _ws(eng, lib, d, "pt"),
            "poses": out["stages"]["poses"].to(dev).contiguous()}
    outs = {"w": torch.zeros(S, device=dev), "z": torch.zeros(b * R, 576, device=dev), "depth": torch.zeros(b * R, device=dev),
            "amax": torch.zeros(n * R, dtype=torch.int32, device=dev)}
    t = {k: base[k] for k in ("logit", "part", "pt", "poses")}
    t.update({k: v.clone() for k, v in outs.items()})
    run_both(lambda p: L.check(lib.car_attend_parts(p("logit"), p("part"), ts, 576, b, V, R, P, p("w"), p("z"), 576, 1, p("pt"), p("poses"), p("depth"),
                                                    p("amax"), stream()), "car_attend_parts"), t, list(outs), tail)
    t = {k: base[k] for k in ("logit", "e", "pt", "poses")}
    t.update({k: v.clone() for k, v in outs.items()})
    run_both(lambda p: L.check(lib.car_attend(p("logit"), None, 128, p("e"), 576, b, V, R, P, None, 0.0, p("w"), p("z"), 576, 1, p("pt"), p("poses"),
                                              p("depth"), p("amax"), stream()), "car_attend"), t, list(outs), tail)
    r2w, r2b = eng._round2_weights(dev)
    torch.cuda.synchronize()
    qry = torch.randn(S, 128, device=dev, generator=torch.Generator(device=dev).manual_seed(3))      # stored query rows: the staged routes' form
    t = {"g": _ws(eng, lib, d, "g"), "uh": _ws(eng, lib, d, "uh"), "qry": qry, "r2w": r2w.clone(), "r2b": r2b.clone(),
         "logit2": torch.zeros(S, device=dev)}
    run_both(lambda p: L.check(lib.car_round2_logits(p("g"), p("uh"), p("qry"), p("r2w"), p("r2b"), b, V, R, P, p("logit2"), stream()),
                               "car_round2_logits"), t, ["logit2"], tail)
    r2qw = torch.empty(lib.car_round2q_packed_floats(), device=dev)
    r2qb = torch.empty(lib.car_round2q_bias_floats(), device=dev)
    ps = [x.detach().float().reshape(x.shape[0], -1).contiguous() for x in
          (md.query_repeat_embed.weight, md.query_repeat_embed.bias, md.query_repeat_embed_2.weight, md.query_repeat_embed_2.bias,
           md.query_embed.weight, md.query_embed.bias, md.query_embed_2.weight, md.query_embed_2.bias)]
    L.check(lib.car_round2q_pack(*[ctypes.c_void_p(x.data_ptr()) for x in ps], ctypes.c_void_p(r2qw.data_ptr()), ctypes.c_void_p(r2qb.data_ptr()),
                                 stream()), "car_round2q_pack")
    torch.cuda.synchronize()
    t = {"g": _ws(eng, lib, d, "g"), "uh": _ws(eng, lib, d, "uh"), "r2qw": r2qw, "r2qb": r2qb, "logit2": torch.zeros(S, device=dev)}
    run_both(lambda p: L.check(lib.car_round2_logits_from_g(p("g"), p("uh"), p("r2qw"), p("r2qb"), b, V, R, P, p("logit2"), stream()),
                               "car_round2_logits_from_g"), t, ["logit2"], tail)


def case_exchange(rows, N, tail=True):
    """car_lattice_encode_linear: lattice, row lists, point table, packed layer, bias, output."""
    lib = L.load()
    g = torch.Generator().manual_seed(rows + N)
    n_maps, C = 2, 576
    sizes = ((8, 8), (16, 16), (32, 32))
    levels = [torch.randn(n_maps, h, w, C, generator=g).to(dev) for h, w in sizes]
    ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in levels])
    hs = (ctypes.c_int * 3)(*[h for h, _ in sizes])
    wsz = (ctypes.c_int * 3)(*[w for _, w in sizes])
    lh, lw, pad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.check(lib.car_merge_lattice(ptrs, hs, wsz, 3, n_maps, None, ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(pad), stream()), "shape")
    lat = torch.empty(n_maps * 2 * lh.value * lw.value * C, device=dev)
    L.check(lib.car_merge_lattice(ptrs, hs, wsz, 3, n_maps, ctypes.c_void_p(lat.data_ptr()), None, None, None, stream()), "car_merge_lattice")
    src = (torch.randint(0, n_maps, (rows,), generator=g) | (torch.randint(0, 2, (rows,), generator=g) << 30)).to(torch.int32)
    src[-1] = (n_maps - 1) | (1 << 30)                              # the last row reads the last map's zeros lattice ...
    grid = torch.rand(rows, 2, generator=g) * 2.8 - 1.4
    grid[-1] = torch.tensor([0.999, 0.999])                         # ... next to its far corner
    W = (torch.randn(N, C, generator=g) / C ** 0.5).to(dev)
    tiles = torch.empty(lib.car_linear_x3_packed_floats(C, N), device=dev)
    L.check(lib.car_linear_x3_pack(ctypes.c_void_p(W.data_ptr()), C, C, N, ctypes.c_void_p(tiles.data_ptr()), stream()), "pack")
    torch.cuda.synchronize()
    t = {"lat": lat, "src": src.to(dev), "grid": grid.to(dev), "pe": torch.tanh(torch.randn(rows, 4, generator=g)).to(dev),
         "wpt": (torch.randn(C, 4, generator=g) * 0.1).to(dev), "tiles": tiles, "bias": torch.randn(N, generator=g).to(dev), "Y": torch.zeros(rows, N, device=dev)}
    run_both(lambda p: L.check(lib.car_lattice_encode_linear(p("lat"), lh.value, lw.value, pad.value, p("src"), p("grid"), p("pe"), p("wpt"), n_maps, rows,
                                                             p("tiles"), p("bias"), C, N, p("Y"), N, 0, stream()), "car_lattice_encode_linear"), t, ["Y"], tail)


def case_attend(R, P, V, D, tail=True):
    """car_attend on rows of D floats (864 = three views: the streaming reduction's last segment is half a wave wide)."""
    lib = L.load()
    g = torch.Generator().manual_seed(R + P + D)
    S = V * R * P
    t = {"qa": torch.randn(S, 128, generator=g).to(dev), "qb": torch.randn(S, 128, generator=g).to(dev), "val": torch.randn(S, D, generator=g).to(dev),
         "w": torch.zeros(S, device=dev), "z": torch.zeros(R, D, device=dev)}
    run_both(lambda p: L.check(lib.car_attend(p("qa"), p("qb"), 128, p("val"), D, 1, V, R, P, None, 0.0, p("w"), p("z"), D, 1, None, None, None, None, stream()),
                               "car_attend"), t, ["w", "z"], tail)


def case_fused_rows(R, P, nsets, tail=True):
    """car_fused_rows: lattice, row lists, packed layers, e."""
    lib = L.load()
    g = torch.Generator().manual_seed(R + P + nsets)
    n_maps, C, ncomp = 2, 576, 3
    sizes = ((8, 8), (16, 16), (32, 32))
    levels = [torch.randn(n_maps, h, w, C, generator=g).to(dev) for h, w in sizes]
    ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in levels])
    hs = (ctypes.c_int * 3)(*[h for h, _ in sizes])
    wsz = (ctypes.c_int * 3)(*[w for _, w in sizes])
    lh, lw, pad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.check(lib.car_merge_lattice(ptrs, hs, wsz, 3, n_maps, None, ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(pad), stream()), "shape")
    lat = torch.empty(n_maps * 2 * lh.value * lw.value * C, device=dev)
    L.check(lib.car_merge_lattice(ptrs, hs, wsz, 3, n_maps, ctypes.c_void_p(lat.data_ptr()), None, None, None, stream()), "car_merge_lattice")
    rows = nsets * R * P * ncomp
    src = torch.empty(nsets, R * P, ncomp, dtype=torch.int32)
    for a_ in range(nsets):
        for k in range(ncomp):
            src[a_, :, k] = ((a_ + k) % n_maps) | ((1 if k else 0) << 30)
    src[-1, :, -1] = (n_maps - 1) | (1 << 30)                      # the last set's last component: the LAST lattice of the buffer
    grid = torch.rand(rows, 2, generator=g) * 2.6 - 1.3
    grid[-1] = torch.tensor([0.999, 0.999])
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    blob = torch.zeros(lib.car_fused_blob_floats(), device=dev)
    fb = torch.empty(lib.car_fused_bias_floats(), device=dev)
    fwpt = torch.empty(C * 4, device=dev)
    ws = [rnd(C, C + 3) * 0.05, rnd(C) * 0.1, rnd(C // 2, C) / 24, rnd(C // 2)]
    L.check(lib.car_fused_pack_rows(*[ctypes.c_void_p(t.data_ptr()) for t in ws], ctypes.c_void_p(blob.data_ptr()), ctypes.c_void_p(fb.data_ptr()),
                                    ctypes.c_void_p(fwpt.data_ptr()), stream()), "car_fused_pack_rows")
    gmeta = lat.abs().max().reshape(1).contiguous()
    torch.cuda.synchronize()
    t = {"lat": lat, "gmeta": gmeta, "wpt": fwpt, "blob": blob, "bias": fb, "src": src.reshape(-1).to(dev), "grid": grid.to(dev),
         "pe": torch.tanh(torch.randn(rows, 4, generator=g)).to(dev), "e": torch.zeros(rows, C // 2, device=dev)}
    run_both(lambda p: L.check(lib.car_fused_rows(p("lat"), lh.value, lw.value, pad.value, p("gmeta"), p("wpt"), p("blob"), p("bias"), p("src"), p("grid"),
                                                  p("pe"), nsets, R, P, ncomp, p("e"), stream()), "car_fused_rows"), t, ["e"], tail)


def case_kq(M, Ce, tail=True):
    """car_key_query_logits: rows of e and g, the packed layers, qry and logit."""
    lib = L.load()
    g_ = torch.Generator().manual_seed(M + Ce)
    rnd = lambda *sh: torch.randn(*sh, generator=g_).to(dev)
    k1w = rnd(128, Ce) / Ce ** 0.5
    tiles = torch.empty(lib.car_linear_x3_packed_floats(Ce, 128), device=dev)
    L.check(lib.car_linear_x3_pack(ctypes.c_void_p(k1w.data_ptr()), Ce, Ce, 128, ctypes.c_void_p(tiles.data_ptr()), stream()), "pack")
    tailw = torch.empty(lib.car_kq_tail_floats(), device=dev)
    tb = torch.empty(lib.car_kq_bias_floats(), device=dev)
    ws = [rnd(128, 128) / 11, rnd(128), rnd(128, 16) / 4, rnd(128), rnd(128, 128) / 11, rnd(128)]
    L.check(lib.car_kq_pack(*[ctypes.c_void_p(t.data_ptr()) for t in ws], ctypes.c_void_p(tailw.data_ptr()), ctypes.c_void_p(tb.data_ptr()), stream()), "car_kq_pack")
    torch.cuda.synchronize()
    t = {"e": rnd(M, Ce), "tiles": tiles, "k1b": rnd(128), "g": rnd(M, 16), "tailw": tailw, "tb": tb, "qry": torch.zeros(M, 128, device=dev),
         "logit": torch.zeros(M, device=dev)}
    run_both(lambda p: L.check(lib.car_key_query_logits(p("e"), Ce, p("tiles"), p("k1b"), Ce, p("g"), p("tailw"), p("tb"), M, p("qry"), p("logit"), stream()),
                               "car_key_query_logits"), t, ["qry", "logit"], tail)


def case_wgrad(M, N, K, flags, tail=True):
    """car_linear_wgrad: dY, X read; dW, db accumulated (the bf16 x 3 kernel for wide layers over >= 4096 rows, else the fp32 pipe's)."""
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K)
    ldy, ldx = (N + 3) // 4 * 4, (K + 3) // 4 * 4
    t = {"dY": torch.randn(M, ldy, generator=g).to(dev), "X": torch.randn(M, ldx, generator=g).to(dev), "dW": torch.zeros(N, K, device=dev),
         "db": torch.zeros(N, device=dev)}

    def both(call_, t_, outs):
        # atomics make the sums' order run-dependent: compare to rounding, keep the margin / NaN checks exact
        plain = {k: v.clone() for k, v in t_.items()}
        call_(lambda k: ctypes.c_void_p(plain[k].data_ptr()))
        torch.cuda.synchronize()
        guarded = {k: Guarded(v) for k, v in t_.items()}
        call_(lambda k: ctypes.c_void_p(guarded[k].ptr))
        torch.cuda.synchronize()
        for k in outs:
            a, b = plain[k], guarded[k].read()
            assert torch.isfinite(b).all(), f"{k}: a NaN from outside an argument reached the result"
            assert (a - b).abs().max().item() <= 1e-4 * a.abs().max().item(), k
        for k in t_:
            assert guarded[k].margins_intact(), f"{k}: bytes outside the argument were written"
            if k not in outs:
                assert torch.equal(t_[k], guarded[k].read()), f"{k}: an input was written"
    both(lambda p: L.check(lib.car_linear_wgrad(p("dY"), ldy, p("X"), ldx, M, N, K, flags, p("dW"), K, p("db"), stream()), "car_linear_wgrad"),
         t, ["dW", "db"])


CASES = {
    "x3_nt18": lambda tail: case_linear_x3(200, 576, 576, 2, tail),
    "x3_nt18_k579": lambda tail: case_linear_x3(4097, 579, 288, 0, tail),
    "x3_nt8": lambda tail: case_linear_x3(333, 128, 128, 1, tail),
    "x3_nt4": lambda tail: case_linear_x3(77, 96, 64, 0, tail),
    "x3_nt2": lambda tail: case_linear_x3(100, 64, 32, 4, tail),
    "x3_nt2_long": lambda tail: case_linear_x3(640, 576, 96, 2, tail),
    "lin_579_576": lambda tail: case_linear(131, 579, 576, 2, tail),
    "lin_16_128": lambda tail: case_linear(513, 16, 128, 2, tail),
    "lin_128_3": lambda tail: case_linear(64, 128, 3, 1, tail),
    "lin_7_5": lambda tail: case_linear(50, 7, 5, 0, tail),
    "lin_glds_off": lambda tail: case_linear(300, 128, 128, 8, tail),
    "gather_wave": lambda tail: case_gather((256, 64, 8), 701, 0, tail),
    "gather_quad": lambda tail: case_gather((8, 12, 4), 333, 1, tail),
    "gather_zeros": lambda tail: case_gather((256, 64, 8), 64, 1, tail),
    "fused_37_13": lambda tail: case_fused(37, 13, 1, tail),
    "fused_48_8_b2": lambda tail: case_fused(48, 8, 2, tail),
    "tail_37_13": lambda tail: case_tail_kernels(37, 13, 1, tail),
    "tail_96_32_b2": lambda tail: case_tail_kernels(96, 32, 2, tail),
    "exchange_288": lambda tail: case_exchange(1000, 288, tail),
    "exchange_ragged": lambda tail: case_exchange(193, 128, tail),
    "attend_864": lambda tail: case_attend(37, 13, 3, 864, tail),
    "attend_100": lambda tail: case_attend(20, 8, 2, 100, tail),
    "fused_rows": lambda tail: case_fused_rows(37, 13, 2, tail),
    "kq_864": lambda tail: case_kq(4099, 864, tail),
    "kq_ragged": lambda tail: case_kq(193, 96, tail),
    "wgrad16_579": lambda tail: case_wgrad(4133, 576, 579, 1, tail),
    "wgrad16_ragged": lambda tail: case_wgrad(4100, 200, 130, 0, tail),
    "wgrad_fp32": lambda tail: case_wgrad(513, 576, 579, 0, tail),
    "wgrad_small": lambda tail: case_wgrad(777, 3, 128, 1, tail),
}
# families of cases (tools/oob_selfcheck.sh runs one family per process)
FAMILIES = {
    "x3": ["x3_nt18", "x3_nt18_k579", "x3_nt8", "x3_nt4", "x3_nt2", "x3_nt2_long"],
    "linear": ["lin_579_576", "lin_16_128", "lin_128_3", "lin_7_5", "lin_glds_off"],
    "gather": ["gather_wave", "gather_quad", "gather_zeros"],
    "fused": ["fused_37_13", "fused_48_8_b2"],
    "tail": ["tail_37_13", "tail_96_32_b2", "attend_864", "attend_100"],
    "exchange": ["exchange_288", "exchange_ragged", "fused_rows", "kq_864", "kq_ragged"],
    "wgrad": ["wgrad16_579", "wgrad16_ragged", "wgrad_fp32", "wgrad_small"],
}

if __name__ == "__main__":
    fam = sys.argv[1]
    failed = 0
    for name in FAMILIES.get(fam, [fam]):
        print(f"RUN {name}", flush=True)
        try:
            CASES[name](True)
            print(f"OK {name}", flush=True)
        except AssertionError as exc:
            failed += 1
            print(f"FAIL {name}: {exc}", flush=True)
    print(f"DONE {fam}: {failed} failed", flush=True)
    sys.exit(1 if failed else 0)
