"""Out-of-bounds harness (run by tests/test_oob_guard.py in a subprocess, one case per process): a stage entry of libcar_hip.so is
called twice on the same inputs — once on ordinary torch tensors, once with EVERY pointer argument in a buffer whose end (or start)
is the end (start) of mapped device memory (tests/host/guard_alloc.cpp: HIP virtual-memory API, the neighbouring addresses are
reserved but unmapped).  A read or write one byte outside any argument is a GPU page fault there — the process dies, the test
fails — whereas torch's caching allocator would have served it silently from a neighbouring tensor.  Outputs must be bit-identical
between the two calls.  Test infrastructure only.

usage: python tests/oob_runner.py <family | case>      (CAR_OOB_LIB=<path>: another build of car_linear_x3, for checking the harness)"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from cross_attention_renderer_amd import _lib as L  # noqa: E402

SRC = os.path.join(ROOT, "tests", "host", "guard_alloc.cpp")
OUT = os.path.join(ROOT, "tests", "host", "_build", "libguard_alloc.so")
dev = torch.device("cuda:0")


def helper():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or os.path.getmtime(SRC) > os.path.getmtime(OUT):
        tmp = f"{OUT}.{os.getpid()}.tmp"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", SRC, "-o", tmp])
        os.replace(tmp, OUT)
    h = ctypes.CDLL(OUT)
    h.guard_error.restype = ctypes.c_char_p
    return h


class Guarded:
    """A copy of tensor `t` flush against unmapped address space (tail: its end; else its start)."""

    def __init__(self, h, t: torch.Tensor, tail: bool):
        self.h, self.t = h, t.contiguous()
        self.bytes = self.t.numel() * self.t.element_size()
        assert self.bytes % 4 == 0
        user, base = ctypes.c_void_p(), ctypes.c_void_p()
        res, mapped, handle = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_ulonglong()
        rc = h.guard_alloc(ctypes.c_size_t(self.bytes), 1 if tail else 0, ctypes.byref(user), ctypes.byref(base), ctypes.byref(res),
                           ctypes.byref(mapped), ctypes.byref(handle))
        if rc != 0:
            print(f"SKIP guard_alloc failed: {h.guard_error(rc).decode()}")
            sys.exit(77)
        self.ptr, self.meta = user.value, (base, res, mapped, handle)
        assert h.guard_copy(ctypes.c_void_p(self.ptr), ctypes.c_void_p(self.t.data_ptr()), ctypes.c_size_t(self.bytes)) == 0

    def read(self) -> torch.Tensor:
        out = torch.empty_like(self.t)
        assert self.h.guard_copy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(self.ptr), ctypes.c_size_t(self.bytes)) == 0
        return out

    def free(self):
        base, res, mapped, handle = self.meta
        self.h.guard_free(base, res, mapped, handle)


def run_both(call, tensors, outputs, tail):
    """call(ptr_of): invokes the entry, taking every pointer through ptr_of(name).  tensors: name -> torch tensor (inputs hold data,
    outputs their initial content).  Returns nothing; asserts the guarded call reproduces the plain one bit for bit."""
    h = helper()
    plain = {k: v.clone() for k, v in tensors.items()}
    call(lambda k: ctypes.c_void_p(plain[k].data_ptr()))
    torch.cuda.synchronize()
    guarded = {k: Guarded(h, v, tail) for k, v in tensors.items()}
    call(lambda k: ctypes.c_void_p(guarded[k].ptr))
    torch.cuda.synchronize()
    for k in outputs:
        a, b = plain[k], guarded[k].read()
        same = torch.equal(a, b) or torch.equal(torch.nan_to_num(a, nan=123.0), torch.nan_to_num(b, nan=123.0))
        assert same, f"{k}: guarded call differs from the plain one (max abs diff {(a.double() - b.double()).abs().max().item():.3g})"
    for k in tensors:
        if k not in outputs:
            assert torch.equal(tensors[k], guarded[k].read()), f"{k}: an input was written"
    for g in guarded.values():
        g.free()


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
         "e": torch.zeros(S, 576, device=dev), "qry": torch.zeros(S, 128, device=dev), "g": torch.zeros(S, 16, device=dev),
         "logit": torch.zeros(S, device=dev), "pt": torch.zeros(S, 3, device=dev), "pixel_val": torch.zeros(S, 2, device=dev),
         "part": torch.zeros(n * R * (-(-P // ts)), 576, device=dev)}
    outs = ["e", "qry", "g", "logit", "pt", "pixel_val", "part"]
    run_both(lambda p: L.check(lib.car_fused_samples_parts(p("poses"), p("rays"), p("steps"), p("lattice"), lh.value, lw.value, lpad.value, p("gmeta"),
                                                           p("wpt"), p("blob"), p("bias"), b, V, R, P, H, H, 0, p("e"), p("qry"), p("g"), p("logit"),
                                                           p("pt"), p("pixel_val"), p("part"), stream()), "car_fused_samples_parts"), t, outs, tail)


def case_tail_kernels(R, P, b, tail):
    """car_attend_parts, car_attend and car_round2_logits on the tensors a real forward left in its workspace."""
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
    t = {"g": _ws(eng, lib, d, "g"), "uh": _ws(eng, lib, d, "uh"), "qry": _ws(eng, lib, d, "qry"), "r2w": r2w.clone(), "r2b": r2b.clone(),
         "logit2": torch.zeros(S, device=dev)}
    run_both(lambda p: L.check(lib.car_round2_logits(p("g"), p("uh"), p("qry"), p("r2w"), p("r2b"), b, V, R, P, p("logit2"), stream()),
                               "car_round2_logits"), t, ["logit2"], tail)


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
}
# one process per family (a page fault takes the HIP context — and the rest of the family — with it; the last "RUN" line names the culprit)
FAMILIES = {
    "x3": ["x3_nt18", "x3_nt18_k579", "x3_nt8", "x3_nt4", "x3_nt2", "x3_nt2_long"],
    "linear": ["lin_579_576", "lin_16_128", "lin_128_3", "lin_7_5", "lin_glds_off"],
    "gather": ["gather_wave", "gather_quad", "gather_zeros"],
    "fused": ["fused_37_13", "fused_48_8_b2"],
    "tail": ["tail_37_13", "tail_96_32_b2"],
}

if __name__ == "__main__":
    fam = sys.argv[1]
    for name in FAMILIES.get(fam, [fam]):
        for where in ("tail", "head"):
            print(f"RUN {name} {where}", flush=True)
            CASES[name](where == "tail")
            print(f"OK {name} {where}", flush=True)
    print(f"DONE {fam}", flush=True)
