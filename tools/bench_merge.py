"""Times car_merge_lattice on the bench pyramid's projected levels (2 maps, 64 / 128 / 256 wide, 576 channels: 2.5 GB of lattice).
usage: python tools/bench_merge.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cross_attention_renderer_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
n_maps, C = 2, 576
sizes = ((64, 64), (128, 128), (256, 256))
levels = [torch.randn(n_maps, h, w, C, device=dev) for h, w in sizes]
ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in levels])
hs = (ctypes.c_int * 3)(*[h for h, _ in sizes])
ws = (ctypes.c_int * 3)(*[w for _, w in sizes])
lh, lw, pad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.car_merge_lattice(ptrs, hs, ws, 3, n_maps, None, ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(pad), st), "shape")
lat = torch.empty(n_maps * 2 * lh.value * lw.value * C, device=dev)
ref = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    ev = []
    for i in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.car_merge_lattice(ptrs, hs, ws, 3, n_maps, ctypes.c_void_p(lat.data_ptr()), None, None, None, st), "merge")
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev[2:])
    if ref is None:
        ref = lat.clone()
    print(f"car_merge_lattice: median {ms[len(ms) // 2]:.3f} ms  min {ms[0]:.3f} ms  ({lat.numel() * 4 / ms[len(ms) // 2] / 1e9:.2f} TB/s written)  equal to the first: {torch.equal(lat, ref)}", flush=True)
