"""Times the stand-alone epipolar gather stage (car_gather_bilinear: a7 + a10 of one 8192-ray chunk at the chunk's real epipolar
sample positions, as bench.py's ``gather_stage`` does) for the (rows per group, items in flight) configurations of the development
build (tools/build_dev.py: ``car_gather_bilinear_cfg``), and reports the HBM roofline fraction: algorithmic bytes = both gathered
tensors written + the pyramid once.
Usage (GPU box): python tools/bench_gather.py [cfg ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
from build_dev import build_dev  # noqa: E402
from cross_attention_renderer_amd import _lib  # noqa: E402
from cross_attention_renderer_amd.engine import RenderEngine  # noqa: E402

CFG = {0: "16 rows, 3 items (product)", 1: "32 rows, 3 items", 2: "64 rows, 3 items", 3: "16 rows, 4 items", 4: "32 rows, 4 items",
       5: "64 rows, 4 items", 6: "64 rows, 6 items", 7: "wave-tasks: 32 rows, a wave-instruction = whole rows of one level"}


def main():
    dev_lib = ctypes.CDLL(build_dev())
    fn = dev_lib.car_gather_bilinear_cfg
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] + _lib.SIGNATURES["car_gather_bilinear"][1]
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    eng = model._engine = RenderEngine(model)
    inp, z = bench.make_frame(0.5, dev)
    R, P, V = 8192, bench.P, bench.V
    maps = eng._channel_last(z)
    C = sum(t.shape[3] for t in maps)
    sub = {"context": inp["context"], "query": dict(inp["query"], uv=inp["query"]["uv"][:, :, 96 * 256:96 * 256 + R].contiguous())}
    with torch.no_grad():
        grid = model(sub, z=z)["pixel_val"].reshape(V, R * P, 2).contiguous()
    out = torch.empty(V * R * P, C, device=dev)
    L = len(maps)
    ptrs = (ctypes.c_void_p * L)(*[m.data_ptr() for m in maps])
    cs = (ctypes.c_int * L)(*[m.shape[3] for m in maps])
    hs = (ctypes.c_int * L)(*[m.shape[1] for m in maps])
    ws = (ctypes.c_int * L)(*[m.shape[2] for m in maps])
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nbytes = 2 * out.numel() * 4 + sum(t.numel() * 4 for t in maps)
    ref = None
    for cfg in [int(a) for a in sys.argv[1:]] or sorted(CFG):
        ev = []
        for _ in range(9):
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for mode in (0, 1):
                rc = fn(cfg, ptrs, cs, hs, ws, L, V, grid.data_ptr(), R * P, P, mode, 0, V, out.data_ptr(), C, 0, st)
                assert rc == 0
            b_.record()
            ev.append((a, b_))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b_) for a, b_ in ev[2:])[3]
        same = ""
        if ref is None:
            ref = out.clone()
        else:
            same = f"  equal to cfg 0: {torch.equal(ref, out)}"
        print(f"cfg {cfg} ({CFG[cfg]}): {ms:.3f} ms for a7 + a10, {nbytes / 1e9:.3f} GB -> {nbytes / ms / 1e9:.2f} TB/s = {nbytes / ms / 1e9 / 8.0 * 100:.1f} % of 8 TB/s{same}",
              flush=True)


if __name__ == "__main__":
    main()
