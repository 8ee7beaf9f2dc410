"""Times car_linear_wgrad on the training step's wide layers: the bf16 x 3 kernel (default for wide layers over >= 4096 rows) against the
fp32-pipe kernel (flag CAR_WGRAD_FP32 = 16).  usage: python tools/bench_wgrad.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cross_attention_renderer_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for M, N, K, bias in ((589824, 576, 579, True), (589824, 288, 576, True), (294912, 128, 576, True), (294912, 128, 864, False), (589824, 576, 576, False)):
    ldx = (K + 3) // 4 * 4
    dy = torch.randn(M, N, device=dev)
    x = torch.randn(M, ldx, device=dev)
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev) if bias else None
    res = {}
    for name, flags in (("bf16 x 3", 0), ("fp32 pipe", 16)):
        ev = []
        for i in range(8):
            dw.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(lib.car_linear_wgrad(P(dy), N, P(x), ldx, M, N, K, flags, P(dw), K, P(db) if bias else None, st), "wgrad")
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev[2:])
        res[name] = (ms[len(ms) // 2], dw.clone())
    flop = 2.0 * M * N * (K + (1 if bias else 0))
    d = (res["bf16 x 3"][1] - res["fp32 pipe"][1]).abs().max().item() / res["fp32 pipe"][1].abs().max().item()
    print(f"M={M} N={N} K={K}{'+b' if bias else ''}: bf16 x 3 {res['bf16 x 3'][0]:.3f} ms ({flop / res['bf16 x 3'][0] / 1e9:.0f} TFLOP/s)   "
          f"fp32 pipe {res['fp32 pipe'][0]:.3f} ms ({flop / res['fp32 pipe'][0] / 1e9:.0f} TFLOP/s)   max |diff| / max {d:.2e}", flush=True)
