"""Round-5 probe p1: what does the attention value reduction cost when the rows of `e` it streams are still on-die?
For slices of N rays (295 KB of e per ray) the SAME car_attend launch is timed (a) right after a kernel that has just WRITTEN the slice
(the rows sit in L2 / Infinity Cache as far as they fit) and (b) cold (2 GB written elsewhere in between).  The hot / cold ratio at slice
sizes below the 256 MB Infinity Cache bounds what any scheme that consumes e "while it is still on-die" across workgroups could gain.
usage: python tools/tail_probe.py"""
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cross_attention_renderer_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
V, P, D = 2, 64, 576
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P_ = lambda t: ctypes.c_void_p(t.data_ptr())
flush = torch.empty(512 * 1024 * 1024, device=dev)          # 2 GB
rows = []
for R in (54, 108, 216, 432, 864, 1728, 8192):
    S = V * R * P
    e = torch.randn(S, D, device=dev)
    logit = torch.randn(S, device=dev)
    w = torch.empty(S, device=dev)
    z = torch.empty(R, D, device=dev)

    def attend():
        _lib.check(lib.car_attend(P_(logit), None, 128, P_(e), D, 1, V, R, P, None, 0.0, P_(w), P_(z), D, 1, None, None, None, None, st), "car_attend")

    def timed(prep, reps=7):
        ts = []
        for _ in range(reps):
            prep()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); attend(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2]
    attend(); torch.cuda.synchronize()
    hot = timed(lambda: (flush.fill_(0.0), e.mul_(1.0)))       # flush, then a kernel that reads and WRITES every row of the slice
    cold = timed(lambda: (e.mul_(1.0), flush.fill_(0.0)))
    mb = S * D * 4 / 2**20
    rows.append((R, mb, hot, cold))
    print(f"R={R:5d}  e slice {mb:8.1f} MB  attend after its writer {hot * 1e3:8.1f} us = {S * D * 4 / hot / 1e9:7.2f} TB/s   cold {cold * 1e3:8.1f} us = {S * D * 4 / cold / 1e9:7.2f} TB/s   ratio {cold / hot:.2f}")
