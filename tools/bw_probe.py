"""Achievable HBM write / copy / read rates on this device with stock fill / copy / reduce kernels (context for the roofline
fractions of the write-dominated stages).  usage: python tools/bw_probe.py"""
import torch

dev = torch.device("cuda:0")
n = 1200 * 1024 * 1024            # 4.8 GB of floats
x = torch.empty(n, device=dev)
y = torch.empty(n, device=dev)


def timed(fn, reps=5):
    fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2] * 1e-3


t = timed(lambda: x.fill_(1.0))
print(f"fill  {4 * n / t / 1e12:.2f} TB/s written")
t = timed(lambda: y.copy_(x))
print(f"copy  {4 * n / t / 1e12:.2f} TB/s read + {4 * n / t / 1e12:.2f} TB/s written = {8 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: x.sum())
print(f"sum   {4 * n / t / 1e12:.2f} TB/s read")
