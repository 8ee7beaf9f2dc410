"""Times one wide linear layer of the staged route on both matrix paths: car_linear (fp32 pipe) and car_linear_x3 (split fp16 x 3).
Usage (GPU box): python tools/bench_linear.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cross_attention_renderer_amd import _lib  # noqa: E402
from cross_attention_renderer_amd.engine import PackedLinear, _ptr  # noqa: E402


def main():
    import __graft_entry__ as ge
    ge.build()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M, K, N in ((589824, 576, 576), (589824, 576, 288), (589824, 288, 128), (589824, 128, 128), (1048576, 576, 576)):
        X = torch.randn(M, K, device=dev)
        W = torch.randn(N, K) / K ** 0.5
        layer = PackedLinear(W, torch.randn(N), dev)
        Y = torch.empty(M, N, device=dev)
        def run(fn, n=12):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, b in ev:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) for a, b in ev[2:])
            return t[len(t) // 2]
        tiles, bias = layer.x3
        t32 = run(lambda: lib.car_linear(_ptr(X), K, _ptr(layer.packed), K, N, _ptr(Y), N, M, 2, st))
        y32 = Y.clone()
        t = run(lambda: lib.car_linear_x3(_ptr(X), K, _ptr(tiles), _ptr(bias), K, N, _ptr(Y), N, M, 2, st))
        res = [f"fp32 pipe {t32:.3f} ms ({2e-9 * M * K * N / t32:.0f} TFLOP/s)", f"split fp16 x3 {t:.3f} ms ({2e-9 * M * K * N / t:.0f} TFLOP/s)"]
        err = ((Y - y32).abs().max() / y32.abs().max()).item()
        print(f"M={M} K={K} N={N}: " + "; ".join(res) + f"; max |x3 - fp32| / max {err:.1e}")


if __name__ == "__main__":
    main()
