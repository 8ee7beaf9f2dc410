"""Times the stand-alone epipolar gather stage (car_gather_bilinear) at the bench shape and reports its HBM roofline
fraction: algorithmic bytes = gathered rows written (V*P*C*4 per ray) + the unique feature-map bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import __graft_entry__ as ge
    ge.build()
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.engine import RenderEngine
    dev = torch.device("cuda:0")
    H, P, R, V = 256, 64, 8192, 2
    eng = RenderEngine.__new__(RenderEngine)
    from cross_attention_renderer_amd import _lib
    eng.lib = _lib.load()
    maps = [t.permute(0, 2, 3, 1).contiguous().to(dev) for t in S.feature_maps(1, V, H, seed=1)]
    C = sum(m.shape[3] for m in maps)
    g = torch.Generator().manual_seed(0)
    # points along random segments inside the image, like epipolar samples
    a = torch.rand(V, R, 1, 2, generator=g) * 2 - 1
    b = torch.rand(V, R, 1, 2, generator=g) * 2 - 1
    grid = (a + (b - a) * torch.linspace(0, 1, P)[None, None, :, None]).reshape(V, R * P, 2).contiguous().to(dev)
    out = torch.empty(V * R * P, C, device=dev)
    for mode, name in ((0, "border"), (1, "zeros")):
        for _ in range(2):
            eng.gather(maps, grid, R * P, mode, 0, V, out, C, 0, run=P)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for s, e in ev:
            s.record(); eng.gather(maps, grid, R * P, mode, 0, V, out, C, 0, run=P); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in ev)[5]
        nbytes = out.numel() * 4 + sum(m.numel() * 4 for m in maps)
        print(f"gather {name}: {ms:.3f} ms per launch, {nbytes / 1e9:.3f} GB algorithmic -> {nbytes / ms / 1e9:.2f} TB/s = "
              f"{nbytes / ms / 1e9 / 8.0 * 100:.1f} % of the 8 TB/s HBM3E peak")


if __name__ == "__main__":
    main()
