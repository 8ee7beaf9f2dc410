"""Where does an announced pair's set-up go?  Frames of the bench with model.prefetch_pair on a side stream, on the launch stream, and
without it; per-frame wall times.  Usage (GPU box): python tools/prefetch_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cross_attention_renderer_amd.engine import RenderEngine  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    eng = model._engine = RenderEngine(model)
    _, z = bench.make_frame(0.5, dev)
    frames = bench.trajectory(12, dev, None, True)
    tile = torch.empty(1, 65536, 5, device=dev)
    pyr = (z, [t.clone() for t in z], [t.clone() for t in z])
    with torch.no_grad():
        bench.render_frame(model, frames[0], z, tile, 1 << 30)
        for mode in ("none", "side", "main", "side", "none"):
            eng.prefetch_side_stream = mode == "side"
            bench.render_frame(model, frames[0], pyr[0], tile, 1 << 30)
            torch.cuda.synchronize()
            times = []
            for i in range(9):
                t0 = time.perf_counter()
                if mode != "none":
                    model.prefetch_pair(pyr[(i + 1) % 3])
                bench.render_frame(model, frames[i], pyr[i % 3], tile, 1 << 30)
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) * 1e3)
            eng.drop_prefetched()
            print(f"prefetch {mode:5s}: " + " ".join(f"{t:6.2f}" for t in times) + f"   mean of the last 6: {sum(times[3:]) / 6:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
