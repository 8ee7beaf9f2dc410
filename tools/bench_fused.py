"""Times the fused per-sample kernel (the engine's default csrc/car_fused4.hip; CAR_FUSED_VERSION=2 / 1: car_fused2.hip / car_fused.hip) alone (one 8192-ray chunk at 256x256x64) for each CAR_FUSED_ABLATE variant.  Timing only:
variants > 0 compute wrong results by construction.  Usage (GPU box): python tools/bench_fused.py [variants...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cross_attention_renderer_amd.engine import RenderEngine  # noqa: E402


def main():
    import __graft_entry__ as ge
    ge.build()
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    model._engine = RenderEngine(model)
    if "CAR_FUSED_VERSION" in os.environ:
        model._engine.fused_version = int(os.environ["CAR_FUSED_VERSION"])     # 1: car_fused.hip, 2: car_fused2.hip, 4: car_fused4.hip
    inp, z = bench.make_frame(0.5, dev)
    uv = inp["query"]["uv"][:, :, 96 * 256: 96 * 256 + 8192].contiguous()
    if "CAR_BENCH_TILE" in os.environ:                 # experiment: rays in 2-D tiles of (rows x cols) pixels instead of row strips
        th, tw = [int(x) for x in os.environ["CAR_BENCH_TILE"].split("x")]
        g = uv.view(1, 1, 32, 256, 2)                  # 32 image rows x 256 columns
        g = g.view(1, 1, 32 // th, th, 256 // tw, tw, 2).permute(0, 1, 2, 4, 3, 5, 6).reshape(1, 1, 8192, 2)
        uv = g.contiguous()
    chunk = {"context": inp["context"], "query": dict(inp["query"], uv=uv)}
    variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 4]
    with torch.no_grad():
        for v in variants:
            os.environ["CAR_FUSED_ABLATE"] = str(v)
            model(chunk, z=z)                           # warm-up (also packs weights / projects maps the first time)
            model._engine.timing = {}
            for _ in range(5):
                model(chunk, z=z)
            torch.cuda.synchronize()
            evs = model._engine.timing["fused_samples"]
            lat = sorted(a.elapsed_time(b) for a, b, *_ in evs)
            flop = evs[0][2]
            print(f"ABL={v}: fused kernel median {lat[len(lat)//2]:.3f} ms  min {lat[0]:.3f} ms  -> {flop / lat[len(lat)//2] / 1e9:.1f} TFLOP/s (nominal flops)")
    os.environ["CAR_FUSED_ABLATE"] = "0"


if __name__ == "__main__":
    main()
