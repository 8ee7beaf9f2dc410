"""Where a training step's wall time goes (reference shape: 12 scenes x 192 rays, 256 x 256, 64 samples; training.py:92-136):
phases timed with a device synchronisation after each (diagnosis), then the plain loop without any (what a run pays).
Usage (GPU box): python tools/train_step_probe.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cross_attention_renderer_amd import harness, synthetic  # noqa: E402
from cross_attention_renderer_amd import training  # noqa: E402
from cross_attention_renderer_amd.training import render_train  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    H, b, R = 256, 12, 192
    model = bench.build_model(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    base = harness.to_device(synthetic.stereo_scene(H, b=b, seed=5), dev)          # cameras stay on the host
    cl = os.environ.get("CAR_PYRAMID", "channels_last") == "channels_last"         # the stand-in pyramid in torch.channels_last memory: no layout copies
    z = [(t.to(dev).contiguous(memory_format=torch.channels_last) if cl else t.to(dev)).requires_grad_(True) for t in synthetic.feature_maps(b, 2, H, seed=1)]
    opt = training.make_adam(params, 5e-5)
    zopt = training.make_adam(z, 5e-5)
    grid = synthetic.pixel_grid(H, H).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    gt = torch.rand(b, 1, R, 3, device=dev) * 2 - 1

    def step(sync):
        t = [time.perf_counter()]

        def mark():
            if sync:
                torch.cuda.synchronize()
            t.append(time.perf_counter())
        idx = torch.stack([torch.randperm(H * H, device=dev, generator=g)[:R] for _ in range(b)])
        inp = {"context": base["context"], "query": dict(base["query"], uv=grid[idx][:, None])}
        mark()
        out = render_train(model, inp, z=z)
        loss = (gt - out["rgb"]).abs().mean()
        mark()
        opt.zero_grad(); zopt.zero_grad()
        loss.backward()
        mark()
        torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
        opt.step(); zopt.step()
        mark()
        return [b_ - a for a, b_ in zip(t[:-1], t[1:])]
    step(True)                                                                    # the engine exists after the first call
    if os.environ.get("CAR_X3_MIN_ROWS"):                                          # development knob: row threshold of the split-fp16 layer kernel
        model._engine.linear_x3_min_rows = int(os.environ["CAR_X3_MIN_ROWS"])
    for _ in range(3):
        step(True)
    rows = [step(True) for _ in range(steps)]
    names = ["rays (device randperm)", "render_train forward + loss", "backward", "clip + Adam"]
    print("per phase, a synchronisation after each (ms): " + ", ".join(f"{n} {1e3 * sum(r[i] for r in rows) / steps:.1f}" for i, n in enumerate(names)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    torch.cuda.synchronize()
    print(f"plain loop: {(time.perf_counter() - t0) / steps * 1e3:.1f} ms per step ({b} scenes x {R} rays)")
    # host time only: how long the CPU needs to queue a step (no waiting)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    host = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    print(f"host time to queue a step: {host:.1f} ms")


if __name__ == "__main__":
    main()
