"""Trajectory rendering entry point (mirrors reference experiment_scripts/render_realestate10k_traj.py).

    python experiment_scripts/render_realestate10k_traj.py --experiment_name demo --views 2 --synthetic [--gpus N]

Per scene: ``z = model.get_z(...)`` once, then one chunked ``model(model_input, z=z)`` pass per trajectory frame
(render_realestate10k_traj.py:97, 118-145); frames are written as PNG + NPY (no mp4 writer in this image)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common  # noqa: E402


def render(rank, opt):
    import torch
    from cross_attention_renderer_amd import harness
    dev = common.init_rank(rank, opt)
    model = common.build_model(opt, dev)
    H = opt.img_sidelength
    if opt.data_root and not opt.synthetic:
        raise SystemExit("dataset readers are not built yet (SURVEY.md §8f row 3); use --synthetic")
    inp, z = harness.synthetic_pair(H, opt.views)
    inp, z = harness.to_device(inp, dev), [t.to(dev) for t in z]
    out_dir = opt.out_dir or os.path.join(opt.logging_root, opt.experiment_name, "renders")
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    for i, frame in enumerate(harness.trajectory(inp, opt.n_frames)):
        tile = harness.render_frame(model, frame, z, rank=rank, world=opt.gpus)
        if rank == 0:
            rgb = tile[0, :, :3].reshape(H, H, 3)
            harness.write_png(os.path.join(out_dir, f"frame_{i:04d}.png"), rgb)
            torch.save({"rgb": rgb.cpu(), "depth": tile[0, :, 3].reshape(H, H).cpu(), "valid": tile[0, :, 4].reshape(H, H).cpu()},
                       os.path.join(out_dir, f"frame_{i:04d}.pt"))
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.time() - t0
        print(f"rendered {opt.n_frames} frames of {H}x{H} in {dt:.2f} s ({opt.n_frames * H * H / dt:,.0f} rays/s) -> {out_dir}")


if __name__ == "__main__":
    opt = common.parser(__doc__).parse_args()
    common.spawn(render, opt)
