"""Trajectory rendering entry point (mirrors reference experiment_scripts/render_realestate10k_traj.py).

    python experiment_scripts/render_realestate10k_traj.py --experiment_name demo --views 2 --synthetic [--gpus N]
    python experiment_scripts/render_realestate10k_traj.py --experiment_name demo --views 2 --data_root SCENES --pose_root CAMERAS

Per scene: ``z = model.get_z(...)`` once, then one chunked ``model(model_input, z=z)`` pass per trajectory frame
(render_realestate10k_traj.py:97, 118-145); frames are written as PNG + NPY (no mp4 writer in this image)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common  # noqa: E402


def _scene_inputs(opt, H, dev, model):
    """(name, model_input with every query frame, z) per scene: real cameras from --data_root / --pose_root
    (cross_attention_renderer_amd/dataio.py) or one seeded synthetic pair."""
    import torch
    from cross_attention_renderer_amd import dataio, harness, synthetic
    if not opt.data_root or opt.synthetic:
        inp, z = harness.synthetic_pair(H, opt.views)
        frames = harness.trajectory(harness.to_device(inp, dev, opt.cameras), opt.n_frames)
        yield "synthetic", frames, [t.to(dev) for t in z]
        return
    if not opt.pose_root:
        raise SystemExit("--data_root needs --pose_root (directory of <scene>.txt camera files)")
    uv = synthetic.pixel_grid(H, H)
    for scene in sorted(p for p in os.listdir(opt.data_root) if os.path.isdir(os.path.join(opt.data_root, p))):
        full = dataio.get_camera_pose(os.path.join(opt.data_root, scene), opt.pose_root, uv, views=opt.views)
        scale = H / 256.0                                  # the reader's intrinsics are pixels of the 256 x 256 crop
        for part in ("query", "context"):
            full[part]["intrinsics"] = full[part]["intrinsics"].clone()
            full[part]["intrinsics"][..., :2, :3] *= scale
        full = harness.to_device(full, dev, opt.cameras)
        nq = min(opt.n_frames, full["query"]["cam2world"].shape[1])
        frames = [{"context": full["context"],
                   "query": {"cam2world": full["query"]["cam2world"][:, i:i + 1], "intrinsics": full["query"]["intrinsics"][:, i:i + 1],
                             "uv": full["query"]["uv"][:, i:i + 1].contiguous(), "rgb": full["query"]["rgb"][:, i:i + 1]}}
                  for i in range(nq)]
        # get_z on the context images when the encoder is built (checkpoint or --with_encoder) and the frames have the 256 x 256 the
        # multi-view encoder needs; otherwise the seeded synthetic pyramid (geometry and timing real, colours not)
        from cross_attention_renderer_amd.models import EncoderNotBuilt
        if not isinstance(model.encoder, EncoderNotBuilt) and tuple(full["context"]["rgb"].shape[2:4]) == (256, 256) and H == 256:
            with torch.no_grad():
                z = model.get_z(full)
        else:
            z = [t.to(dev) for t in synthetic.feature_maps(1, opt.views, H, seed=1)]
        yield scene, frames, z


def render(rank, opt):
    import torch
    from cross_attention_renderer_amd import harness
    dev = common.init_rank(rank, opt)
    model = common.build_model(opt, dev)
    H = opt.img_sidelength
    out_root = opt.out_dir or os.path.join(opt.logging_root, opt.experiment_name, "renders")
    t0, n_done = time.time(), 0
    for scene, frames, z in _scene_inputs(opt, H, dev, model):
        out_dir = out_root if scene == "synthetic" else os.path.join(out_root, scene)
        if rank == 0:
            os.makedirs(out_dir, exist_ok=True)
        for i, frame in enumerate(frames):
            tile = harness.render_frame(model, frame, z, rank=rank, world=opt.gpus)
            n_done += 1
            if rank == 0:
                rgb = tile[0, :, :3].reshape(H, H, 3)
                harness.write_png(os.path.join(out_dir, f"frame_{i:04d}.png"), rgb)
                torch.save({"rgb": rgb.cpu(), "depth": tile[0, :, 3].reshape(H, H).cpu(), "valid": tile[0, :, 4].reshape(H, H).cpu()},
                           os.path.join(out_dir, f"frame_{i:04d}.pt"))
                gt = frame["query"].get("rgb")
                if gt is not None and tuple(gt.shape[2:4]) == (H, H):
                    print(f"{scene} frame {i}: PSNR {harness.psnr((rgb.clamp(-1, 1) + 1) / 2, (gt[0, 0].clamp(-1, 1) + 1) / 2):.2f} dB")
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.time() - t0
        print(f"rendered {n_done} frames of {H}x{H} in {dt:.2f} s ({n_done * H * H / dt:,.0f} rays/s) -> {out_root}")


if __name__ == "__main__":
    opt = common.parser(__doc__).parse_args()
    common.spawn(render, opt)
