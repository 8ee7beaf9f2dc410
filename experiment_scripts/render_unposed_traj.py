"""Novel views from two images of unknown pose (mirrors reference experiment_scripts/render_unposed_traj.py).

    python experiment_scripts/render_unposed_traj.py --experiment_name demo --checkpoint_path model.pth --im1 a.png --im2 b.png --pose rt.npz

The reference estimates the relative pose of the second image with SuperPoint + SuperGlue + ``cv2.findEssentialMat`` /
``recoverPose`` (dataset/load_video_superglue.py:114-139, 421-460); the matcher's weights are not part of the reference tree and cv2
is not installed here, so (R, t) — ``recoverPose``'s convention, x_2 = R x_1 + t — comes from ``--pose`` (an .npz with ``R`` (3, 3) and
``t`` (3,)).  Everything after that is the reference's path: the two images centre-cropped to squares and scaled to [-1, 1], first camera =
world frame, second at inv([R | t]) with its position divided by 1.2, fixed RealEstate10K intrinsics, 76 query poses on a helix
between the two (``trajectory.unposed_pair_input``, pinned in tests/test_trajectory.py), ``get_z`` once, one chunked forward per
pose, frames written as PNG.  Images must already be 256 pixels high (the reference resizes with skimage, not installed here).
--synthetic renders the same trajectory over the seeded synthetic pair and feature pyramid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common  # noqa: E402


def _read_image(path):
    import numpy as np
    if path.endswith(".npy"):
        im = np.load(path)
    else:
        from PIL import Image
        im = np.asarray(Image.open(path).convert("RGB"))
    im = im.astype(np.float64) / 255.0 if im.dtype == np.uint8 else im.astype(np.float64)
    from cross_attention_renderer_amd.trajectory import center_crop_square
    im = center_crop_square(im)
    if im.shape[:2] != (256, 256):
        raise SystemExit(f"{path}: {im.shape[0]}x{im.shape[1]} after the centre crop; the renderer works at 256x256 — resize the image first")
    return im


def render(rank, opt):
    import numpy as np
    import torch
    from cross_attention_renderer_amd import harness, synthetic, trajectory
    dev = common.init_rank(rank, opt)
    H = 256
    opt.img_sidelength = H
    uv = synthetic.pixel_grid(H, H)
    if opt.synthetic or not (opt.im1 and opt.im2 and opt.pose):
        model = common.build_model(opt, dev)
        g = np.random.default_rng(0)
        yaw = np.deg2rad(-12.0)
        R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]]).T
        inp = trajectory.unposed_pair_input(g.random((H, H, 3)), g.random((H, H, 3)), R, -R @ np.array([0.6, 0.03, 0.05]), uv)
        z = [t.to(dev) for t in synthetic.feature_maps(1, 2, H, seed=1)]
        inp = harness.to_device(inp, dev, opt.cameras)
    else:
        model = common.build_model(opt, dev, with_encoder=True)
        rt = np.load(opt.pose)
        inp = harness.to_device(trajectory.unposed_pair_input(_read_image(opt.im1), _read_image(opt.im2), rt["R"], rt["t"], uv), dev, opt.cameras)
        with torch.no_grad():
            z = model.get_z(inp)
    out_dir = opt.out_dir or os.path.join(opt.logging_root, opt.experiment_name, "unposed")
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    nq = inp["query"]["cam2world"].shape[1] if not opt.n_frames else min(opt.n_frames, inp["query"]["cam2world"].shape[1])
    for i in range(nq):
        frame = {"context": inp["context"], "query": {"cam2world": inp["query"]["cam2world"][:, i:i + 1], "intrinsics": inp["query"]["intrinsics"][:, i:i + 1],
                                                       "uv": inp["query"]["uv"][:, i:i + 1].contiguous()}}
        tile = harness.render_frame(model, frame, z, rank=rank, world=opt.gpus)          # 8192-ray chunks (render_unposed_traj.py:73)
        if rank == 0:
            harness.write_png(os.path.join(out_dir, f"frame_{i:04d}.png"), tile[0, :, :3].reshape(H, H, 3))
    torch.cuda.synchronize()
    if rank == 0:
        print(f"rendered {nq} frames -> {out_dir}")


if __name__ == "__main__":
    p = common.parser(__doc__)
    p.add_argument("--im1", type=str, default=None)
    p.add_argument("--im2", type=str, default=None)
    p.add_argument("--pose", type=str, default=None, help=".npz with R (3,3), t (3,) of the second camera relative to the first")
    opt = p.parse_args()
    common.spawn(render, opt)
