"""Writes the reader fixture (build container only): a tiny synthetic scene in the on-disk format of RealEstate10K
(tests/golden/dataio_scene/) and what the REFERENCE's ``get_camera_pose`` returns for it (dataio_expected.npz).
Run:  python tests/golden/make_dataio_golden.py"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

SCENE = os.path.join(HERE, "dataio_scene", "scene0")
POSES = os.path.join(HERE, "dataio_scene", "poses")


def write_scene(n_frames=7, seed=3):
    rng = np.random.default_rng(seed)
    stamps = rng.permutation(np.arange(n_frames) * 33367 + 100100)          # stored out of order on purpose
    frames = {f"{int(t)}.jpg": rng.integers(0, 256, size=(4, 6, 3), dtype=np.uint8) for t in stamps}
    os.makedirs(SCENE, exist_ok=True)
    os.makedirs(POSES, exist_ok=True)
    np.savez(os.path.join(SCENE, "data.npz"), **frames)
    with open(os.path.join(POSES, "scene0.txt"), "w") as f:
        f.write("https://www.youtube.com/watch?v=synthetic\n")
        for k, t in enumerate(sorted(int(x) for x in stamps)):
            a = 0.05 * k
            R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
            tvec = np.array([0.1 * k, -0.02 * k, 0.03 * k * k])
            w2c = np.concatenate([R, tvec[:, None]], axis=1).reshape(-1)
            vals = [0.48 + 0.001 * k, 0.86 + 0.002 * k, 0.5 + 0.003 * k, 0.5 - 0.002 * k, 0.0, 0.0] + list(w2c)
            f.write(str(t) + " " + " ".join(repr(float(v)) for v in vals) + "\n")


def main():
    write_scene()
    ref_import._install_stubs()
    for name in ("imageio", "skimage", "skimage.transform", "lpips", "h5py"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = ref_import._Anything(name)
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    ref = importlib.import_module("dataset.realestate10k_dataio")
    from pathlib import Path
    uv = torch.stack(torch.meshgrid(torch.arange(3.0), torch.arange(2.0), indexing="xy"), dim=-1)
    out = {}
    for views in (1, 2, 3):
        got = ref.get_camera_pose(Path(SCENE), POSES, uv, views=views)
        for part in ("query", "context"):
            for k, v in got[part].items():
                out[f"v{views}.{part}.{k}"] = v.numpy()
    np.savez(os.path.join(HERE, "dataio_expected.npz"), uv=uv.numpy(), **out)
    print("wrote", len(out), "arrays")
    make_vis(ref)


VIS_ROOT = os.path.join(HERE, "dataio_scene_vis")


def write_vis_scene(n_frames=50):
    """A scene in the evaluation layout: frames at the reference's working size 256 x 455 (a smooth pattern, so the npz stays small)
    and a .mat pose table keyed by scene name."""
    from scipy.io import savemat
    ys, xs = np.meshgrid(np.arange(256), np.arange(455), indexing="ij")
    frames, rows = {}, []
    for k in range(n_frames):
        t = 200200 + 33367 * k
        frames[f"{t}.png"] = np.stack([(xs // 16 * 3 + ys // 16 + 7 * k) % 256, (xs // 8 + ys // 32 * 2 + 11 * k) % 256, ((xs // 16) * (ys // 16) + 5 * k) % 256], axis=-1).astype(np.uint8)
        a = 0.03 * k
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        tv = np.array([0.05 * k, 0.01 * k, -0.02 * k])
        rows.append([t + 0.3, 0.49 + 0.001 * k, 0.87, 0.5, 0.5 + 0.001 * k, 0.0, 0.0] + list(np.concatenate([R, tv[:, None]], axis=1).reshape(-1)))
    os.makedirs(os.path.join(VIS_ROOT, "scenes", "sceneA"), exist_ok=True)
    np.savez_compressed(os.path.join(VIS_ROOT, "scenes", "sceneA", "data.npz"), **frames)
    savemat(os.path.join(VIS_ROOT, "poses.mat"), {"sceneA": np.asarray(rows, dtype=np.float64)})


def make_vis(ref):
    """What the reference's RealEstate10kVis returns for that scene (eval_realestate10k.py:101-105: augment=False), for 1 / 2 / 3
    context views and two seeds of the query draw."""
    import random
    write_vis_scene()
    out = {}
    for views in (1, 2, 3):
        ds = ref.RealEstate10kVis(img_root=os.path.join(VIS_ROOT, "scenes"), pose_root=os.path.join(VIS_ROOT, "poses.mat"),
                                  num_ctxt_views=views, num_query_views=1, augment=False)
        for seed in (0, 1):
            random.seed(seed)
            item, gt = ds[0]
            for part in ("query", "context"):
                for k, v in item[part].items():
                    v = np.asarray(v)
                    if k == "rgb":                       # bulky: a strided probe and a checksum
                        out[f"v{views}.s{seed}.{part}.rgb_sum"] = np.asarray([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum()])
                        v = v.reshape(-1, 3)[::997]
                    if k == "uv":
                        v = v.reshape(-1, 2)[::257]
                    out[f"v{views}.s{seed}.{part}.{k}"] = v
    np.savez(os.path.join(HERE, "dataio_vis_expected.npz"), **out)
    print("wrote", len(out), "arrays (vis)")


if __name__ == "__main__":
    main()
