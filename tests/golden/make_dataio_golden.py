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


if __name__ == "__main__":
    main()
