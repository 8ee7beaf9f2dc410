"""Writes tests/golden/trajectory_expected.npz (build container only) by calling the REFERENCE's own trajectory helpers and unposed-pair
input builder (dataset/load_video_superglue.py:33-111, 419-506) on seeded inputs.

``roma`` (rotation slerp), the SuperGlue matcher, cv2's essential-matrix solver and the image readers are not installed here: slerp is
provided to the reference by scipy's ``Slerp`` (an implementation independent of the one in cross_attention_renderer_amd/
trajectory.py), the matcher / pose solver are replaced by a stub that returns the seeded (R, t) — they are inputs of the function under
test, not part of it — and ``imread`` / ``imresize`` hand over seeded images.  Run:  python tests/golden/make_trajectory_golden.py"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402


def seeded_poses(seed):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(2):
        T = np.eye(4)
        T[:3, :3] = Rotation.from_rotvec(rng.normal(size=3) * 0.4).as_matrix()
        T[:3, 3] = rng.normal(size=3) * 0.5
        out.append(T)
    return np.stack(out)


def main():
    from scipy.spatial.transform import Rotation, Slerp
    roma = types.ModuleType("roma")

    def rotmat_slerp(R0, R1, steps):
        rots = Rotation.from_matrix(np.stack([R0.double().numpy(), R1.double().numpy()]))
        return torch.from_numpy(Slerp([0.0, 1.0], rots)(steps.double().numpy()).as_matrix())
    roma.rotmat_slerp = rotmat_slerp
    sys.modules["roma"] = roma
    ref_import._install_stubs()
    for name in ("imageio", "skimage", "skimage.transform", "skimage.color", "lpips", "h5py", "cv2", "scipy.io"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = ref_import._Anything(name)
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    ref = importlib.import_module("dataset.load_video_superglue")
    out = {}
    for k, seed in enumerate((1, 2, 3)):
        poses = seeded_poses(seed)
        out[f"poses{k}"] = poses
        out[f"linear{k}"] = ref.linear_interpolate(poses, 9)
        out[f"rotate{k}"] = ref.rotate_interpolate(poses, 24)
        d = poses[1][:3, 3] - poses[0][:3, 3]
        out[f"circle{k}"] = ref.make_circle(d / np.linalg.norm(d), 11, radius=0.04)
    # the unposed-pair dict: get_camera_pose(path, path2, uv) with the matcher replaced by the seeded (R, t)
    rng = np.random.default_rng(9)
    im = [rng.random((256, 300, 3)) for _ in range(2)]                     # landscape frames, cropped to 256 x 256 by the reference
    Rt = seeded_poses(4)[1]
    R, t = Rt[:3, :3], Rt[:3, 3] / np.linalg.norm(Rt[:3, 3])
    ref.imread = lambda path: im[int(path)]
    ref.imresize = lambda a, shape: a if a.shape[:2] == tuple(shape) else (_ for _ in ()).throw(AssertionError("resize needed"))
    ref.rgb2gray = lambda a: a.mean(-1)

    class Matching(torch.nn.Module):
        def __init__(self, config):
            super().__init__()

        def forward(self, d):
            k = torch.zeros(1, 8, 2)
            return {"keypoints0": k, "keypoints1": k, "matches0": torch.arange(8)[None], "matching_scores0": torch.ones(1, 8)}
    ref.Matching = Matching
    ref.estimate_pose = lambda *a, **k: (R, t, None)
    uv = torch.stack(torch.meshgrid(torch.arange(4.0), torch.arange(3.0), indexing="xy"), dim=-1)
    got = ref.get_camera_pose("0", "1", uv)
    out.update(R=R, t=t, uv=uv.numpy(), im_probe=np.stack([a[::64, ::64] for a in im]))     # the images are regenerated from the seed
    for part in ("query", "context"):
        for k, v in got[part].items():
            if k == "rgb":
                v = v[:, :, ::32, ::32]                                    # the images themselves are inputs: keep a strided probe
            out[f"dict.{part}.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "trajectory_expected.npz"), **out)
    print("wrote", len(out), "arrays;", {k: v.shape for k, v in out.items() if k.startswith("dict.")})


if __name__ == "__main__":
    main()
