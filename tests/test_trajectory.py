"""cross_attention_renderer_amd/trajectory.py against vectors produced by the reference's own functions
(dataset/load_video_superglue.py:33-111, 419-506; tests/golden/make_trajectory_golden.py)."""
import os

import numpy as np
import torch

from cross_attention_renderer_amd import trajectory as T

FX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory_expected.npz"))


def test_interpolated_trajectories_match_the_reference():
    for k in range(3):
        poses = FX[f"poses{k}"]
        # rotations: float32 precision (the reference interpolates float32 copies with roma, which is not installed: the vectors
        # were made with scipy's Slerp); positions: float64, exact
        lin = T.linear_interpolate(poses, 9)
        np.testing.assert_allclose(lin[:, :3, :3], FX[f"linear{k}"][:, :3, :3], rtol=0, atol=2e-7)
        np.testing.assert_allclose(lin[:, :, 3], FX[f"linear{k}"][:, :, 3], rtol=0, atol=1e-12)
        got = T.rotate_interpolate(poses, 24)
        assert got.shape == (20, 4, 4)                                   # first and last two poses dropped
        np.testing.assert_allclose(got[:, :3, :3], FX[f"rotate{k}"][:, :3, :3], rtol=0, atol=2e-7)
        np.testing.assert_allclose(got[:, :, 3], FX[f"rotate{k}"][:, :, 3], rtol=0, atol=1e-12)
        d = poses[1][:3, 3] - poses[0][:3, 3]
        np.testing.assert_allclose(T.make_circle(d / np.linalg.norm(d), 11, radius=0.04), FX[f"circle{k}"], rtol=0, atol=1e-14)


def test_slerp_endpoints_and_orthonormality():
    poses = FX["poses0"]
    R = T.rotmat_slerp(poses[0][:3, :3], poses[1][:3, :3], np.array([0.0, 0.3, 1.0]))
    np.testing.assert_allclose(R[0], poses[0][:3, :3], atol=2e-7)
    np.testing.assert_allclose(R[2], poses[1][:3, :3], atol=2e-7)
    np.testing.assert_allclose(R[1] @ R[1].T, np.eye(3), atol=5e-7)
    # a rotation by more than 180 degrees about one axis is interpolated the short way round
    from math import cos, sin, pi
    a = 1.2 * pi
    Rz = np.array([[cos(a), -sin(a), 0], [sin(a), cos(a), 0], [0, 0, 1.0]])
    mid = T.rotmat_slerp(np.eye(3), Rz, np.array([0.5]))[0]
    b = -0.4 * pi
    np.testing.assert_allclose(mid, np.array([[cos(b), -sin(b), 0], [sin(b), cos(b), 0], [0, 0, 1.0]]), atol=2e-7)


def test_unposed_pair_input_matches_the_reference():
    rng = np.random.default_rng(9)
    im = [rng.random((256, 300, 3)) for _ in range(2)]
    np.testing.assert_array_equal(np.stack([a[::64, ::64] for a in im]), FX["im_probe"])      # same seeded images as the maker's
    uv = torch.as_tensor(FX["uv"])
    got = T.unposed_pair_input(T.center_crop_square(im[0]), T.center_crop_square(im[1]), FX["R"], FX["t"], uv)
    for part in ("query", "context"):
        for k, v in got[part].items():
            want = FX[f"dict.{part}.{k}"]
            v = v.numpy()
            if k == "rgb":
                v = v[:, :, ::32, ::32]
            assert v.shape == want.shape and v.dtype == want.dtype, (part, k, v.shape, want.shape)
            np.testing.assert_allclose(v, want, rtol=0, atol=1e-6 if k == "cam2world" else 0, err_msg=f"{part}.{k}")
    assert got["query"]["cam2world"].shape[1] == 76
