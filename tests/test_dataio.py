"""Scene reader (cross_attention_renderer_amd/dataio.py) against what the reference's ``get_camera_pose`` returns for the committed
synthetic scene (tests/golden/dataio_scene + dataio_expected.npz, written by tests/golden/make_dataio_golden.py)."""
import os

import numpy as np
import pytest
import torch

from cross_attention_renderer_amd import dataio

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "golden", "dataio_scene", "scene0")
POSES = os.path.join(HERE, "golden", "dataio_scene", "poses")


@pytest.mark.parametrize("views", [1, 2, 3])
def test_get_camera_pose_matches_reference(views):
    want = np.load(os.path.join(HERE, "golden", "dataio_expected.npz"))
    got = dataio.get_camera_pose(SCENE, POSES, torch.from_numpy(want["uv"]), views=views)
    for part in ("query", "context"):
        for k, v in got[part].items():
            ref = want[f"v{views}.{part}.{k}"]
            assert tuple(v.shape) == ref.shape, (part, k, v.shape, ref.shape)
            assert v.dtype == torch.float32
            assert np.array_equal(v.numpy(), ref), (part, k, np.abs(v.numpy() - ref).max())


def test_pose_file_parsing():
    cams = dataio.parse_pose_file(os.path.join(POSES, "scene0.txt"))
    assert len(cams) == 7 and min(cams) == 100100
    c = cams[100100 + 33367]
    assert np.allclose(c.w2c_mat @ c.c2w_mat, np.eye(4), atol=1e-12)
    K = dataio.unnormalize_intrinsics(c.intrinsics, 256, 456)
    assert np.isclose(K[0, 0], c.intrinsics[0, 0] * 456) and np.isclose(K[1, 2], c.intrinsics[1, 2] * 256)
    with pytest.raises(ValueError):
        dataio.get_camera_pose(SCENE, POSES, torch.zeros(2, 2), views=4)
