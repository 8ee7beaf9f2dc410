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


# ---- evaluation items: RealEstate10kVis / ACIDVis (realestate10k_dataio.py:469-719) --------------------------------------------
VIS = os.path.join(HERE, "golden", "dataio_scene_vis")


@pytest.mark.parametrize("views", [1, 2, 3])
@pytest.mark.parametrize("seed", [0, 1])
def test_eval_item_matches_reference(views, seed):
    """One item of the evaluation reader against what the reference's RealEstate10kVis returns for the committed scene (50 frames at
    256 x 455, .mat pose table; tests/golden/make_dataio_golden.py): context = first / middle / last frame, the query drawn with
    random.choice from the frames more than 10 away from every context frame, centre crop, intrinsics, cameras, pixel grid."""
    import random
    want = np.load(os.path.join(HERE, "golden", "dataio_vis_expected.npz"))
    ds = dataio.RealEstate10kVis(os.path.join(VIS, "scenes"), os.path.join(VIS, "poses.mat"), num_ctxt_views=views, num_query_views=1, augment=False)
    assert len(ds) == 1
    random.seed(seed)
    item, gt = ds[0]
    assert gt is item["query"] and item["query"]["mask"] == 0.0
    for part in ("query", "context"):
        for k, v in item[part].items():
            if k == "mask":
                continue
            v = v.numpy()
            ref = want[f"v{views}.s{seed}.{part}.{k}"]
            if k == "rgb":
                s2 = np.asarray([v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum()])
                assert np.array_equal(s2, want[f"v{views}.s{seed}.{part}.rgb_sum"]), (part, k)
                v = v.reshape(-1, 3)[::997]
            if k == "uv":
                v = v.reshape(-1, 2)[::257]
            assert v.shape == ref.shape and v.dtype == ref.dtype, (part, k, v.shape, ref.shape)
            assert np.array_equal(v, ref), (part, k, np.abs(v - ref).max())
    assert tuple(item["context"]["rgb"].shape) == (views, 256, 256, 3) and tuple(item["query"]["rgb"].shape) == (1, 65536, 3)


def test_eval_reader_rejects_training_options_and_unresized_frames():
    with pytest.raises(ValueError, match="training-time"):
        dataio.RealEstate10kVis(os.path.join(VIS, "scenes"), os.path.join(VIS, "poses.mat"), num_ctxt_views=2, augment=True)
    img = np.zeros((256, 455, 3), np.uint8)
    assert dataio.square_crop_img(img).shape == (256, 256, 3)
    assert dataio.ACIDVis is dataio.RealEstate10kVis


def test_resize_of_360_line_frames_follows_opencv_bilinear():
    """cv2.resize(INTER_LINEAR) restated (dataio.resize_linear_u8): against float bilinear interpolation with the same centre-aligned
    coordinates it may differ by the fixed-point rounding only (one grey level); constants and edges are exact."""
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (360, 640, 3), generator=g, dtype=torch.uint8).numpy()
    out = dataio.resize_linear_u8(img, 455, 256)
    assert out.shape == (256, 455, 3) and out.dtype == np.uint8
    ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(256, 455), mode="bilinear",
                                          align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.0
    assert np.abs(out.astype(np.float64) - ref).mean() < 0.3
    flat = np.full((360, 640, 3), 77, np.uint8)
    assert np.array_equal(dataio.resize_linear_u8(flat, 455, 256), np.full((256, 455, 3), 77, np.uint8))
    assert np.array_equal(dataio.resize_linear_u8(img, 640, 360), img)                    # identity size: weights (2048, 0)


def test_eval_reader_resizes_raw_360_line_scenes(tmp_path):
    """A scene as the download scripts write it (raw 360 x 640 frames, generate_realestate.py:58-72) is read like the reference does:
    resized to 256 x 455 per frame, then centre-cropped."""
    import random
    import shutil
    from scipy.io import loadmat
    src = np.load(os.path.join(VIS, "scenes", "sceneA", "data.npz"))
    g = np.random.default_rng(0)
    raw = {k: g.integers(0, 256, (360, 640, 3), dtype=np.uint8) for k in src.keys()}
    os.makedirs(tmp_path / "scenes" / "sceneA")
    np.savez(tmp_path / "scenes" / "sceneA" / "data.npz", **raw)
    shutil.copy(os.path.join(VIS, "poses.mat"), tmp_path / "poses.mat")
    ds = dataio.RealEstate10kVis(str(tmp_path / "scenes"), str(tmp_path / "poses.mat"), num_ctxt_views=2)
    random.seed(0)
    item, _ = ds[0]
    assert tuple(item["context"]["rgb"].shape) == (2, 256, 256, 3) and tuple(item["query"]["rgb"].shape) == (1, 65536, 3)
    first = sorted(raw.keys(), key=lambda n: int(n.split(".")[0]))[0]
    want = dataio.square_crop_img(dataio.resize_linear_u8(raw[first], 455, 256)).astype(np.float32) / 127.5 - 1
    assert np.array_equal(item["context"]["rgb"][0].numpy(), want)
    assert loadmat(str(tmp_path / "poses.mat"))["sceneA"].shape[1] == 19
