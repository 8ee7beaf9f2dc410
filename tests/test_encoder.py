"""``CrossAttentionRenderer.get_z`` with the multi-view DPT-hybrid encoder (cross_attention_renderer_amd/encoder.py) against the
reference's own ``get_z`` (models.py:148-188 with midas/dpt_depth.py, midas/vit.py, midas/blocks.py, vit_models.py), replayed from the
fixtures ``tests/golden/getz_<variant>.npz`` that ``make_encoder_golden.py`` wrote by RUNNING the reference (its timm 0.5.4 layers
restated by ``tests/golden/timm_stub.py``, see there).  Pinned: the state_dict name -> shape table (a reference checkpoint loads with
strict=True), image normalisation, the relative-pose embedding incl. ``no_multiview``, the cross-view token sequence, read-out and
re-assembly, RefineNet fusion, ``conv_map`` incl. ``no_high_freq`` and the order of the returned levels."""
import numpy as np
import pytest
import torch

import encoder_cases as EC
from cross_attention_renderer_amd.models import CrossAttentionRenderer


def _model(variant):
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, **EC.VARIANTS[variant]).eval()
    fx = np.load(EC.fixture_path(variant))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sorted(shapes) == list(fx["names"]), "parameter names differ from the reference's state_dict"
    assert [str(shapes[n]) for n in sorted(shapes)] == list(fx["shapes"]), "parameter shapes differ from the reference's state_dict"
    m.load_state_dict(EC.seeded_weights(shapes), strict=True)
    return m, fx


@pytest.mark.parametrize("variant", list(EC.VARIANTS))
def test_get_z_reproduces_the_reference(variant):
    m, fx = _model(variant)
    inp = EC.context_pair()
    inp64 = {k: {kk: vv.double() for kk, vv in v.items()} for k, v in inp.items()}
    with torch.no_grad():
        z32 = m.get_z(inp)
        assert (m.H, m.W) == (EC.H, EC.H)                       # get_z records the image size for forward (models.py:162)
        z64 = m.double().get_z(inp64)
    assert [tuple(t.shape) for t in z32] == [(2, 256, 64, 64), (2, 256, 128, 128), (2, 64, 256, 256)]      # [path_2, path_1, conv_map]
    for i, (got, got32) in enumerate(zip(EC.sample(z64), EC.sample(z32))):
        want = fx[f"z{i}"]
        err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
        assert err.max() < 1e-8, (variant, i, err.max())                                   # float64: the computation is the reference's
        rms = max(float(np.sqrt((want ** 2).mean())), 1e-30)
        assert np.abs(got32 - want).max() <= 1e-3 * rms + 1e-6, (variant, i)               # float32: summation-order noise only
    np.testing.assert_allclose(EC.stats(z64), fx["stats"], rtol=1e-9, atol=1e-12)
    if variant == "no_high_freq":
        assert float(z32[2].abs().max()) == 0.0


def test_pose_embedding_reaches_the_features():
    """no_multiview zeroes the 16-vector of the relative pose (models.py:170-174): the pyramids of the two variants must differ in
    the transformer levels and agree in conv_map, which never sees the pose."""
    a, b_ = np.load(EC.fixture_path("default")), np.load(EC.fixture_path("no_multiview"))
    assert np.abs(a["z0"] - b_["z0"]).max() > 1e-3 and np.abs(a["z1"] - b_["z1"]).max() > 1e-3
    assert np.array_equal(a["z2"], b_["z2"])


def test_encoder_rejects_other_image_sizes():
    from cross_attention_renderer_amd.encoder import MultiViewDPTEncoder
    enc = MultiViewDPTEncoder().eval()
    with pytest.raises(ValueError, match="256x256"):
        enc(torch.zeros(2, 3, 128, 128), torch.zeros(2, 16), 2)


def test_renderer_only_module_has_no_encoder():
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, with_encoder=False)
    assert not any(k.startswith("encoder.") for k in m.state_dict())
    with pytest.raises(NotImplementedError, match="with_encoder=True"):
        m.get_z(EC.context_pair())
