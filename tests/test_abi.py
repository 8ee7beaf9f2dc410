"""CPU checks of the C-ABI boundary: the library builds, loads and exports every symbol of include/car_hip.h,
argument validation returns error codes (never aborts), and the product package never touches the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from cross_attention_renderer_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from cross_attention_renderer_amd import _lib
    header = open(os.path.join(ROOT, "include", "car_hip.h")).read()
    declared = set(re.findall(r"\b(car_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.car_version() == 300


def test_bad_arguments_return_codes_not_crashes(lib):
    from cross_attention_renderer_amd import _lib
    assert lib.car_linear(None, 4, None, 4, 4, None, 4, 1, 0, None) == -1
    assert b"null pointer" in lib.car_last_error()
    assert lib.car_ray_setup(None, None, 1, 2, 1, 8, 8, 4, 0, None, None, None, None, 0, None) == -1
    assert lib.car_linear_packed_floats(0, 4) == 0
    # K=579 (+1 bias column) -> 19 chunks of 32; N=576 -> 18 tiles; 1024 floats per (chunk, tile)
    assert lib.car_linear_packed_floats(579, 576) == 19 * 18 * 1024
    with pytest.raises(RuntimeError):
        _lib.check(-1, "probe")


def test_round5_entries_validate_their_arguments(lib):
    """The entries added in round 5 return codes (never abort) on null pointers and impossible shapes, and report their sizes without a GPU."""
    assert lib.car_fused_tile_steps() == 8
    assert lib.car_kq_tail_floats() == 72 * 512 and lib.car_kq_bias_floats() == 2 * 128 + 16
    assert lib.car_attend_parts(None, None, 8, 576, 1, 2, 8, 8, None, None, 576, 1, None, None, None, None, None) == -1
    assert b"null pointer" in lib.car_last_error()
    assert lib.car_fused_samples_parts(*([None] * 4), 521, 521, 5, *([None] * 4), 1, 2, 8, 8, 256, 256, 0, *([None] * 6), None) == -1
    assert lib.car_lattice_encode_linear(None, 37, 37, 5, None, None, None, None, 1, 16, None, None, 576, 288, None, 288, 0, None) == -1
    assert lib.car_fused_rows(None, 37, 37, 5, *([None] * 7), 1, 8, 8, 3, None, None) == -1
    assert lib.car_key_query_logits(None, 576, None, None, 576, None, None, None, 16, None, None, None) == -1
    assert lib.car_kq_pack(*([None] * 9)) == -1 and lib.car_fused_pack_rows(*([None] * 8)) == -1
    assert lib.car_exchange_rows(None, None, None, None, 1, 3, 8, 64, 64, None, None, None, None) == -1
    assert lib.car_linear_wgrad(None, 4, None, 4, 8, 4, 4, 16, None, 4, None, None) == -1


def test_round6_entries_validate_their_arguments(lib):
    """The entries added in round 6: sizes without a GPU, codes (never aborts) on null pointers."""
    # the folded 128 x 128 layer and two 128 x 16 layers as fp16 hi / lo tiles (64 + 8 + 8 KB); four vectors + scales, then 128 x 128 of scratch
    assert lib.car_round2q_packed_floats() == lib.car_round2_packed_floats() + 2048 and lib.car_round2q_bias_floats() == 4 * 128 + 8 + 128 * 128
    assert lib.car_fused_bias_floats() == 288 + 3 * 128 + 16 + 128 * 128                                # the table + the packer's scratch
    assert lib.car_round2_logits_from_g(None, None, None, None, 1, 2, 8, 8, None, None) == -1
    assert b"null pointer" in lib.car_last_error()
    assert lib.car_round2q_pack(*([None] * 11)) == -1
    assert lib.car_attend_parts(None, None, 4, 576, 1, 2, 8, 8, None, None, 576, 1, None, None, None, None, None) == -1


def test_round6_training_entries_validate_their_arguments(lib):
    """The backward entries added in round 6 (second session): the binned scatter's workspace size without a GPU, codes on bad arguments."""
    import ctypes
    hs, ws = (ctypes.c_int * 3)(256, 128, 64), (ctypes.c_int * 3)(256, 128, 64)
    texels = 256 * 256 + 128 * 128 + 64 * 64
    n_maps, pts, gathers = 24, 12288, 2
    n = n_maps * texels + 1
    blocks = (n + 1023) // 1024
    want = ((2 * n + blocks + 4) * 4 + 15) // 16 * 16 + gathers * n_maps * pts * 3 * 4 * 8       # counters | start | block sums, then 8-byte records
    assert lib.car_scatter_workspace_bytes(hs, ws, 3, n_maps, pts, gathers) == want
    assert lib.car_scatter_workspace_bytes(None, ws, 3, n_maps, pts, gathers) == 0 and lib.car_scatter_workspace_bytes(hs, ws, 5, n_maps, pts, gathers) == 0
    assert lib.car_gather_bilinear_backward_binned(None, None, hs, ws, 3, n_maps, None, None, None, gathers, pts, 2, None, 608, 0, None, 0, None) == -1
    assert b"null pointer" in lib.car_last_error()
    assert lib.car_linear_x3_masked(None, 4, None, None, 4, 32, None, 32, 8, 0, None, 32, None) == -1
    assert b"act" in lib.car_last_error()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cross_attention_renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f"{f} mentions the oracle"


def test_forward_refuses_cpu_tensors():
    """No CPU fallback: the render path fails loudly off-device."""
    import torch
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    m = CrossAttentionRenderer(model="tiny", n_view=2, npoints=4).eval()
    inp = S.stereo_scene(16, b=1, uv=S.pixel_grid(16, 16)[:8].contiguous())
    z = S.feature_maps(1, 2, 16, channels=(16, 16), strides=(2, 1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(inp, z=z)


def test_module_state_dict_matches_reference_table():
    """Parameter names/shapes of SURVEY.md §8b (checked against the real reference in make_golden.py)."""
    from cross_attention_renderer_amd.models import renderer_param_shapes
    t = renderer_param_shapes("midas_vit", n_view=2)
    assert t["query_encode_latent.weight"] == (576, 579, 1, 1)
    assert t["query_encode_latent_2.weight"] == (288, 576, 1, 1)
    assert t["update_val_merge.weight"] == (288, 582, 1, 1)
    assert t["latent_value.weight"] == (288, 576, 1, 1)
    assert t["key_map.weight"] == (128, 576, 1, 1)
    assert t["query_repeat_embed.weight"] == (128, 144, 1, 1)
    assert t["latent_avg_repeat_query.weight"] == (128, 153, 1, 1)
    assert t["encode_latent.weight"] == (128, 288, 1)
    assert t["phi.lin_in.weight"] == (128, 18) and t["phi.lin_z.2.weight"] == (128, 576)
    assert t["conv_map.weight"] == (64, 3, 7, 7)
    assert sum(int(__import__("math").prod(s)) for s in t.values()) == 1457955



def test_one_call_abi_host_side(lib):
    """Host-only parts of the one-call forward (include/car_hip.h): sizes, argument validation, linspace."""
    import torch
    from cross_attention_renderer_amd import _lib
    d = _lib.CarDims()
    d.b, d.V, d.R, d.P, d.H, d.W, d.n_levels, d.repeat_attention = 1, 2, 8192, 64, 256, 256, 3, 1
    for l, (c, h) in enumerate(((256, 64), (256, 128), (64, 256))):
        d.level_c[l], d.level_h[l], d.level_w[l] = c, h, h
    S = 2 * 8192 * 64
    assert 4 * S * (576 + 128) > lib.car_workspace_bytes(ctypes.byref(d)) >= 4 * S * (576 + 16)      # e, g dominate; no 128-wide query rows
    assert lib.car_plan_bytes(ctypes.byref(d)) >= 4 * (lib.car_fused_blob_floats() + lib.car_round2_packed_floats())
    # the three levels (64, 128, 256 wide: factors 4, 2, 1) share the lattice u in [-5, 515] (pad = r_max + 1 = 5)
    lh, lw, pad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.car_lattice_shape(ctypes.byref(d), ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(pad)) == 0
    assert (lh.value, lw.value, pad.value) == (521, 521, 5)
    lattice = 2 * 2 * 521 * 521 * 576
    up64 = lambda n: (n + 63) // 64 * 64
    texels = 2 * (64 * 64 + 128 * 128 + 256 * 256)
    assert lib.car_gmeta_offset(ctypes.byref(d)) == up64(lattice)
    assert lib.car_gmaps_floats(ctypes.byref(d)) == up64(lattice) + 64 + texels * 576        # lattice, gmeta, the projected levels
    d.level_h[0] = 60                                                                       # 128 is no multiple of 60: no common lattice
    assert lib.car_gmaps_floats(ctypes.byref(d)) == 0 and b"lattice" in lib.car_last_error()
    d.level_h[0] = 64
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.car_workspace_find(ctypes.byref(d), b"e", ctypes.byref(off), ctypes.byref(cnt)) == 0 and cnt.value == S * 576
    assert lib.car_workspace_find(ctypes.byref(d), b"nope", ctypes.byref(off), ctypes.byref(cnt)) == -1
    assert lib.car_profile_count() == 0
    d.V = 3                                                                                 # not covered by the one-call entry
    assert lib.car_workspace_bytes(ctypes.byref(d)) == 0 and b"n_view = 2" in lib.car_last_error()
    assert lib.car_render_forward(ctypes.byref(d), None, None, None, None, 0, None) == -1
    # car_linspace: torch's scalar formula; torch's vectorised CPU kernel may differ in the last ulp for n >= 16
    for n in (2, 5, 8, 15, 16, 32, 64, 128):
        buf = (ctypes.c_float * n)()
        lib.car_linspace(0.0, 1.0, n, buf)
        got, want = torch.tensor(list(buf)), torch.linspace(0, 1, n)
        assert (got - want).abs().max() <= 6e-8, n
        if n < 8:
            assert torch.equal(got, want)
