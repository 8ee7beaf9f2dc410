"""``-m gpu`` parity tests: the HIP path (libcar_hip.so, called through its C ABI) against the CPU oracle and against
the committed reference outputs, on the same seeded inputs.

Tolerance (north_star): |a-b| <= 1e-4 * max(1,|b|) for rgb / depth_ray / at_wt.  The per-sample Pluecker intersection
is ill-conditioned where a sample's pixel ray is nearly parallel to the query ray, so a last-ulp difference in any
fp32 input (e.g. tanhf vs SLEEF tanh, or LAPACK on a different host CPU) can move a handful of samples by more than
that; the tests therefore allow a small, stated budget of outlier elements (OUTLIER_FRAC) and bound the worst one.
"""
import ctypes

import numpy as np
import pytest
import torch

import cases as C
from golden_util import load_case, rel_err
from hip_harness import run_native, build_module, err_stats, oracle_cfg, run_case, to_device
from oracle import car_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4                   # the contract: |a-b| <= 1e-4 * max(1,|b|)
# Only for the comparison that runs torch.inverse on *this* host against vectors made on another CPU (see
# test_forward_host_poses_vs_reference): at most 2 % of an output's elements may exceed TOL, none by more than 5e-2.
OUTLIER_FRAC = 2e-2
OUTLIER_MAX = 5e-2

HIP_CASES = list(C.CASES)


def _lib():
    from cross_attention_renderer_amd import _lib
    return _lib.load()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


# ----------------------------------------------------------------------------------------------------------
# the MFMA linear kernel against torch (fp64 reference of the same op)
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N,flags", [
    (128, 32, 32, 0), (1000, 579, 576, 2), (4096, 576, 288, 0), (777, 576, 128, 2), (300, 128, 128, 0),
    (513, 16, 128, 2), (129, 18, 128, 0), (64, 128, 3, 1), (256, 35, 32, 2), (100, 32, 16, 0), (31, 144, 128, 3),
    (2048, 288, 128, 0), (200, 576, 416, 0), (50, 7, 5, 0),
])
@pytest.mark.parametrize("no_glds", [0, 8])
def test_linear_matches_torch(M, K, N, flags, no_glds):
    from cross_attention_renderer_amd.engine import PackedLinear
    lib = _lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M * 7 + K * 3 + N)
    ldx = (K + 3) // 4 * 4 + 4
    ldy = N + 4 if M % 2 == 0 else N + 5          # even M: float4 store path, odd M: scalar path
    X = torch.randn(M, ldx, generator=g)
    Wt = torch.randn(N, K, generator=g) / K ** 0.5          # asymmetric by construction
    bias = torch.randn(N, generator=g)
    Y0 = torch.randn(M, ldy, generator=g)
    want = (torch.relu(X[:, :K]) if flags & 1 else X[:, :K]).double() @ Wt.double().T + bias.double()
    if flags & 2:
        want = torch.relu(want)
    Xd, Yd = X.to(dev), Y0.clone().to(dev)
    layer = PackedLinear(Wt, bias, dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.car_linear(_ptr(Xd), ldx, _ptr(layer.packed), K, N, _ptr(Yd), ldy, M, flags | no_glds, st)
    assert rc == 0, lib.car_last_error()
    # accumulate variant on top
    Ya = Y0.clone().to(dev)
    rc = lib.car_linear(_ptr(Xd), ldx, _ptr(layer.packed), K, N, _ptr(Ya), ldy, M, (flags & 1) | 4 | no_glds, st)
    assert rc == 0, lib.car_last_error()
    torch.cuda.synchronize()
    got = Yd.cpu()
    assert rel_err(got[:, :N], want) < 2e-5
    assert torch.equal(got[:, N:], Y0[:, N:]), "columns beyond N were touched"
    want_acc = Y0[:, :N].double() + (torch.relu(X[:, :K]) if flags & 1 else X[:, :K]).double() @ Wt.double().T + bias.double()
    assert rel_err(Ya.cpu()[:, :N], want_acc) < 2e-5


# ----------------------------------------------------------------------------------------------------------
# stage kernels
# ----------------------------------------------------------------------------------------------------------
def test_gather_matches_grid_sample():
    from cross_attention_renderer_amd.engine import RenderEngine
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    n, pts = 4, 700
    z = [torch.randn(n, 8, 5, 7, generator=g), torch.randn(n, 12, 9, 6, generator=g), torch.randn(n, 4, 16, 16, generator=g)]
    grid = torch.rand(n, pts, 2, generator=g) * 2.6 - 1.3
    grid[0, 0] = torch.tensor([1e10, 1e10]); grid[0, 1] = torch.tensor([-1.0, 1.0]); grid[1, 2] = torch.tensor([7.8e7, -0.2])

    class Dummy:
        pass
    eng = RenderEngine.__new__(RenderEngine)
    eng.lib = _lib()
    maps = [t.permute(0, 2, 3, 1).contiguous().to(dev) for t in z]
    Ct = 24
    for mode, name in ((0, "border"), (1, "zeros")):
        out = torch.full((n * pts, 32), -7.0, device=dev)
        eng.gather(maps, grid.to(dev), pts, mode, 0, 1, out, 32, 4)
        torch.cuda.synchronize()
        want = torch.cat([torch.nn.functional.grid_sample(t, grid[:, :, None, :], mode="bilinear", padding_mode=name,
                                                          align_corners=False)[..., 0].permute(0, 2, 1) for t in z], dim=-1)
        got = out.cpu().view(n, pts, 32)
        assert (got[..., 4:4 + Ct] - want).abs().max() < 1e-5, name
        assert (got[..., :4] == -7).all() and (got[..., 4 + Ct:] == -7).all()


@pytest.mark.parametrize("name", ["t0_default", "t0_query_at_ctx0", "t0_diverging", "t1_c1", "t2_c5", "t0_no_sample", "t0_nview1"])
def test_geometry_stages_match_oracle(name):
    c, fx, ora, out = run_case(name, fuse_samples=False)
    st, hs = ora["stages"], out["stages"]
    rays = hs["rays"]
    # with the host computing the reference's own pose algebra the device geometry reproduces the oracle to the bit,
    # except where libm-vs-device transcendental/rounding differences enter (tanh) -> compare tightly, not bitwise
    assert rel_err(rays[..., 0:6], st["lf"]) < 1e-6
    assert rel_err(out["coords"], ora["coords"]) < 1e-6
    if "overlaps" in st:
        assert (rays[..., 10] != st["overlaps"].float()).float().mean() == 0.0
    assert rel_err(out["pixel_val"], st["pixel_val"]) < 1e-5
    assert rel_err(hs["pt"], st["pt"]) < 1e-6
    assert rel_err(hs["local_coords"][..., :9], st["local_coords"][..., :9]) < 1e-5
    assert (out["valid_mask"] == ora["valid_mask"]).all()


# ----------------------------------------------------------------------------------------------------------
# the whole forward: HIP vs oracle and HIP vs the committed reference outputs
# ----------------------------------------------------------------------------------------------------------
def _check_outputs(got, want_of, what, frac=0.0, worst=TOL):
    for k in ("rgb", "depth_ray", "at_wt"):
        e = err_stats(got[k], want_of(k))
        assert e["f1e-4"] <= frac and e["max"] <= worst, f"{what} {k}: {e}"
    assert (np.asarray(got["valid_mask"]) == np.asarray(want_of("valid_mask"))).all(), what
    same = (np.asarray(got["at_wt_max"]) == np.asarray(want_of("at_wt_max"))).mean()
    assert same > 0.995, f"{what} at_wt_max agreement {same}"


@pytest.mark.parametrize("name", HIP_CASES)
def test_forward_matches_oracle(name):
    """HIP vs the CPU oracle on this host, both running the reference's pose algebra here: strict 1e-4."""
    c, fx, ora, out = run_case(name)
    assert tuple(out["rgb"].shape) == fx["out_rgb"].shape
    assert tuple(out["at_wt_max"].shape) == fx["out_at_wt_max"].shape and out["at_wt_max"].dtype == torch.int64
    assert tuple(out["coords"].shape) == fx["out_coords"].shape
    _check_outputs(out, lambda k: ora[k], "vs oracle")
    assert err_stats(out["stages"]["interp_val"], ora["stages"]["interp_val"])["max"] <= TOL
    assert err_stats(out["stages"]["pt"], ora["stages"]["pt"])["max"] <= TOL


@pytest.mark.parametrize("name", HIP_CASES)
def test_forward_matches_reference_fixture(name):
    """HIP vs the outputs of the reference itself (committed fixture), using the pose matrices the reference computed:
    strict 1e-4, every element."""
    c, fx, ora, out = run_case(name, fixture_poses=True)
    _check_outputs(out, lambda k: fx["out_" + k], "vs reference fixture")
    assert rel_err(out["pixel_val"], fx["out_pixel_val"]) < 1e-6
    assert rel_err(out["coords"], fx["out_coords"]) < 1e-6
    if c["tier"] == 0:
        assert rel_err(out["stages"]["pt"], fx["stage_pt"]) < 1e-6
        assert err_stats(out["stages"]["interp_val"], fx["stage_interp_val"])["max"] <= TOL


@pytest.mark.parametrize("name", ["t1_c1", "t1_c1_diverging", "t2_c2", "t2_c4", "t2_c5"])
def test_forward_host_poses_vs_reference(name):
    """The production path (torch.inverse on this host's CPU) against vectors made on another CPU: LAPACK's last-ulp
    results are host dependent and the fp64 intersection amplifies them on a few ill-conditioned samples, so this
    comparison — and only this one — carries an outlier budget."""
    c, fx, ora, out = run_case(name)
    _check_outputs(out, lambda k: fx["out_" + k], "host poses vs reference fixture", frac=OUTLIER_FRAC, worst=OUTLIER_MAX)


@pytest.mark.parametrize("name", ["t0_default", "t0_diverging", "t1_c1", "t2_c2", "t2_c5"])
def test_literal_gather_gemm_pipeline_matches_oracle(name):
    """The literal pipeline (materialised 579-wide rows -> K=579 GEMM), kept for A/B against the default path that
    applies the first point-MLP layer per texel (csrc/car_encode.hip): both must sit within 1e-4 of the oracle."""
    c, fx, ora, out = run_case(name, project_maps=False)
    _check_outputs(out, lambda k: ora[k], "literal pipeline vs oracle")
    _, _, _, out2 = run_case(name, project_maps=True, fuse_samples=False)
    assert rel_err(out["rgb"], out2["rgb"]) < 2e-5
    assert err_stats(out["stages"]["interp_val"], out2["stages"]["interp_val"])["max"] < 5e-5


@pytest.mark.parametrize("version", [4, 2, 1])
@pytest.mark.parametrize("name", ["t1_c1", "t1_c1_diverging", "t2_c2", "t2_c3", "t2_c5"])
def test_fused_sample_kernel_matches_stage_pipeline(name, version):
    """A/B of csrc/car_fused2.hip / car_fused.hip (geometry + encode + e + key/query MLPs + logits in one kernel, everything
    chained through the MFMA accumulators) against the stage-by-stage kernels; both against the oracle at 1e-4."""
    c, fx, ora, fused = run_case(name, fuse_samples=True, fused_version=version)
    assert fused["stages"]["local_coords"] is None, "the fused kernel was not selected"
    _, _, _, staged = run_case(name, fuse_samples=False)
    assert rel_err(fused["stages"]["pt"], staged["stages"]["pt"]) < 1e-6
    assert rel_err(fused["pixel_val"], staged["pixel_val"]) < 1e-6
    assert err_stats(fused["stages"]["interp_val"], staged["stages"]["interp_val"])["max"] < 5e-5
    assert rel_err(fused["at_wt"], staged["at_wt"]) < 1e-5
    assert rel_err(fused["rgb"], staged["rgb"]) < 2e-5
    _check_outputs(fused, lambda k: ora[k], "fused vs oracle")


@pytest.mark.parametrize("name", ["t1_c1", "t1_c1_diverging", "t2_c2", "t2_c3", "t2_c5"])
def test_split_fp16_layer_is_fp32_class(name):
    """The 576->288 layer on the f16 matrix pipe with fp16 hi/lo operand splits (three exact products per term) against the
    same kernel on the fp32 pipe: the per-sample features must agree to ~1e-6, i.e. far inside the 1e-4 contract."""
    c, fx, ora, a = run_case(name, split_fp16=True, fused_version=1)
    _, _, _, b_ = run_case(name, split_fp16=False, fused_version=1)
    e = err_stats(a["stages"]["interp_val"], b_["stages"]["interp_val"])
    assert e["max"] < 5e-6, e
    assert rel_err(a["rgb"], b_["rgb"]) < 1e-5
    _check_outputs(a, lambda k: ora[k], "split-fp16 vs oracle")
    _check_outputs(b_, lambda k: ora[k], "fp32 pipe vs oracle")


@pytest.mark.parametrize("name", ["t1_c1", "t1_c1_diverging", "t2_c3", "t2_c4"])
def test_fused_v4_equals_v2(name):
    """Three waves per SIMD (csrc/car_fused4.hip) against two (car_fused2.hip): same arithmetic per sample; only the order in which
    the very first chunk adds its pyramid levels and the order of the key layer's two halves (e_1 first) differ, i.e. rounding."""
    c, fx, ora, a = run_case(name, fused_version=4)
    _, _, _, b_ = run_case(name, fused_version=2)
    assert torch.equal(a["stages"]["pt"], b_["stages"]["pt"])
    assert err_stats(a["stages"]["interp_val"], b_["stages"]["interp_val"])["max"] < 2e-6
    assert rel_err(a["at_wt"], b_["at_wt"]) < 1e-5
    assert rel_err(a["rgb"], b_["rgb"]) < 1e-5
    _check_outputs(a, lambda k: ora[k], "fused v4 vs oracle")


@pytest.mark.parametrize("name", ["t1_c1", "t2_c2", "t2_c4"])
def test_fused_v2_matches_fp32_pipe_v1(name):
    """The two-waves-per-SIMD kernel (16x16x32 f16 tiles, every layer split-fp16) against the first-generation kernel on the
    fp32 matrix pipe: per-sample features to ~1e-6, attention weights and colours to 1e-5."""
    c, fx, ora, a = run_case(name, fused_version=2)
    _, _, _, b_ = run_case(name, fused_version=1, split_fp16=False)
    assert rel_err(a["stages"]["pt"], b_["stages"]["pt"]) == 0
    e = err_stats(a["stages"]["interp_val"], b_["stages"]["interp_val"])
    assert e["max"] < 5e-6, e
    assert rel_err(a["at_wt"], b_["at_wt"]) < 1e-5
    assert rel_err(a["rgb"], b_["rgb"]) < 1e-5
    _check_outputs(a, lambda k: ora[k], "fused v2 vs oracle")


@pytest.mark.parametrize("name", ["t0_default", "t0_p5", "t0_nview1", "t0_nview3", "t1_c1", "t2_c3"])
def test_round2_logit_kernel_matches_stage_kernels(name):
    """csrc/car_round2.hip (second-round query layer + logits, q2 never stored) against add_ray_bias_relu + car_linear +
    the logits phase of car_attend."""
    c, fx, ora, a = run_case(name, fuse_round2=True)
    _, _, _, b_ = run_case(name, fuse_round2=False)
    assert rel_err(a["stages"]["at_wt2"], b_["stages"]["at_wt2"]) < 1e-5
    assert rel_err(a["rgb"], b_["rgb"]) < 1e-5
    _check_outputs(a, lambda k: ora[k], "fused round 2 vs oracle")


def test_register_staged_weights_agree_with_lds_dma():
    """A/B of the two weight-staging variants of the MFMA kernel on a whole forward."""
    _, _, _, a = run_case("t1_c1", fuse_samples=False)
    _, _, _, b_ = run_case("t1_c1", linear_flags=8, fuse_samples=False)
    assert rel_err(a["rgb"], b_["rgb"]) < 1e-6


# ----------------------------------------------------------------------------------------------------------
# size-independent properties at the full bench shape (256x256, 64 samples, one 8192-ray chunk)
# ----------------------------------------------------------------------------------------------------------
def test_full_size_properties():
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H, P, R = 256, 64, 8192
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P).eval()
    S.perturb_parameters(m, seed=0)
    m.H = m.W = H
    m = m.to(dev)
    uv = S.pixel_grid(H, H)[96 * H:96 * H + R].contiguous()
    inp = to_device(S.stereo_scene(H, b=1, uv=uv, seed=5), dev)
    z = [t.to(dev) for t in S.feature_maps(1, 2, H, seed=1)]
    with torch.no_grad():
        full = m(inp, z=z)
        # rays are independent: any sub-chunk reproduces the corresponding slice
        sub = {"context": inp["context"], "query": dict(inp["query"], uv=inp["query"]["uv"][:, :, 1000:1777].contiguous())}
        part = m(sub, z=z)
    torch.cuda.synchronize()
    assert torch.isfinite(full["rgb"]).all() and torch.isfinite(full["depth_ray"]).all()
    assert rel_err(part["rgb"].cpu(), full["rgb"][:, :, 1000:1777].cpu()) < 1e-5
    assert rel_err(part["at_wt"].cpu(), full["at_wt"][:, 1000:1777].cpu()) < 1e-5
    # softmax weights of a ray sum to one over both views' samples
    s = full["at_wt"].view(1, 2, R, P).sum(dim=(1, 3))
    assert (s - 1).abs().max() < 1e-5
    assert (full["depth_ray"] >= 0).all() and (full["depth_ray"] <= 10).all()
    # white where no view sees the ray
    inval = full["valid_mask"][..., 0] == 0
    if inval.any():
        assert (full["rgb"][:, 0][inval] == 1).all()
    # spot-check 64 rays of the chunk against the oracle
    idx = torch.linspace(0, R - 1, 64).long()
    cpu_inp = {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in inp.items()}
    cpu_inp["query"]["uv"] = cpu_inp["query"]["uv"][:, :, idx].contiguous()
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ora = O.render_forward(sd, cpu_inp, [t.cpu() for t in z], O.RenderConfig(n_view=2, npoints=P, H=H, W=H))
    e = err_stats(full["rgb"][:, :, idx].cpu(), ora["rgb"])
    assert e["max"] <= TOL, e



def test_whole_frame_call_equals_chunked_calls():
    """One forward call over all 65 536 rays of a 256x256 frame (bench.py's default; 8.4 M samples, 19 GB of per-sample features:
    exercises every 64-bit row offset) against the reference render script's 8 calls of 8192 rays: rays are independent, and the
    sample groups of the fused kernel never straddle a call boundary, so the results are identical to rounding."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H, P = 256, 64
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P).eval()
    S.perturb_parameters(m, seed=0)
    m.H = m.W = H
    m = m.to(dev)
    inp = to_device(S.stereo_scene(H, b=1, seed=5), dev)
    z = [t.to(dev) for t in S.feature_maps(1, 2, H, seed=1)]
    uv = inp["query"]["uv"]
    with torch.no_grad():
        full = m(inp, z=z)
        keep = {k: full[k].cpu() for k in ("rgb", "depth_ray", "valid_mask", "at_wt", "at_wt_max")}
        del full
        for c0 in range(0, H * H, 8192):
            sub = {"context": inp["context"], "query": dict(inp["query"], uv=uv[:, :, c0:c0 + 8192].contiguous())}
            part = m(sub, z=z)
            assert rel_err(part["rgb"].cpu(), keep["rgb"][:, :, c0:c0 + 8192]) < 1e-5
            assert rel_err(part["depth_ray"].cpu(), keep["depth_ray"][:, c0:c0 + 8192]) < 1e-5
            assert rel_err(part["at_wt"].cpu(), keep["at_wt"][:, c0:c0 + 8192]) < 1e-5
            assert torch.equal(part["valid_mask"].cpu(), keep["valid_mask"][:, c0:c0 + 8192])
            assert (part["at_wt_max"].cpu() == keep["at_wt_max"][:, c0:c0 + 8192]).float().mean() > 0.999
    assert torch.isfinite(keep["rgb"]).all()



@pytest.mark.parametrize("name", ["t1_c1", "t1_c1_diverging", "t2_c2", "t2_c3", "t2_c4", "t2_c5"])
def test_one_call_c_abi_equals_python_engine(name):
    """car_plan_build + car_project_maps + car_render_forward (csrc/car_render.hip: weights packed on the device, the launch sequence
    issued from C++) against the Python engine on the same module and inputs: same kernels in the same order, so every output tensor
    is identical bit for bit — the C ABI is a complete boundary for the default configuration, not a helper of the Python host."""
    eng, nat = run_native(name)
    for k in eng:
        assert eng[k].shape == nat[k].shape, k
        assert torch.equal(eng[k], nat[k]), f"{name} {k}: max abs diff {(eng[k].double() - nat[k].double()).abs().max().item()}"



@pytest.mark.parametrize("R,P,b", [(37, 13, 1), (16, 8, 2), (131, 70, 1), (96, 32, 5)])
def test_fused_path_ragged_sizes_against_the_oracle(R, P, b):
    """Ray counts that are no multiple of 16 and sample counts that are no multiple of 8 (the fused kernel works on groups of
    16 rays x 8 steps, the round-2 kernel on blocks of 32 samples): real widths, H = 64, rays picked across the frame."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H = 64
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P).eval()
    S.perturb_parameters(m, seed=4)
    m.H = m.W = H
    uv = C.select_rays(H, R)
    inp = S.stereo_scene(H, b=b, uv=uv, seed=7, alpha=0.35)
    z = S.feature_maps(b, 2, H, seed=2)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ora = O.render_forward(sd, inp, z, O.RenderConfig(n_view=2, npoints=P, H=H, W=H), debug=True)
        md = m.to(dev)
        out = md(to_device(inp, dev), z=[t.to(dev) for t in z], debug=True)
    torch.cuda.synchronize()
    assert out["stages"]["local_coords"] is None, "the fused kernel was not selected"
    assert torch.equal(out["stages"]["pt"].cpu(), ora["stages"]["pt"])
    for k in ("rgb", "depth_ray", "at_wt"):
        e = err_stats(out[k].cpu(), ora[k])
        assert e["max"] <= TOL, (k, e)
    assert torch.equal(out["valid_mask"].cpu(), ora["valid_mask"])
    assert (out["at_wt_max"].cpu() == ora["at_wt_max"]).float().mean() > 0.99



def test_one_call_c_abi_without_second_round():
    """repeat_attention=False (models.py:547) through car_render_forward, the Python engine and the oracle at real widths."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    from cross_attention_renderer_amd.native import NativeRenderer
    dev = torch.device("cuda:0")
    H, P, R = 64, 32, 200
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, repeat_attention=False).eval()
    S.perturb_parameters(m, seed=6)
    m.H = m.W = H
    inp = S.stereo_scene(H, b=1, uv=C.select_rays(H, R), seed=9, alpha=0.6)
    z = S.feature_maps(1, 2, H, seed=3)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ora = O.render_forward(sd, inp, z, O.RenderConfig(n_view=2, npoints=P, H=H, W=H, repeat_attention=False))
        md = m.to(dev)
        dinp, dz = to_device(inp, dev), [t.to(dev) for t in z]
        eng = md(dinp, z=dz)
        nat = NativeRenderer(md, dev).forward(dinp, dz)
    torch.cuda.synchronize()
    for k in ("rgb", "depth_ray", "at_wt", "valid_mask", "at_wt_max"):
        assert torch.equal(eng[k].cpu(), nat[k].cpu()), k
    for k in ("rgb", "depth_ray", "at_wt"):
        assert err_stats(eng[k].cpu(), ora[k])["max"] <= TOL, k
