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
from hip_harness import argmax_exact_where_decided, build_module, err_stats, oracle_cfg, run_case, to_device
from oracle import car_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4                   # the contract: |a-b| <= 1e-4 * max(1,|b|)
# Only for the comparisons that run torch.inverse on *this* host against vectors made on another CPU (test_forward_host_poses_vs_reference,
# test_forward_with_the_whole_input_dict_on_the_device): LAPACK's last ulp is host dependent.  Measured on the MI355X box (EPYC host) against
# the fixtures (made on a Xeon), all 21 cases: 19 with every element inside 1e-4; t1_c1_diverging 1.0 % of rgb beyond it (worst 2.0e-4),
# t2_c2 one ray of 64 = 1.6 % (worst 1.6e-3); depth_ray / at_wt always inside.  Budget: one ray of a 48-ray fixture, none beyond 5e-3.
OUTLIER_FRAC = 2.1e-2
OUTLIER_MAX = 5e-3
# car_pose_setup's fp64 Gauss-Jordan against the fixture's matrices (the reference's fp32 LAPACK on the build host): a whole frame moves
# 0.8-2 % of its elements by up to 1.3e-2 (profiles/round2_whole_frame_parity.md).  On the 48-256-ray fixtures: 0-1.6 % of the elements,
# two rays of 48 (4.2 %) on t1_c1_diverging and t2_c5, worst 7.1e-3 (t2_c2) — the budget is two rays of the smallest fixture, and 2e-2
DEVICE_POSE_FRAC = 4.5e-2
DEVICE_POSE_MAX = 2e-2

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


@pytest.mark.parametrize("M,K,N,flags", [
    (4096, 576, 576, 2), (1000, 579, 576, 0), (777, 576, 288, 1), (300, 128, 128, 3), (5000, 288, 128, 0), (129, 144, 128, 2), (640, 64, 32, 0),
    (2048, 576, 64, 2), (333, 100, 96, 1),
])
def test_split_fp16_linear_matches_torch(M, K, N, flags):
    """car_linear_x3: the stage entries' wide layers on the f16 matrix pipe (fp16 hi / lo halves, three products, fp32 accumulate, a power of
    two per row and per layer) against an fp64 reference — the same bound as the fp32-pipe kernel — with rows of very different magnitudes."""
    from cross_attention_renderer_amd.engine import PackedLinear
    lib = _lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M * 7 + K * 3 + N)
    ldx = (K + 3) // 4 * 4 + 4
    ldy = N + 4
    X = torch.randn(M, ldx, generator=g) * torch.logspace(-6, 4, M).unsqueeze(1)          # rows from 1e-6 to 1e4
    if M % 2:                                                                               # and magnitudes that grow 1e6-fold along the row: the
        X = X * torch.logspace(-3, 3, ldx).unsqueeze(0)                                     # row's power of two is re-chosen chunk after chunk
    X[::5, :min(K, 96)] = 0.0                                                               # rows that start with whole chunks of zeros
    X[7::11] = 0.0                                                                          # and all-zero rows (the output is the bias)
    Wt = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    Y0 = torch.randn(M, ldy, generator=g)
    xin = (torch.relu(X[:, :K]) if flags & 1 else X[:, :K]).double()
    want = xin @ Wt.double().T + bias.double()
    if flags & 2:
        want = torch.relu(want)
    layer = PackedLinear(Wt, bias, dev)
    assert layer.x3 is not None
    tiles, bdev = layer.x3
    Xd, Yd, Ya = X.to(dev), Y0.clone().to(dev), Y0.clone().to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.car_linear_x3(_ptr(Xd), ldx, _ptr(tiles), _ptr(bdev), K, N, _ptr(Yd), ldy, M, flags, st) == 0, lib.car_last_error()
    assert lib.car_linear_x3(_ptr(Xd), ldx, _ptr(tiles), _ptr(bdev), K, N, _ptr(Ya), ldy, M, (flags & 1) | 4, st) == 0
    torch.cuda.synchronize()
    got = Yd.cpu()
    # per row: the error is relative to the row's own scale (|x| |W| summed), whatever the row's magnitude
    bound = (xin.abs() @ Wt.double().abs().T + bias.double().abs())
    assert ((got[:, :N].double() - want).abs() / bound).max().item() < 4e-6
    assert torch.equal(got[:, N:], Y0[:, N:]), "columns beyond N were touched"
    want_acc = Y0[:, :N].double() + xin @ Wt.double().T + bias.double()
    assert ((Ya.cpu()[:, :N].double() - want_acc).abs() / (bound + Y0[:, :N].double().abs())).max().item() < 4e-6
    # shapes the entry refuses (car_linear serves them)
    assert lib.car_linear_x3(_ptr(Xd), ldx, _ptr(tiles), _ptr(bdev), K, N - 1, _ptr(Yd), ldy, M, 0, st) != 0


@pytest.mark.parametrize("M,K,N,flags", [(4100, 288, 576, 0), (1000, 128, 128, 4), (333, 576, 64, 0), (777, 64, 32, 2)])
def test_split_fp16_linear_with_the_relu_mask_in_its_store(M, K, N, flags):
    """car_linear_x3_masked = car_linear_x3 followed by car_relu_mask, bit for bit (the backward's data gradient of a layer behind a ReLU),
    for every tile count of the kernel, with ACCUM and RELU_OUT, a strided activation, and -0.0 / NaN activations (not > 0: zeroed)."""
    from cross_attention_renderer_amd.engine import PackedLinear
    lib = _lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + K + N)
    ldx, ldy, lda = (K + 3) // 4 * 4, N + 4, N + 8
    X = torch.randn(M, ldx, generator=g).to(dev)
    act = torch.randn(M, lda, generator=g)
    act[::7, ::3] = -0.0
    act[3::11, 1::5] = float("nan")
    act = act.to(dev)
    Y0 = torch.randn(M, ldy, generator=g).to(dev)
    layer = PackedLinear(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), dev)
    tiles, bdev = layer.x3
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    want, got = Y0.clone(), Y0.clone()
    assert lib.car_linear_x3(_ptr(X), ldx, _ptr(tiles), _ptr(bdev), K, N, _ptr(want), ldy, M, flags, st) == 0, lib.car_last_error()
    assert lib.car_relu_mask(_ptr(want), ldy, _ptr(act), lda, M, N, st) == 0
    assert lib.car_linear_x3_masked(_ptr(X), ldx, _ptr(tiles), _ptr(bdev), K, N, _ptr(got), ldy, M, flags, _ptr(act), lda, st) == 0, lib.car_last_error()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert (got[:, :N] == 0).float().mean().item() > 0.4                                   # the mask did something
    # refused: no activation, a row stride that does not hold a row or is not a multiple of 4
    assert lib.car_linear_x3_masked(_ptr(X), ldx, _ptr(tiles), _ptr(bdev), K, N, _ptr(got), ldy, M, flags, None, lda, st) != 0
    assert lib.car_linear_x3_masked(_ptr(X), ldx, _ptr(tiles), _ptr(bdev), K, N, _ptr(got), ldy, M, flags, _ptr(act), N - 4, st) != 0
    assert lib.car_linear_x3_masked(_ptr(X), ldx, _ptr(tiles), _ptr(bdev), K, N, _ptr(got), ldy, M, flags, _ptr(act), N + 2, st) != 0


# ----------------------------------------------------------------------------------------------------------
# stage kernels
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("chans", [(8, 12, 4), (8, 16, 4), (4, 4, 4), (256, 64, 8), (4,)])
def test_gather_matches_grid_sample(chans):
    """car_gather_bilinear against F.grid_sample.  (8, 12, 4): a quad count that is no power of two -> the per-float4 kernel;
    (8, 16, 4) / (4, 4, 4) / (4,): powers of two only, but a 4-channel level would need 64 rows per wave-task, more than a work group's 32
    -> must fall back too (it used to leave that level's columns unwritten); (256, 64, 8): the wave-task kernel proper."""
    from cross_attention_renderer_amd.engine import RenderEngine
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    n, pts = 4, 700
    sizes = [(5, 7), (9, 6), (16, 16)]
    z = [torch.randn(n, c_, *sizes[l], generator=g) for l, c_ in enumerate(chans)]
    grid = torch.rand(n, pts, 2, generator=g) * 2.6 - 1.3
    grid[0, 0] = torch.tensor([1e10, 1e10]); grid[0, 1] = torch.tensor([-1.0, 1.0]); grid[1, 2] = torch.tensor([7.8e7, -0.2])

    class Dummy:
        pass
    eng = RenderEngine.__new__(RenderEngine)
    eng.lib = _lib()
    maps = [t.permute(0, 2, 3, 1).contiguous().to(dev) for t in z]
    Ct = sum(chans)
    ld = Ct + 8
    for mode, name in ((0, "border"), (1, "zeros")):
        out = torch.full((n * pts, ld), -7.0, device=dev)
        eng.gather(maps, grid.to(dev), pts, mode, 0, 1, out, ld, 4)
        torch.cuda.synchronize()
        want = torch.cat([torch.nn.functional.grid_sample(t, grid[:, :, None, :], mode="bilinear", padding_mode=name,
                                                          align_corners=False)[..., 0].permute(0, 2, 1) for t in z], dim=-1)
        got = out.cpu().view(n, pts, ld)
        assert (got[..., 4:4 + Ct] - want).abs().max() < 1e-5, name
        assert (got[..., :4] == -7).all() and (got[..., 4 + Ct:] == -7).all()
        # the same points declared as rays x samples (a work group takes one sample of 16 neighbouring rays; 700 = 35 rays x 20,
        # 28 x 25: ragged last ray block): another order of the same rows, bit-identical; a run that does not divide pts is ignored
        for run in (20, 25, 700, 13):
            out2 = torch.full((n * pts, ld), -7.0, device=dev)
            eng.gather(maps, grid.to(dev), pts, mode, 0, 1, out2, ld, 4, run=run)
            assert torch.equal(out2, out), (name, run)


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
    # argmax: exact wherever the reference's own weights decide it by more than 1e-6; the rest are ties
    decided, wrong = argmax_exact_where_decided(got["at_wt_max"], want_of("at_wt"))
    assert decided > 0 and wrong == 0, f"{what} at_wt_max: {wrong} of {decided} decided rays differ"


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


@pytest.mark.parametrize("name", HIP_CASES)
def test_forward_with_the_whole_input_dict_on_the_device(name):
    """The reference scripts' call (render_realestate10k_traj.py:85, 128-130: dict_to_gpu moves EVERYTHING, cameras included, and then
    model(model_input, z=z)): the boundary's default route downloads the four camera tensors and runs the reference's own host pose
    algebra, so the result is bit-identical to the forward with the cameras left on the host, and strict (1e-4, every element)
    against the oracle on this host.  Against the reference's own outputs (made on another CPU) see
    test_forward_host_poses_vs_reference: same budget, because the two forwards are the same bits."""
    c, fx, ora, out = run_case(name, cameras_on_host=False)
    _check_outputs(out, lambda k: ora[k], "whole dict on the device vs oracle")
    _, _, _, host = run_case(name, cameras_on_host=True)
    for k in ("rgb", "depth_ray", "at_wt", "valid_mask", "at_wt_max", "pixel_val"):
        assert torch.equal(out[k], host[k]), k
    worst = {k: err_stats(out[k], fx["out_" + k]) for k in ("rgb", "depth_ray", "at_wt")}
    print(f"whole dict on the device vs reference fixture {name}: " + ", ".join(f"{k} beyond 1e-4 {e['f1e-4']:.5f} worst {e['max']:.2e}" for k, e in worst.items()))
    for k, e in worst.items():
        assert e["f1e-4"] <= OUTLIER_FRAC and e["max"] <= OUTLIER_MAX, f"whole dict on the device vs reference fixture {name} {k}: {e}"


@pytest.mark.parametrize("name", ["t0_default", "t0_diverging", "t1_c1", "t2_c2", "t2_c5", "t0_nview3", "t1_nview3"])
def test_literal_gather_gemm_pipeline_matches_oracle(name):
    """The literal pipeline (materialised 579-wide rows -> K=579 GEMM), kept for A/B against the default path that
    applies the first point-MLP layer per texel (csrc/car_encode.hip; for three views through car_gather_encode_rows' explicit row
    list): both must sit within 1e-4 of the oracle."""
    c, fx, ora, out = run_case(name, project_maps=False)
    _check_outputs(out, lambda k: ora[k], "literal pipeline vs oracle")
    _, _, _, out2 = run_case(name, project_maps=True, fuse_samples=False)
    assert rel_err(out["rgb"], out2["rgb"]) < 2e-5
    assert err_stats(out["stages"]["interp_val"], out2["stages"]["interp_val"])["max"] < 5e-5


@pytest.mark.parametrize("name", ["t1_c1", "t1_c1_diverging", "t1_no_sample", "t2_c2", "t2_c3", "t2_c4", "t2_c5"])
def test_one_call_route_matches_stage_pipeline(name):
    """A/B of the product route (one-call C ABI: csrc/car_render.hip sequencing csrc/car_fused.hip — geometry + encode + e +
    key/query MLPs + logits in one kernel on the f16 matrix pipe with fp16 hi/lo splits — and csrc/car_round2.hip) against the
    stage-by-stage kernels of engine.py (every layer on the fp32 matrix pipe); both against the oracle at 1e-4."""
    c, fx, ora, fused = run_case(name, fuse_samples=True)
    assert fused["stages"]["local_coords"] is None, "the one-call route was not selected"
    _, _, _, staged = run_case(name, fuse_samples=False)
    assert staged["stages"]["local_coords"] is not None
    assert rel_err(fused["stages"]["pt"], staged["stages"]["pt"]) < 1e-6
    assert rel_err(fused["stages"]["g"], staged["stages"]["local_coords"]) < 1e-6
    assert rel_err(fused["pixel_val"], staged["pixel_val"]) < 1e-6
    e = err_stats(fused["stages"]["interp_val"], staged["stages"]["interp_val"])
    assert e["max"] < 5e-5, e                                         # split-fp16 is fp32 class: inside the 1e-4 contract
    assert rel_err(fused["at_wt"], staged["at_wt"]) < 1e-5
    assert rel_err(fused["stages"]["at_wt2"], staged["stages"]["at_wt2"]) < 1e-5
    assert rel_err(fused["rgb"], staged["rgb"]) < 2e-5
    _check_outputs(fused, lambda k: ora[k], "one-call route vs oracle")
    _check_outputs(staged, lambda k: ora[k], "staged route vs oracle")


@pytest.mark.parametrize("name", ["t0_default", "t0_p5", "t0_nview1", "t0_nview3", "t1_c1", "t2_c3"])
def test_round2_logit_kernel_matches_stage_kernels(name):
    """csrc/car_round2.hip (local half of query_repeat_embed from g, second-round query layer and logits in one kernel, nothing of
    it stored) against car_linear + add_ray_bias_relu + car_linear + the logits phase of car_attend, on the staged route."""
    c, fx, ora, a = run_case(name, fuse_round2=True, fuse_samples=False)
    _, _, _, b_ = run_case(name, fuse_round2=False, fuse_samples=False)
    assert rel_err(a["stages"]["at_wt2"], b_["stages"]["at_wt2"]) < 1e-5
    assert rel_err(a["rgb"], b_["rgb"]) < 1e-5
    _check_outputs(a, lambda k: ora[k], "fused round 2 vs oracle")


def test_register_staged_weights_agree_with_lds_dma():
    """A/B of the two weight-staging variants of the fp32-pipe MFMA kernel (car_linear) on a whole forward: every layer on car_linear in
    both runs (the NO_GLDS knob keeps the split-fp16 kernel out anyway: engine.linear)."""
    fp32_pipe = lambda e: setattr(e, "linear_x3", False)
    _, _, _, a = run_case("t1_c1", fuse_samples=False, engine_setup=fp32_pipe)
    _, _, _, b_ = run_case("t1_c1", linear_flags=8, fuse_samples=False, engine_setup=fp32_pipe)
    assert rel_err(a["rgb"], b_["rgb"]) < 1e-6
    # and the knob alone must select the same kernels
    _, _, _, c_ = run_case("t1_c1", linear_flags=8, fuse_samples=False)
    assert torch.equal(b_["rgb"], c_["rgb"])


# ----------------------------------------------------------------------------------------------------------
# size-independent properties at the full bench shape (256x256, 64 samples, one 8192-ray chunk)
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,P", [(256, 64), (256, 128), (384, 64)])
def test_full_size_properties(H, P):
    """One 8192-ray chunk at the sizes of BASELINE.json's configs 2 (256 x 256, 64 samples), 4 (128 samples) and 5 (384 x 384):
    size-independent properties of the whole chunk, and 64 of its rays against the oracle."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    R = 8192
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).eval()
    S.perturb_parameters(m, seed=0)
    m.H = m.W = H
    m = m.to(dev)
    uv = S.pixel_grid(H, H)[96 * H:96 * H + R].contiguous()
    inp = to_device(S.stereo_scene(H, b=1, uv=uv, seed=5), dev, cameras_on_host=True)
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
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).eval()
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
            assert torch.equal(part["at_wt_max"].cpu(), keep["at_wt_max"][:, c0:c0 + 8192])
    assert torch.isfinite(keep["rgb"]).all()



@pytest.mark.parametrize("name,ws_mib,level_mib,pair_mib", [("t2_c3", 30, None, None), ("t2_c3", None, 2600, None), ("t2_c2", 20, None, None),
                                                           ("t2_c3", None, None, 4000)])
def test_forward_split_into_several_calls_is_bit_identical(name, ws_mib, level_mib, pair_mib):
    """The engine splits a forward into several car_render_forward calls when the workspace would not fit the free memory (ray
    chunks, then scene groups), and builds the lattices of the scenes in several groups when all of them would not leave room for it
    (each group its own car_project_maps).  Rays and scenes are independent and every call sees whole sample groups, so the result
    must not change by a single bit.  The limits are shrunk here to force the splits on small cases (level_mib: lattice bytes per
    call; pair_mib: bytes of one lattice buffer)."""
    c, fx, ora, one = run_case(name, debug=False)

    def setup(eng):
        eng.max_workspace_bytes = None if ws_mib is None else ws_mib << 20
        if level_mib is not None:
            eng.max_level_bytes = level_mib << 20
        if pair_mib is not None:
            eng.max_pair_bytes = pair_mib << 20
    calls = {}
    _, _, _, many = run_case(name, debug=False, engine_setup=lambda e: (setup(e), calls.setdefault("eng", e)))
    assert calls["eng"].last_calls > 1, "the limits did not force a split"
    assert (calls["eng"].last_pair_groups > 1) == (pair_mib is not None)
    for k in ("rgb", "valid_mask", "depth_ray", "at_wt", "at_wt_max", "coords", "pixel_val"):
        assert torch.equal(one[k], many[k]), f"{name} {k}: max abs diff {(one[k].double() - many[k].double()).abs().max().item()}"


@pytest.mark.parametrize("R,P,b", [(37, 13, 1), (16, 8, 2), (131, 70, 1), (96, 32, 5)])
def test_fused_path_ragged_sizes_against_the_oracle(R, P, b):
    """Ray counts that are no multiple of 24 and sample counts that are no multiple of 8 (the fused kernel works on tiles of
    24 rays x 8 steps — rows past the end are clamped duplicates that write their sample's values again — the round-2 kernel on blocks
    of 32 samples): real widths, H = 64, rays picked across the frame."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H = 64
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).eval()
    S.perturb_parameters(m, seed=4)
    m.H = m.W = H
    uv = C.select_rays(H, R)
    inp = S.stereo_scene(H, b=b, uv=uv, seed=7, alpha=0.35)
    z = S.feature_maps(b, 2, H, seed=2)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ora = O.render_forward(sd, inp, z, O.RenderConfig(n_view=2, npoints=P, H=H, W=H), debug=True)
        md = m.to(dev)
        out = md(to_device(inp, dev, cameras_on_host=True), z=[t.to(dev) for t in z], debug=True)
    torch.cuda.synchronize()
    assert out["stages"]["local_coords"] is None, "the fused kernel was not selected"
    assert torch.equal(out["stages"]["pt"].cpu(), ora["stages"]["pt"])
    for k in ("rgb", "depth_ray", "at_wt"):
        e = err_stats(out[k].cpu(), ora[k])
        assert e["max"] <= TOL, (k, e)
    assert torch.equal(out["valid_mask"].cpu(), ora["valid_mask"])
    decided, wrong = argmax_exact_where_decided(out["at_wt_max"].cpu(), ora["at_wt"])
    assert decided > 0 and wrong == 0



@pytest.mark.parametrize("R,P,b", [(37, 13, 1), (131, 70, 2), (96, 32, 1), (48, 5, 1)])
def test_first_round_partial_sums_equal_the_row_reduction(R, P, b):
    """car_fused_samples_parts leaves sum_j exp(logit_j - m_g) e_j per (view, ray, group of 8 steps); car_attend_parts folds the groups —
    against car_attend over the rows of e of the SAME forward (the workspace still holds e, the logits and the partial sums): the
    softmax weights, depth and argmax bit for bit (they are computed the same way), the value average to fp32 rounding.  Ragged ray and
    step counts: tiles of 24 rays x 8 steps with clamped duplicates past the end, a last step group of 1-7 live steps."""
    from cross_attention_renderer_amd import _lib as L, synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H = 64
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).eval()
    S.perturb_parameters(m, seed=4)
    m.H = m.W = H
    inp = S.stereo_scene(H, b=b, uv=C.select_rays(H, R), seed=7, alpha=0.35)
    z = S.feature_maps(b, 2, H, seed=2)
    with torch.no_grad():
        md = m.to(dev)
        out = md(to_device(inp, dev, cameras_on_host=True), z=[t.to(dev) for t in z], debug=True)
    torch.cuda.synchronize()
    eng, lib = md._engine, _lib()
    assert eng.last_calls == 1
    d = eng._dims(b, R, [t.to(dev) for t in z])
    V, n = 2, 2 * b

    def ws(name):
        off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
        L.check(lib.car_workspace_find(ctypes.byref(d), name.encode(), ctypes.byref(off), ctypes.byref(cnt)), "car_workspace_find")
        return eng._work[off.value:off.value + cnt.value]
    e, logit, part, pt = ws("e"), ws("logit"), ws("part"), ws("pt")
    ts = lib.car_fused_tile_steps()
    assert part.numel() == n * R * (-(-P // ts)) * 576
    poses = out["stages"]["poses"].to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = []
    for parts in (False, True):
        w = torch.empty(n, R, P, device=dev)
        zz = torch.full((b * R, 576), float("nan"), device=dev)
        depth = torch.empty(b, R, device=dev)
        amax = torch.empty(n, R, dtype=torch.int32, device=dev)
        if parts:
            L.check(lib.car_attend_parts(_ptr(logit), _ptr(part), ts, 576, b, V, R, P, _ptr(w), _ptr(zz), 576, 1, _ptr(pt), _ptr(poses),
                                         _ptr(depth), _ptr(amax), st), "car_attend_parts")
        else:
            L.check(lib.car_attend(_ptr(logit), None, 128, _ptr(e), 576, b, V, R, P, None, 0.0, _ptr(w), _ptr(zz), 576, 1, _ptr(pt),
                                   _ptr(poses), _ptr(depth), _ptr(amax), st), "car_attend")
        torch.cuda.synchronize()
        res.append((w, zz, depth, amax))
    (w0, z0, d0, a0), (w1, z1, d1, a1) = res
    assert torch.equal(w0, w1) and torch.equal(d0, d1) and torch.equal(a0, a1)
    assert torch.equal(w1, out["at_wt"])
    assert torch.isfinite(z1).all()
    scale = z0.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    assert ((z0 - z1).abs() / scale).max().item() < 2e-6


@pytest.mark.parametrize("name", ["t1_c1", "t1_no_sample", "t2_c2", "t2_c3"])
def test_first_round_from_partial_sums_equals_the_row_form(name):
    """The product route (first attention round over the fused kernel's partial sums) against CAR_PHASE_ROWS_FIRST_ROUND (the same round
    over the rows of e, rounds 1-4): at_wt, depth_ray, valid_mask and the argmax bit for bit, rgb to fp32 rounding."""
    c, fx, ora, a = run_case(name)
    _, _, _, b_ = run_case(name, engine_setup=lambda e: setattr(e, "first_round_parts", False))
    for k in ("at_wt", "depth_ray", "valid_mask", "at_wt_max", "pixel_val"):
        assert torch.equal(a[k], b_[k]), k
    assert rel_err(a["rgb"], b_["rgb"]) < 1e-5                         # ebar moves by 2e-6 of its largest entry; the decoder's layers follow
    _check_outputs(b_, lambda k: ora[k], "row form vs oracle")


def test_prefetched_pair_equals_the_inline_projection():
    """model.prefetch_pair (the eval loop's hook): the next pair's channel-last pyramid and lattice made on a side stream beside the current
    render — the forward that then receives those tensors returns exactly what it returns when it projects the pair itself; a pair that is
    announced and never rendered is dropped; the pair in place is not re-projected."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H, P, R, b = 64, 16, 120, 2
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).eval()
    S.perturb_parameters(m, seed=3)
    m.H = m.W = H
    m = m.to(dev)
    inp = to_device(S.stereo_scene(H, b=b, uv=C.select_rays(H, R), seed=4, alpha=0.4), dev, cameras_on_host=True)
    za, zb, zc = ([t.to(dev) for t in S.feature_maps(b, 2, H, seed=sd)] for sd in (1, 2, 3))
    keys = ("rgb", "depth_ray", "at_wt", "valid_mask", "pixel_val")
    with torch.no_grad():
        assert m.prefetch_pair(za) is False                            # no engine / plan yet: the first forward projects its pair itself
        want_b = {k: v.clone() for k, v in m(inp, z=zb).items() if k in keys}
        want_a = {k: v.clone() for k, v in m(inp, z=za).items() if k in keys}
        assert m.prefetch_pair(za) is False                            # the pair in place
        zd = [t.clone() for t in zc]
        assert m.prefetch_pair(zc) and m.prefetch_pair(zd) and m.prefetch_pair(zb)     # two may wait: zc, announced first and never rendered, is dropped
        assert len(m._engine._pf) == 2
        got_a = m(inp, z=za)                                           # renders beside the side stream's projections
        pair_before = m._engine._pair
        got_b = m(inp, z=zb)
        assert len(m._engine._pf) == 1 and m._engine._pair is not pair_before, "the forward did not take the announced pair over"
        m._engine.drop_prefetched()
        torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(got_a[k], want_a[k]), k
        assert torch.equal(got_b[k], want_b[k]), k


def test_one_call_c_abi_without_second_round():
    """repeat_attention=False (models.py:547) through car_render_forward and the oracle at real widths."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H, P, R = 64, 32, 200
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, repeat_attention=False, with_encoder=False).eval()
    S.perturb_parameters(m, seed=6)
    m.H = m.W = H
    inp = S.stereo_scene(H, b=1, uv=C.select_rays(H, R), seed=9, alpha=0.6)
    z = S.feature_maps(1, 2, H, seed=3)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ora = O.render_forward(sd, inp, z, O.RenderConfig(n_view=2, npoints=P, H=H, W=H, repeat_attention=False))
        md = m.to(dev)
        dinp, dz = to_device(inp, dev, cameras_on_host=True), [t.to(dev) for t in z]
        eng = md(dinp, z=dz)
    torch.cuda.synchronize()
    for k in ("rgb", "depth_ray", "at_wt"):
        assert err_stats(eng[k].cpu(), ora[k])["max"] <= TOL, k


def test_two_phase_entry_on_two_streams_equals_the_one_call():
    """car_render_forward_phase (include/car_hip.h): the per-sample phase of every ray batch on one stream, the per-ray phase on a second
    one behind an event, each batch in its own workspace — bit-identical to car_render_forward on the whole set of rays."""
    import ctypes
    from cross_attention_renderer_amd import _lib, synthetic as S
    from cross_attention_renderer_amd.engine import _ptr
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H, P, R, nb = 64, 32, 192, 2
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).eval()
    S.perturb_parameters(m, seed=6)
    m.H = m.W = H
    inp = S.stereo_scene(H, b=1, uv=C.select_rays(H, R), seed=9, alpha=0.6)
    z = S.feature_maps(1, 2, H, seed=3)
    md = m.to(dev)
    dinp, dz = to_device(inp, dev, cameras_on_host=True), [t.to(dev) for t in z]
    with torch.no_grad():
        ref = md(dinp, z=dz)
    eng, lib = md._engine, _lib.load()
    assert eng.last_calls == 1
    f32 = dict(device=dev, dtype=torch.float32)
    poses = eng._poses(dinp, H, 2, dev)
    uv = dinp["query"]["uv"].reshape(1, R, 2).float().contiguous()
    steps = eng._linspace(0.0, 1.0, P, dev)
    plan = eng._plan_for(eng._dims(1, R, dz), dev)
    pair, d_pair = eng._pair_for(plan, dz, dev, 0, 1, R)
    rc = R // nb
    d = eng._dims(1, rc, dz)
    need = lib.car_workspace_bytes(ctypes.byref(d))
    order = ("rgb", "valid_mask", "depth_ray", "at_wt", "at_wt_max", "coords", "pixel_val")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    keep, outs = [], []
    for c in range(nb):
        o = {"rgb": torch.empty(1, 1, rc, 3, **f32), "valid_mask": torch.empty(1, rc, 1, **f32), "depth_ray": torch.empty(1, rc, 1, **f32),
             "at_wt": torch.empty(2, rc, P, **f32), "at_wt_max": torch.empty(2, rc, 1, device=dev, dtype=torch.int32),
             "coords": torch.empty(2, rc, 9, **f32), "pixel_val": torch.empty(2, rc, P, 2, **f32)}
        u = uv[:, c * rc:(c + 1) * rc].contiguous()
        work = torch.empty(need // 4, **f32)
        ci = _lib.CarInputs()
        ci.poses, ci.uv, ci.lattice, ci.steps = poses.data_ptr(), u.data_ptr(), pair.data_ptr(), steps.data_ptr()
        ci.gmeta = pair.data_ptr() + 4 * lib.car_gmeta_offset(ctypes.byref(d_pair))
        co = _lib.CarOutputs(*[o[k].data_ptr() for k in order])
        ev = torch.cuda.Event()
        for phase, stream in ((1, sa), (2, sb)):
            if phase == 2:
                sb.wait_event(ev)
            _lib.check(lib.car_render_forward_phase(ctypes.byref(d), _ptr(plan), ctypes.byref(ci), ctypes.byref(co), _ptr(work), need, phase,
                                                    ctypes.c_void_p(stream.cuda_stream)), "car_render_forward_phase")
            if phase == 1:
                ev.record(sa)
        keep += [u, work, ci, co, ev]
        outs.append(o)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([o["rgb"] for o in outs], dim=2), ref["rgb"])
    for k in ("depth_ray", "at_wt", "valid_mask"):
        assert torch.equal(torch.cat([o[k] for o in outs], dim=1), ref[k]), k
    assert lib.car_render_forward_phase(ctypes.byref(d), _ptr(plan), ctypes.byref(ci), ctypes.byref(co), _ptr(work), need, 4, None) != 0


# ----------------------------------------------------------------------------------------------------------
# dynamic range of the split-fp16 arithmetic (csrc/car_fused_mma.h): fp16 halves keep 11 bits only between 2^-14 and 65504
# ----------------------------------------------------------------------------------------------------------
def _rescale(sd, s):
    """Moves every intermediate of the sample path by the factor s without changing the function: a layer feeding a ReLU is
    scaled by s (weights and bias), the layer consuming it by 1/s.  h, k1, q1 and the round-2 hidden layer end up s times larger."""
    sd = dict(sd)
    for first, second in (("query_encode_latent", "query_encode_latent_2"), ("key_map", "key_map_2"), ("query_embed", "query_embed_2"),
                          ("query_repeat_embed", "query_repeat_embed_2")):
        sd[first + ".weight"] = sd[first + ".weight"] * s
        sd[first + ".bias"] = sd[first + ".bias"] * s
        sd[second + ".weight"] = sd[second + ".weight"] / s
    return sd


@pytest.mark.parametrize("name", ["t1_c1", "t2_c2"])
@pytest.mark.parametrize("s", [1e-4, 1e-2, 1e2, 1e4])
def test_split_fp16_dynamic_range_of_activations(name, s):
    """Hidden activations 1e-4 ... 1e4 times their usual size (and the consuming weights the inverse): in plain fp16 halves the
    small end sinks into subnormals and the large end overflows 65504; with the power-of-two operand scaling the one-call route
    stays inside the 1e-4 contract, on the per-sample features as well as on the outputs."""
    c, fx, ora, out = run_case(name, sd_edit=lambda sd: _rescale(sd, s))
    assert out["stages"]["local_coords"] is None
    _check_outputs(out, lambda k: ora[k], f"activations x{s:g}")
    e = err_stats(out["stages"]["interp_val"], ora["stages"]["interp_val"])
    assert e["max"] <= TOL, e
    assert err_stats(out["stages"]["at_wt2"], ora["stages"]["at_wt2"])["max"] <= TOL


@pytest.mark.parametrize("s", [1e-4, 1e4])
def test_split_fp16_dynamic_range_of_the_feature_maps(s):
    """The pyramid itself s times larger with the first layer's feature columns 1/s: same function, other operand ranges for the
    per-texel projection and the gather."""
    def sd_edit(sd):
        w = sd["query_encode_latent.weight"].clone()
        w[:, :576] = w[:, :576] / s
        return dict(sd, **{"query_encode_latent.weight": w})
    c, fx, ora, out = run_case("t1_c1", sd_edit=sd_edit, z_edit=lambda z: [t * s for t in z])
    _check_outputs(out, lambda k: ora[k], f"feature maps x{s:g}")
    assert err_stats(out["stages"]["interp_val"], ora["stages"]["interp_val"])["max"] <= TOL


def test_split_fp16_with_outliers_in_the_feature_maps():
    """A handful of 1e5 outliers in an otherwise N(0,1) pyramid: the launch-wide power of two of the first layer is set by the
    outliers, the ordinary texels must keep their accuracy.  Errors are measured against the size of the values, as everywhere."""
    def z_edit(z):
        g = torch.Generator().manual_seed(11)
        z = [t.clone() for t in z]
        for t in z:
            idx = torch.randint(0, t.numel(), (12,), generator=g)
            t.view(-1)[idx] = 1e5 * torch.sign(torch.randn(12, generator=g))
        return z
    c, fx, ora, out = run_case("t1_c1", z_edit=z_edit)
    _, _, _, staged = run_case("t1_c1", z_edit=z_edit, fuse_samples=False)
    # per-sample features: a sample next to an outlier sums terms of ~1e2-1e3 per channel, so single channels that cancel to O(1)
    # cannot be reproduced to 1e-4 of THEMSELVES by any fp32 evaluation order; the error is measured against the sample's largest
    # feature instead (ordinary samples: O(1-10), i.e. the usual bar)
    want = ora["stages"]["interp_val"].double()
    scale = want.abs().amax(dim=-1, keepdim=True).clamp_min(1.0)
    for got, what in ((out["stages"]["interp_val"], "one-call"), (staged["stages"]["interp_val"], "staged")):
        assert ((got.double() - want).abs() / scale).max() <= TOL, what
    # samples next to an outlier carry features of ~1e3-1e4, which the later layers difference away: the colours are
    # ill-conditioned in fp32 itself.  The split-fp16 route must not be worse than the route that runs every layer on the fp32
    # matrix pipe (both against the oracle's own fp32).
    for k in ("rgb", "depth_ray", "at_wt"):
        e1, e0 = err_stats(out[k], ora[k]), err_stats(staged[k], ora[k])
        assert e1["max"] <= max(TOL, 3 * e0["max"]), (k, e1, e0)
    assert (np.asarray(out["valid_mask"]) == np.asarray(ora["valid_mask"])).all()


@pytest.mark.parametrize("route", ["one-call", "staged"])
def test_all_zero_layers_and_dead_relu_rows_stay_finite(route):
    """The reference initialises ResnetBlockFC.fc_1.weight to zero (resnet_block_fc.py:39) and a ReLU can kill a whole row: the
    split-fp16 power-of-two scaling sees an all-zero weight matrix / input vector there.  The scale is clamped to [2^-90, 2^43], so the
    products are exact zeros and the fp32 side passes the residual / bias through — no Inf * 0."""
    def sd_edit(sd):
        sd = dict(sd)
        for i in range(3):
            sd[f"phi.blocks.{i}.fc_1.weight"] = torch.zeros_like(sd[f"phi.blocks.{i}.fc_1.weight"])
        sd["phi.blocks.1.fc_0.bias"] = torch.full_like(sd["phi.blocks.1.fc_0.bias"], -1e3)      # relu(net) == 0 for every ray
        sd["key_map.bias"] = torch.full_like(sd["key_map.bias"], -1e3)                           # relu(k1) == 0: key = bias of key_map_2
        sd["query_repeat_embed_2.weight"] = torch.zeros_like(sd["query_repeat_embed_2.weight"])
        return sd
    c, fx, ora, out = run_case("t1_c1", sd_edit=sd_edit, fuse_samples=(route == "one-call"))
    for k in ("rgb", "depth_ray", "at_wt"):
        assert torch.isfinite(out[k]).all(), k
    _check_outputs(out, lambda k: ora[k], f"zero layers, {route}")


# ----------------------------------------------------------------------------------------------------------
# a3 on the device: car_pose_setup (fp64 Gauss-Jordan) against the host pose algebra (torch.inverse, the reference's call)
# ----------------------------------------------------------------------------------------------------------
def _device_poses(inp, H):
    lib = _lib()
    dev = torch.device("cuda:0")
    ctx, q = inp["context"], inp["query"]
    b, V = ctx["cam2world"].shape[:2]
    t = [x.float().contiguous().to(dev) for x in (ctx["cam2world"], q["cam2world"][:, 0], ctx["intrinsics"], q["intrinsics"][:, 0])]
    poses = torch.empty(b * V, 96, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.car_pose_setup(_ptr(t[0]), _ptr(t[1]), _ptr(t[2]), _ptr(t[3]), b, V, H, _ptr(poses), st)
    assert rc == 0, lib.car_last_error()
    torch.cuda.synchronize()
    return poses.cpu()


@pytest.mark.parametrize("name", HIP_CASES)
def test_device_pose_records_match_the_host_algebra(name):
    """Every one of the 96 floats of every camera of every fixture: the device's records (what a C host gets) against
    poses.pack_poses (the reference's torch.inverse / matmul on this host) and against the matrices the reference itself
    computed in the build container (stored in the fixture): a few ulp of the matrix scale."""
    from cross_attention_renderer_amd.poses import pack_poses
    from golden_util import load_case
    c, inp, z, sd, fx = load_case(name)
    got = _device_poses(inp, c["H"])
    host = pack_poses(inp, c["H"])
    ref = torch.as_tensor(fx["poses"]).float()
    V = c["n_view"]
    used = torch.ones(96, dtype=torch.bool)
    used[89:] = False                                   # padding
    used[24 + 12 * V:60] = False                        # T[s] of views that do not exist
    for want, what in ((host, "host algebra"), (ref, "reference's matrices")):
        err = (got - want).abs()[:, used]
        scale = want.abs()[:, used].clamp_min(1.0)
        assert (err / scale).max() < 2e-6, f"{name} vs {what}: {(err / scale).max().item()}"


@pytest.mark.parametrize("name", ["t1_c1", "t1_c1_diverging", "t2_c2", "t2_c3", "t2_c5"])
def test_forward_on_device_made_poses(name):
    """The whole forward on car_pose_setup's records: strict against the oracle fed the same records, and against the reference's
    own outputs with the outlier budget of test_forward_host_poses_vs_reference (last-ulp pose differences move the few samples
    whose pixel ray is nearly parallel to the query ray)."""
    from golden_util import load_case
    c, inp, z, sd, fx = load_case(name)
    poses = _device_poses(inp, c["H"])
    c, fx, ora, out = run_case(name, poses=poses)
    _check_outputs(out, lambda k: ora[k], "device poses vs oracle")
    # the opt-in device route (module.pose_route = "device", cameras on the GPU): identical to handing the engine car_pose_setup's records
    _, _, _, auto = run_case(name, cameras_on_host=False, pose_route="device")
    for k in ("rgb", "depth_ray", "at_wt"):
        assert torch.equal(auto[k], out[k]), k
    for k in ("rgb", "depth_ray", "at_wt"):
        e = err_stats(out[k], fx["out_" + k])
        print(f"device poses vs reference fixture {name} {k}: beyond 1e-4: {e['f1e-4']:.4f}, worst {e['max']:.2e}")
        assert e["f1e-4"] <= DEVICE_POSE_FRAC and e["max"] <= DEVICE_POSE_MAX, f"device poses vs reference fixture {k}: {e}"
    assert (np.asarray(out["valid_mask"]) == fx["out_valid_mask"]).all()


@pytest.mark.parametrize("sizes", [((8, 8), (16, 16), (32, 32)), ((6, 10), (24, 40)), ((16, 16),), ((4, 4), (8, 8), (16, 16), (32, 32)),
                                   ((64, 64), (128, 128), (256, 256)), ((4, 840),), ((2, 420), (4, 840))])
def test_merged_lattice_kernel_matches_grid_sample(sizes):
    """car_merge_lattice (csrc/car_render.hip merge_kernel: one 16-lane group per node writes both padding modes, zero-weight taps are
    out-of-range buffer loads, the taps come from per-axis tables a workgroup keeps in LDS) against torch's grid_sample of every level at
    the lattice nodes, summed: border and zeros padding, interior, ring and corner nodes, several maps; the bench's pyramid; and lattices too
    wide for the tables (840-texel rows: the kernel's table-free form)."""
    import torch.nn.functional as F
    from cross_attention_renderer_amd import _lib as L
    lib = _lib()
    dev = torch.device("cuda:0")
    n_maps, C = (3 if max(w for _, w in sizes) < 200 else 2), 576
    g = torch.Generator().manual_seed(5)
    levels = [torch.randn(n_maps, h, w, C, generator=g).to(dev) for h, w in sizes]
    nl = len(levels)
    ptrs = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in levels])
    hs = (ctypes.c_int * nl)(*[h for h, _ in sizes])
    wsz = (ctypes.c_int * nl)(*[w for _, w in sizes])
    lh, lw, pad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.car_merge_lattice(ptrs, hs, wsz, nl, n_maps, None, ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(pad), st), "shape")
    lat = torch.full((n_maps, 2, lh.value, lw.value, C), float("nan"), device=dev)
    L.check(lib.car_merge_lattice(ptrs, hs, wsz, nl, n_maps, _ptr(lat), None, None, None, st), "car_merge_lattice")
    torch.cuda.synchronize()
    hm, wm = max(h for h, _ in sizes), max(w for _, w in sizes)
    uy = torch.arange(lh.value, device=dev, dtype=torch.float64) - pad.value
    ux = torch.arange(lw.value, device=dev, dtype=torch.float64) - pad.value
    gy, gx = torch.meshgrid((uy + 1) / hm - 1, (ux + 1) / wm - 1, indexing="ij")
    grid = torch.stack([gx, gy], dim=-1)[None].expand(n_maps, -1, -1, -1)
    big = lh.value * lw.value > 100000                                 # the bench's lattice: fp32 reference (2.5 GB per mode in fp64)
    for mode, name in enumerate(("border", "zeros")):
        ref = [t.permute(0, 3, 1, 2) if big else t.permute(0, 3, 1, 2).double() for t in levels]
        want = sum(F.grid_sample(t, grid.to(t.dtype), mode="bilinear", padding_mode=name, align_corners=False) for t in ref)
        got = lat[:, mode].permute(0, 3, 1, 2).to(want.dtype)
        assert torch.isfinite(got).all()
        assert (got - want).abs().max().item() < (5e-5 if big else 2e-5), (name, (got - want).abs().max().item())
        del want, got, ref


@pytest.mark.parametrize("rows,N", [(1000, 288), (4097, 288), (193, 128), (64, 32)])
def test_fused_exchange_layers_equal_the_two_launches(rows, N):
    """car_lattice_encode_linear (the gather-fed instance of csrc/car_linear16.hip) against car_lattice_encode_rows followed by car_linear_x3
    on the same rows: bit-identical — border and zeros rows, points on / beyond the lattice's outer ring, several maps, ragged row counts."""
    from cross_attention_renderer_amd import _lib as L
    lib = _lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + N)
    n_maps, C = 3, 576
    sizes = ((8, 8), (16, 16), (32, 32))
    levels = [torch.randn(n_maps, h, w, C, generator=g).to(dev) for h, w in sizes]
    ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in levels])
    hs = (ctypes.c_int * 3)(*[h for h, _ in sizes])
    wsz = (ctypes.c_int * 3)(*[w for _, w in sizes])
    lh, lw, pad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.car_merge_lattice(ptrs, hs, wsz, 3, n_maps, None, ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(pad), st), "shape")
    lat = torch.empty(n_maps, 2, lh.value, lw.value, C, device=dev)
    L.check(lib.car_merge_lattice(ptrs, hs, wsz, 3, n_maps, _ptr(lat), None, None, None, st), "car_merge_lattice")
    src = (torch.randint(0, n_maps, (rows,), generator=g) | (torch.randint(0, 2, (rows,), generator=g) << 30)).to(torch.int32).to(dev)
    grid = (torch.rand(rows, 2, generator=g) * 2.8 - 1.4)
    grid[0] = torch.tensor([1e10, -1e10]); grid[1] = torch.tensor([-1.0, 1.0]); grid[2] = torch.tensor([1.0, 1.0]); grid[3] = torch.tensor([float("nan"), 0.0])
    grid = grid.to(dev)
    pe = torch.tanh(torch.randn(rows, 4, generator=g)).to(dev)
    wpt = (torch.randn(C, 4, generator=g) * 0.1).to(dev)
    W = (torch.randn(N, C, generator=g) / C ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    tiles = torch.empty(lib.car_linear_x3_packed_floats(C, N), device=dev)
    L.check(lib.car_linear_x3_pack(_ptr(W), C, C, N, _ptr(tiles), st), "pack")
    h1 = torch.empty(rows, C, device=dev)
    L.check(lib.car_lattice_encode_rows(_ptr(lat), lh.value, lw.value, pad.value, C, _ptr(src), _ptr(grid), _ptr(pe), _ptr(wpt), n_maps, rows, _ptr(h1), C, st),
            "car_lattice_encode_rows")
    for flags in (0, 2):
        want = torch.empty(rows, N, device=dev)
        L.check(lib.car_linear_x3(_ptr(h1), C, _ptr(tiles), _ptr(bias), C, N, _ptr(want), N, rows, flags, st), "car_linear_x3")
        got = torch.full((rows, N), float("nan"), device=dev)
        L.check(lib.car_lattice_encode_linear(_ptr(lat), lh.value, lw.value, pad.value, _ptr(src), _ptr(grid), _ptr(pe), _ptr(wpt), n_maps, rows, _ptr(tiles),
                                              _ptr(bias), C, N, _ptr(got), N, flags, st), "car_lattice_encode_linear")
        torch.cuda.synchronize()
        assert torch.isfinite(want).all()
        assert torch.equal(got, want), (flags, (got - want).abs().max().item())


@pytest.mark.parametrize("R,P,nsets", [(48, 16, 2), (37, 13, 3), (131, 8, 1)])
def test_fused_rows_kernel_matches_the_two_exchange_layers(R, P, nsets):
    """car_fused_rows (the fused per-sample kernel's source pass over explicit rows: tiles of 24 rays x 8 steps per (set, component), ragged ray
    and step counts) against car_lattice_encode_rows + car_linear_x3 on the same rows — the split-fp16 first layer with one power of two per
    launch against the fp32 gather: fp32 class, 5e-5 of the row's scale like the two-view routes' A/B."""
    from cross_attention_renderer_amd import _lib as L
    lib = _lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(R + P)
    n_maps, C, ncomp = 3, 576, 3
    sizes = ((8, 8), (16, 16), (32, 32))
    levels = [torch.randn(n_maps, h, w, C, generator=g).to(dev) for h, w in sizes]
    ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in levels])
    hs = (ctypes.c_int * 3)(*[h for h, _ in sizes])
    wsz = (ctypes.c_int * 3)(*[w for _, w in sizes])
    lh, lw, pad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.car_merge_lattice(ptrs, hs, wsz, 3, n_maps, None, ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(pad), st), "shape")
    lat = torch.empty(n_maps, 2, lh.value, lw.value, C, device=dev)
    L.check(lib.car_merge_lattice(ptrs, hs, wsz, 3, n_maps, _ptr(lat), None, None, None, st), "car_merge_lattice")
    S = nsets * R * P
    rows = S * ncomp
    # one (map, padding mode) per (set, component), as the exchange's row list has it
    src = torch.empty(nsets, R * P, ncomp, dtype=torch.int32)
    for a_ in range(nsets):
        for k in range(ncomp):
            src[a_, :, k] = ((a_ + k) % n_maps) | ((1 if k else 0) << 30)
    src = src.reshape(-1).to(dev)
    grid = torch.rand(rows, 2, generator=g) * 2.6 - 1.3
    grid[0] = torch.tensor([1e10, -1e10]); grid[1] = torch.tensor([-1.0, 1.0]); grid[5] = torch.tensor([float("nan"), 0.3])
    grid = grid.to(dev)
    pe = torch.tanh(torch.randn(rows, 4, generator=g)).to(dev)
    w1 = torch.randn(C, C + 3, generator=g) * 0.05
    b1 = torch.randn(C, generator=g) * 0.1
    w2 = torch.randn(C // 2, C, generator=g) / C ** 0.5
    b2 = torch.randn(C // 2, generator=g)
    w1d, b1d, w2d, b2d = [t.to(dev).contiguous() for t in (w1, b1, w2, b2)]
    wpt = torch.cat([w1[:, C:], b1[:, None]], dim=1).contiguous().to(dev)
    tiles = torch.empty(lib.car_linear_x3_packed_floats(C, C // 2), device=dev)
    L.check(lib.car_linear_x3_pack(_ptr(w2d), C, C, C // 2, _ptr(tiles), st), "pack")
    h1 = torch.empty(rows, C, device=dev)
    L.check(lib.car_lattice_encode_rows(_ptr(lat), lh.value, lw.value, pad.value, C, _ptr(src), _ptr(grid), _ptr(pe), _ptr(wpt), n_maps, rows, _ptr(h1), C, st),
            "car_lattice_encode_rows")
    want = torch.empty(rows, C // 2, device=dev)
    L.check(lib.car_linear_x3(_ptr(h1), C, _ptr(tiles), _ptr(b2d), C, C // 2, _ptr(want), C // 2, rows, 0, st), "car_linear_x3")
    blob = torch.zeros(lib.car_fused_blob_floats(), device=dev)
    fb = torch.empty(lib.car_fused_bias_floats(), device=dev)
    fwpt = torch.empty(C * 4, device=dev)
    L.check(lib.car_fused_pack_rows(_ptr(w1d), _ptr(b1d), _ptr(w2d), _ptr(b2d), _ptr(blob), _ptr(fb), _ptr(fwpt), st), "car_fused_pack_rows")
    gmeta = lat.abs().max().reshape(1).contiguous()
    got = torch.full((rows, C // 2), float("nan"), device=dev)
    L.check(lib.car_fused_rows(_ptr(lat), lh.value, lw.value, pad.value, _ptr(gmeta), _ptr(fwpt), _ptr(blob), _ptr(fb), _ptr(src), _ptr(grid), _ptr(pe),
                               nsets, R, P, ncomp, _ptr(got), st), "car_fused_rows")
    torch.cuda.synchronize()
    assert torch.equal(fwpt.view(C, 4), wpt)
    assert torch.isfinite(got).all() and torch.isfinite(want).all()
    scale = want.abs().amax(dim=1, keepdim=True).clamp_min(1.0)
    assert ((got - want).abs() / scale).max().item() < 5e-5


def test_exchange_row_lists_match_their_definition():
    """car_exchange_rows against the row lists written out with torch indexing and car_project_points (the form engine._encode_three_views
    used to build them in, ~40 launches): bit-identical."""
    from cross_attention_renderer_amd import _lib as L
    lib = _lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    b, V, pts, H = 2, 3, 500, 64
    poses = torch.randn(b * V, 96, generator=g).to(dev)
    poses[:, :] = poses.abs() + 0.5                                    # intrinsics well away from zero
    pixel_val = (torch.rand(b * V, pts, 2, generator=g) * 2 - 1).to(dev)
    pt_in = (torch.randn(b * V, pts, V, 3, generator=g) + torch.tensor([0.0, 0.0, 3.0])).to(dev).contiguous()
    ptenc = torch.tanh(torch.randn(b * V * pts * V, 4, generator=g)).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    src = torch.empty(b, V, pts, 3, dtype=torch.int32, device=dev)
    rgrid = torch.empty(b, V, pts, 3, 2, device=dev)
    rpe = torch.zeros(b, V, pts, 3, 4, device=dev)
    pe, pin, pv = ptenc.view(b, V, pts, V, 4), pt_in.view(b, V, pts, V, 3), pixel_val.view(b, V, pts, 2)
    sc = torch.arange(b, device=dev, dtype=torch.int32).view(b, 1)
    grid = torch.empty(b, pts, 2, device=dev)
    for c in range(V):
        src[:, c, :, 0] = sc * V + c
        rgrid[:, c, :, 0] = pv[:, c]
        rpe[:, c, :, 0, :3] = pe[:, c, :, c, :3]
        k = 1
        for o in range(V):
            if o == c:
                continue
            q = pin[:, o, :, c, :].contiguous()
            L.check(lib.car_project_points(_ptr(poses), _ptr(q), b, pts, V, o, H, H, _ptr(grid), st), "car_project_points")
            src[:, c, :, k] = (sc * V + o) | (1 << 30)
            rgrid[:, c, :, k] = grid
            rpe[:, c, :, k, :3] = pe[:, o, :, c, :3]
            k += 1
    s2 = torch.empty_like(src)
    g2 = torch.full_like(rgrid, float("nan"))
    p2 = torch.full_like(rpe, float("nan"))
    L.check(lib.car_exchange_rows(_ptr(poses), _ptr(pixel_val), _ptr(pt_in), _ptr(ptenc), b, V, pts, H, H, _ptr(s2), _ptr(g2), _ptr(p2), st), "car_exchange_rows")
    torch.cuda.synchronize()
    assert torch.equal(s2, src) and torch.equal(p2, rpe)
    assert torch.equal(torch.nan_to_num(g2, nan=7.0), torch.nan_to_num(rgrid, nan=7.0))


def test_three_view_route_rows_kernel_against_the_other_forms():
    """The n_view = 3 forward with the exchange on car_fused_rows (default) against the gather-fed linear kernel: fp32 rounding; against the oracle:
    the contract."""
    _, _, ora, a = run_case("t1_nview3", fuse_samples=False)
    _, _, _, b_ = run_case("t1_nview3", fuse_samples=False, engine_setup=lambda e: setattr(e, "fuse_exchange", True))
    assert rel_err(a["at_wt"], b_["at_wt"]) < 1e-5 and rel_err(a["rgb"], b_["rgb"]) < 2e-5
    assert err_stats(a["stages"]["interp_val"], b_["stages"]["interp_val"])["max"] < 5e-5
    _check_outputs(a, lambda k: ora[k], "rows kernel vs oracle")


def test_three_view_route_with_fused_exchange_equals_two_launches():
    """The n_view = 3 forward with the exchange's two layers fused (default) against the two-launch form: every output bit-identical."""
    outs = []
    for fuse in (True, False):
        _, _, ora, out = run_case("t1_nview3", fuse_samples=False, engine_setup=lambda e, f=fuse: setattr(e, "fuse_exchange", f))
        outs.append(out)
    for k in ("rgb", "depth_ray", "at_wt", "valid_mask"):
        assert torch.equal(outs[0][k], outs[1][k]), k
    _check_outputs(outs[0], lambda k: ora[k], "fused exchange vs oracle")


@pytest.mark.parametrize("M,Ce", [(5000, 576), (4099, 864), (8192, 288), (193, 96)])
def test_key_query_chain_kernel_matches_the_separate_layers(M, Ce):
    """car_key_query_logits (key_map -> relu -> key_map_2, query_embed -> relu -> query_embed_2, <key, qry> / 16 in one kernel) against
    fp64 matmuls of the same layers: qry and the logits at the fp32-class bound of the split-fp16 layers, ragged row counts, the widths of
    the one-, two- and three-view routes."""
    from cross_attention_renderer_amd import _lib as L
    lib = _lib()
    dev = torch.device("cuda:0")
    g_ = torch.Generator().manual_seed(M + Ce)
    rnd = lambda *sh: torch.randn(*sh, generator=g_)
    e = (rnd(M, Ce) * torch.logspace(-3, 2, M).unsqueeze(1)).to(dev)            # rows of very different magnitude
    gq = rnd(M, 16).to(dev)
    k1w, k1b = (rnd(128, Ce) / Ce ** 0.5).to(dev), rnd(128).to(dev)
    k2w, k2b = (rnd(128, 128) / 128 ** 0.5).to(dev), rnd(128).to(dev)
    q1w, q1b = (rnd(128, 16) / 4).to(dev), rnd(128).to(dev)
    q2w, q2b = (rnd(128, 128) / 128 ** 0.5).to(dev), rnd(128).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    tiles = torch.empty(lib.car_linear_x3_packed_floats(Ce, 128), device=dev)
    L.check(lib.car_linear_x3_pack(_ptr(k1w), Ce, Ce, 128, _ptr(tiles), st), "pack")
    tail = torch.empty(lib.car_kq_tail_floats(), device=dev)
    tb = torch.empty(lib.car_kq_bias_floats(), device=dev)
    L.check(lib.car_kq_pack(_ptr(k2w), _ptr(k2b), _ptr(q1w), _ptr(q1b), _ptr(q2w), _ptr(q2b), _ptr(tail), _ptr(tb), st), "car_kq_pack")
    qry = torch.full((M, 128), float("nan"), device=dev)
    logit = torch.full((M,), float("nan"), device=dev)
    L.check(lib.car_key_query_logits(_ptr(e), Ce, _ptr(tiles), _ptr(k1b), Ce, _ptr(gq), _ptr(tail), _ptr(tb), M, _ptr(qry), _ptr(logit), st),
            "car_key_query_logits")
    torch.cuda.synchronize()
    d = lambda t: t.double()
    k1 = torch.relu(d(e) @ d(k1w).T + d(k1b))
    key = k1 @ d(k2w).T + d(k2b)
    q = torch.relu(d(gq) @ d(q1w).T + d(q1b)) @ d(q2w).T + d(q2b)
    want = (key * q).sum(1) / 16
    qb = torch.relu(d(gq) @ d(q1w).T + d(q1b)).abs() @ d(q2w).abs().T + d(q2b).abs()
    assert ((d(qry) - q).abs() / qb).max().item() < 4e-6
    kb = k1.abs() @ d(k2w).abs().T + d(k2b).abs()
    bound = (kb * q.abs()).sum(1) / 16 + (key.abs() * qb).sum(1) / 16
    assert torch.isfinite(logit).all()
    assert ((d(logit) - want).abs() / bound).max().item() < 8e-6


@pytest.mark.parametrize("b,R,P,wscale", [(1, 77, 32, 1.0), (2, 50, 5, 1.0), (1, 300, 64, 30.0), (1, 33, 16, 1e-3)])
def test_second_round_bilinear_form_matches_the_two_layers(b, R, P, wscale):
    """car_round2_logits_from_g: <q2, qry> / 16 evaluated as y^T (M x + v) + u^T x + c with M = Wr2^T Wq2 folded by car_round2q_pack, against
    fp64 matmuls of the four separate layers (models.py:529, 549-556) and against car_round2_logits fed the fp64-made query rows; ragged sizes,
    weights of very different magnitude."""
    from cross_attention_renderer_amd import _lib as L
    lib = _lib()
    dev = torch.device("cuda:0")
    V = 2
    S = b * V * R * P
    g_ = torch.Generator().manual_seed(S)
    rnd = lambda *sh: torch.randn(*sh, generator=g_)
    gq = (rnd(S, 16) * torch.logspace(-2, 1, S).unsqueeze(1)).to(dev)
    uh = rnd(b * R, 128).to(dev)
    wr1, br1 = (rnd(128, 144) / 4).to(dev), rnd(128).to(dev)
    wr2, br2 = (rnd(128, 128) / 128 ** 0.5 * wscale).to(dev), rnd(128).to(dev)
    wq1, bq1 = (rnd(128, 16) / 4).to(dev), rnd(128).to(dev)
    wq2, bq2 = (rnd(128, 128) / 128 ** 0.5 / wscale).to(dev), rnd(128).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    wp, bp = torch.empty(lib.car_round2q_packed_floats(), device=dev), torch.empty(lib.car_round2q_bias_floats(), device=dev)
    L.check(lib.car_round2q_pack(_ptr(wr1), _ptr(br1), _ptr(wr2), _ptr(br2), _ptr(wq1), _ptr(bq1), _ptr(wq2), _ptr(bq2), _ptr(wp), _ptr(bp), st), "car_round2q_pack")
    logit = torch.full((S,), float("nan"), device=dev)
    L.check(lib.car_round2_logits_from_g(_ptr(gq), _ptr(uh), _ptr(wp), _ptr(bp), b, V, R, P, _ptr(logit), st), "car_round2_logits_from_g")
    d = lambda t: t.double()
    ray = (torch.arange(S, device=dev) // P)                               # (scene-view, ray) -> (scene, ray)
    ray = (ray // R // V) * R + ray % R
    y = torch.relu(d(gq) @ d(wr1[:, 128:]).T + d(br1) + d(uh)[ray])
    x = torch.relu(d(gq) @ d(wq1).T + d(bq1))
    q2, qry = y @ d(wr2).T + d(br2), x @ d(wq2).T + d(bq2)
    want = (q2 * qry).sum(1) / 16
    # fp32-class bound of either evaluation order: the magnitudes that enter the sums
    q2b, qb = y.abs() @ d(wr2).abs().T + d(br2).abs(), x.abs() @ d(wq2).abs().T + d(bq2).abs()
    bound = (q2b * qb).sum(1) / 16
    torch.cuda.synchronize()
    assert torch.isfinite(logit).all()
    assert ((d(logit) - want).abs() / bound).max().item() < 4e-6
    # the stored-query kernel on the same inputs (the staged routes' form)
    w0, b0 = torch.empty(lib.car_round2_packed_floats(), device=dev), torch.empty(lib.car_round2_bias_floats(), device=dev)
    L.check(lib.car_round2_pack(_ptr(wr1), _ptr(br1), _ptr(wr2), _ptr(br2), _ptr(w0), _ptr(b0), st), "car_round2_pack")
    qrows = qry.float().contiguous()
    logit0 = torch.empty(S, device=dev)
    L.check(lib.car_round2_logits(_ptr(gq), _ptr(uh), _ptr(qrows), _ptr(w0), _ptr(b0), b, V, R, P, _ptr(logit0), st), "car_round2_logits")
    torch.cuda.synchronize()
    assert ((d(logit) - d(logit0)).abs() / bound).max().item() < 4e-6


@pytest.mark.parametrize("name", ["t1_nview1", "t1_nview3", "t1_no_latent_concat"])
def test_staged_route_with_the_key_query_chain_kernel_equals_the_separate_layers(name):
    """The stage route's default (car_key_query_logits) against its five-launch form: attention weights and rgb to fp32 rounding; both
    against the oracle at the contract."""
    c, fx, ora, a = run_case(name)
    _, _, _, b_ = run_case(name, engine_setup=lambda e: setattr(e, "fuse_kq", False))
    assert rel_err(a["at_wt"], b_["at_wt"]) < 1e-5
    assert rel_err(a["rgb"], b_["rgb"]) < 2e-5
    assert torch.equal(a["valid_mask"], b_["valid_mask"])
    _check_outputs(b_, lambda k: ora[k], "five launches vs oracle")
    _check_outputs(a, lambda k: ora[k], "key / query chain kernel vs oracle")


def test_project_maps_records_the_lattice_maximum():
    """car_project_maps takes the largest |lattice value| inside the merge kernel (it used to be a second pass over the 2.5 GB): gmeta[0]
    must be exactly the maximum of the lattice it wrote."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.engine import RenderEngine
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H = 64
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=16, with_encoder=False).eval().to(dev)
    m.H = m.W = H
    eng = RenderEngine(m)
    lib = eng.lib
    for b, seed in ((1, 1), (3, 2)):
        z = [t.to(dev) for t in S.feature_maps(b, 2, H, seed=seed)]
        d = eng._dims(b, 48, z)
        plan = eng._plan_for(d, dev)
        pair, dp = eng._pair_for(plan, z, dev, 0, b, 48)
        torch.cuda.synchronize()
        off = lib.car_gmeta_offset(ctypes.byref(dp))
        lh, lw, lpad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert lib.car_lattice_shape(ctypes.byref(dp), ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(lpad)) == 0
        lattice = pair[:b * 2 * 2 * lh.value * lw.value * 576]
        assert torch.isfinite(lattice).all()
        assert pair[off].item() == lattice.abs().max().item() > 0


def test_lattice_beyond_the_fused_kernels_range_takes_the_stage_route():
    """A finest level wider than ~470 pixels makes the lattice of one (view, padding mode) exceed the 2 GiB the fused kernel can
    address with its 32-bit tap offsets: the engine must select the stage route by itself (it used to fail with a hard error), and
    car_fused_samples must refuse such a lattice with an error that names the limit.  The route test only asks the engine which
    route it would take (a 512-pixel pyramid is 3.6 GB per scene: nothing is rendered here)."""
    from cross_attention_renderer_amd.engine import RenderEngine
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=8, with_encoder=False).eval()
    eng = RenderEngine(m)
    for H, fits in ((256, True), (384, True), (464, True), (512, False)):
        m.H = m.W = H
        z = [torch.empty(2, 256, H // 4, H // 4, device="meta"), torch.empty(2, 256, H // 2, H // 2, device="meta"), torch.empty(2, 64, H, H, device="meta")]
        assert eng._common_lattice(z)
        assert eng._lattice_fits(1, 64, z) == fits, H
    lib = _lib()
    one = torch.zeros(64, device="cuda:0")
    rc = lib.car_fused_samples(_ptr(one), _ptr(one), _ptr(one), _ptr(one), 1033, 1033, 5, _ptr(one), _ptr(one), _ptr(one), _ptr(one), 1, 2, 8, 8, 512, 512, 0,
                               _ptr(one), _ptr(one), _ptr(one), _ptr(one), _ptr(one), _ptr(one), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"2 GiB" in lib.car_last_error()


def test_pyramid_without_a_common_lattice_takes_the_stage_route():
    """A pyramid whose coarser levels are no integer refinements of each other (20 x 20 under 32 x 32) has no merged lattice: the
    engine renders it through the stage kernels (per-level gathers), same contract against the oracle."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H, P, R = 64, 16, 96
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).eval()
    S.perturb_parameters(m, seed=3)
    m.H = m.W = H
    uv = S.pixel_grid(H, H)[20 * H:20 * H + R].contiguous()
    inp = S.stereo_scene(H, b=1, uv=uv, seed=9, alpha=0.3)
    g = torch.Generator().manual_seed(4)
    z = [torch.randn(2, 256, 20, 20, generator=g), torch.randn(2, 256, 32, 32, generator=g), torch.randn(2, 64, H, H, generator=g)]
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev)
    calls = []
    with torch.no_grad():
        m(to_device(inp, dev, cameras_on_host=True), z=[t.to(dev) for t in S.feature_maps(1, 2, H, seed=8)])     # builds the engine
        eng = m._engine
        one_call = eng._render_one_call
        eng._render_one_call = lambda *a, **k: (calls.append(1), one_call(*a, **k))[1]
        out = m(to_device(inp, dev, cameras_on_host=True), z=[t.to(dev) for t in z])
    torch.cuda.synchronize()
    assert not calls, "this pyramid must not reach the one-call route"
    with torch.no_grad():
        ora = O.render_forward(sd, inp, z, O.RenderConfig(n_view=2, npoints=P, H=H, W=H))
    for k in ("rgb", "depth_ray", "at_wt"):
        e = err_stats(out[k].cpu(), ora[k])
        assert e["max"] <= TOL, (k, e)
    assert torch.equal(out["valid_mask"].cpu(), ora["valid_mask"])


# ----------------------------------------------------------------------------------------------------------
# config C3 at its size: 12 scenes per call
# ----------------------------------------------------------------------------------------------------------
def test_c3_twelve_scenes_at_full_size():
    """BASELINE config 3 (batch_size 12 at 256x256, 64 samples): one forward over 12 scenes x 8192 rays — the per-GPU share when
    the frame's rays are banded over 8 ranks — through the one-call route in ONE call (12.6 M samples; the lattices of 12 scenes are
    30 GB, the workspace 43 GB; the kernel's 32-bit node offsets span one view's lattice, so the scene count is not limited), 64 rays
    of every scene against the oracle."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    H, P, R, b = 256, 64, 8192, 12
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).eval()
    S.perturb_parameters(m, seed=0)
    m.H = m.W = H
    uv = S.pixel_grid(H, H)[64 * H:64 * H + R].contiguous()
    inp = S.stereo_scene(H, b=b, uv=uv, seed=21, alpha=0.4)
    z = S.feature_maps(b, 2, H, seed=8)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(dev)
    with torch.no_grad():
        out = m(to_device(inp, dev, cameras_on_host=True), z=[t.to(dev) for t in z])
    torch.cuda.synchronize()
    assert m._engine.last_calls == 1
    assert torch.isfinite(out["rgb"]).all()
    idx = torch.linspace(0, R - 1, 64).long()
    sub = {"context": inp["context"], "query": dict(inp["query"], uv=inp["query"]["uv"][:, :, idx].contiguous())}
    with torch.no_grad():
        ora = O.render_forward(sd, sub, z, O.RenderConfig(n_view=2, npoints=P, H=H, W=H))
    for k, got in (("rgb", out["rgb"][:, :, idx]), ("depth_ray", out["depth_ray"][:, idx]), ("at_wt", out["at_wt"][:, idx])):
        e = err_stats(got.cpu(), ora[k])
        assert e["max"] <= TOL, (k, e)
    assert torch.equal(out["valid_mask"][:, idx].cpu(), ora["valid_mask"])
    decided, wrong = argmax_exact_where_decided(out["at_wt_max"][:, idx].cpu(), ora["at_wt"])
    assert wrong == 0
    assert m._engine.max_level_bytes is None          # no scene limit: the kernel's 32-bit offsets span one (view, padding mode) lattice


def test_forward_without_z_runs_get_z_on_the_device():
    """``model(input)`` with z=None (the reference's training / validation call, training.py:92): get_z — ImageNet normalisation, the
    multi-view DPT-hybrid encoder, conv_map — runs on the GPU in stock PyTorch-ROCm ops, its pyramid goes straight into the HIP
    render.  The pyramid is compared with the same module's CPU get_z (summation-order noise only), the render with the oracle fed
    that very pyramid (1e-4)."""
    import encoder_cases as EC
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    dev = torch.device("cuda:0")
    P, R = 32, 96
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P).eval()
    sd = EC.seeded_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=11)
    # the untrained encoder's pyramid has magnitudes of ~1e2 and the seeded value path gives colours of ~1e2 (fp32 summation noise
    # of 1e-4 x |rgb| on both sides); bring the first point-MLP layer's view of the pyramid and the value layer back to O(1) so that
    # the render is as well conditioned as with a trained model
    sd["query_encode_latent.weight"][:, :576] /= 100.0
    sd["latent_value.weight"] /= 30.0
    sd["latent_value.bias"] /= 30.0
    m.load_state_dict(sd, strict=True)
    inp = EC.context_pair()
    inp["query"]["uv"] = C.select_rays(EC.H, R)[None, None].contiguous()
    with torch.no_grad():
        z_cpu = m.get_z(inp)
        md = m.to(dev)
        out = md(to_device(inp, dev, cameras_on_host=True))
    torch.cuda.synchronize()
    z_gpu = [t.cpu() for t in out["z"]]
    for a, b_ in zip(z_gpu, z_cpu):
        assert a.shape == b_.shape
        assert (a - b_).abs().max() <= 2e-3 * b_.pow(2).mean().sqrt() + 1e-5
    with torch.no_grad():
        ora = O.render_forward({k: v for k, v in sd.items() if not k.startswith("encoder.")}, inp, z_gpu,
                               O.RenderConfig(n_view=2, npoints=P, H=EC.H, W=EC.H))
    for k in ("rgb", "depth_ray", "at_wt"):
        e = err_stats(out[k].cpu(), ora[k])
        assert e["max"] <= TOL, (k, e)
    assert torch.equal(out["valid_mask"].cpu(), ora["valid_mask"])
