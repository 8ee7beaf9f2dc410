"""SURVEY.md §8 f4 on the device: ``training.render_train`` (forward of the staged route + HIP backward, csrc/car_backward.hip) against
the REFERENCE's autograd gradients (tests/golden/grad_*.npz) — every renderer parameter on the path and every pyramid level, for the
scalar L = sum(rgb * c_rgb) + sum(depth_ray * c_depth) with the fixtures' seeded cotangents."""
import ctypes

import numpy as np
import pytest
import torch

import grad_cases as G
from golden_util import load_case, rel_err
from hip_harness import build_module, oracle_cfg, to_device

pytestmark = pytest.mark.gpu


def _lib():
    from cross_attention_renderer_amd import _lib as L
    return L.load()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("name", G.GRAD_CASES)
def test_hip_backward_matches_the_reference_gradients(name):
    from cross_attention_renderer_amd.training import render_train
    dev = torch.device("cuda:0")
    c, inp, z, sd, fx_fwd = load_case(name)
    fx = np.load(G.grad_fixture_path(name))
    m = build_module(c, sd, dev).train()
    zr = [t.to(dev).requires_grad_(True) for t in z]
    out = render_train(m, to_device(inp, dev, cameras_on_host=True), z=zr)
    # the training forward is the staged route: same results as the inference path's fixtures
    assert rel_err(out["rgb"].detach().cpu(), fx_fwd["out_rgb"]) < 1e-4
    assert rel_err(out["depth_ray"].detach().cpu(), fx_fwd["out_depth_ray"]) < 1e-4
    c_rgb, c_depth = G.cotangents(out["rgb"].shape, out["depth_ray"].shape)
    loss = (out["rgb"] * c_rgb.to(dev)).sum() + (out["depth_ray"] * c_depth.to(dev)).sum()
    assert abs(loss.item() - float(fx["loss"])) <= 1e-4 * max(1.0, abs(float(fx["loss"])))
    loss.backward()
    torch.cuda.synchronize()
    got = {"param." + k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    got.update({f"z.{l}": t.grad for l, t in enumerate(zr)})
    stored = G.keys(fx)
    assert sorted(got) == stored, sorted(set(got) ^ set(stored))
    report = {k: G.compare(fx, k, got[k], tol=1e-3) for k in stored}
    worst = max(v[0] for v in report.values())
    print(f"{name}: worst deviation {worst:.2e} of a tensor's largest entry; beyond 1e-3: " +
          ", ".join(f"{k} {v[1]:.1e}" for k, v in report.items() if v[1] > 0))
    for k in fx["unused"].tolist():
        p = dict(m.named_parameters())[k]
        assert p.grad is None, k


def test_wgrad_kernel_matches_torch():
    """car_linear_wgrad alone: dW += dY^T X, db += sum dY, ragged sizes, strides, the relu-on-load flag and accumulation."""
    lib, dev = _lib(), torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    # (4099, 288, 576), (8221, 576, 579), (5000, 128, 576), (4100, 200, 100): wide layers over >= 4096 rows -> the bf16 x 3 kernel (ragged
    # tiles in both directions, the bias column at 579 = a tile's last live column); the rest: the fp32-pipe kernels
    # (2304, ...): the per-ray layers of a training step; (2304 / 4099, 128, 16): a sparsely filled tile of the bf16 kernel (round 6: every layer
    # over >= 2048 rows with N K >= 1024 takes it); (2047, ...) and (3000, 3, 128): just below either threshold, the fp32 pipe's kernels
    for M, N, K, relu in ((1000, 128, 16, False), (777, 3, 128, True), (4099, 288, 576, False), (513, 576, 579, False), (64, 128, 144, True),
                          (8221, 576, 579, True), (5000, 128, 576, False), (4100, 200, 100, True), (2304, 288, 576, False), (2304, 128, 16, True),
                          (4099, 128, 16, False), (2304, 128, 128, True), (2047, 288, 576, False), (3000, 3, 128, False), (2048, 32, 32, True)):
        ldy, ldx = N + (4 - N % 4) % 4 + 4, K + (4 - K % 4) % 4
        dy = torch.randn(M, ldy, generator=g).to(dev)
        x = torch.randn(M, ldx, generator=g).to(dev)
        # the row padding of both operands is poison: the kernels stage columns past an operand's edge as they come (they only reach
        # entries of dW that are never written) or zero them — either way nothing of it may show in dW / db
        dy[:, N:] = float("nan")
        x[:, K:] = float("inf")
        dw = torch.randn(N, K + 5, generator=g).to(dev)
        db = torch.randn(N, generator=g).to(dev)
        dw0, db0 = dw.clone(), db.clone()
        rc = lib.car_linear_wgrad(_ptr(dy), ldy, _ptr(x), ldx, M, N, K, 1 if relu else 0, _ptr(dw), K + 5, _ptr(db), _stream())
        assert rc == 0, lib.car_last_error()
        torch.cuda.synchronize()
        xs = (x[:, :K].clamp_min(0) if relu else x[:, :K]).double()
        want_w = dw0[:, :K].double() + dy[:, :N].double().t() @ xs
        want_b = db0.double() + dy[:, :N].double().sum(0)
        scale = want_w.abs().max().item()
        assert (dw[:, :K].double() - want_w).abs().max().item() <= 2e-5 * scale, (M, N, K)
        assert torch.equal(dw[:, K:], dw0[:, K:]), "columns beyond K were touched"
        assert (db.double() - want_b).abs().max().item() <= 2e-5 * want_b.abs().max().item(), (M, N, K)
        # the same product kept on the fp32 matrix pipe (CAR_WGRAD_FP32 = 16): the two kernels agree to the same bound
        dw2, db2 = dw0.clone(), db0.clone()
        assert lib.car_linear_wgrad(_ptr(dy), ldy, _ptr(x), ldx, M, N, K, (1 if relu else 0) | 16, _ptr(dw2), K + 5, _ptr(db2), _stream()) == 0
        torch.cuda.synchronize()
        assert (dw2[:, :K].double() - want_w).abs().max().item() <= 2e-5 * scale, (M, N, K, "fp32 pipe")
        assert (dw2 - dw).abs().max().item() <= 2e-5 * scale


def test_wgrad_bf16_split_keeps_small_and_large_rows():
    """The bf16 x 3 weight-gradient kernel on operands whose rows span twelve orders of magnitude (bf16 keeps fp32's exponent: no scale is
    chosen anywhere) and on columns that cancel: the error stays rounding-like — 2e-5 of sum |dY||X| per entry against fp64."""
    lib, dev = _lib(), torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    M, N, K = 6000, 288, 576
    dy = (torch.randn(M, N, generator=g) * torch.logspace(-8, 4, M).unsqueeze(1)).to(dev)
    x = (torch.randn(M, K, generator=g) * torch.logspace(3, -6, M).unsqueeze(1)).to(dev)
    x[:, :8] = 1.0                                                      # constant columns: entries that are sums with little cancellation
    dw = torch.zeros(N, K, device=dev)
    assert lib.car_linear_wgrad(_ptr(dy), N, _ptr(x), K, M, N, K, 0, _ptr(dw), K, None, _stream()) == 0, lib.car_last_error()
    torch.cuda.synchronize()
    want = dy.double().t() @ x.double()
    bound = dy.double().abs().t() @ x.double().abs()
    assert torch.isfinite(dw).all()
    assert ((dw.double() - want).abs() / bound).max().item() <= 2e-5


def test_gather_backward_is_the_adjoint_of_the_gather():
    """<gather(maps), dout> == <maps, gather_backward(dout)> for both padding modes and every placement (the adjoint identity of a
    linear map), with coordinates inside, on the edge of and far outside the maps."""
    from cross_attention_renderer_amd.engine import PLACE_OTHER2, PLACE_OWN, PLACE_PLAIN
    lib, dev = _lib(), torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    n_maps, pts, V = 4, 300, 2
    maps = [torch.randn(n_maps, h, h, cch, generator=g).to(dev) for h, cch in ((5, 8), (9, 12), (17, 4))]
    C = sum(t.shape[3] for t in maps)
    grid = (torch.rand(n_maps, pts, 2, generator=g) * 3 - 1.5)
    grid[0, 0] = torch.tensor([1e10, -1e10])
    grid[1, 1] = torch.tensor([-1.0, 1.0])
    grid = grid.to(dev)
    L = len(maps)
    cs = (ctypes.c_int * L)(*[t.shape[3] for t in maps])
    hs = (ctypes.c_int * L)(*[t.shape[1] for t in maps])
    ws = (ctypes.c_int * L)(*[t.shape[2] for t in maps])
    for mode in (0, 1):
        for place, rows, ld in ((PLACE_PLAIN, n_maps * pts, C + 4), (PLACE_OWN, n_maps * pts * V, C + 8), (PLACE_OTHER2, n_maps * pts * V, C)):
            out = torch.zeros(rows, ld, device=dev)
            ptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in maps])
            assert lib.car_gather_bilinear(ptrs, cs, hs, ws, L, n_maps, _ptr(grid), pts, 1, mode, place, V, _ptr(out), ld, 0, _stream()) == 0
            dout = torch.randn(rows, ld, generator=g).to(dev)
            dmaps = [torch.zeros_like(t) for t in maps]
            dptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in dmaps])
            rc = lib.car_gather_bilinear_backward(dptrs, cs, hs, ws, L, n_maps, _ptr(grid), pts, mode, place, V, _ptr(dout), ld, 0, _stream())
            assert rc == 0, lib.car_last_error()
            torch.cuda.synchronize()
            lhs = (out[:, :C].double() * dout[:, :C].double()).sum().item()
            rhs = sum((a.double() * b_.double()).sum().item() for a, b_ in zip(maps, dmaps))
            assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs)), (mode, place, lhs, rhs)


def test_binned_scatter_equals_the_atomic_scatter():
    """car_gather_bilinear_backward_binned (taps binned by texel, every texel written once) against car_gather_bilinear_backward (fp32
    atomics into zeroed maps), launch for launch summed: one and two gathers with different grids, padding modes and placements reading
    one dout; coordinates inside, on the edge of and far outside the maps, points piled on one texel; channel counts of 1, 3 and 64 float4s
    per texel; maps the binned form must overwrite (NaN before the call)."""
    from cross_attention_renderer_amd.engine import PLACE_OTHER2, PLACE_OWN, PLACE_PLAIN
    lib, dev = _lib(), torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    for (n_maps, pts, V, shapes) in ((4, 300, 2, ((5, 8), (9, 12), (17, 4))), (2, 5000, 2, ((32, 256), (16, 256), (64, 64))), (3, 77, 1, ((7, 4),))):
        maps = [torch.randn(n_maps, h, h + 1, cch, generator=g).to(dev) for h, cch in shapes]
        C = sum(t.shape[3] for t in maps)
        L = len(maps)
        cs = (ctypes.c_int * L)(*[t.shape[3] for t in maps])
        hs = (ctypes.c_int * L)(*[t.shape[1] for t in maps])
        ws = (ctypes.c_int * L)(*[t.shape[2] for t in maps])
        grids = []
        for k in range(2):
            grid = (torch.rand(n_maps, pts, 2, generator=g) * 3 - 1.5)
            grid[0, 0] = torch.tensor([1e10, -1e10])
            grid[-1, 1] = torch.tensor([-1.0, 1.0])
            grid[0, 10:60] = torch.tensor([0.123, -0.4])                          # fifty points on one texel
            grids.append(grid.to(dev))
        combos = [[(grids[0], 0, PLACE_PLAIN)], [(grids[1], 1, PLACE_PLAIN)]]
        if V == 2:
            combos += [[(grids[0], 0, PLACE_OWN), (grids[1], 1, PLACE_OTHER2)], [(grids[1], 1, PLACE_OWN)]]
        for gathers in combos:
            rows = n_maps * pts * (1 if gathers[0][2] == PLACE_PLAIN else V)
            ld = C + 8
            dout = torch.randn(rows, ld, generator=g).to(dev)
            want = [torch.zeros_like(t) for t in maps]
            dptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in want])
            for grid, mode, place in gathers:
                assert lib.car_gather_bilinear_backward(dptrs, cs, hs, ws, L, n_maps, _ptr(grid), pts, mode, place, V, _ptr(dout), ld, 4, _stream()) == 0
            got = [torch.full_like(t, float("nan")) for t in maps]
            gptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in got])
            G = len(gathers)
            gr = (ctypes.c_void_p * G)(*[x[0].data_ptr() for x in gathers])
            mo = (ctypes.c_int * G)(*[x[1] for x in gathers])
            pl = (ctypes.c_int * G)(*[x[2] for x in gathers])
            nbytes = lib.car_scatter_workspace_bytes(hs, ws, L, n_maps, pts, G)
            assert nbytes > 0
            work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.car_gather_bilinear_backward_binned(gptrs, cs, hs, ws, L, n_maps, gr, mo, pl, G, pts, V, _ptr(dout), ld, 4, _ptr(work), nbytes, _stream())
            assert rc == 0, lib.car_last_error()
            torch.cuda.synchronize()
            for a, b_ in zip(got, want):
                assert torch.isfinite(a).all()
                assert (a - b_).abs().max().item() <= 2e-5 * max(1.0, b_.abs().max().item()), (n_maps, pts, len(gathers))
            # a workspace one byte short is refused
            assert lib.car_gather_bilinear_backward_binned(gptrs, cs, hs, ws, L, n_maps, gr, mo, pl, G, pts, V, _ptr(dout), ld, 4, _ptr(work), nbytes - 1,
                                                           _stream()) != 0


def test_a_gradient_step_lowers_the_image_loss_by_the_predicted_amount():
    """The reference's training step (training.py:92-136) with render_train in the place of model(model_input): L1 image loss on
    192 random rays of two synthetic scenes, backward through the HIP kernels, one plain gradient step on the renderer's parameters AND
    on the feature pyramid (standing in for the encoder's output).  The step is sized for a first-order decrease of 1 % of the loss;
    the loss measured after it must fall by that amount to within a factor 1.5 — an end-to-end check of every gradient at once, and of
    the engine picking up the updated parameters (packed weights and channel-last maps are cached on the tensors' versions)."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    from cross_attention_renderer_amd.training import render_train
    dev = torch.device("cuda:0")
    H, P, R = 64, 32, 192
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).train()
    S.perturb_parameters(m, seed=0)
    m.H = m.W = H
    m = m.to(dev)
    g = torch.Generator().manual_seed(3)
    uv = S.pixel_grid(H, H)[torch.randperm(H * H, generator=g)[:R]].contiguous()
    inp = S.stereo_scene(H, b=2, uv=uv, seed=5)
    inp = {k: {kk: (vv if kk in ("cam2world", "intrinsics") else vv.to(dev)) for kk, vv in v.items()} for k, v in inp.items()}
    z = [t.to(dev).requires_grad_(True) for t in S.feature_maps(2, 2, H, seed=1)]
    target = (torch.rand(2, 1, R, 3, generator=g) * 2 - 1).to(dev)

    def loss_of():
        return (render_train(m, inp, z=z)["rgb"] - target).abs().mean()           # loss_functions.image_loss
    loss0 = loss_of()
    loss0.backward()
    leaves = [p for p in list(m.parameters()) + z if p.grad is not None]
    assert len(leaves) == 45                                                         # every layer on the path (42 tensors) and the three levels
    gnorm2 = sum((p.grad.double() ** 2).sum().item() for p in leaves)
    predicted = 0.01 * loss0.item()
    eta = predicted / gnorm2
    with torch.no_grad():
        for p in leaves:
            p -= eta * p.grad
    loss1 = loss_of().item()
    drop = loss0.item() - loss1
    assert np.isfinite(loss1) and predicted / 1.5 <= drop <= predicted * 1.5, (loss0.item(), loss1, predicted)


def test_round6_default_routes_agree_with_the_conservative_ones_at_a_training_steps_row_counts():
    """At the row counts of the reference's training step (2304 rays: b x R = 2 x 1152 here; 147 456 sample-source rows) the default routes of
    round 6 — weight gradients on the bf16 x 3 kernel from 2048 rows on, the split-fp16 layer kernel from 2048 rows on with the ReLU mask in
    its store, the pyramid's gradient by binned taps — against the conservative ones (fp32-pipe weight gradients, the 4096-row threshold, the
    fp32-atomic scatter): every gradient within 1e-3 of its tensor's largest entry (the fixtures' bound), the forward bit for bit where the
    kernels are the same and to 2e-5 where the per-ray layers changed pipes."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    from cross_attention_renderer_amd.training import render_train
    dev = torch.device("cuda:0")
    H, P, R = 64, 32, 1152
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).train()
    S.perturb_parameters(m, seed=4)
    m.H = m.W = H
    m = m.to(dev)
    g = torch.Generator().manual_seed(7)
    uv = S.pixel_grid(H, H)[torch.randperm(H * H, generator=g)[:R]].contiguous()
    inp = to_device(S.stereo_scene(H, b=2, uv=uv, seed=5), dev, cameras_on_host=True)
    base = [t.to(dev) for t in S.feature_maps(2, 2, H, seed=1)]
    cot = torch.randn(2, 1, R, 3, generator=g).to(dev)

    def run(conservative):
        z = [t.clone().requires_grad_(True) for t in base]
        m.zero_grad(set_to_none=True)
        out = render_train(m, inp, z=z)                                               # the engine exists after the first call
        eng = m._engine
        eng.wgrad_fp32, eng.scatter_atomics, eng.linear_x3_min_rows = (True, True, 4096) if conservative else (False, False, 2048)
        z = [t.clone().requires_grad_(True) for t in base]
        m.zero_grad(set_to_none=True)
        out = render_train(m, inp, z=z)
        (out["rgb"] * cot).sum().backward()
        grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        grads.update({f"z{i}": t.grad.clone() for i, t in enumerate(z)})
        return out["rgb"].detach().clone(), grads
    try:
        rgb_a, ga = run(False)
        rgb_b, gb = run(True)
    finally:
        m._engine.wgrad_fp32, m._engine.scatter_atomics, m._engine.linear_x3_min_rows = False, False, 2048
    assert (rgb_a - rgb_b).abs().max().item() <= 2e-5 * max(1.0, rgb_b.abs().max().item())
    assert ga.keys() == gb.keys() and len(ga) == 45
    for k in ga:
        scale = gb[k].abs().max().item()
        assert torch.isfinite(ga[k]).all(), k
        assert (ga[k] - gb[k]).abs().max().item() <= 1e-3 * max(scale, 1e-12), (k, (ga[k] - gb[k]).abs().max().item(), scale)


def test_model_call_in_train_mode_carries_gradients_and_channel_last_levels_are_taken_as_they_lie():
    """The reference's training loop calls model(model_input) (training.py:92): on a module in train() mode under autograd that call IS
    render_train here; under no_grad() or in eval() mode it is the inference engine.  A pyramid level in torch.channels_last memory is used
    as a view (no NCHW -> NHWC copy) and gets its gradient back in the same layout; the outputs equal the NCHW call's bit for bit, the gradients to the rounding of their atomics' order."""
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    from cross_attention_renderer_amd.training import render_train
    dev = torch.device("cuda:0")
    H, P, R = 64, 16, 96
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).train()
    S.perturb_parameters(m, seed=2)
    m.H = m.W = H
    m = m.to(dev)
    inp = to_device(S.stereo_scene(H, b=2, uv=S.pixel_grid(H, H)[::43][:R].contiguous(), seed=6), dev, cameras_on_host=True)
    base = [t.to(dev) for t in S.feature_maps(2, 2, H, seed=1)]
    g = torch.Generator().manual_seed(1)
    cot = torch.randn(2, 1, R, 3, generator=g).to(dev)

    def grads(call, zs):
        for p in m.parameters():
            p.grad = None
        out = call(zs)
        assert out["rgb"].requires_grad and out["depth_ray"].requires_grad
        ((out["rgb"] * cot).sum() + out["depth_ray"].sum()).backward()
        return out, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, [t.grad for t in zs]
    z_nchw = [t.clone().requires_grad_(True) for t in base]
    z_cl = [t.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True) for t in base]
    out_a, pg_a, zg_a = grads(lambda zs: render_train(m, inp, z=zs), z_nchw)
    out_b, pg_b, zg_b = grads(lambda zs: m(inp, z=zs), z_cl)             # the module call itself, channel-last levels
    eng = m._engine
    assert all(mp.data_ptr() == t.data_ptr() for mp, t in zip(eng._maps, z_cl)), "a channel-last level was copied"
    assert torch.equal(out_a["rgb"], out_b["rgb"]) and sorted(pg_a) == sorted(pg_b)
    # the gradients meet in fp32 atomics (row slabs of the weight gradients, the pyramid's scatter): the same sums in another order from
    # run to run, so "equal" means to fp32 rounding of the largest entry
    close = lambda a, b_: (a - b_).abs().max().item() <= 2e-6 * max(a.abs().max().item(), 1e-30)
    for k in pg_a:
        assert close(pg_a[k], pg_b[k]), k
    for a, b_, t in zip(zg_a, zg_b, z_cl):
        assert b_.stride() == t.stride() and close(a, b_)
    with torch.no_grad():
        assert not m(inp, z=z_cl)["rgb"].requires_grad               # no_grad: the inference engine
    assert not m.eval()(inp, z=z_cl)["rgb"].requires_grad            # eval(): the inference engine (its forward never builds a graph)
