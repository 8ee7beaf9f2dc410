"""CPU unit tests of the product's geometry header (csrc/car_geom.h) against the oracle.

The header is `__host__ __device__`: g++ compiles the same inline functions the HIP kernels execute into a
small shim (tests/host/car_geom_host.cpp), so ray clipping, the fp64 Pluecker intersection, the cross-view
projection and the bilinear tap logic are checked here without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

import cases as C
from golden_util import load_case, rel_err
from oracle import car_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "car_geom_host.cpp")
INC = os.path.join(ROOT, "cross_attention_renderer_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "host", "_build", "libcar_geom_host.so")


@pytest.fixture(scope="module")
def shim():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = [SRC, os.path.join(INC, "car_geom.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-shared", "-fPIC",
                               "-I", INC, SRC, "-o", OUT])
    lib = ctypes.CDLL(OUT)
    assert lib.host_sizeof_pose() == 96 * 4 and lib.host_sizeof_ray() == 12 * 4
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def run_shim(lib, c, inp, host_poses=True):
    """host_poses=True: pose records packed by the product's host code (torch.inverse, like the reference);
    False: the library's own fp64 Gauss-Jordan ``car_pose_setup``."""
    from cross_attention_renderer_amd.poses import pack_poses
    b, V, P, H = c["b"], c["n_view"], c["P"], c["H"]
    R = inp["query"]["uv"].shape[2]
    f = lambda t: np.ascontiguousarray(t.numpy(), dtype=np.float32)
    if host_poses:
        poses = np.ascontiguousarray(pack_poses(inp, H).numpy())
    else:
        poses = np.zeros((b * V, 96), np.float32)
        lib.host_pose_setup(_ptr(f(inp["context"]["cam2world"])), _ptr(f(inp["query"]["cam2world"])),
                            _ptr(f(inp["context"]["intrinsics"])), _ptr(f(inp["query"]["intrinsics"])),
                            b, V, H, _ptr(poses))
    rays = np.zeros((b * V, R, 12), np.float32)
    lib.host_ray_setup(_ptr(poses), _ptr(f(inp["query"]["uv"])), b, V, R, _ptr(rays))
    ssz = lib.host_sizeof_sample() // 4
    samples = np.zeros((b * V, R, P, ssz), np.float32)
    interval = np.ascontiguousarray(torch.linspace(0, 1, P).numpy())
    lib.host_sample_setup(_ptr(poses), _ptr(rays), _ptr(interval), b, V, R, P, H, H, _ptr(samples))
    return poses, rays, samples


@pytest.mark.parametrize("name", ["t0_default", "t0_query_at_ctx0", "t0_query_at_ctx1", "t0_diverging", "t0_p5",
                                  "t1_c1_diverging", "t2_c2", "t2_c5"])
def test_geometry_header_matches_oracle(shim, name):
    c, inp, z, sd, fx = load_case(name)
    b, V, P, H = c["b"], c["n_view"], c["P"], c["H"]
    cfg = O.RenderConfig(n_view=V, npoints=P, H=H, W=H)
    with torch.no_grad():
        ora = O.render_forward(sd, inp, z, cfg, debug=True)
    st = ora["stages"]
    # the library's own pose kernel (fp64 Gauss-Jordan) agrees with torch.inverse/LAPACK to a few ulp
    poses_gj, _, _ = run_shim(shim, c, inp, host_poses=False)
    poses, rays, samples = run_shim(shim, c, inp, host_poses=True)
    assert rel_err(poses_gj, poses) < 2e-6

    # with the reference's own pose algebra on the host, the header reproduces the oracle's geometry exactly
    assert np.array_equal(rays[..., 0:6], st["lf"].numpy())
    assert np.array_equal(samples[..., 0:2], st["pixel_val"].numpy())
    assert np.array_equal(samples[..., 2:5], st["pt"].numpy())
    flips = (rays[..., 10] != st["overlaps"].float().numpy()).mean()
    assert flips == 0.0, f"overlap flag flips: {flips}"
    start = torch.as_tensor(rays[..., 6:8]); end = torch.as_tensor(rays[..., 8:10])
    pv = start[:, :, None] + (end - start)[:, :, None] * torch.linspace(0, 1, P)[None, None, :, None]
    assert rel_err(pv, st["pixel_val"]) < 2e-5
    # samples (the shim recomputes pixel_val itself): pixel_val, 3-D point, cross-view grid, geometric query
    assert rel_err(samples[..., 0:2], st["pixel_val"]) < 2e-5
    pt, pt_ref = torch.as_tensor(samples[..., 2:5]).double(), st["pt"].double()
    # the intersection is ill-conditioned where the pixel ray is nearly parallel to the query ray: judge it on
    # a relative scale and allow a tiny budget of outliers driven by last-ulp differences in the pose inverse
    err = ((pt - pt_ref).abs() / pt_ref.abs().clamp_min(1.0)).max(dim=-1).values
    assert (err > 1e-3).double().mean() < 0.01, f"pt mismatch fraction {(err > 1e-3).double().mean()}"
    g = torch.as_tensor(samples[..., 5:21])
    assert rel_err(g[..., :9], st["local_coords"][..., :9]) < 2e-5
    assert (g[..., 9:13] - st["local_coords"][..., 9:13]).abs().max() < 1e-6      # tanh: libm vs SLEEF
    # where the point lands in the *other* view
    gi = torch.as_tensor(samples[..., 21 + 9:21 + 9 + 6]).reshape(b, V, -1, P, 3, 2)
    other = torch.stack([gi[:, 1, :, :, 0], gi[:, 0, :, :, 1]], dim=1).flatten(0, 1)
    ref = st["pixel_val_stack"]
    d = ((other - ref).abs() / ref.abs().clamp_min(1.0)).max(dim=-1).values
    assert (d > 1e-3).double().mean() < 0.01


def test_bilinear_taps_match_grid_sample(shim):
    g = torch.Generator().manual_seed(0)
    Hl, Wl, Cc = 6, 7, 5
    feat = torch.randn(1, Cc, Hl, Wl, generator=g)
    grid = torch.rand(1, 1, 400, 2, generator=g) * 3 - 1.5
    grid[0, 0, 0] = torch.tensor([1e10, -1e10]); grid[0, 0, 1] = torch.tensor([-1.0, 1.0])
    grid[0, 0, 2] = torch.tensor([7.8e7, 0.1]); grid[0, 0, 3] = torch.tensor([1.0, -1.0])
    gnp = np.ascontiguousarray(grid.reshape(-1, 2).numpy())
    fl = feat[0].permute(1, 2, 0).reshape(Hl * Wl, Cc)
    for mode, name in ((0, "border"), (1, "zeros")):
        idx = np.zeros((400, 4), np.int32); w = np.zeros((400, 4), np.float32)
        shim.host_bilinear_taps(_ptr(gnp), 400, Wl, Hl, mode, _ptr(idx), _ptr(w))
        assert idx.min() >= 0 and idx.max() < Hl * Wl
        got = (fl[torch.as_tensor(idx).long()] * torch.as_tensor(w)[..., None]).sum(1)
        want = torch.nn.functional.grid_sample(feat, grid, mode="bilinear", padding_mode=name,
                                               align_corners=False)[0, :, 0].T
        assert (got - want).abs().max() < 1e-5, name


@pytest.mark.parametrize("sizes", [((8, 8), (16, 16)), ((4, 6), (16, 24), (8, 12)), ((16, 16),)])
def test_merged_lattice_equals_the_sum_of_the_levels(shim, sizes):
    """car_lattice_taps / the merge kernel's arithmetic: four taps of the lattice on which the levels are summed (built with the
    levels' own padding rule) equal the sum of the levels' grid_sample, for border and zeros padding, inside, on and far outside
    the maps (the identity behind the fused kernel's gather)."""
    g = torch.Generator().manual_seed(1)
    Cc = 5
    feats = [torch.randn(1, Cc, h, w, generator=g) for h, w in sizes]
    hm, wm = max(h for h, _ in sizes), max(w for _, w in sizes)
    r = [hm // h for h, _ in sizes]
    assert all(hm % h == 0 and wm // w == hm // h for h, w in sizes)
    pad = max(r) + 1
    lh, lw = 2 * hm + 2 * max(r) + 1, 2 * wm + 2 * max(r) + 1
    lat = np.zeros((2, lh, lw, Cc), np.float32)
    cl = [np.ascontiguousarray(f[0].permute(1, 2, 0).numpy()) for f in feats]
    ptrs = (ctypes.c_void_p * len(cl))(*[a.ctypes.data for a in cl])
    ia = lambda v: (ctypes.c_int * len(v))(*v)
    shim.host_lattice_build(ptrs, ia([h for h, _ in sizes]), ia([w for _, w in sizes]), ia(r), len(sizes), Cc, lh, lw, pad, _ptr(lat))
    n = 2000
    grid = torch.rand(1, 1, n, 2, generator=g) * 2.6 - 1.3
    grid[0, 0, 0] = torch.tensor([1e10, -1e10]); grid[0, 0, 1] = torch.tensor([-1.0, 1.0]); grid[0, 0, 2] = torch.tensor([7.8e7, 0.1])
    grid[0, 0, 3] = torch.tensor([1.0, -1.0]); grid[0, 0, 4] = torch.tensor([-1.0 + 1.0 / wm, 1.0 - 1.0 / hm])     # outermost texel centres
    gnp = np.ascontiguousarray(grid.reshape(-1, 2).numpy())
    node = np.zeros(n, np.int32); flags = np.zeros(n, np.int32); w = np.zeros((n, 4), np.float32)
    shim.host_lattice_taps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    shim.host_lattice_taps(_ptr(gnp), n, lw, lh, pad, float(wm), float(hm), _ptr(node), _ptr(flags), _ptr(w))
    east, south, ring = flags & 1, (flags >> 1) & 1, (flags >> 2) & 1
    assert node.min() >= 0 and (node + east + south * lw).max() < lh * lw
    # the fused kernel addresses the taps as "node, +1 node, +1 row, +both" without a decode: the north-west node is never in the last
    # column / row (a point clamped onto the far edge takes the cell before it with weights (0, 1))
    assert (east == 1).all() and (south == 1).all() and (node % lw).max() <= lw - 2 and (node // lw).max() <= lh - 2
    far = (gnp[:, 0] > 1.2) & (np.abs(gnp[:, 1]) < 1.0)
    assert far.any() and np.all(w[far][:, 0] + w[far][:, 2] == 0.0)          # clamped in x: everything on the east taps
    assert np.all(np.abs(w.sum(1) - 1.0) < 1e-6) and w.min() >= 0.0
    idx = np.stack([node, node + east, node + south * lw, node + east + south * lw], 1)
    # bit 2: on or beyond the outer ring, where the zeros-padding lattice is exactly zero (the fused kernel does not fetch those taps)
    assert ring[0] == 1 and ring[2] == 1 and 0 < ring.mean() < 0.5
    # (a tap with a non-zero value there is the far one of a point clamped onto the ring: its weight is exactly zero)
    assert (lat[1].reshape(lh * lw, Cc)[idx[ring == 1]] * w[ring == 1][..., None] == 0).all()
    for mode, name in ((0, "border"), (1, "zeros")):
        flat = torch.as_tensor(lat[mode].reshape(lh * lw, Cc))
        got = (flat[torch.as_tensor(idx).long()] * torch.as_tensor(w)[..., None]).sum(1)
        want = sum(torch.nn.functional.grid_sample(f, grid, mode="bilinear", padding_mode=name, align_corners=False) for f in feats)[0, :, 0].T
        assert (got - want).abs().max() < 2e-5, (name, (got - want).abs().max())
