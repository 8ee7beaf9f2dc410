"""Generate the golden vectors by running the *reference* forward in this container.

    python tests/golden/make_golden.py [case ...]

For each case of ``cases.CASES`` this imports yilundu/cross_attention_renderer read-only from
``/root/reference`` (through the stubs of ``ref_import.py``), loads it with seeded weights, runs
``CrossAttentionRenderer.forward(input, z=z)`` on the CPU, checks the in-repo oracle against it and writes
``tests/golden/<case>.npz`` holding ONLY data: the small inputs (poses, intrinsics, pixel coordinates), the
case knobs, checksums of the regenerated bulky inputs, and the reference outputs.  The reference itself
never travels; the GPU box replays these files.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cases as C                                      # noqa: E402
import ref_import                                      # noqa: E402
from cross_attention_renderer_amd import synthetic as S   # noqa: E402
from oracle import car_oracle as O                     # noqa: E402


def reference_stages(model, inp, z, c):
    """Re-runs the reference's own helper functions to capture intermediates the module does not return
    (tier 0 only).  Every call below is a call INTO the reference, not a restatement."""
    ref = ref_import.load_reference()
    out = {}
    captured = {}

    def hook(name):
        def fn(mod, args, output):
            captured[name] = output.detach()
        return fn

    handles = []
    if hasattr(model, "latent_value"):
        handles.append(model.latent_value.register_forward_hook(
            lambda mod, args, output: captured.__setitem__("interp_val", args[0].detach())))
    orig_pt = ref.geometry.get_3d_point_epipolar

    def pt_spy(*a, **k):
        r = orig_pt(*a, **k)
        captured["pt"] = r[0].detach().clone()
        return r

    ref.geometry.get_3d_point_epipolar = pt_spy
    phi_handle = model.phi.register_forward_hook(
        lambda mod, args, output: captured.__setitem__("phi_in", args[0].detach()))
    try:
        with torch.no_grad():
            res = model(inp, z=z)
    finally:
        ref.geometry.get_3d_point_epipolar = orig_pt
        phi_handle.remove()
        for h in handles:
            h.remove()
    if "interp_val" in captured:
        out["interp_val"] = captured["interp_val"].permute(0, 2, 3, 1).contiguous()     # -> (bV,R,P,C)
    if "pt" in captured:
        out["pt"] = captured["pt"]
    if "phi_in" in captured:
        D = model.latent_dim
        bR = captured["phi_in"]
        out["z_final"] = bR[..., :D].contiguous()                                        # (b,R,D) view-1 copy
    return res, out


def run_case(name: str) -> None:
    c = C.case_config(name)
    torch.manual_seed(0)
    inp, z = C.build_inputs(c)
    shapes = C.param_shapes(c)
    sd = S.seeded_state_dict(shapes, seed=c["w_seed"])

    model = ref_import.build_reference_model(
        n_view=c["n_view"], npoints=c["P"], model=c["model"], H=c["H"], no_sample=c["no_sample"],
        no_latent_concat=c["no_latent_concat"], repeat_attention=c["repeat_attention"])
    ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.startswith("encoder.")}
    assert ref_shapes == shapes, f"parameter table differs from the reference for {name}: " \
        f"{set(ref_shapes.items()) ^ set(shapes.items())}"
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("encoder.") for k in missing.missing_keys), missing

    t0 = time.time()
    if c["tier"] == 0:
        ref_out, ref_st = reference_stages(model, inp, z, c)
    else:
        with torch.no_grad():
            ref_out = model(inp, z=z)
        ref_st = {}
    t_ref = time.time() - t0

    # oracle vs reference, right here where the reference exists
    cfg = O.RenderConfig(n_view=c["n_view"], npoints=c["P"], no_sample=c["no_sample"],
                         no_latent_concat=c["no_latent_concat"], repeat_attention=c["repeat_attention"],
                         H=c["H"], W=c["H"])
    with torch.no_grad():
        ora = O.render_forward(sd, inp, z, cfg, debug=True)
    worst = {}
    for k in C.OUT_KEYS:
        a, b_ = ref_out[k].double(), ora[k].double()
        assert a.shape == b_.shape, (k, a.shape, b_.shape)
        worst[k] = ((a - b_).abs() / b_.abs().clamp_min(1.0)).max().item()
    for k, v in ref_st.items():
        o = ora["stages"][k]
        if k == "z_final":
            b, V = c["b"], c["n_view"]
            o = o.reshape(b, V, *o.shape[1:])[:, 0]
        worst["stage:" + k] = ((v.double() - o.double()).abs() / o.double().abs().clamp_min(1.0)).max().item()
    print(f"{name:24s} ref {t_ref:6.2f}s  oracle-vs-reference max rel err: " +
          "  ".join(f"{k}={v:.1e}" for k, v in worst.items()))
    assert worst["valid_mask"] == 0.0 and worst["at_wt_max"] == 0.0, "discrete outputs differ"
    assert max(worst.values()) < 2e-5, "oracle does not reproduce the reference"

    from cross_attention_renderer_amd.poses import pack_poses
    poses = pack_poses(inp, c["H"])
    with torch.no_grad():   # the pose records must reproduce the reference run they are stored beside
        chk = O.render_forward(sd, inp, z, cfg, poses96=poses)
    assert torch.equal(chk["pixel_val"], ref_out["pixel_val"]) and torch.equal(chk["coords"], ref_out["coords"])
    payload = {
        "poses": poses.numpy(),
        "ctx_cam2world": inp["context"]["cam2world"].numpy(),
        "ctx_intrinsics": inp["context"]["intrinsics"].numpy(),
        "qry_cam2world": inp["query"]["cam2world"].numpy(),
        "qry_intrinsics": inp["query"]["intrinsics"].numpy(),
        "uv": inp["query"]["uv"].numpy(),
        "z_checksum": C.checksum(z),
        "w_checksum": C.checksum([sd[k] for k in sorted(sd)]),
    }
    for k in C.OUT_KEYS:
        v = ref_out[k]
        payload["out_" + k] = v.numpy() if v.dtype != torch.int64 else v.numpy().astype(np.int32)
    for k, v in ref_st.items():
        payload["stage_" + k] = v.numpy()
    np.savez_compressed(C.fixture_path(name), **payload)
    print(f"   wrote {C.fixture_path(name)} ({os.path.getsize(C.fixture_path(name)) / 1024:.0f} KiB)")


if __name__ == "__main__":
    if not ref_import.reference_available():
        sys.exit("the reference tree is not present: golden vectors can only be generated in the build container")
    names = sys.argv[1:] or list(C.CASES)
    for n in names:
        run_case(n)
