"""Helpers shared by the golden-vector tests (data plumbing only)."""
from __future__ import annotations

import numpy as np
import torch

import cases as C
from cross_attention_renderer_amd import synthetic as S


def load_case(name: str):
    """Returns (config, input dict, z, state_dict, fixture) for a golden case, after verifying that the
    regenerated bulky inputs (feature maps, weights) are the ones the fixture was made with."""
    c = C.case_config(name)
    fx = np.load(C.fixture_path(name))
    inp, z = C.build_inputs(c)
    sd = S.seeded_state_dict(C.param_shapes(c), seed=c["w_seed"])
    np.testing.assert_allclose(C.checksum(z), fx["z_checksum"], rtol=1e-12, err_msg="feature-map RNG drift")
    np.testing.assert_allclose(C.checksum([sd[k] for k in sorted(sd)]), fx["w_checksum"], rtol=1e-12,
                               err_msg="weight RNG drift")
    for k_in, k_fx in (("cam2world", "ctx_cam2world"), ("intrinsics", "ctx_intrinsics")):
        np.testing.assert_array_equal(inp["context"][k_in].numpy(), fx[k_fx])
    np.testing.assert_array_equal(inp["query"]["cam2world"].numpy(), fx["qry_cam2world"])
    np.testing.assert_array_equal(inp["query"]["uv"].numpy(), fx["uv"])
    return c, inp, z, sd, fx


def rel_err(a, b) -> float:
    """max |a-b| / max(1,|b|): the parity metric of SURVEY.md §8c."""
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return ((a - b).abs() / b.abs().clamp_min(1.0)).max().item()
