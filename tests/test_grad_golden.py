"""SURVEY.md §8 f4, the pin: the CPU oracle's autograd gradients against the REFERENCE's autograd gradients
(tests/golden/grad_*.npz, written by tests/golden/make_grad_golden.py where the reference exists).  The scalar is
L = sum(rgb * c_rgb) + sum(depth_ray * c_depth) with seeded cotangents: rgb and depth_ray are what the reference's loss reads
(loss_functions.py:74-132).  Gradients with respect to every renderer parameter that receives one and to every pyramid level."""
import numpy as np
import pytest

import grad_cases as G
from golden_util import load_case
from hip_harness import oracle_cfg


@pytest.mark.parametrize("name", G.GRAD_CASES)
def test_oracle_autograd_reproduces_the_reference_gradients(name):
    c, inp, z, sd, _ = load_case(name)
    fx = np.load(G.grad_fixture_path(name))
    og = G.oracle_gradients(sd, inp, z, oracle_cfg(c))
    stored = G.keys(fx)
    assert sorted(og) == stored, (sorted(set(og) ^ set(stored)))
    unused = set(fx["unused"].tolist())
    assert "latent_avg_query.weight" in unused               # declared by the reference, never on any forward path
    if c["n_view"] == 2 and not c["no_latent_concat"]:
        assert "update_val_merge.weight" in unused           # only the single-view forward uses it (models.py:485)
    if c["no_latent_concat"]:
        assert "feature_map.weight" in unused                # declared for no_latent_concat, never applied
    worst = max(G.compare(fx, k, og[k], tol=1e-4)[0] for k in stored)
    assert worst <= G.FLIP_WORST
