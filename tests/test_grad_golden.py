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
    assert "latent_avg_query.weight" in unused and "update_val_merge.weight" in unused       # declared, never on the n_view = 2 forward path
    worst = max(G.compare(fx, k, og[k], tol=1e-4)[0] for k in stored)
    assert worst <= G.FLIP_WORST
