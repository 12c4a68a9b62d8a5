"""CPU: the oracle's training step (autograd over the restatement + Adam written out) against fixtures produced by
the REAL reference network and torch.optim.Adam (tools/gen_golden.py:train_case)."""
import os

import numpy as np
import pytest
import torch

from lft_amd.params import deterministic_state, synthetic_lr
from oracle import lft_oracle as O
from fixture_util import stats, sub_indices

CASES = ["train_a3_s2_b2_6x6", "train_a2_s4_b1_8x5"]


def load_case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    A, s, B, h, w, wseed, iseed, tseed, steps = [int(v) for v in g["meta"]]
    sd = O.state_from_numpy(deterministic_state(64, s, seed=wseed, flavor=str(g["flavor"])))
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=iseed))
    hr = torch.from_numpy(g["hr"])
    return g, sd, lr, hr, A, s, steps


@pytest.mark.parametrize("name", CASES)
def test_train_step_matches_reference_fixture(name, golden_dir):
    g, sd, lr, hr, A, s, steps = load_case(golden_dir, name)
    losses, grads, post = O.train_steps(sd, lr, hr, A, s, steps)
    assert np.allclose(losses, g["losses"], rtol=0, atol=2e-6), (losses, g["losses"])
    for k, gr in grads.items():
        mine = gr.contiguous().numpy().ravel()
        ref = g[f"grad_{k}_sub"]
        scale = max(float(np.abs(ref).max()), 1e-6)
        assert np.abs(mine[sub_indices(mine.size)] - ref).max() <= 2e-4 * scale + 1e-9, k      # fp32 re-association in backward
        if f"grad_{k}_full" in g.files:
            assert np.abs(mine - g[f"grad_{k}_full"]).max() <= 2e-4 * max(float(np.abs(g[f"grad_{k}_full"]).max()), 1e-6) + 1e-9, k
        assert stats(mine)[0] == g[f"grad_{k}_stats"][0]
    for k, v in post.items():
        mine = v.contiguous().numpy().ravel()
        # Adam moves every weight by ~lr per step whatever the gradient's size: compare the displacement, loosely where
        # |g| is tiny (sign-like sensitivity), tightly in aggregate
        ref = g[f"post_{k}_sub"]
        assert np.abs(mine[sub_indices(mine.size)] - ref).max() <= 1.05 * steps * 2e-4, k
        assert np.mean(np.abs(mine[sub_indices(mine.size)] - ref)) <= 2e-5, k
