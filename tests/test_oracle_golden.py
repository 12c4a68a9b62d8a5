"""Pin the CPU oracle (oracle/lft_oracle.py) against fixtures captured from the real reference
(tools/gen_golden.py).  Reference citations: model/LFT.py:52-83 (forward), :147-162 (mask),
:86-115 (position encoding), :255-266 (bicubic), :269-277 (loss)."""
import os

import numpy as np
import pytest
import torch

from lft_amd.params import deterministic_state, synthetic_lr
from oracle import lft_oracle as O
from fixture_util import stats, sub_indices

CASES = ["tiny_a5_s2_b2_6x6", "small_a5_s4_b1_8x8", "small_a9_s4_b1_8x8", "rect_a5_s2_b1_8x6", "wide_a2_s2_b1_6x12",
         "cfg1_a5_s2_b1_32x32", "cfg2_a5_s4_b1_32x32"]
TOL = 2e-6   # fp32 re-association between two stock-op formulations of the same maths


def _run(g):
    A, s, B, h, w, wseed, iseed = [int(v) for v in g["meta"]]
    sd = O.state_from_numpy(deterministic_state(64, s, seed=wseed, flavor=str(g["flavor"])))
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=iseed))
    taps = {}
    out = O.forward(sd, lr, A, s, taps)
    return out, taps


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_fixture(name, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    out, taps = _run(g)
    assert out.shape == g["out"].shape
    assert np.abs(out.numpy() - g["out"]).max() <= TOL
    assert O.psnr(out, torch.from_numpy(g["out"])) > 120.0
    for key in [k[4:-4] for k in g.files if k.startswith("tap_") and k.endswith("_sub")]:
        mine = taps[key].contiguous().numpy().ravel()
        ref_sub = g[f"tap_{key}_sub"]
        scale = max(1.0, float(np.abs(ref_sub).max()))
        assert np.abs(mine[sub_indices(mine.size)] - ref_sub).max() <= TOL * scale, key
        st, rs = stats(mine), g[f"tap_{key}_stats"]
        assert st[0] == rs[0]
        assert abs(st[3] - rs[3]) <= 1e-5 * max(1.0, rs[3]), key
        if f"tap_{key}_full" in g.files:
            assert np.abs(taps[key].numpy() - g[f"tap_{key}_full"]).max() <= TOL * scale, key


def test_masks_and_quirk(golden_dir):
    g = np.load(os.path.join(golden_dir, "misc.npz"))
    for (h, w) in [(8, 8), (6, 4), (4, 8)]:
        allowed = torch.isfinite(O.window_mask(h, w)).numpy()
        assert (allowed == g[f"mask_{h}x{w}"]).all()
    # h < w: the reference's column clamp with h leaves some queries without any key
    assert (~g["mask_4x8"]).all(axis=1).sum() == 8


def test_position_encodings(golden_dir):
    g = np.load(os.path.join(golden_dir, "misc.npz"))
    spa = O.spatial_pe(8, 6, 64).numpy()
    assert np.abs(spa - g["pe_spa_8x6"][0, :, 0]).max() <= 1e-6
    ang = O.angular_pe(25, 64).numpy()
    assert np.abs(ang - g["pe_ang_25"][0, :, :, 0, 0].T).max() <= 1e-6


def test_bicubic_and_loss(golden_dir):
    g = np.load(os.path.join(golden_dir, "misc.npz"))
    x = torch.from_numpy(synthetic_lr(1, 3, 7, 5, seed=3))
    for s in (2, 4):
        assert np.abs(O.bicubic_skip(x, 3, s).numpy() - g[f"bicubic_a3_7x5_s{s}"]).max() <= 1e-6
    a, b = torch.from_numpy(g["l1_pair"][0]), torch.from_numpy(g["l1_pair"][1])
    assert abs(float(O.l1_loss(a, b)) - float(g["l1_value"])) <= 1e-7


def test_scene_tiling_matches_reference(golden_dir):
    """LFdivide / LFintegrate restatement (oracle) vs outputs of the reference functions (utils/utils.py:91-157)."""
    g = np.load(os.path.join(golden_dir, "tiling.npz"))
    for name in ("a2_50x41", "a3_32x48", "a2_20x23_p8"):
        A, h0, w0, patch, stride, s, nu, nv = [int(v) for v in g[f"{name}_meta"]]
        rng = np.random.Generator(np.random.PCG64([11, A, h0, w0]))
        scene = torch.from_numpy(rng.random((A * h0, A * w0), dtype=np.float32))
        assert O.lf_divide_counts(h0, w0, patch, stride) == (nu, nv)
        sub = O.lf_divide(scene, A, patch, stride)
        assert np.array_equal(sub.numpy(), g[f"{name}_divide"])
        srp = torch.from_numpy(rng.random((nu, nv, A * patch * s, A * patch * s), dtype=np.float32))
        out = O.lf_integrate(srp, A, patch * s, stride * s, h0 * s, w0 * s)
        assert np.array_equal(out.numpy(), g[f"{name}_integrate"])
