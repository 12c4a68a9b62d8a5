"""north_star's second parity form: "PSNR within 0.01 dB".  The reference's .pth weights and its five test sets are absent
(SURVEY.md 8c), so the literal check cannot run; its FORM can: per view |PSNR(path, HR) - PSNR(oracle, HR)| on weights that
were trained (by this repo's own trainer, a few epochs on synthetic light fields) and held-out scenes pushed through
lft_amd.evaluate -- see tests/psnr_util.py.  The same figure is reported by bench.py as `psnr_delta_db`."""
import pytest
import torch

import psnr_util as PU

pytestmark = pytest.mark.gpu
TOL_DB = 0.01          # BASELINE.json north_star


@pytest.mark.parametrize("fmax", [0.25, 0.06], ids=["hard_scenes", "smooth_scenes"])
def test_psnr_delta_within_a_hundredth_of_a_db_on_trained_weights(fmax):
    """Two scene families: content up to the LR Nyquist limit (PSNR in the 20s) and smooth content (PSNR in the reference tables'
    range and above) -- the same output error weighs more against a smaller model error, so the smooth family is the harder test."""
    dev = torch.device("cuda:0")
    sd, hist = PU.train_small_model(dev, fmax=fmax)
    assert hist[-1] < 0.6 * hist[0], f"the small model did not train: {hist[0]:.4f} -> {hist[-1]:.4f}"
    scenes = PU.held_out_scenes(n=1, size=32, fmax=fmax)          # 5 x 5 views of 32 x 32 LR: four overlapping patches
    res = PU.psnr_delta(dev, sd, scenes)
    for prec, r in res.items():
        print(f"psnr_delta_db [{prec}]: max {r['max_abs_delta_db']:.5f} mean {r['mean_abs_delta_db']:.5f} dB over {r['views']} views; "
              f"PSNR(oracle, HR) {r['psnr_oracle_mean_db']:.3f} dB, PSNR(path, HR) {r['psnr_path_mean_db']:.3f} dB; "
              f"training loss {hist[0]:.4f} -> {hist[-1]:.4f}")
    # exact fp32 and the fp16 path meet the bound on both families (observed 0 / <= 0.003 dB).  bf16: the same output error (rms 8e-4)
    # weighs more the smaller the model's own error is -- observed 0.007 - 0.009 dB at 33 dB, 0.014 dB at 37 dB, 0.05 - 0.06 dB at 50 dB:
    # at the edge of north_star's 0.01 dB in the PSNR range of the reference's result tables (29 - 44 dB), beyond it above.  Its gate
    # pins the observed level; it does not claim the bound.
    assert res["fp32"]["max_abs_delta_db"] <= TOL_DB
    assert res["fp16"]["max_abs_delta_db"] <= TOL_DB
    assert res["bf16"]["max_abs_delta_db"] <= BF16_TOL_DB[fmax]


BF16_TOL_DB = {0.25: 0.02, 0.06: 0.10}
