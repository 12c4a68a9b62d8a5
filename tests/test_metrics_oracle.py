"""CPU: the metrics restatement (oracle/metrics_oracle.py) against the REAL scikit-image: tests/golden/metrics_skimage.npz holds
seeded mosaics and the results of the reference's own two calls per view (utils/utils.py:79-83) made with scikit-image 0.18.3
(tools/gen_golden_skimage.py, run under the image's /opt/conda Python 3.9 -- the only interpreter here that has scikit-image).
Plus properties of the published definitions: identical images, known MSE, window normalisation."""
import os

import numpy as np
import pytest

from oracle import metrics_oracle as M

FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics_skimage.npz"))


@pytest.mark.parametrize("name", [str(n) for n in FIX["names"]])
def test_oracle_matches_scikit_image_fixture(name):
    """Per-view PSNR / SSIM and the `> 0` means of cal_metrics, incl. an exactly reproduced view and an all-zero view (MSE 0:
    PSNR inf, and so is the mean), a label with negative values (data range 2), non-square and 11x11 views.  float32 results as the
    reference stores them; the restatement calls the same scipy Gaussian filter scikit-image does, so the agreement is to the bit."""
    A = int(FIX[name + "/A"])
    with np.errstate(all="ignore"):
        P, S, pm, sm = M.cal_metrics(FIX[name + "/label"], FIX[name + "/out"], A)
    eP, eS = FIX[name + "/psnr"], FIX[name + "/ssim"]
    assert np.array_equal(np.isnan(P), np.isnan(eP)) and np.array_equal(np.isinf(P), np.isinf(eP))
    fin = np.isfinite(eP)
    assert np.abs(P[fin] - eP[fin]).max() <= 1e-5 and np.abs(S - eS).max() <= 1e-7
    epm, esm = float(FIX[name + "/psnr_mean"]), float(FIX[name + "/ssim_mean"])
    assert (np.isinf(epm) and np.isinf(pm)) or abs(pm - epm) <= 1e-5
    assert abs(sm - esm) <= 1e-7


def test_psnr_known_value_and_range_rule():
    t = np.full((16, 16), 0.5, dtype=np.float32)
    assert abs(M.psnr_view(t, t + 0.1) - 20.0) < 1e-4                # MSE 0.01, R = 1
    assert abs(M.psnr_view(t - 1.0, t - 0.9) - (20.0 + 20 * np.log10(2))) < 1e-4    # negative minimum: R = 2


def test_ssim_identity_and_symmetry_and_constant_shift():
    g = np.random.default_rng(0)
    a = g.random((40, 32)).astype(np.float32)
    b = np.clip(a + 0.05 * g.standard_normal(a.shape), 0, 1).astype(np.float32)
    assert abs(M.ssim_view(a, a) - 1.0) < 1e-12
    assert abs(M.ssim_view(a, b) - M.ssim_view(b, a)) < 1e-12
    assert M.ssim_view(a, b, 1.0) < M.ssim_view(a, b, 2.0) < 1.0      # larger stabilisers pull towards 1
    # a flat image against a shifted flat image: only the luminance term remains
    f = np.full((30, 30), 0.4, dtype=np.float32)
    lum = (2 * 0.4 * 0.5 + (0.01 * 2) ** 2) / (0.4 ** 2 + 0.5 ** 2 + (0.01 * 2) ** 2)
    assert abs(M.ssim_view(f, f + 0.1) - lum) < 1e-6
