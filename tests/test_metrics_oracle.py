"""CPU: sanity of the metrics restatement (oracle/metrics_oracle.py).  scikit-image is absent here, so the restatement
is pinned only by properties of the published definitions: identical images, known MSE, window normalisation."""
import numpy as np

from oracle import metrics_oracle as M


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
