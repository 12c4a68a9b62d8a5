"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the per-view PSNR / SSIM the reference reports (utils/utils.py:56-88,
``cal_metrics``).  Never imported by the product path.

The arithmetic lives in scikit-image (``skimage.metrics.peak_signal_noise_ratio`` / ``structural_similarity(...,
gaussian_weights=True)``), which is NOT vendored in /root/reference, has no pinned version there (README.md lists only
PyTorch / torchvision) and is not importable by the interpreter this framework runs on.  The image does carry an Anaconda
Python 3.9 under /opt/conda with scikit-image 0.18.3 (a release contemporary with the reference's PyTorch 1.3-1.8):
**pinned** by tests/golden/metrics_skimage.npz -- the reference's own two calls per view (utils.py:79-83) made with that
scikit-image on seeded mosaics (tools/gen_golden_skimage.py; views of 11x11 .. 64x64, non-square, 81 views, a label with
negative values, MSE = 0 views) -- which this file reproduces to the bit (tests/test_metrics_oracle.py).  It restates the
published algorithm (Wang et al. 2004 as implemented by scikit-image 0.16-0.18) on top of scipy.ndimage.gaussian_filter, the
very function scikit-image calls:
  * PSNR: 10 log10(R^2 / MSE), MSE in float64, R = 1 for float images whose minimum is >= 0 (else 2);
  * SSIM: sigma 1.5, truncate 3.5 -> 11x11 Gaussian window, sample covariance (N/(N-1), N = 121), K1 0.01, K2 0.03,
    data_range R = 2 for float images in those releases (dtype range [-1, 1]; newer releases require an explicit
    data_range -- pass ``ssim_range``), mean of the SSIM map cropped by 5 pixels on every side, float64;
  * cal_metrics: per view of the [B,1,(a1 h),(a2 w)] mosaics; means over the views whose value is > 0.
"""
import numpy as np
from scipy.ndimage import gaussian_filter


def psnr_view(true: np.ndarray, test: np.ndarray) -> float:
    t, x = true.astype(np.float64), test.astype(np.float64)
    rng = 1.0 if t.min() >= 0 else 2.0
    mse = np.mean((t - x) ** 2)
    return float(10.0 * np.log10(rng * rng / mse))


def ssim_view(a: np.ndarray, b: np.ndarray, ssim_range: float = 2.0) -> float:
    x, y = a.astype(np.float64), b.astype(np.float64)
    f = lambda z: gaussian_filter(z, sigma=1.5, truncate=3.5, mode="reflect")    # noqa: E731
    NP = 11 * 11
    cov_norm = NP / (NP - 1.0)
    ux, uy = f(x), f(y)
    vx = cov_norm * (f(x * x) - ux * ux)
    vy = cov_norm * (f(y * y) - uy * uy)
    vxy = cov_norm * (f(x * y) - ux * uy)
    C1, C2 = (0.01 * ssim_range) ** 2, (0.03 * ssim_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    return float(S[5:-5, 5:-5].mean())


def cal_metrics(label: np.ndarray, out: np.ndarray, A: int, ssim_range: float = 2.0):
    """label / out: [B,1,A*h,A*w].  Returns (PSNR [B,A,A], SSIM [B,A,A], PSNR_mean, SSIM_mean) as utils.py:56-88."""
    B, _, H, W = label.shape
    h, w = H // A, W // A
    P = np.zeros((B, A, A), dtype=np.float32)
    S = np.zeros((B, A, A), dtype=np.float32)
    for b in range(B):
        for u in range(A):
            for v in range(A):
                t = label[b, 0, u * h:(u + 1) * h, v * w:(v + 1) * w]
                x = out[b, 0, u * h:(u + 1) * h, v * w:(v + 1) * w]
                P[b, u, v] = psnr_view(t, x)
                S[b, u, v] = ssim_view(t, x, ssim_range)
    return P, S, float(P.sum() / np.sum(P > 0)), float(S.sum() / np.sum(S > 0))
