#!/opt/conda/bin/python3.9
"""Generate tests/golden/metrics_skimage.npz with the REAL scikit-image (build container only).

The reference's metrics are two scikit-image calls per view (utils/utils.py:79-83):
    metrics.peak_signal_noise_ratio(label_view, out_view)
    metrics.structural_similarity(label_view, out_view, gaussian_weights=True)
on float32 arrays, then means over the views whose value is > 0 (utils.py:85-86).  scikit-image is not importable by the
interpreter the framework runs on (/usr/bin/python3, 3.10), but the image carries an Anaconda Python 3.9 under /opt/conda
with scikit-image 0.18.3 -- a release contemporary with the reference.  This script runs under THAT interpreter (numpy +
skimage only, no torch), makes the very calls above on seeded mosaics and writes inputs and expected outputs: data only.

    /opt/conda/bin/python3.9 tools/gen_golden_skimage.py        (writes tests/golden/metrics_skimage.npz)

The fixture pins oracle/metrics_oracle.py (tests/test_oracle_metrics.py, CPU) and, through it and directly, the HIP metrics
kernels (tests/test_gpu_metrics.py)."""
import os
import warnings

import numpy as np
import skimage
from skimage import metrics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scene(rng, A, h, w, noise, lo=0.0):
    """A smooth random mosaic in [lo, 1] (label) and a noisy copy (out), [1,1,A*h,A*w] float32 -- what cal_metrics sees."""
    yy, xx = np.meshgrid(np.linspace(0, 1, A * h), np.linspace(0, 1, A * w), indexing="ij")
    img = np.zeros((A * h, A * w))
    for _ in range(6):
        fy, fx, ph = rng.uniform(1, 9), rng.uniform(1, 9), rng.uniform(0, 6.28)
        img += rng.uniform(0.2, 1.0) * np.sin(6.28 * (fy * yy + fx * xx) + ph)
    img = (img - img.min()) / (img.max() - img.min())
    img = lo + (1 - lo) * img
    label = img.astype(np.float32)
    out = (img + noise * rng.standard_normal(img.shape)).astype(np.float32)
    return label[None, None], out[None, None]


def reference_loop(label, out, A):
    """utils.py:74-86 on numpy arrays: the same two calls per view, float32 result arrays, means over positive entries."""
    B, _, H, W = label.shape
    h, w = H // A, W // A
    P = np.zeros((B, A, A), dtype="float32")
    S = np.zeros((B, A, A), dtype="float32")
    for b in range(B):
        for u in range(A):
            for v in range(A):
                t = label[b, 0, u * h:(u + 1) * h, v * w:(v + 1) * w]
                x = out[b, 0, u * h:(u + 1) * h, v * w:(v + 1) * w]
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    P[b, u, v] = metrics.peak_signal_noise_ratio(t, x)
                    S[b, u, v] = metrics.structural_similarity(t, x, gaussian_weights=True)
    with np.errstate(invalid="ignore"):
        return P, S, np.float64(P.sum() / np.sum(P > 0)), np.float64(S.sum() / np.sum(S > 0))


def main():
    rng = np.random.default_rng(20240607)
    cases = {}
    # name: (A, h, w, noise, lowest label value)
    spec = {
        "a5_32x32_n02": (5, 32, 32, 0.02, 0.0),          # the SR regime: ~34 dB
        "a3_64x64_n005": (3, 64, 64, 0.005, 0.0),        # 2x of a 32x32 patch, ~46 dB
        "a2_24x40_n05": (2, 24, 40, 0.05, 0.0),          # non-square views
        "a3_11x11_n03": (3, 11, 11, 0.03, 0.0),          # the smallest view the 11x11 window accepts (one SSIM sample)
        "a2_48x48_neg": (2, 48, 48, 0.02, -0.25),        # label with negative values: PSNR data range 2 instead of 1
        "a9_16x16_n01": (9, 16, 16, 0.01, 0.0),          # 81 views
    }
    for name, (A, h, w, noise, lo) in spec.items():
        label, out = scene(rng, A, h, w, noise, lo)
        if name == "a5_32x32_n02":
            # one view reproduced exactly and one view all zero in both images: MSE 0 -> PSNR = inf in the reference (counted by
            # `PSNR > 0`, so the mean is inf), SSIM 1.0 -- the corner cases of the `> 0` means
            out[0, 0, 0:32, 32:64] = label[0, 0, 0:32, 32:64]
            label[0, 0, 32:64, 0:32] = 0.0
            out[0, 0, 32:64, 0:32] = 0.0
        P, S, pm, sm = reference_loop(label, out, A)
        cases[name + "/label"] = label
        cases[name + "/out"] = out
        cases[name + "/A"] = np.int32(A)
        cases[name + "/psnr"] = P
        cases[name + "/ssim"] = S
        cases[name + "/psnr_mean"] = pm
        cases[name + "/ssim_mean"] = sm
        print(f"{name:16s} PSNR {np.nanmin(P):7.3f} .. {np.nanmax(P):7.3f} mean {pm:8.4f}   SSIM {S.min():.5f} .. {S.max():.5f} mean {sm:.6f}")
    cases["skimage_version"] = np.array(skimage.__version__)
    cases["names"] = np.array(list(spec))
    path = os.path.join(ROOT, "tests", "golden", "metrics_skimage.npz")
    np.savez_compressed(path, **cases)
    print("wrote", path, os.path.getsize(path), "bytes; scikit-image", skimage.__version__)


if __name__ == "__main__":
    main()
