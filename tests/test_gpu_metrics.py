"""GPU: per-view PSNR / SSIM kernels (C ABI lft_view_metrics) against scikit-image's own results (tests/golden/metrics_skimage.npz:
the reference's two calls per view, utils/utils.py:79-83, made with scikit-image 0.18.3) and against the restatement that fixture
pins (oracle/metrics_oracle.py)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from lft_amd import metrics
from oracle import metrics_oracle as M

pytestmark = pytest.mark.gpu
FIX = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics_skimage.npz"))


@pytest.mark.parametrize("name", [str(n) for n in FIX["names"]])
def test_view_metrics_match_scikit_image_fixture(name):
    A = int(FIX[name + "/A"])
    label, out = torch.from_numpy(FIX[name + "/label"]).cuda(), torch.from_numpy(FIX[name + "/out"]).cuda()
    p, s = metrics.view_metrics(label, out, A)
    p, s = p.cpu().numpy().reshape(FIX[name + "/psnr"].shape), s.cpu().numpy().reshape(FIX[name + "/ssim"].shape)
    eP, eS = FIX[name + "/psnr"], FIX[name + "/ssim"]
    fin = np.isfinite(eP)
    assert np.array_equal(np.isinf(p), np.isinf(eP)) and np.array_equal(np.isnan(p), np.isnan(eP)), (p, eP)   # MSE 0 (exactly reproduced view, all-zero view): inf
    assert np.abs(p[fin] - eP[fin]).max() <= 1e-4                      # dB
    assert np.abs(s - eS).max() <= 2e-6
    pm, sm = metrics.cal_metrics(SimpleNamespace(angRes=A), label, out)
    epm, esm = float(FIX[name + "/psnr_mean"]), float(FIX[name + "/ssim_mean"])
    assert (np.isinf(epm) and np.isinf(pm)) or abs(pm - epm) <= 1e-4
    assert abs(sm - esm) <= 2e-6


@pytest.mark.parametrize("B,A,h,w,rng", [(2, 3, 40, 33, 2.0), (1, 2, 64, 64, 1.0), (1, 5, 11, 17, 2.0)])
def test_view_metrics_match_oracle(B, A, h, w, rng):
    g = np.random.default_rng(B * 100 + A)
    yy, xx = np.meshgrid(np.arange(A * h), np.arange(A * w), indexing="ij")
    label = (0.5 + 0.3 * np.sin(yy / 5.0) * np.cos(xx / 7.0))[None, None].repeat(B, 0).astype(np.float32)
    label += 0.1 * g.random(label.shape).astype(np.float32)
    out = np.clip(label + 0.03 * g.standard_normal(label.shape).astype(np.float32), 0, 1).astype(np.float32)
    P, S, pm, sm = M.cal_metrics(label, out, A, ssim_range=rng)
    p, s = metrics.view_metrics(torch.from_numpy(label).cuda(), torch.from_numpy(out).cuda(), A, ssim_range=rng)
    assert np.abs(p.cpu().numpy() - P).max() <= 1e-4, (p.cpu().numpy(), P)            # dB
    assert np.abs(s.cpu().numpy() - S).max() <= 1e-6
    pm2, sm2 = metrics.cal_metrics(SimpleNamespace(angRes=A), torch.from_numpy(label).cuda(), torch.from_numpy(out).cuda(), ssim_range=rng)
    assert abs(pm2 - pm) <= 1e-4 and abs(sm2 - sm) <= 1e-6


def test_negative_label_switches_psnr_range_and_2d_input():
    g = np.random.default_rng(3)
    label = (g.random((24, 24)) - 0.2).astype(np.float32)
    out = (label + 0.01).astype(np.float32)
    P, S, _, _ = M.cal_metrics(label[None, None], out[None, None], 2)
    p, s = metrics.view_metrics(torch.from_numpy(label).cuda(), torch.from_numpy(out).cuda(), 2)
    assert np.abs(p.cpu().numpy() - P).max() <= 1e-4 and np.abs(s.cpu().numpy() - S).max() <= 1e-6
    assert float(p.min()) > 40.0                                                          # 10 log10(4 / 1e-4) = 46 dB: R = 2
