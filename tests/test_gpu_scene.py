"""GPU tests of the scene tiling kernels (C ABI: lft_scene_divide / lft_scene_integrate) and of whole-scene
inference, against the oracle's restatement of the reference's LFdivide / LFintegrate (utils/utils.py:91-157)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from lft_amd import scene as S
from lft_amd.params import deterministic_state
from oracle import lft_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("A,h0,w0,patch,stride,s", [(2, 50, 41, 32, 16, 2), (3, 32, 48, 32, 16, 4), (2, 20, 23, 8, 4, 2), (5, 33, 32, 32, 16, 4)])
def test_divide_and_integrate_bit_exact(A, h0, w0, patch, stride, s):
    rng = np.random.Generator(np.random.PCG64([11, A, h0, w0]))
    scene = torch.from_numpy(rng.random((A * h0, A * w0), dtype=np.float32))
    nu, nv = O.lf_divide_counts(h0, w0, patch, stride)
    assert S.scene_counts(h0, w0, patch, stride) == (nu, nv)
    sub = S.divide(scene.to("cuda:0"), A, patch, stride).cpu()
    assert torch.equal(sub.reshape(nu, nv, A * patch, A * patch), O.lf_divide(scene, A, patch, stride))
    srp = torch.from_numpy(rng.random((nu * nv, 1, A * patch * s, A * patch * s), dtype=np.float32))
    got = S.integrate(srp.to("cuda:0"), A, h0, w0, s, patch, stride).cpu()
    ref = O.views_to_scene_mosaic(O.lf_integrate(srp.reshape(nu, nv, A * patch * s, A * patch * s), A, patch * s, stride * s, h0 * s, w0 * s))
    assert torch.equal(got, ref)


def test_whole_scene_matches_patchwise_oracle():
    """End to end on a small scene: divide -> batched HIP forward -> integrate == oracle run patch by patch."""
    from model import LFT
    A, s, patch, stride, h0, w0 = 3, 2, 8, 4, 14, 11
    sd = deterministic_state(64, s, seed=1, flavor="stress")
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s), precision="fp32")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to("cuda:0").eval()
    rng = np.random.Generator(np.random.PCG64([12, A, h0, w0]))
    scene = torch.from_numpy(rng.random((A * h0, A * w0), dtype=np.float32))
    got = S.super_resolve_scene(net, scene.to("cuda:0"), patch, stride, max_batch=5).cpu()
    sub = O.lf_divide(scene, A, patch, stride)
    nu, nv = sub.shape[:2]
    osd = O.state_from_numpy(sd)
    outs = torch.stack([torch.stack([O.forward(osd, sub[u, v][None, None], A, s)[0, 0] for v in range(nv)]) for u in range(nu)])
    ref = O.views_to_scene_mosaic(O.lf_integrate(outs, A, patch * s, stride * s, h0 * s, w0 * s))
    assert got.shape == ref.shape == (A * h0 * s, A * w0 * s)
    assert float((got - ref).abs().max() / ref.abs().max()) <= 1e-4


def test_evaluate_scene_matches_oracle_pipeline():
    """lft_amd.evaluate.test_scene (divide -> batched network -> integrate -> per-view PSNR / SSIM, all on the GPU) against
    the same pipeline built from the CPU oracles (tiling, network, metrics), reference test.py:75-104."""
    from types import SimpleNamespace
    from lft_amd import evaluate
    from lft_amd.params import deterministic_state
    from oracle import metrics_oracle as M
    from model import LFT
    A, s, h0, w0 = 2, 2, 40, 36
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s), precision="fp32")
    sd = deterministic_state(64, s, seed=1, flavor="default")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to("cuda:0")
    g = np.random.default_rng(3)
    lr_scene = torch.from_numpy(g.random((A * h0, A * w0), dtype=np.float32))
    hr_scene = torch.from_numpy(g.random((A * h0 * s, A * w0 * s), dtype=np.float32))
    psnr, ssim, sr = evaluate.test_scene(net, lr_scene, hr_scene, patch=16, stride=8)
    # oracle pipeline
    osd = O.state_from_numpy(sd)
    sub = O.lf_divide(lr_scene, A, 16, 8)
    nu, nv = sub.shape[:2]
    outs = torch.stack([torch.stack([O.forward(osd, sub[u, v][None, None], A, s)[0, 0] for v in range(nv)]) for u in range(nu)])
    ref = O.views_to_scene_mosaic(O.lf_integrate(outs, A, 16 * s, 8 * s, h0 * s, w0 * s))
    assert float((sr.cpu() - ref).abs().max()) <= 1e-5
    _, _, pm, sm = M.cal_metrics(hr_scene[None, None].numpy(), ref[None, None].numpy(), A)
    assert abs(psnr - pm) <= 1e-3 and abs(ssim - sm) <= 1e-5
    p2, s2 = evaluate.test(net, [(lr_scene, hr_scene), (lr_scene, hr_scene)], patch=16, stride=8)
    assert abs(p2 - psnr) <= 1e-6 and abs(s2 - ssim) <= 1e-6
