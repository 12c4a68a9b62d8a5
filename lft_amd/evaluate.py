"""Test-set evaluation around the hot path: the loop of the reference's test.py:72-110 with every step on the GPU --
LFdivide (lft_scene_divide), ALL numU x numV patches of a scene through the network as one batch instead of the
reference's batch-1 double loop, LFintegrate (lft_scene_integrate), per-view PSNR / SSIM (lft_view_metrics) with the
reference's aggregation (mean over views with a positive value per scene, then the mean over scenes)."""
from __future__ import annotations

from typing import Iterable, Tuple

import numpy as np
import torch

from . import dp, metrics, scene


def test_scene(net, lr_scene: torch.Tensor, hr_scene: torch.Tensor, patch: int = 32, stride: int = 16,
               ssim_range: float = 2.0) -> Tuple[float, float, torch.Tensor]:
    """One scene: lr_scene [A*h0, A*w0], hr_scene [A*h0*s, A*w0*s] -> (PSNR, SSIM, SR mosaic) as test.py:75-104."""
    dev = next(net.parameters()).device
    with torch.no_grad():
        net.eval()
        sr = scene.super_resolve_scene(net, lr_scene.to(dev).float().contiguous(), patch=patch, stride=stride)
    psnr, ssim = metrics.cal_metrics(net, hr_scene.to(dev).float().contiguous(), sr, ssim_range=ssim_range)
    return psnr, ssim, sr


def test(net, scenes: Iterable[Tuple[torch.Tensor, torch.Tensor]], patch: int = 32, stride: int = 16,
         ssim_range: float = 2.0) -> Tuple[float, float]:
    """Mean PSNR / SSIM over the scenes of one test set (reference test.py:72-110).  Under torch.distributed every rank
    takes a contiguous share of the scenes (lft_amd.dp.shard_range) and the sums are combined."""
    import torch.distributed as dist
    scenes = list(scenes)
    rank, _, world = dp.env_world()
    if not (dist.is_available() and dist.is_initialized()):
        rank, world = 0, 1
    b, e = dp.shard_range(len(scenes), rank, world)
    acc = np.zeros(3, dtype=np.float64)
    for lr_scene, hr_scene in scenes[b:e]:
        p, s, _ = test_scene(net, lr_scene, hr_scene, patch, stride, ssim_range)
        acc += (p, s, 1.0)
    if world > 1:
        t = torch.from_numpy(acc)
        if dist.get_backend() != "gloo":
            t = t.to(next(net.parameters()).device)
        dist.all_reduce(t)
        acc = t.cpu().numpy()
    return float(acc[0] / acc[2]), float(acc[1] / acc[2])


def test_sets(net, args, log=print, patch: int = None, stride: int = None, ssim_range: float = 2.0):
    """The reference's test.py main loop (:60-69) over the test tree of utils_datasets.MultiTestSetDataLoader:
    ``<path_for_test>/SR_{A}x{A}_{s}x/<dataset>/<scene>.h5`` read by lft_amd.datasets (h5lite), every scene through `test_scene`.
    args: path_for_test, angRes, scale_factor (+ patch_size_for_test / stride_for_test as option.py names them).
    Returns {dataset: (psnr, ssim)} in the loader's order and logs the reference's line per dataset."""
    from . import datasets
    patch = patch or getattr(args, "patch_size_for_test", 32)
    stride = stride or getattr(args, "stride_for_test", 16)
    names, loaders, _ = datasets.MultiTestSetDataLoader(args)
    out = {}
    for name, loader in zip(names, loaders):
        scenes = ((lr.squeeze(), hr.squeeze()) for lr, hr in loader)                       # test.py:76-77
        p, s = test(net, scenes, patch, stride, ssim_range)
        out[name] = (p, s)
        log("Test on %s, psnr/ssim is %.2f/%.3f" % (name, p, s))                           # test.py:66
    return out
