"""Whole-scene inference around the hot path: the reference's test loop (test.py:83-101) cuts a scene into
overlapping 32x32-LR patches with LFdivide (utils/utils.py:91-123), runs the network ONE PATCH AT A TIME, and
re-assembles with LFintegrate (utils.py:141-157).  Here the cut and the re-assembly are GPU gathers
(lft_scene_divide / lft_scene_integrate) and all patches of the scene go through the network as batches."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def scene_counts(h0: int, w0: int, patch: int = 32, stride: int = 16):
    nu, nv = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.lib().lft_scene_counts(h0, w0, patch, stride, ctypes.byref(nu), ctypes.byref(nv)), "lft_scene_counts")
    return nu.value, nv.value


def divide(scene: torch.Tensor, ang: int, patch: int = 32, stride: int = 16) -> torch.Tensor:
    """scene: float32 [A*h0, A*w0] on a HIP device -> [numU*numV, 1, A*patch, A*patch]."""
    if not scene.is_cuda or scene.dim() != 2:
        raise _lib.LftError("scene must be a 2-D float32 tensor on a HIP device")
    H, W = scene.shape
    h0, w0 = H // ang, W // ang
    nu, nv = scene_counts(h0, w0, patch, stride)
    x = scene.contiguous().float()
    out = torch.empty((nu * nv, 1, ang * patch, ang * patch), dtype=torch.float32, device=scene.device)
    with torch.cuda.device(scene.device):
        _lib.check(_lib.lib().lft_scene_divide(x.data_ptr(), out.data_ptr(), ang, h0, w0, patch, stride,
                                               torch.cuda.current_stream(scene.device).cuda_stream), "lft_scene_divide")
    return out


def integrate(sr_patches: torch.Tensor, ang: int, h0: int, w0: int, scale: int, patch: int = 32, stride: int = 16) -> torch.Tensor:
    """sr_patches: float32 [numU*numV, 1, A*patch*s, A*patch*s] -> SR scene mosaic [A*h0*s, A*w0*s]."""
    x = sr_patches.contiguous().float()
    out = torch.empty((ang * h0 * scale, ang * w0 * scale), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().lft_scene_integrate(x.data_ptr(), out.data_ptr(), ang, h0, w0, patch, stride, scale,
                                                  torch.cuda.current_stream(x.device).cuda_stream), "lft_scene_integrate")
    return out


@torch.no_grad()
def super_resolve_scene(net, scene: torch.Tensor, patch: int = 32, stride: int = 16, max_batch: int = 64) -> torch.Tensor:
    """Equivalent of the body of the reference's test() for one scene (test.py:79-101), batched."""
    A, s = net.angRes, net.factor
    h0, w0 = scene.shape[0] // A, scene.shape[1] // A
    patches = divide(scene, A, patch, stride)
    outs = [net(patches[i:i + max_batch]) for i in range(0, patches.shape[0], max_batch)]
    return integrate(torch.cat(outs, dim=0), A, h0, w0, s, patch, stride)
