"""Per-view PSNR / SSIM of SR mosaics on the GPU, with the reference's aggregation (utils/utils.py:56-88 cal_metrics):
means over the views whose metric is > 0."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def view_metrics(label: torch.Tensor, out: torch.Tensor, angRes: int, ssim_range: float = 2.0):
    """label, out: [B,1,A*h,A*w] (or [A*h,A*w]) float32 on a HIP device -> (psnr [B,A,A], ssim [B,A,A])."""
    if label.dim() == 2:
        label, out = label[None, None], out[None, None]
    if not label.is_cuda:
        raise _lib.LftError("lft_amd.metrics runs on a HIP device only")
    label, out = label.contiguous().float(), out.contiguous().float()
    B, _, H, W = label.shape
    A = int(angRes)
    h, w = H // A, W // A
    n = ctypes.c_size_t(0)
    _lib.check(_lib.lib().lft_view_metrics_scratch_bytes(B, A, h, w, ctypes.byref(n)), "lft_view_metrics_scratch_bytes")
    with torch.cuda.device(label.device):
        scratch = torch.empty(n.value, dtype=torch.uint8, device=label.device)
        psnr = torch.empty(B * A * A, dtype=torch.float32, device=label.device)
        ssim = torch.empty_like(psnr)
        _lib.check(_lib.lib().lft_view_metrics(label.data_ptr(), out.data_ptr(), B, A, h, w, float(ssim_range), psnr.data_ptr(), ssim.data_ptr(),
                                               scratch.data_ptr(), torch.cuda.current_stream(label.device).cuda_stream), "lft_view_metrics")
    return psnr.view(B, A, A), ssim.view(B, A, A)


def cal_metrics(args, label: torch.Tensor, out: torch.Tensor, ssim_range: float = 2.0):
    """Drop-in for the reference's cal_metrics(args, label, out) -> (PSNR_mean, SSIM_mean)."""
    p, s = view_metrics(label, out, args.angRes, ssim_range)
    return float(p.sum() / (p > 0).sum()), float(s.sum() / (s > 0).sum())
