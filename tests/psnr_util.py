"""Parity in north_star's second form -- "PSNR within 0.01 dB" -- without the reference's shipped weights or test sets (both
absent): a 5x5 2x network is trained for a few epochs with the repo's own trainer (so the weights are not random
draws), held-out synthetic scenes go through lft_amd.evaluate (LFdivide -> network -> LFintegrate) in every precision and
through the CPU oracle (oracle/lft_oracle.py, the checker), and per view

    delta = | PSNR(path, HR) - PSNR(oracle, HR) |            PSNR = 10 log10(1 / MSE) on [0, 1] data, float64, CPU

is the figure north_star bounds by 0.01 dB.  Used by tests/test_gpu_psnr.py and by bench.py's checker leg (`psnr_delta_db`)."""
from types import SimpleNamespace

import numpy as np
import torch

from oracle import lft_oracle as O          # checker only


def train_small_model(dev, A=5, s=2, n_patches=64, epochs=20, batch=8, rate=1e-3, seed=3, fmax=0.25):
    """Returns (state dict as numpy, per-epoch mean losses).  Weights start from the deterministic default-init state."""
    from lft_amd import trainer
    from lft_amd.params import deterministic_state
    from model import LFT
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, s, seed=1).items()})
    net = net.to(dev).train()
    src = trainer.SyntheticPatchSource(n_patches, A, s, patch=32, seed=seed, device=dev, fmax=fmax)
    hist = trainer.fit(net, src, epochs=epochs, batch_size=batch, lr=rate, n_steps=10**6, seed=seed, log=lambda m: None)
    torch.cuda.synchronize()
    return {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}, hist


def held_out_scenes(A=5, s=2, n=2, size=48, seed=11, fmax=0.25):
    """n synthetic light fields of A x A views, size x size LR pixels per view: (lr [A*size, A*size], hr [A*size*s, A*size*s])."""
    from lft_amd import trainer
    src = trainer.SyntheticPatchSource(n, A, s, patch=size, seed=seed, fmax=fmax)
    return [(src.lr[i, 0].contiguous(), src.hr[i, 0].contiguous()) for i in range(n)]


def view_psnrs(sr: torch.Tensor, hr: torch.Tensor, A: int) -> np.ndarray:
    """Per-view PSNR of a mosaic against its ground truth, float64 on the CPU: [A, A]."""
    sr, hr = sr.double().cpu(), hr.double().cpu()
    H, W = hr.shape[-2] // A, hr.shape[-1] // A
    out = np.empty((A, A))
    for u in range(A):
        for v in range(A):
            mse = float(((sr[u * H:(u + 1) * H, v * W:(v + 1) * W] - hr[u * H:(u + 1) * H, v * W:(v + 1) * W]) ** 2).mean())
            out[u, v] = 10.0 * np.log10(1.0 / mse)
    return out


def oracle_scene(sd_np, lr_scene: torch.Tensor, A: int, s: int, patch=32, stride=16) -> torch.Tensor:
    """The reference's test loop (test.py:79-101) on the CPU oracle: LFdivide, forward per patch, LFintegrate -> SR mosaic."""
    sd = O.state_from_numpy(sd_np)
    h0, w0 = lr_scene.shape[0] // A, lr_scene.shape[1] // A
    sub = O.lf_divide(lr_scene.float(), A, patch, stride)                     # [numU, numV, A*patch, A*patch]
    nu, nv = sub.shape[0], sub.shape[1]
    outs = torch.empty((nu, nv, A * patch * s, A * patch * s))
    for u in range(nu):
        for v in range(nv):
            outs[u, v] = O.forward(sd, sub[u, v][None, None], A, s)[0, 0]
    return O.views_to_scene_mosaic(O.lf_integrate(outs, A, patch * s, stride * s, h0 * s, w0 * s))


def psnr_delta(dev, sd_np, scenes, A=5, s=2, precisions=("fp32", "fp16", "bf16")):
    """-> {precision: {"max_abs_delta_db", "mean_abs_delta_db", "psnr_oracle_mean_db", "psnr_path_mean_db", "views"}}"""
    from lft_amd import evaluate
    from model import LFT
    ref = [view_psnrs(oracle_scene(sd_np, lr, A, s), hr, A) for lr, hr in scenes]
    res = {}
    for prec in precisions:
        net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s), precision=prec)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        net = net.to(dev).eval()
        deltas, mine = [], []
        for (lr, hr), r in zip(scenes, ref):
            _, _, sr = evaluate.test_scene(net, lr, hr)
            p = view_psnrs(sr, hr, A)
            mine.append(p)
            deltas.append(np.abs(p - r))
        d = np.stack(deltas)
        res[prec] = {"max_abs_delta_db": float(d.max()), "mean_abs_delta_db": float(d.mean()),
                     "psnr_oracle_mean_db": float(np.mean(ref)), "psnr_path_mean_db": float(np.mean(mine)), "views": int(d.size)}
    return res
