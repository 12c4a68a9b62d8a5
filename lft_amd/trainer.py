"""Data-parallel training driver around the hot path: the loop of the reference's train.py (:74-132) -- Adam, StepLR
(step 15 epochs, gamma 0.5), per-epoch checkpoints in the reference's ``{'epoch', 'state_dict'}`` format -- with one
process per GPU, every global batch sharded over the ranks and one gradient all-reduce per step (lft_amd.train.TrainStep).
The per-batch skimage PSNR/SSIM of train.py:121-124 is not part of the step; an epoch reports the
mean loss and the PSNR of the last batch computed on the GPU.

Data comes from a *patch source*: anything with ``__len__`` and ``get(indices) -> (lr [n,1,A*p,A*p], hr [n,1,A*p*s,A*p*s])``
float32 tensors.  ``TensorPatchSource`` wraps arrays already in memory; ``lft_amd.datasets.H5PatchSource`` reads the reference's
``Lr_SAI_y`` / ``Hr_SAI_y`` .h5 training tree (own HDF5 reader);
``SyntheticPatchSource`` makes band-limited random light fields for rehearsals and benchmarks.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import dp


# ---------------------------------------------------------------------------------------------- checkpoints
def checkpoint_name(model_name: str, angRes: int, scale: int, epoch: int) -> str:
    """File name used by the reference (train.py:99-100)."""
    return "%s_%dx%d_%dx_epoch_%02d_model.pth" % (model_name, angRes, angRes, scale, epoch)


def save_checkpoint(net, path: str, epoch: int) -> None:
    """``{'epoch': int, 'state_dict': OrderedDict}`` with CPU tensors (reference train.py:101-105)."""
    sd = net.module.state_dict() if hasattr(net, "module") else net.state_dict()
    state = {"epoch": int(epoch), "state_dict": OrderedDict((k, v.detach().cpu().clone()) for k, v in sd.items())}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(state, path)


def load_checkpoint(net, path: str) -> int:
    """Load a reference-format checkpoint, with or without the ``module.`` prefix DataParallel adds (reference
    train.py:42-58, test.py:35-51).  Parameters are updated IN PLACE (they may be views of a flat buffer).
    Returns the stored epoch."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    clean = OrderedDict((k[len("module."):] if k.startswith("module.") else k, v) for k, v in sd.items())
    own = net.state_dict()
    missing = [k for k in own if k not in clean]
    extra = [k for k in clean if k not in own]
    if missing or extra:
        raise KeyError(f"checkpoint does not match the model: missing {missing[:3]}..., unexpected {extra[:3]}...")
    with torch.no_grad():
        for k, p in net.named_parameters():
            p.copy_(clean[k].to(p.device, p.dtype).reshape(p.shape))
    if hasattr(net, "_packed"):
        net._packed = None
    return int(ckpt.get("epoch", 0)) if isinstance(ckpt, dict) else 0


# ---------------------------------------------------------------------------------------------- schedule / sampling
def step_lr(base_lr: float, epoch: int, n_steps: int = 15, gamma: float = 0.5) -> float:
    """torch.optim.lr_scheduler.StepLR as the reference uses it (train.py:84, stepped once per epoch)."""
    return base_lr * gamma ** (epoch // n_steps)


def epoch_batches(n_items: int, global_batch: int, epoch: int, seed: int, rank: int, world: int) -> List[np.ndarray]:
    """This rank's index batches for one epoch: a permutation seeded by (seed, epoch) -- identical on every rank --
    cut into global batches (the reference's DataLoader(shuffle=True, batch_size=--batch_size), train.py:26-27), each
    split contiguously over the ranks.  global_batch must divide by world; the tail wraps around so that all ranks
    always hold equal shards (the gradient average over ranks is then the global-batch gradient)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by the world size {world}")
    perm = np.random.Generator(np.random.PCG64([seed, epoch])).permutation(n_items)
    nb = (n_items + global_batch - 1) // global_batch
    idx = np.resize(perm, nb * global_batch)                       # wrap-around padding of the last batch
    per = global_batch // world
    return [idx[b * global_batch + rank * per: b * global_batch + (rank + 1) * per] for b in range(nb)]


def augment(lr: torch.Tensor, hr: torch.Tensor, rng: np.random.Generator) -> Tuple[torch.Tensor, torch.Tensor]:
    """The reference's three augmentations (utils_datasets.py:123-135), decided per sample: mirror the whole mosaic
    left-right, up-down, and transpose it (each flips / swaps the angular and the spatial axes together)."""
    lo, ho = [], []
    for i in range(lr.shape[0]):
        a, b = lr[i], hr[i]
        if rng.random() < 0.5:
            a, b = a.flip(-1), b.flip(-1)
        if rng.random() < 0.5:
            a, b = a.flip(-2), b.flip(-2)
        if rng.random() < 0.5:
            a, b = a.transpose(-1, -2), b.transpose(-1, -2)
        lo.append(a)
        ho.append(b)
    return torch.stack(lo).contiguous(), torch.stack(ho).contiguous()


# ---------------------------------------------------------------------------------------------- patch sources
class TensorPatchSource:
    def __init__(self, lr: torch.Tensor, hr: torch.Tensor):
        if lr.dim() == 3:
            lr, hr = lr[:, None], hr[:, None]
        assert lr.shape[0] == hr.shape[0] and lr.dim() == 4 and hr.dim() == 4
        self.lr, self.hr = lr.float(), hr.float()

    def __len__(self):
        return self.lr.shape[0]

    def get(self, indices: Sequence[int]):
        ix = torch.as_tensor(np.asarray(indices), dtype=torch.long, device=self.lr.device)
        return self.lr[ix], self.hr[ix]


class SyntheticPatchSource(TensorPatchSource):
    """Smooth random light fields: HR views = a random low-frequency image shifted by a per-view disparity; LR = box
    down-sampling of each view.  Learnable (the network must undo the blur), deterministic in ``seed``."""

    def __init__(self, n: int, angRes: int, scale: int, patch: int = 32, seed: int = 0, device="cpu", fmax: float = 0.25):
        # fmax: highest spatial frequency of the scene in cycles per HR pixel (0.25 = the LR Nyquist limit at 2x: hard scenes,
        # bicubic ~24 dB; 0.06: smooth scenes on which a trained network reaches the PSNR range of the reference's tables)
        g = np.random.Generator(np.random.PCG64([seed, n, angRes, scale, patch]))
        P = patch * scale
        hr = np.empty((n, angRes * P, angRes * P), dtype=np.float32)
        yy, xx = np.meshgrid(np.arange(P, dtype=np.float32), np.arange(P, dtype=np.float32), indexing="ij")
        for i in range(n):
            k = 6
            fy, fx = g.uniform(0.02, fmax, k), g.uniform(0.02, fmax, k)
            ph, am = g.uniform(0, 2 * np.pi, k), g.uniform(0.2, 1.0, k)
            disp = g.uniform(-1.5, 1.5)
            for u in range(angRes):
                for v in range(angRes):
                    img = sum(am[j] * np.sin(2 * np.pi * (fy[j] * (yy + disp * u) + fx[j] * (xx + disp * v)) + ph[j]) for j in range(k))
                    hr[i, u * P:(u + 1) * P, v * P:(v + 1) * P] = 0.5 + 0.5 * img / am.sum()
        hr_t = torch.from_numpy(hr)
        lr_t = hr_t.reshape(n, angRes, patch, scale, angRes, patch, scale).mean(dim=(3, 6)).reshape(n, angRes * patch, angRes * patch)
        super().__init__(lr_t.to(device), hr_t.to(device))


# ---------------------------------------------------------------------------------------------- the loop
def psnr_gpu(sr: torch.Tensor, hr: torch.Tensor) -> float:
    mse = float(((sr - hr) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * float(np.log10(1.0 / mse))


def fit(net, source, epochs: int, batch_size: int, lr: float = 2e-4, n_steps: int = 15, gamma: float = 0.5,
        start_epoch: int = 0, ckpt_dir: Optional[str] = None, model_name: str = "LFT", seed: int = 0,
        use_augmentation: bool = True, log=print, max_batches_per_epoch: Optional[int] = None, decay_rate: float = 0.0,
        batch_metrics: bool = False, ssim_range: float = 2.0):
    """Train ``net`` (lft_amd.module.get_model on this rank's GPU) like reference train.py:86-110.  ``batch_size`` is
    the GLOBAL batch (reference --batch_size).  Returns the list of per-epoch mean losses (global).
    batch_metrics: also compute the reference's per-batch ``cal_metrics(args, label, out)`` (train.py:121-124: per-view PSNR / SSIM of
    the step's own output, means over positive views, then the mean over the epoch's batches) -- on the GPU (lft_view_metrics), with
    no host synchronisation inside the epoch -- and log the reference's line ('... loss is: %.5f, psnr is %.5f, ssim is %.5f');
    ``fit.last_metrics`` then holds the per-epoch (psnr, ssim) pairs."""
    import torch.distributed as dist
    from .train import TrainStep
    rank, _, world = dp.env_world()
    if not (dist.is_available() and dist.is_initialized()):
        rank, world = 0, 1
    dev = next(net.parameters()).device
    ts = TrainStep(net, lr=lr, weight_decay=decay_rate)            # reference train.py:82 weight_decay=args.decay_rate
    history = []
    fit.last_metrics = []
    for epoch in range(start_epoch, epochs):
        ts.lr = step_lr(lr, epoch, n_steps, gamma)
        rng = np.random.Generator(np.random.PCG64([seed, epoch, rank, 17]))
        batches = epoch_batches(len(source), batch_size, epoch, seed, rank, world)
        if max_batches_per_epoch:
            batches = batches[:max_batches_per_epoch]
        total = torch.zeros(3 if batch_metrics else 1, device=dev)
        last = None
        for ix in batches:
            a, b = source.get(ix)
            a, b = a.to(dev, non_blocking=True), b.to(dev, non_blocking=True)
            if use_augmentation:
                a, b = augment(a, b, rng)
            loss = ts.step(a, b)
            if batch_metrics:
                from . import metrics
                p, s = metrics.view_metrics(b, ts.last_out, net.angRes, ssim_range)
                total += torch.stack([loss[0], p.sum() / (p > 0).sum(), s.sum() / (s > 0).sum()])
            else:
                total += loss
            last = (a, b)
        mean = total / max(1, len(batches))
        if world > 1:
            host = mean.cpu() if dist.get_backend() == "gloo" else mean
            dist.all_reduce(host)
            mean = host.to(dev) / world
        history.append(float(mean[0]))
        if batch_metrics:
            fit.last_metrics.append((float(mean[1]), float(mean[2])))
            msg = "The %dth Train, loss is: %.5f, psnr is %.5f, ssim is %.5f" % (epoch + 1, history[-1], *fit.last_metrics[-1])   # train.py:90-91
        else:
            msg = "The %dth Train, loss is: %.5f, lr %.3g" % (epoch + 1, history[-1], ts.lr)
        if last is not None and not batch_metrics:
            with torch.no_grad():
                msg += ", psnr(last batch) %.3f" % psnr_gpu(net(last[0]), last[1])
        if rank == 0:
            log(msg)
            if ckpt_dir:
                save_checkpoint(net, os.path.join(ckpt_dir, checkpoint_name(model_name, net.angRes, net.factor, epoch + 1)), epoch + 1)
    return history
