"""Data-parallel plumbing for the LFT hot path: one process per GPU, patches sharded over ranks.

Inference (BASELINE configs 2, 4, 5) needs no data-path collective: light-field patches never interact
(SURVEY.md 8e), so each rank owns a contiguous slice of the batch.  torch.distributed ("nccl" = RCCL on
ROCm, "gloo" in the CPU tests) is only used for the barrier / max-over-ranks of the timing protocol and
for gathering results where a caller wants them on one rank.

Training (BASELINE config 3) has exactly one exchange per step: the sum of the flat gradient buffer
(1.11 M floats, 4.5 MB) over the ranks; the division by the world size is folded into the Adam kernel
(lft_amd/train.py:TrainStep).  The buffer is summed in the three contiguous buckets in which the backward pass finishes
it (``sum_gradients_start_`` / ``sum_gradients_finish``): the all-reduce of a bucket runs on the process group's own
stream beside the backward kernels of the next one.  ``sum_gradients_`` is the one-call form.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of n_items for `rank`; sizes differ by at most one, earlier ranks larger."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def barrier_max_seconds(seconds: float, device: torch.device) -> float:
    """Max of a per-rank duration over all ranks (the bench contract); identity without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_patches(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather per-rank output patches [n_local, ...] (ragged over ranks by at most one) into [n_total, ...]."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    pad = max(sizes)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


def _force() -> bool:
    # LFT_DP_FORCE_COLLECTIVES=1: issue the collectives even in a process group of one rank (a one-GPU box can then
    # exercise the RCCL path end to end; numerically the identity).
    return os.environ.get("LFT_DP_FORCE_COLLECTIVES", "0") == "1"


def broadcast_(flat: torch.Tensor, src: int = 0, group=None, force: bool = False) -> torch.Tensor:
    """In-place broadcast of a flat buffer from rank ``src`` (what DistributedDataParallel does with the parameters at
    construction): data-parallel replicas must start from identical weights, and a freshly constructed network draws
    them from each process's own RNG.  Identity without a process group.  gloo + device tensor: through the host."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    if dist.get_world_size(group) == 1 and not (force or _force()):
        return flat
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.cpu()
        dist.broadcast(host, src=src, group=group)
        flat.copy_(host)
    else:
        dist.broadcast(flat, src=src, group=group)
    return flat


def sum_gradients_(flat: torch.Tensor, group=None, force: bool = False) -> float:
    """In-place SUM all-reduce of the flat gradient buffer over the data-parallel ranks; returns the factor
    (1 / world) that turns the sum of per-shard mean-loss gradients into the global-batch gradient (equal shards:
    L1Loss is a mean over the local shard, reference LFT.py:272).  One collective per step.
    With the gloo backend (CPU rehearsals, or several ranks sharing one GPU) device tensors go through the host."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world == 1 and not (force or _force()):
        return 1.0
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def dp_active(group=None, force: bool = False) -> bool:
    """True when gradients have to be exchanged: a process group of more than one rank (or a forced one-rank rehearsal)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or force or _force()


def grad_scale(group=None) -> float:
    """1 / world: turns the SUM of per-shard mean-loss gradients into the global-batch gradient (equal shards)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    return 1.0 / dist.get_world_size(group)


def sum_gradients_start_(bucket: torch.Tensor, group=None):
    """Start the in-place SUM all-reduce of one contiguous gradient bucket; returns a handle for sum_gradients_finish.
    NCCL (= RCCL) backend: asynchronous -- the collective is ordered after the work already enqueued on the current
    stream and runs on the process group's stream, so kernels enqueued afterwards overlap it.  gloo with a device tensor
    (CPU rehearsals, several ranks sharing one GPU): through the host, finished on return."""
    import torch.distributed as dist
    if bucket.is_cuda and dist.get_backend(group) == "gloo":
        host = bucket.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        bucket.copy_(host)
        return None
    return dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group, async_op=True)


def sum_gradients_finish(handles) -> None:
    """Order the current stream after the collectives started by sum_gradients_start_ (no host synchronisation with NCCL)."""
    for h in handles:
        if h is not None:
            h.wait()
