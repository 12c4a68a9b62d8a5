"""Data-parallel plumbing for the LFT hot path: one process per GPU, patches sharded over ranks.

Inference (BASELINE configs 2, 4, 5) needs no data-path collective: light-field patches never interact
(SURVEY.md 8e), so each rank owns a contiguous slice of the batch.  torch.distributed ("nccl" = RCCL on
ROCm, "gloo" in the CPU tests) is only used for the barrier / max-over-ranks of the timing protocol and
for gathering results where a caller wants them on one rank.
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
