"""world_size-2 gloo tests (CPU) of the data-parallel plumbing used by bench.py --gpus N: shard partition,
max-over-ranks timing, ragged gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lft_amd import dp


def test_shard_range_partitions():
    for n in (1, 4, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        dp.shard_range(4, 2, 2)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert dp.env_world() == (rank, rank, world)
        n_total = 5                                                     # ragged: 3 + 2
        b, e = dp.shard_range(n_total, rank, world)
        local = torch.arange(b, e, dtype=torch.float32).view(-1, 1, 1) * torch.ones(1, 2, 3)   # "patch i" filled with i
        dist.barrier()
        t = dp.barrier_max_seconds(1.0 + rank, torch.device("cpu"))
        full = dp.gather_patches(local, n_total)
        ok = (t == float(world)) and full.shape == (n_total, 2, 3) and bool((full[:, 0, 0] == torch.arange(n_total)).all())
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_protocol():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1)
