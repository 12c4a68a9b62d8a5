"""Worker of tests/test_gpu_train.py::test_two_rank_dp_step_equals_single_rank: torch.distributed.run starts 2 ranks that
SHARE cuda:0 (gloo backend, flat gradients summed through the host); each takes half of a 2-patch batch for 2 Adam
steps and rank 0 saves the resulting weights."""
import os, sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lft_amd import dp, train as T                                     # noqa: E402
from lft_amd.params import deterministic_state, synthetic_lr          # noqa: E402
from model import LFT                                                  # noqa: E402


def run(world_batch, out_path, scratch=False):
    rank, _, world = dp.env_world()
    if world > 1:
        dist.init_process_group("gloo")
    A, s, B, h, w = 3, 2, world_batch, 6, 6
    if scratch:
        torch.manual_seed(100 + rank)          # a network built from scratch: every process draws its own weights
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s))
    if not scratch:
        net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, s, seed=1, flavor="stress").items()})
    net = net.to("cuda:0").train()
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0))
    hr = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).random((B, 1, A * h * s, A * w * s), dtype=np.float32))
    b, e = dp.shard_range(B, rank, world)
    ts = T.TrainStep(net, lr=2e-4)
    losses = [float(ts.step(lr[b:e].cuda(), hr[b:e].cuda())) for _ in range(2)]
    if rank == 0:
        torch.save({"flat": ts.flat_params.cpu(), "losses": losses}, out_path)
    if scratch:
        torch.save({"flat": ts.flat_params.cpu()}, f"{out_path}.rank{rank}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), sys.argv[2], scratch=len(sys.argv) > 3 and sys.argv[3] == "scratch")
