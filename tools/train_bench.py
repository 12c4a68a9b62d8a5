"""Training-step throughput (BASELINE configs[2]: LFT 5x5 angRes 2xSR training, Adam + gradient all-reduce), fp32.

  python tools/train_bench.py --batch 8 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/train_bench.py --gpus N ...

One process per GPU, each with its own --batch patches (weak scaling); a step = forward-with-tape, L1 loss + its
gradient, backward, ONE all-reduce of the flat gradient buffer (RCCL), fused Adam.  Same timing protocol as bench.py
(barrier + synchronize on both sides, max over ranks).  Prints one JSON line on rank 0.  --phases adds a per-phase
breakdown from HIP events on the step's stream (forward / loss / backward / all-reduce / adam)."""
import argparse, json, os, sys, time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="LF patches per GPU per step")
    ap.add_argument("--ang", type=int, default=5)
    ap.add_argument("--lr", type=int, default=32)
    ap.add_argument("--scale", type=int, default=2, choices=[2, 4])
    ap.add_argument("--phases", action="store_true")
    ap.add_argument("--math", default="bf16x3", choices=["fp32", "bf16x3", "bf16x6"])
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    from lft_amd import dp, train as T
    from lft_amd.params import deterministic_state, synthetic_lr
    from model import LFT
    rank, local, world = dp.env_world()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    A, S, H = args.ang, args.scale, args.lr
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=S))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, S, seed=1).items()})
    net = net.to(dev).train()
    lr = torch.from_numpy(synthetic_lr(args.batch, A, H, H, seed=rank)).to(dev)
    hr = torch.from_numpy(np.random.Generator(np.random.PCG64([2, rank])).random((args.batch, 1, A * H * S, A * H * S), dtype=np.float32)).to(dev)
    ts = T.TrainStep(net, lr=2e-4, math=args.math, graph=not args.no_graph)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for _ in range(args.warmup):
        losses.append(ts.step(lr, hr))
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(ts.step(lr, hr))
    sync()
    dt = dp.barrier_max_seconds(time.perf_counter() - t0, dev)
    lv = [float(x) for x in losses]
    assert all(np.isfinite(lv)), lv
    if rank == 0:
        V = A * A
        flops_fwd = {2: 58.85e9, 4: 61.73e9}[S] * (H * H / 1024.0) * (V / 25.0)      # SURVEY 8d (approximate outside cfg shapes)
        out = {"metric": f"LF patches/sec training ({A}x{A} angRes, {H}x{H} LR, {S}xSR, fp32, Adam)", "value": args.batch * world * args.steps / dt,
               "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "dtype": "f32" if args.math == "fp32" else "f32 (split-bf16 products)", "data": "synthetic",
               "config": {"workload": f"LFT {A}x{A} angRes {S}xSR training step, batch={args.batch} per GPU, {H}x{H} LR patches",
                          "global_batch": args.batch * world, "parallelism": f"dp{world} (one flat-gradient all-reduce per step)"},
               "loss_first_last": [lv[0], lv[-1]],
               "tflops_algorithmic": 3 * flops_fwd * args.batch * world * args.steps / dt / 1e12,
               "tape_gib": T.tape_bytes(args.batch, A, H, H, S) / 2**30}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
