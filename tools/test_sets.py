#!/usr/bin/env python3
"""The reference's test.py on this framework: every scene of every test dataset under ``--path_for_test`` (the reference's
``./data_for_test/SR_{A}x{A}_{s}x/<dataset>/<scene>.h5``, read by lft_amd.h5lite) through LFdivide -> LFT -> LFintegrate ->
per-view PSNR / SSIM, all on the GPU (lft_amd.evaluate.test_sets).  Option names are the reference's (option.py).

    python tools/test_sets.py --angRes 5 --scale_factor 4 --use_pre_pth --path_pre_pth ./pth/LFT_5x5_4x_epoch_50_model.pth
    python -m torch.distributed.run --nproc-per-node 8 tools/test_sets.py ...      (scenes of a dataset sharded over the ranks)"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--angRes", type=int, default=5)
    ap.add_argument("--scale_factor", type=int, default=4)
    ap.add_argument("--model_name", default="LFT")
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--use_pre_pth", action="store_true")
    ap.add_argument("--path_pre_pth", default="./pth/LFT_5x5_4x_epoch_50_model.pth")
    ap.add_argument("--path_for_test", default="./data_for_test/")
    ap.add_argument("--patch_size_for_test", type=int, default=32)
    ap.add_argument("--stride_for_test", type=int, default=16)
    ap.add_argument("--num_workers", type=int, default=0)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "bf16"])
    args = ap.parse_args()
    from lft_amd import dp, evaluate, trainer
    rank, local, world = dp.env_world()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    MODEL = importlib.import_module("model." + args.model_name)          # reference test.py:27-29
    args.lft_precision = args.precision
    net = MODEL.get_model(args).to(dev)
    if args.use_pre_pth:
        trainer.load_checkpoint(net, args.path_pre_pth)
    else:
        net.apply(MODEL.weights_init)
    evaluate.test_sets(net, args, log=(print if rank == 0 else (lambda *_: None)))


if __name__ == "__main__":
    main()
