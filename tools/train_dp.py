"""One-process-per-GPU training launcher with the reference's option names (option.py) where they apply.

  python tools/train_dp.py --angRes 5 --scale_factor 2 --batch_size 8 --epoch 50 --data patches.npz --path_log ./log
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 tools/train_dp.py ...

--path_for_train DIR: the reference's own training tree (DIR/SR_AxA_sx/<dataset>/*.h5 with Lr_SAI_y / Hr_SAI_y, as
Generate_Data_for_Training.m writes it), read by lft_amd.h5lite (--data_name as in option.py).
--data: an .npz with arrays Lr_SAI_y [n, A*32, A*32] and Hr_SAI_y [n, A*32*s, A*32*s] (the two datasets of the
reference's training .h5 patches, stacked); --synthetic N instead makes N band-limited random light fields."""
import argparse, os, sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--angRes", type=int, default=5)
    ap.add_argument("--scale_factor", type=int, default=4)
    ap.add_argument("--model_name", default="LFT")
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--use_pre_pth", action="store_true")
    ap.add_argument("--path_pre_pth", default="./pth/LFT_5x5_4x_epoch_50_model.pth")
    ap.add_argument("--path_log", default="./log/")
    ap.add_argument("--batch_size", type=int, default=4, help="GLOBAL batch, as in the reference")
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--n_steps", type=int, default=15)
    ap.add_argument("--gamma", type=float, default=0.5)
    ap.add_argument("--decay_rate", type=float, default=0.0, help="Adam weight_decay, as the reference's option.py")
    ap.add_argument("--epoch", type=int, default=50)
    ap.add_argument("--data", default=None)
    ap.add_argument("--path_for_train", default=None, help="the reference's training tree (./data_for_train/): SR_AxA_sx/<dataset>/*.h5, read by lft_amd.h5lite")
    ap.add_argument("--data_name", default="ALL")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--max_batches", type=int, default=0)
    ap.add_argument("--batch_metrics", action="store_true", help="per-batch PSNR / SSIM of train.py:121-124 (on the GPU) and the reference's epoch line")
    args = ap.parse_args()
    import importlib
    from lft_amd import dp, trainer
    rank, local, world = dp.env_world()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    MODEL = importlib.import_module("model." + args.model_name)          # reference train.py:31-33
    net = MODEL.get_model(args).to(dev)
    start = 0
    if args.use_pre_pth:
        start = trainer.load_checkpoint(net, args.path_pre_pth)
    else:
        net.apply(MODEL.weights_init)
    if args.path_for_train:
        from lft_amd import datasets
        src = datasets.H5PatchSource(args.path_for_train, args.angRes, args.scale_factor, args.data_name, cache=True)
    elif args.data:
        z = np.load(args.data)
        src = trainer.TensorPatchSource(torch.from_numpy(z["Lr_SAI_y"]), torch.from_numpy(z["Hr_SAI_y"]))
    else:
        src = trainer.SyntheticPatchSource(args.synthetic or 64, args.angRes, args.scale_factor, 32, seed=0)
    ckpt_dir = os.path.join(args.path_log, "SR_%dx%d_%dx" % (args.angRes, args.angRes, args.scale_factor), args.model_name, "checkpoints")
    trainer.fit(net, src, args.epoch, args.batch_size, lr=args.lr, n_steps=args.n_steps, gamma=args.gamma, start_epoch=start,
                ckpt_dir=ckpt_dir, model_name=args.model_name, max_batches_per_epoch=args.max_batches or None,
                decay_rate=args.decay_rate, batch_metrics=args.batch_metrics)


if __name__ == "__main__":
    main()
