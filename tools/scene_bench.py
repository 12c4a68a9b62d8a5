"""Whole-scene inference throughput (reference test.py:83-101 as one batched call): LFdivide on the GPU, every
numU x numV patch through the network as one batch, LFintegrate.  python tools/scene_bench.py [--ang 5 --scale 4 --size 128]"""
import argparse, json, os, sys, time
from types import SimpleNamespace
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from lft_amd import scene                                          # noqa: E402
from lft_amd.params import deterministic_state                     # noqa: E402
from model import LFT                                              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ang", type=int, default=5); ap.add_argument("--scale", type=int, default=4)
ap.add_argument("--size", type=int, default=128, help="LR view height = width"); ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--precision", default="bf16")
a = ap.parse_args()
net = LFT.get_model(SimpleNamespace(channels=64, angRes=a.ang, scale_factor=a.scale), precision=a.precision)
net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, a.scale, seed=1).items()})
net = net.cuda().eval()
sc = torch.from_numpy(np.random.default_rng(0).random((a.ang * a.size, a.ang * a.size), dtype=np.float32)).cuda()
with torch.no_grad():
    for _ in range(3):
        out = scene.super_resolve_scene(net, sc, patch=32, stride=16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.reps):
        out = scene.super_resolve_scene(net, sc, patch=32, stride=16)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.reps
nu, nv = scene.scene_counts(a.size, a.size, 32, 16)
print(json.dumps({"metric": "LF scenes/sec", "value": 1 / dt, "ms_per_scene": 1e3 * dt, "patches_per_scene": nu * nv,
                  "patches_per_s": nu * nv / dt, "config": f"{a.ang}x{a.ang} views of {a.size}x{a.size} LR, {a.scale}x, patch 32 stride 16, {a.precision}",
                  "out_shape": list(out.shape)}))
