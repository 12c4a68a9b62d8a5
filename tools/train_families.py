#!/usr/bin/env python3
"""Per-kernel-family time of the training step (BASELINE configs[2] shape, batch 8), both math modes: one forward + backward on one
stream with a HIP event after every kernel (lft_train_step_profiled, through bench.train_object).  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

for math in (sys.argv[1:] or ["bf16x3", "fp32"]):
    t = bench.train_object(torch.device("cuda", 0), math, with_roofline=True)
    r = t["roofline"]
    print(math, "step ms", round(t["ms_per_step"], 2), "patches/s", round(t["patches_per_s"], 1), r["families_ms"], "dominant", r["kernel"], r["bound"],
          "frac", round(r["frac"], 3))
    for k, (ms, calls) in r["gemm_shapes_ms"].items():
        print(f"    {k:28s} {ms:7.3f} ms  {calls:3d} calls  {ms / calls * 1e3:7.1f} us each")
