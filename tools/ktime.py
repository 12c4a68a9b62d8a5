#!/usr/bin/env python3
"""Kernel timing A/B without any result check: for each library given (LFT_LIB_PATH-style paths or names under ab_so/), run one
forward of BASELINE configs[1] and time the named kernels with lft_kernel_time (launches back to back between two HIP events).
For timing-only builds whose results may be garbage (knock-out experiments) -- bench.py refuses those.

  tools/ktime.py [--rounds 3] [--reps 50] [--kernels k_spa1,k_spa_b,k_ang,k_conv64] name1 name2 ...
Each library is loaded in a fresh subprocess (one library per process), rounds interleaved."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    import ctypes
    sys.path.insert(0, ROOT)
    import torch
    from lft_amd import _lib
    from lft_amd.params import deterministic_state, param_table, synthetic_lr
    A, s, B, h, w = args.ang, args.scale, args.batch, args.lr, args.lr
    prec = {"bf16": _lib.PREC_BF16, "fp16": _lib.PREC_F16, "fp32": _lib.PREC_F32}[args.precision]
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    sd = deterministic_state(64, s, seed=1)
    params = [torch.from_numpy(sd[n]).cuda() for n, _, _ in param_table(64, s)]
    packed = torch.empty(_lib.packed_bytes(A, h, w, s, prec), dtype=torch.uint8, device="cuda")
    arr = (ctypes.c_void_p * len(params))(*[p.data_ptr() for p in params])
    _lib.check(L.lft_pack_weights(arr, len(params), packed.data_ptr(), A, h, w, s, prec, st), "pack")
    work = torch.empty(_lib.workspace_bytes(B, A, h, w, s, prec), dtype=torch.uint8, device="cuda")
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).cuda()
    out = torch.empty(B, 1, A * h * s, A * w * s, device="cuda")
    for _ in range(3):
        _lib.check(L.lft_forward(packed.data_ptr(), lr.data_ptr(), out.data_ptr(), work.data_ptr(), B, A, h, w, s, prec, st), "forward")
    torch.cuda.synchronize()
    res = {}
    for k in args.kernels.split(","):
        ms = ctypes.c_float(0)
        best = []
        for _ in range(3):
            _lib.check(L.lft_kernel_time(k.encode(), packed.data_ptr(), work.data_ptr(), B, A, h, w, s, prec, args.reps, st, ctypes.byref(ms)), k)
            best.append(ms.value * 1e3)
        res[k] = sorted(best)[1]                     # median of three
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--kernels", default="k_spa1,k_spa_b,k_ang,k_conv64")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--ang", type=int, default=5)
    ap.add_argument("--scale", type=int, default=4)
    ap.add_argument("--lr", type=int, default=32)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for rnd in range(args.rounds):
        for name in args.libs:
            path = name if os.path.exists(name) else os.path.join(ROOT, "ab_so", f"liblft_{name}.so")
            env = dict(os.environ, LFT_LIB_PATH=path)
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--reps", str(args.reps), "--kernels", args.kernels, "--precision", args.precision,
                   "--ang", str(args.ang), "--scale", str(args.scale), "--lr", str(args.lr), "--batch", str(args.batch)]
            r = subprocess.run(cmd, env=env, capture_output=True, text=True)
            try:
                j = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"[{rnd}] {name:12s} " + "  ".join(f"{k[2:]}={v:6.1f}us" for k, v in j.items()), flush=True)
            except Exception as e:
                print(name, "FAILED", e, r.stderr[-500:], flush=True)


if __name__ == "__main__":
    main()
