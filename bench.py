#!/usr/bin/env python3
"""Benchmark of the LFT forward hot path on MI355X.

Workload (BASELINE.json configs[1]): LFT 5x5 angRes, 4x SR, 32x32 LR patches, batch 4 per GPU, bf16 MFMA
operands with fp32 accumulation; synthetic U[0,1) inputs and seeded default-init weights; inputs resident
in HBM before the timed region.  A "step" is one forward over the batch.  Multi-GPU = independent
data-parallel shards (weak scaling, no data-path collective); one process per GPU, launched by
torch.distributed.run, barrier + synchronize on both sides of the timed region, max over ranks.

Prints ONE JSON line on rank 0 with the extra objects
  roofline     : for the dominant kernel (largest share of GPU time; HIP events on the launch stream, inside this run)
                 its algorithmic FLOPs and HBM bytes per launch decide the bound -- dense bf16 MFMA peak 2.5 PFLOP/s
                 (fp32 path 157.3 TFLOP/s) or HBM 8 TB/s -- and `achieved`/`frac` are quoted against that peak
  cpu_baseline : the CPU oracle (a port of the reference's operator sequence, oracle/lft_oracle.py) timed on
                 this node's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

A, S, H, W = 5, 4, 32, 32          # BASELINE.json metric: 5x5 angRes, 32x32 LR, 4x SR (overridable for the other configs)
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # /opt/skills/guides/MI355X_MICROARCH.md (dense)


def flops_per_token(s: int, V: int, wbar: float) -> dict:
    """Algorithmic FLOPs (2*MAC) per token and kernel, SURVEY.md 8(d): window-limited attention,
    LN/softmax/activations excluded, position-token embedding excluded (cached)."""
    C, E = 64, 128
    return {
        "k_conv0": 18 * C,
        "k_conv64": 18 * C * C,
        "k_ang": 16 * C * C + 4 * V * C,
        "k_spa1": 18 * C * E + 6 * E * E,
        "k_spa_attn": 4 * wbar * E,
        "k_spa2": 2 * E * E + 8 * E * E + 2 * E * C,
        "k_spa_b": 4 * wbar * E + 2 * E * E + 8 * E * E + 2 * E * C,     # bf16: attention + out_proj + FFN + 1x1x1 in one kernel
        "k_up": 2 * C * C * s * s + 18 * C * s * s,
        "k_assemble": 32 * s * s,
    }


def bytes_per_token(s: int, esz: int) -> dict:
    """Algorithmic HBM bytes per token and kernel: every activation tensor a kernel must read or write once
    (esz = bytes per stored activation element; weights and tables are amortised over the batch and ignored)."""
    C, E, gp = 64, 128, (s + 2) * (s + 2)
    return {
        "k_conv0": 4 + C * esz,                       # LR pixel in, 64-channel token out
        "k_conv64": 2 * C * esz,                      # token in, token out (the residual read of the 3rd conv is ignored)
        "k_ang": 2 * C * esz,
        "k_spa1": C * esz + 4 * E * esz,              # x in; tok, Q, K, V out
        "k_spa_attn": 4 * E * esz,                    # Q, K, V in (halo re-reads are not algorithmic); O out
        "k_spa2": 2 * E * esz + C * esz,              # tok, O in; x out
        "k_spa_b": 4 * E * esz + C * esz,             # tok, Q, K, V in (halo re-reads are not algorithmic); x out
        "k_up": C * esz + gp * 4,                     # x in; (s+2)^2 fp32 footprint out
        "k_assemble": gp * 4 + 4 + s * s * 4,         # footprint + LR pixel in; s*s HR pixels out
    }


HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def mean_window(h: int, w: int) -> float:
    cnt = lambda n: sum(min(n, i + 3) - max(0, i - 2) for i in range(n))  # noqa: E731
    return cnt(h) * cnt(w) / (h * w)


def kernel_breakdown(net, lr, reps: int):
    """Per-kernel mean milliseconds per launch via lft_forward_profiled (HIP events on torch's current stream)."""
    from lft_amd import _lib
    from lft_amd.module import _PREC
    B = lr.shape[0]
    prec = _PREC[net.precision]
    stream = torch.cuda.current_stream().cuda_stream
    packed = net._ensure_packed(lr.device, H, W, prec, stream)
    work = net._ensure_work(lr.device, B, H, W, prec, slot="profile")
    out = torch.empty((B, 1, A * H * S, A * W * S), dtype=torch.float32, device=lr.device)
    n_max = 64
    ms = (ctypes.c_float * n_max)()
    names = (ctypes.c_char_p * n_max)()
    n = ctypes.c_int(0)
    acc = {}
    for _ in range(reps):
        _lib.check(_lib.lib().lft_forward_profiled(packed.data_ptr(), lr.data_ptr(), out.data_ptr(), work.data_ptr(),
                                                   B, A, H, W, S, prec, stream, n_max, ms, names, ctypes.byref(n)),
                   "lft_forward_profiled")
        for i in range(n.value):
            k = names[i].decode()
            t, c = acc.get(k, (0.0, 0))
            acc[k] = (t + ms[i], c + 1)
    return {k: (t / c, c // reps) for k, (t, c) in acc.items()}     # name -> (mean ms per launch, launches per forward)


def cpu_baseline(seconds: float):
    from lft_amd.params import deterministic_state, synthetic_lr
    from oracle import lft_oracle as O            # the checker, timed as the CPU baseline
    # threads = this process's CPU share (the GPU box gives 16 CPUs per GPU; os.cpu_count() reports the whole host)
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("LFT_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    sd = O.state_from_numpy(deterministic_state(64, S, seed=1))
    lr = torch.from_numpy(synthetic_lr(1, A, H, W, seed=0))
    O.forward(sd, lr, A, S)                        # warm
    n, t0 = 0, time.perf_counter()
    while True:
        O.forward(sd, lr, A, S)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 64:
            break
    return {"value": n / dt, "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} single-patch forwards (A5, 4x, 32x32 LR, fp32, torch {torch.__version__} CPU ops) in {dt:.1f} s"}


TRAFFIC_PROFILE = os.path.join(ROOT, "profiles", "r01_v7_hbm_traffic.json")


def traffic_from_profile(kernel: str, args) -> dict:
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
    runs, gfx950 correction; tools/collect_traffic.py).  PMC counters cannot be read from inside this process, so
    the value is only reported for the exact workload the profile was taken on; otherwise null."""
    default = (A, S, H, W, args.batch, args.precision) == (5, 4, 32, 32, 4, "bf16")
    if default and os.path.exists(TRAFFIC_PROFILE):
        k = json.load(open(TRAFFIC_PROFILE))["kernels"].get(kernel)
        if k:
            return {"traffic": k["total"], "traffic_source": "profiles/r01_v7_hbm_traffic.json"}
    return {"traffic": None}


def note(msg: str) -> None:
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4, help="LF patches per GPU per step (BASELINE configs[1]: 4)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--ang", type=int, default=5, help="angular resolution (default: the BASELINE metric's 5)")
    ap.add_argument("--lr", type=int, default=32, help="LR view size (default 32)")
    ap.add_argument("--scale", type=int, default=4, choices=[2, 4])
    ap.add_argument("--streams", type=int, default=1, help="HIP streams ONE step's batch is split over (1 = whole-batch kernels; overlap then comes from --inflight)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the host each step instead of replaying a HIP graph")
    ap.add_argument("--inflight", type=int, default=2, help="steps in flight: captured forwards replayed round-robin on this many streams (1 = strictly one after the other)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    global A, S, H, W
    A, S, H, W = args.ang, args.scale, args.lr, args.lr

    from lft_amd import dp
    rank, local, world = dp.env_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            # Plain `python bench.py --gpus N`: start the N ranks ourselves, as child processes of a parent that never
            # touches the GPU (nothing below this point has run yet), and leave with the launcher's exit code.
            import socket
            import subprocess
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            note("spawning: " + " ".join(cmd))
            raise SystemExit(subprocess.call(cmd))
        args.gpus = world
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from lft_amd.params import deterministic_state, synthetic_lr
    from model import LFT
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=S), precision=args.precision, streams=args.streams)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, S, seed=1).items()})
    net = net.to(dev).eval()
    lr = torch.from_numpy(synthetic_lr(args.batch, A, H, W, seed=rank)).to(dev)   # resident in HBM

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()                  # device-wide: covers the pipeline's streams

    note(f"rank {rank}/{world}: model on {dev}, precision {args.precision}, batch {args.batch}")
    step = net
    pipe = None
    if not args.no_graph:
        from lft_amd.module import GraphedForward, PipelinedForward
        if args.inflight > 1:                     # every step: a whole batch through the whole network; consecutive steps overlap
            pipe = PipelinedForward(net, lr, depth=args.inflight)
            step = lambda x: pipe()               # noqa: E731  (each captured forward owns a resident copy of the input)
        else:
            step = GraphedForward(net, lr)        # one HIP-graph launch per step; lr is the graph's resident input buffer
    with torch.no_grad():                         # setup, before the contract's W warm-up steps: let clocks and caches settle
        for _ in range(30):
            step(lr)
        torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(args.warmup):
            out = step(lr)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(lr)
        sync()
        dt = time.perf_counter() - t0
    assert bool(torch.isfinite(out).all())
    note(f"rank {rank}: {args.steps} steps in {dt:.3f} s")
    dt = dp.barrier_max_seconds(dt, dev)          # MAX over ranks

    result = None
    if rank == 0:
        V = A * A
        ntok = args.batch * V * H * W
        fpt = flops_per_token(S, V, mean_window(H, W))
        with torch.no_grad():
            kb = kernel_breakdown(net, lr, reps=10)
        total_ms = sum(ms * cnt for ms, cnt in kb.values())
        note("kernel ms/launch: " + ", ".join(f"{k}={ms:.3f}x{c}" for k, (ms, c) in kb.items()))
        dom = max(kb, key=lambda k: kb[k][0] * kb[k][1])
        dom_ms, dom_cnt = kb[dom]
        peak = PEAK_TFLOPS[args.precision]
        bpt = bytes_per_token(S, 2 if args.precision == "bf16" else 4)

        def roof(k, ms):
            """Roofline of one kernel: the larger of (FLOPs / MFMA peak) and (bytes / HBM peak) names the bound."""
            gflop = fpt[k] * ntok / 1e9
            gbyte = bpt[k] * ntok / 1e9
            t_mfma, t_hbm = gflop / (peak * 1e3), gbyte / HBM_PEAK_GBS
            if t_mfma >= t_hbm:
                return {"bound": "mfma", "achieved": gflop / ms, "peak": peak, "unit": "TFLOP/s", "frac": gflop / ms / peak}
            return {"bound": "hbm", "achieved": gbyte / (ms * 1e-3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbyte / (ms * 1e-3) / HBM_PEAK_GBS}

        flops_patch = sum(fpt[k] * (V * H * W) * c for k, (_, c) in kb.items())
        result = {
            "metric": f"LF patches/sec ({A}x{A} angRes, {H}x{W} LR, {S}xSR)",
            "value": world * args.batch * args.steps / dt,
            "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"LFT {A}x{A} angRes {S}xSR inference, batch={args.batch} per GPU, {H}x{W} LR patches",
                       "global_batch": world * args.batch, "parallelism": f"dp{world} (independent shards)",
                       "streams_per_gpu": args.streams, "hip_graph": not args.no_graph, "steps_in_flight": (args.inflight if not args.no_graph else 1),
                       "algorithmic_gflop_per_patch": flops_patch / 1e9},
            "roofline": dict(roof(dom, dom_ms), kernel=dom, **traffic_from_profile(dom, args), launch_ms=dom_ms, launches_per_forward=dom_cnt,
                             gpu_ms_per_forward=total_ms,
                             algorithmic_per_launch={"gflop": fpt[dom] * ntok / 1e9, "gbyte": bpt[dom] * ntok / 1e9},
                             kernels={k: dict(ms=round(ms, 4), n=c, **{kk: (round(vv, 3) if isinstance(vv, float) else vv)
                                                                         for kk, vv in roof(k, ms).items() if kk in ("bound", "achieved", "frac")})
                                      for k, (ms, c) in kb.items()}),
        }
        if world == 1 and not args.no_cpu_baseline:
            note("timing the CPU oracle on host cores ...")
            result["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
