#!/usr/bin/env python3
"""Benchmark of the LFT forward hot path on MI355X.

Workload (BASELINE.json configs[1]): LFT 5x5 angRes, 4x SR, 32x32 LR patches, batch 4 per GPU, bf16 MFMA
operands with fp32 accumulation; synthetic U[0,1) inputs and seeded default-init weights; inputs resident
in HBM before the timed region.  A "step" is one forward over the batch.  Multi-GPU = independent
data-parallel shards (weak scaling, no data-path collective); one process per GPU, launched by
torch.distributed.run, barrier + synchronize on both sides of the timed region, max over ranks.

--config cfg1|cfg2|cfg3|cfg4|cfg5 selects a BASELINE.json configuration by name (cfg2 = configs[1], the metric's own and the
default; cfg3 = the data-parallel training step, see run_training; cfg4 = 64x64 LR views, 8 per GPU; cfg5 = 9x9 views); explicit
--ang/--lr/--scale/--batch/--precision flags override the named shape.

Prints ONE JSON line on rank 0 with the extra objects
  roofline     : for the dominant kernel (largest share of GPU time).  Its algorithmic FLOPs and HBM bytes per launch
                 decide the bound -- dense bf16 MFMA peak 2.5 PFLOP/s (fp32 path 157.3 TFLOP/s) or HBM 8 TB/s -- and
                 `achieved`/`frac` are quoted against that peak.  `launch_ms` = mean of 50 launches of that kernel back to
                 back between two HIP events on the launch stream (lft_kernel_time: no event between launches, so it is
                 comparable with the rocprofv3 kernel trace under profiles/).  `traffic` = HBM-side bytes per launch from the
                 committed PMC passes, only when they were taken on the very sources that are being timed (`source_hash`).
  parity_path  : the paths that meet BASELINE.json's 1e-3 relative tolerance, same workload, same run, SAME timing protocol as the
                 headline (settle, --warmup, --steps, steps in flight): the fp16 path (the headline kernels with IEEE-half operands
                 and tensors) with patches/s and its measured error against the CPU oracle on one patch (max-norm and element-wise on
                 |ref| >= 0.05), the headline path's error beside it, and `exact_fp32` (the exact-fp32 MFMA path).  The top-level
                 `meets_tolerance` / `value_within_tolerance` say whether `value` itself is inside 1e-3 and what the fastest path
                 inside it delivers.
  latency_path : the same workload strictly one step after the other (one captured forward, nothing in flight beside it); the
                 batch whole (`unsplit`) and as two half-batches on two streams inside the step (`split2`); value = the better.
  sustained_path : the headline's step over a timed region of >= 3 000 steps (> 1 s); `value` itself is timed behind one second of
                 settle load (timed_protocol), so the two must agree.  burst_path: the round-3 protocol (100 settle steps from idle).
  per_rank_ms  : every rank's own ms per step (a straggler shows here; `ms_per_step` is the maximum).
  train        : BASELINE configs[2] on this GPU (A5, 2x, batch 8, Adam; one process = no all-reduce partner): ms/step, patches/s.
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
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # /opt/skills/guides/MI355X_MICROARCH.md (dense)


def flops_per_token(s: int, V: int, wbar: float, q_in_part_b: bool = True) -> dict:
    """Algorithmic FLOPs (2*MAC) per token and kernel, SURVEY.md 8(d): window-limited attention,
    LN/softmax/activations excluded, position-token embedding excluded (cached).  q_in_part_b (the 16-bit paths on
    32-column-aligned views, i.e. every BASELINE shape): the Q projection runs in k_spa_b, not in k_spa1."""
    C, E = 64, 128
    qa, qb = (4, 2) if q_in_part_b else (6, 0)
    return {
        "k_conv0": 18 * C,
        "k_conv64": 18 * C * C,
        "k_ang": 16 * C * C + 4 * V * C,
        "k_spa1": 18 * C * E + qa * E * E,
        "k_spa_attn": 4 * wbar * E,
        "k_spa2": 2 * E * E + 8 * E * E + 2 * E * C,
        "k_spa_b": 4 * wbar * E + qb * E * E + 2 * E * E + 8 * E * E + 2 * E * C,     # 16-bit: (Q +) attention + out_proj + FFN + 1x1x1 in one kernel
        "k_up": 2 * C * C * s * s + 18 * C * s * s,
        "k_assemble": 32 * s * s,
    }


def bytes_per_token(s: int, esz: int) -> dict:
    """Algorithmic HBM bytes per token and kernel: every activation tensor a kernel must read or write once
    (esz = bytes per stored activation element; weights and tables are amortised over the batch and ignored).  16-bit paths:
    Q never exists in memory (k_spa1 writes tok, K, V; k_spa_b reads them)."""
    C, E, gp = 64, 128, (s + 2) * (s + 2)
    nqkv = 3 if esz == 2 else 4
    return {
        "k_conv0": 4 + C * esz,                       # LR pixel in, 64-channel token out
        "k_conv64": 2 * C * esz,                      # token in, token out (the residual read of the 3rd conv is ignored)
        "k_ang": 2 * C * esz,
        "k_spa1": C * esz + nqkv * E * esz,           # x in; tok, (Q,) K, V out
        "k_spa_attn": 4 * E * esz,                    # Q, K, V in (halo re-reads are not algorithmic); O out
        "k_spa2": 2 * E * esz + C * esz,              # tok, O in; x out
        "k_spa_b": 3 * E * esz + C * esz,             # tok, K, V in (halo re-reads and the second read of tok are not algorithmic); x out
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


def cpu_baseline(seconds: float, lr=None):
    from lft_amd.params import deterministic_state, synthetic_lr
    from oracle import lft_oracle as O            # the checker, timed as the CPU baseline
    # threads = this process's CPU share (the GPU box gives 16 CPUs per GPU; os.cpu_count() reports the whole host)
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("LFT_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    sd = O.state_from_numpy(deterministic_state(64, S, seed=1))
    if lr is None:
        lr = torch.from_numpy(synthetic_lr(1, A, H, W, seed=0))
    O.forward(sd, lr, A, S)                        # warm
    n, t0 = 0, time.perf_counter()
    while True:
        O.forward(sd, lr, A, S)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 64:
            break
    ref = O.forward(sd, lr, A, S)                  # the checker's output for patch 0 of rank 0's batch
    return {"value": n / dt, "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} single-patch forwards (A{A}, {S}x, {H}x{W} LR, fp32, torch {torch.__version__} CPU ops) in {dt:.1f} s"}, ref


PROFILE_TAG = "r04"                    # the round whose committed PMC passes carry the hash of the sources being timed
TRAFFIC_PROFILE = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_hbm_traffic.json")


def source_hash() -> str:
    """sha256 over the kernel sources: ties a PMC traffic file to the build it was measured on."""
    import hashlib
    from lft_amd import _lib
    hsh = hashlib.sha256()
    for name in sorted(_lib.SOURCES):
        hsh.update(open(os.path.join(_lib.CSRC, name), "rb").read())
    return hsh.hexdigest()[:16]


def traffic_from_profile(kernel: str, args) -> dict:
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
    runs, gfx950 correction; tools/collect_traffic.py).  PMC counters cannot be read from inside this process, so
    the value is only reported for the exact workload AND the exact sources the profile was taken on; otherwise null."""
    cur = source_hash()
    default = (A, S, H, W, args.batch, args.precision) == (5, 4, 32, 32, 4, "bf16")
    if default and os.path.exists(TRAFFIC_PROFILE):
        prof = json.load(open(TRAFFIC_PROFILE))
        k = prof["kernels"].get(kernel)
        if k and prof.get("source_hash") == cur:
            return {"traffic": k["total"], "traffic_source": os.path.relpath(TRAFFIC_PROFILE, ROOT), "source_hash": cur}
        return {"traffic": None, "traffic_note": "%s was taken on other sources (hash %s)" % (os.path.relpath(TRAFFIC_PROFILE, ROOT), prof.get("source_hash")),
                "source_hash": cur}
    return {"traffic": None, "source_hash": cur}


def kernel_time_ms(net, lr, kernel: str, reps: int = 50):
    """Mean ms of one kernel, `reps` launches back to back between two HIP events (lft_kernel_time); None if unsupported."""
    from lft_amd import _lib
    from lft_amd.module import _PREC
    B = lr.shape[0]
    prec = _PREC[net.precision]
    stream = torch.cuda.current_stream().cuda_stream
    packed = net._ensure_packed(lr.device, H, W, prec, stream)
    work = net._ensure_work(lr.device, B, H, W, prec, slot="profile")      # holds the activations of kernel_breakdown's forwards
    ms = ctypes.c_float(0.0)
    rc = _lib.lib().lft_kernel_time(kernel.encode(), packed.data_ptr(), work.data_ptr(), B, A, H, W, S, prec, reps, stream, ctypes.byref(ms))
    return float(ms.value) if rc == 0 else None


def make_step(net, lr, args, inflight: int):
    """The step callable for `inflight` steps in flight (1 = one captured forward replayed strictly one after the other;
    --no-graph: eager launches).  Returns (step, owner); owner.sync(check=True) / owner.check() reads the overflow status."""
    from lft_amd.module import GraphedForward, PipelinedForward
    if args.no_graph:
        return net, None
    if inflight > 1:                          # every step: a whole batch through the whole network; consecutive steps overlap
        pipe = PipelinedForward(net, lr, depth=inflight)
        return (lambda x: pipe()), pipe       # each captured forward owns a resident copy of the input
    g = GraphedForward(net, lr)               # one HIP-graph launch per step; lr is the graph's resident input buffer
    return g, g


SETTLE_SECONDS = 3.0      # untimed load in front of --warmup, see timed_protocol
SETTLE_MIN_STEPS = 100


def settle(step, lr, seconds=None):
    """Run the step under load for `seconds` (and at least SETTLE_MIN_STEPS steps); returns the number of steps it took."""
    seconds = SETTLE_SECONDS if seconds is None else seconds
    n, t0 = 0, time.perf_counter()
    while n < SETTLE_MIN_STEPS or time.perf_counter() - t0 < seconds:
        for _ in range(50):
            step(lr)
        torch.cuda.synchronize()
        n += 50
    return n


def timed_protocol(step, lr, args, sync):
    """THE timing protocol, shared by the headline, the parity paths and the latency path: SETTLE_SECONDS of untimed load, --warmup
    untimed steps, then exactly --steps steps bracketed by barrier + device synchronisation.
    The settle time is chosen A PRIORI, not by which value reads best (round-3 advice): the card's power management has several time
    constants -- clocks come up within tens of milliseconds of load; on some boxes they are taken down again after ~0.4 s of it (round 3,
    gpurun_out/r4k: a 20-step window read 8 640 - 9 000 patches/s behind 30 settle steps, 9 140 - 9 510 behind 100, 8 670 - 8 820 behind
    1 000); on others the rate keeps RISING for more than a second (round 4, gpurun_out/r4k: 9 390 behind 1 s of load, 9 780 in the
    regions timed a few seconds later in the same process).  Three seconds of load is past all of them, so a short timed window (the
    driver's --steps 20) measures the SUSTAINED rate; `sustained_path` (> 1 s timed) cross-checks it and `burst_path` is the round-3
    protocol (100 settle steps from idle).
    """
    with torch.no_grad():
        settle(step, lr)
        for _ in range(args.warmup):
            out = step(lr)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(lr)
        sync()
        dt = time.perf_counter() - t0
    return dt, out


def make_net(precision, dev, streams=1):
    from lft_amd.params import deterministic_state
    from model import LFT
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=S), precision=precision, streams=streams)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, S, seed=1).items()})
    net.check_finite = False                  # the bench checks the status words itself, after the timed regions
    return net.to(dev).eval()


def parity_path(args, dev, lr, head_net):
    """The paths that meet north_star's 1e-3 on the same workload under the SAME protocol as the headline (timed_protocol, same
    steps in flight): fp16 (the bf16 kernels with IEEE-half operands and tensors) and exact fp32; returns their timings and
    patch-0 outputs for the oracle check."""
    timed, outs = {}, {}
    for precision in ("fp16", "fp32"):
        net = make_net(precision, dev)
        step, owner = make_step(net, lr, args, args.inflight)
        dt, _ = timed_protocol(step, lr, args, torch.cuda.synchronize)
        net.check_status()                    # an fp16 overflow is a loud error, never a silently wrong number
        with torch.no_grad():
            outs[precision] = net(lr[:1]).float().cpu()
        timed[precision] = {"value": args.batch * args.steps / dt, "unit": "patches/s", "steps": args.steps, "warmup": args.warmup,
                            "ms_per_step": dt / args.steps * 1e3}
        del step, owner, net
    with torch.no_grad():
        outs["headline"] = head_net(lr[:1]).float().cpu()
    res = dict(timed["fp16"], precision="fp16 (v_mfma_f32_32x32x16_f16, fp16 storage, fp32 accumulation / softmax / LayerNorm)",
               tolerance="max|out - ref| <= 1e-3 * max|ref| (BASELINE.json north_star; the output lives in [0, 1], so this is an absolute bound)",
               protocol=f"as the headline: {SETTLE_SECONDS} s of settle load, --warmup, --steps, same steps in flight",
               overflow_check="status words read after the timed region: clear",
               exact_fp32=dict(timed["fp32"], precision="fp32 (v_mfma_f32_32x32x2_f32, fp32 storage)"))
    return res, outs


def error_figures(out, ref) -> dict:
    """Max-norm relative error (the tolerance's definition) and the element-wise relative error on pixels with |ref| >= 0.05."""
    e = (out - ref).abs()
    m = ref.abs() >= 0.05
    return {"rel_max_err_vs_oracle": float(e.max() / ref.abs().max()),
            "elementwise_rel_err_on_ref_ge_0.05": {"max": float((e[m] / ref.abs()[m]).max()), "mean": float((e[m] / ref.abs()[m]).mean()),
                                                   "pixels": int(m.sum())}}


def train_gemm_work(s: int) -> dict:
    """Algorithmic work per token of the two GEMM kernel families of the training step, from the step's call list
    (lft_train_host.cuh: train_forward / train_backward).  k_lin = every Linear / conv forward and every input gradient:
    reads X (Cin floats; a 3x3 conv reads each input token once algorithmically), optionally a residual R or an activation
    mask M (Cout floats each), writes Y (Cout floats).  k_wgrad = every weight gradient dW = dY^T X: reads dY and X."""
    ss = s * s
    gt32 = 32 * (((s + 2) * (s + 2) + 31) // 32)
    lin, wg = [], []                                   # (Cin, Cout, taps, extra Cout-sized reads)
    lin += [(64, 64, 9, 0)] * 3                                                      # conv_init forward
    for _ in range(4):
        lin += [(64, 128, 1, 0), (64, 64, 1, 0), (64, 64, 1, 1), (64, 128, 1, 0), (128, 64, 1, 1)]            # ang: QK, V, out(+x), FF1, FF2(+t1)
        lin += [(64, 128, 9, 0), (128, 256, 1, 0), (128, 128, 1, 0), (128, 128, 1, 1), (128, 256, 1, 0), (256, 128, 1, 1), (128, 64, 1, 0)]   # spa
        # backward, spa then ang (input gradients)
        lin += [(64, 128, 1, 0), (128, 256, 1, 1), (256, 128, 1, 0), (128, 128, 1, 0), (128, 128, 1, 1), (256, 128, 1, 0), (128, 64, 9, 0)]
        lin += [(64, 128, 1, 1), (128, 64, 1, 0), (64, 64, 1, 0), (64, 64, 1, 1), (128, 64, 1, 0)]
        wg += [(64, 128, 1), (128, 256, 1), (256, 128, 1), (128, 128, 1), (128, 128, 1), (256, 128, 1), (128, 64, 9)]           # spa weights
        wg += [(64, 128, 1), (128, 64, 1), (64, 64, 1), (64, 64, 1), (128, 64, 1)]                                               # ang weights
    lin += [(64, 64 * ss, 1, 0), (64 * ss, gt32, 1, 0), (gt32, 64 * ss, 1, 1), (64 * ss, 64, 1, 0)]          # up-sampler fwd (2) + bwd (2)
    lin += [(64, 64, 9, 1)] * 3                                                      # conv_init backward
    wg += [(gt32, 64 * ss, 1), (64 * ss, 64, 1)] + [(64, 64, 9)] * 3
    return {"k_lin": {"bytes": sum(4 * (ci + co + x * co) for ci, co, _, x in lin), "flops": sum(2 * ci * co * t for ci, co, t, _ in lin), "calls": len(lin)},
            "k_wgrad": {"bytes": sum(4 * (a + b) for a, b, _ in wg), "flops": sum(2 * a * b * t for a, b, t in wg), "calls": len(wg)}}


def train_object(dev, math: str, with_roofline: bool = False):
    """BASELINE configs[2] shape on this GPU: A5, 2x, 32x32 LR, batch 8, fp32 tape, Adam (reference train.py:77-83)."""
    import numpy as np
    from lft_amd import _lib, train as T
    from lft_amd.params import deterministic_state, synthetic_lr
    from model import LFT
    A3, S3, B3 = 5, 2, 8
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A3, scale_factor=S3))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, S3, seed=1).items()})
    net = net.to(dev).train()
    lr = torch.from_numpy(synthetic_lr(B3, A3, 32, 32, seed=0)).to(dev)
    hr = torch.from_numpy(np.random.Generator(np.random.PCG64([2, 0])).random((B3, 1, A3 * 32 * S3, A3 * 32 * S3), dtype=np.float32)).to(dev)
    ts = T.TrainStep(net, lr=2e-4, math=math)
    for _ in range(3):
        ts.step(lr, hr)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        loss = ts.step(lr, hr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tflops = 3 * 58.85e9 * B3 * n / dt / 1e12                         # forward + 2x for backward, SURVEY.md 8d
    out = {"workload": "LFT 5x5 angRes 2xSR training step, batch=8, 32x32 LR, Adam (BASELINE configs[2] on 1 GPU: no all-reduce partner)",
           "math": math, "ms_per_step": dt / n * 1e3, "patches_per_s": B3 * n / dt, "steps": n,
           "algorithmic_tflops": tflops, "fp32_peak_tflops": PEAK_TFLOPS["fp32"], "frac_of_fp32_peak": tflops / PEAK_TFLOPS["fp32"],
           "tape_bytes": T.tape_bytes(B3, A3, 32, 32, S3), "loss": float(loss)}
    if with_roofline:
        # The step's dominant kernel family, timed live: one forward + backward on one stream with a HIP event after every kernel
        # (lft_train_step_profiled), the family's algorithmic bytes / FLOPs from the step's call list.
        ntok = B3 * A3 * A3 * 32 * 32
        tape = torch.empty(T.tape_bytes(B3, A3, 32, 32, S3), dtype=torch.uint8, device=dev)
        o = torch.empty((B3, 1, A3 * 32 * S3, A3 * 32 * S3), device=dev)
        g = torch.empty(T.grad_floats(S3), device=dev)
        dout = torch.full_like(o, 1.0 / o.numel())
        n_max = 1024
        ms = (ctypes.c_float * n_max)()
        names = (ctypes.c_char_p * n_max)()
        cnt = ctypes.c_int(0)
        acc, shapes = {}, {}
        reps = 3
        for _ in range(reps):
            _lib.check(_lib.lib().lft_train_step_profiled(T._ptr_array(ts.params), len(ts.params), lr.data_ptr(), o.data_ptr(), tape.data_ptr(), dout.data_ptr(),
                                                          g.data_ptr(), B3, A3, 32, 32, S3, T.MATH[math], torch.cuda.current_stream().cuda_stream,
                                                          n_max, ms, names, ctypes.byref(cnt)), "lft_train_step_profiled")
            for i in range(cnt.value):
                full = names[i].decode()                     # "k_lin:128>256 +R": family, then the call's shape
                k = full.split(":")[0]
                t, c = acc.get(k, (0.0, 0))
                acc[k] = (t + ms[i], c + 1)
                if full != k:
                    t, c = shapes.get(full, (0.0, 0))
                    shapes[full] = (t + ms[i], c + 1)
        fam = {k: {"ms_per_step": t / reps, "launches_per_step": c // reps} for k, (t, c) in acc.items()}
        dom = max(fam, key=lambda k: fam[k]["ms_per_step"])
        work = train_gemm_work(S3).get(dom)
        roof = {"kernel": dom, "ms_per_step_all_launches": fam[dom]["ms_per_step"], "launches_per_step": fam[dom]["launches_per_step"],
                "method": "HIP event after every kernel of one forward + backward on one stream (lft_train_step_profiled), mean of 3",
                "gpu_ms_per_step_sum": sum(v["ms_per_step"] for v in fam.values()),
                "families_ms": {k: round(v["ms_per_step"], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_per_step"])},
                "gemm_shapes_ms": {k: [round(t / reps, 3), c // reps] for k, (t, c) in sorted(shapes.items(), key=lambda kv: -kv[1][0])}}
        if work:
            gbyte, gflop = work["bytes"] * ntok / 1e9, work["flops"] * ntok / 1e9
            t_ms = fam[dom]["ms_per_step"]
            peak = PEAK_TFLOPS["fp32"] if math == "fp32" else PEAK_TFLOPS["bf16"] / 3.0          # split-bf16: three bf16 MFMAs per product
            t_mfma, t_hbm = gflop / (peak * 1e3), gbyte / HBM_PEAK_GBS
            if t_mfma >= t_hbm:
                roof.update(bound="mfma", achieved=gflop / t_ms, peak=peak, unit="TFLOP/s", frac=gflop / t_ms / peak)
            else:
                roof.update(bound="hbm", achieved=gbyte / (t_ms * 1e-3), peak=HBM_PEAK_GBS, unit="GB/s", frac=gbyte / (t_ms * 1e-3) / HBM_PEAK_GBS)
            roof["algorithmic_per_step"] = {"gbyte": gbyte, "gflop": gflop, "calls": work["calls"]}
            # HBM bytes per step of the family from the committed PMC passes of the training step (tools/collect_train_traffic.py),
            # only for the sources and the math mode they were taken on
            roof["traffic"] = None
            tp = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_train_hbm_traffic_{math}.json")
            if os.path.exists(tp):
                prof = json.load(open(tp))
                if prof.get("source_hash") == source_hash() and dom in prof["kernels"]:
                    roof["traffic"] = prof["kernels"][dom]["total"]
                    roof["traffic_source"] = os.path.relpath(tp, ROOT)
                else:
                    roof["traffic_note"] = "%s was taken on other sources (hash %s)" % (os.path.relpath(tp, ROOT), prof.get("source_hash"))
        out["roofline"] = roof
        del tape, o, g
    del ts, net
    torch.cuda.empty_cache()
    return out


def note(msg: str) -> None:
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


CONFIGS = {   # BASELINE.json configs by name: (kind, angRes, scale, LR view size, patches per GPU and step, precision)
    "cfg1": ("infer", 5, 2, 32, 1, "fp32"),    # configs[0]: single 32x32 patch, 2x, batch 1 (the reference's CPU-runnable case, here on the GPU)
    "cfg2": ("infer", 5, 4, 32, 4, "bf16"),    # configs[1]: the metric's own configuration
    "cfg3": ("train", 5, 2, 32, 8, "fp32"),    # configs[2]: training step, Adam + gradient all-reduce (8 patches per GPU: weak scaling; --batch 1 = global batch 8 on 8 GPUs)
    "cfg4": ("infer", 5, 4, 64, 8, "bf16"),    # configs[3]: 64x64 LR views, batch 64 over 8 GPUs = 8 per GPU
    "cfg5": ("infer", 9, 4, 32, 2, "bf16"),    # configs[4]: 9x9 views (81-view AngTrans), batch 2
}


def per_rank_ms(dt: float, steps: int, world: int, dist, device) -> list:
    """Every rank's own ms per step, on rank 0 (all_gather of one double)."""
    if dist is None:
        return [dt / steps * 1e3]
    t = torch.tensor([dt / steps * 1e3], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def run_training(args, rank, world, dev, dist, rehearsal):
    """BASELINE configs[2]: the data-parallel training step (reference train.py:89-107 per rank + the gradient exchange of SURVEY
    8e).  A step = forward-with-tape, L1 loss and its gradient, backward, the flat gradient buffer summed over the ranks in the
    three buckets the backward pass finishes (each all-reduce started while the next bucket's kernels run), fused Adam.  Beside
    the contract's fields the line reports `allreduce`: the three bucket all-reduces timed alone, the step timed with and without
    the exchange, and from those the share of the exchange that is hidden under the backward pass."""
    import numpy as np
    from lft_amd import dp, train as T
    from lft_amd.params import deterministic_state, synthetic_lr
    from model import LFT
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=S))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, S, seed=1).items()})
    net = net.to(dev).train()
    B = args.batch
    lr = torch.from_numpy(synthetic_lr(B, A, H, W, seed=rank)).to(dev)
    hr = torch.from_numpy(np.random.Generator(np.random.PCG64([2, rank])).random((B, 1, A * H * S, A * W * S), dtype=np.float32)).to(dev)
    math = args.train_math
    ts = T.TrainStep(net, lr=2e-4, math=math, graph=not args.no_graph)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, warm):
        for _ in range(warm):
            loss = ts.step(lr, hr)
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = ts.step(lr, hr)
        sync()
        return time.perf_counter() - t0, loss

    t_settle = time.perf_counter()                 # the same a-priori settle as the inference protocol: one second of load
    while time.perf_counter() - t_settle < SETTLE_SECONDS:
        ts.step(lr, hr)
        torch.cuda.synchronize()
    dt, loss = timed(args.steps, args.warmup)
    assert bool(torch.isfinite(loss).all())
    ranks_ms = per_rank_ms(dt, args.steps, world, dist, torch.device("cpu") if rehearsal else dev)
    dt_max = dp.barrier_max_seconds(dt, torch.device("cpu") if rehearsal else dev)
    ar = None
    if dp.dp_active(ts.group):
        # the exchange alone: the three bucket all-reduces back to back on an otherwise idle GPU
        buckets = [T.grad_bucket(S, b) for b in range(3)]
        reps = 20

        def exchange_only():
            hs = [dp.sum_gradients_start_(ts.flat_grads[f:f + c], ts.group) for f, c in buckets]
            dp.sum_gradients_finish(hs)
        ts.flat_grads.zero_()                  # repeated sums of zeros stay zero; the next step overwrites the buffer anyway
        for _ in range(3):
            exchange_only()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            exchange_only()
        sync()
        ar_ms = (time.perf_counter() - t0) / reps * 1e3
        bucket_ms = []                         # each bucket's all-reduce on its own (latency of one collective of that size)
        for f, c in buckets:
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                dp.sum_gradients_finish([dp.sum_gradients_start_(ts.flat_grads[f:f + c], ts.group)])
            sync()
            bucket_ms.append((time.perf_counter() - t0) / reps * 1e3)
        # the same step without the exchange (each rank on its own shard): what the exchange adds to a step is its exposed part
        ts.exchange = False
        dt_solo, _ = timed(max(5, args.steps // 2), 3)
        ts.exchange = True
        solo_ms = dt_solo / max(5, args.steps // 2) * 1e3
        step_ms = dt / args.steps * 1e3
        exposed = max(0.0, step_ms - solo_ms)
        ar = {"buckets_bytes": [4 * c for _, c in buckets], "bucket_ms_alone": bucket_ms, "ms_alone": ar_ms, "step_ms_with_exchange": step_ms, "step_ms_without_exchange": solo_ms,
              "ms_exposed": exposed, "hidden_frac": (1.0 - exposed / ar_ms) if ar_ms > 0 else None,
              "method": "3 bucket all-reduces timed back to back on an idle GPU; exposed = step with exchange - step without (this rank)"}
    if rank != 0:
        return
    V = A * A
    fpt = flops_per_token(S, V, mean_window(H, W))
    launches = {"k_conv0": 1, "k_conv64": 3, "k_ang": 4, "k_spa1": 4, "k_spa_b": 4, "k_up": 1, "k_assemble": 1}      # per forward
    flops_fwd = sum(fpt[k] * c for k, c in launches.items()) * V * H * W                                               # SURVEY 8d per patch
    out = {"metric": f"LF patches/sec training ({A}x{A} angRes, {H}x{W} LR, {S}xSR, Adam)", "value": world * B * args.steps / dt_max,
           "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
           "per_rank_ms": ranks_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           **({"rehearsal": f"{world} ranks share one GPU over gloo: not a scaling measurement"} if rehearsal else {}),
           "dtype": {"fp32": "f32", "bf16x3": "f32 (split-bf16 products)", "bf16x6": "f32 (weight gradients: six bf16 products, fp32-class)"}[math], "data": "synthetic",
           "config": {"workload": f"LFT {A}x{A} angRes {S}xSR training step (Adam, bucketed gradient all-reduce), batch={B} per GPU, {H}x{W} LR patches",
                      "name": args.config, "global_batch": world * B, "parallelism": f"dp{world} (flat gradient buffer summed in 3 buckets under the backward pass)",
                      "hip_graph": not args.no_graph, "algorithmic_gflop_per_patch_step": 3 * flops_fwd / 1e9},
           "allreduce": ar,
           # the exchange's figures again as flat top-level keys, so that a record which keeps only scalars / lists still shows them
           "allreduce_hidden_frac": ar["hidden_frac"] if ar else None, "allreduce_ms_alone": ar["ms_alone"] if ar else None,
           "allreduce_ms_exposed": ar["ms_exposed"] if ar else None, "allreduce_bucket_ms": ar["bucket_ms_alone"] if ar else None,
           "allreduce_bucket_bytes": ar["buckets_bytes"] if ar else None,
           "tape_bytes": T.tape_bytes(B, A, H, W, S), "loss": float(loss),
           "algorithmic_tflops": 3 * flops_fwd * world * B * args.steps / dt_max / 1e12}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS), help="BASELINE.json configuration by name (cfg2 = configs[1], the metric's own)")
    ap.add_argument("--batch", type=int, default=None, help="LF patches per GPU per step (default: the named configuration's)")
    ap.add_argument("--precision", default=None, choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--ang", type=int, default=None, help="angular resolution (default: the named configuration's)")
    ap.add_argument("--lr", type=int, default=None, help="LR view size")
    ap.add_argument("--scale", type=int, default=None, choices=[2, 4])
    ap.add_argument("--streams", type=int, default=1, help="HIP streams ONE step's batch is split over (1 = whole-batch kernels; overlap then comes from --inflight)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the host each step instead of replaying a HIP graph")
    ap.add_argument("--inflight", type=int, default=2, help="steps in flight: captured forwards replayed round-robin on this many streams (1 = strictly one after the other)")
    ap.add_argument("--train-math", default="fp32", choices=["fp32", "bf16x3", "bf16x6"], help="cfg3: GEMM arithmetic of the training kernels")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the parity_path, latency_path and train objects (A/B timing runs)")
    ap.add_argument("--settle-seconds", type=float, default=None, help="untimed load in front of --warmup (default SETTLE_SECONDS; profiling passes under counters use a short one)")
    args = ap.parse_args()
    global SETTLE_SECONDS
    if args.settle_seconds is not None:
        SETTLE_SECONDS = args.settle_seconds
    kind, c_ang, c_scale, c_lr, c_batch, c_prec = CONFIGS[args.config]
    args.ang = args.ang or c_ang
    args.scale = args.scale or c_scale
    args.lr = args.lr or c_lr
    args.batch = args.batch or c_batch
    args.precision = args.precision or c_prec
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
    # LFT_BENCH_ONE_GPU_REHEARSAL=1: all ranks share cuda:0 and talk over gloo -- exercises the spawn / barrier / max-over-ranks
    # path of `--gpus N` on a one-GPU box (tests/test_gpu_module.py); the number it prints is not a scaling measurement.
    rehearsal = os.environ.get("LFT_BENCH_ONE_GPU_REHEARSAL", "0") == "1" and world > 1
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # LFT_DP_FORCE_COLLECTIVES=1 under a launcher with ONE rank: initialise RCCL anyway, so that a one-GPU box runs the
    # collectives of the multi-rank path end to end (loop-back; its timings say nothing about scaling)
    forced = world == 1 and os.environ.get("LFT_DP_FORCE_COLLECTIVES", "0") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or forced:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    if kind == "train":
        run_training(args, rank, world, dev, dist, rehearsal)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from lft_amd.params import synthetic_lr
    net = make_net(args.precision, dev, streams=args.streams)
    lr = torch.from_numpy(synthetic_lr(args.batch, A, H, W, seed=rank)).to(dev)   # resident in HBM

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()                  # device-wide: covers the pipeline's streams

    note(f"rank {rank}/{world}: {args.config} on {dev}, precision {args.precision}, batch {args.batch}")
    step, owner = make_step(net, lr, args, args.inflight)
    dt, out = timed_protocol(step, lr, args, sync)
    net.check_status()                            # non-finite activations / outputs (an fp16 overflow) anywhere in the run: loud error
    assert bool(torch.isfinite(out).all())
    note(f"rank {rank}: {args.steps} steps in {dt:.3f} s")
    ranks_ms = per_rank_ms(dt, args.steps, world, dist, torch.device("cpu") if rehearsal else dev)
    dt = dp.barrier_max_seconds(dt, torch.device("cpu") if rehearsal else dev)          # MAX over ranks

    result = None
    if rank == 0:
        V = A * A
        ntok = args.batch * V * H * W
        fpt = flops_per_token(S, V, mean_window(H, W), q_in_part_b=args.precision != "fp32" and W % 32 == 0 and (H * W) % 128 == 0)
        with torch.no_grad():
            kb = kernel_breakdown(net, lr, reps=10)
        total_ms = sum(ms * cnt for ms, cnt in kb.values())
        note("kernel ms/launch: " + ", ".join(f"{k}={ms:.3f}x{c}" for k, (ms, c) in kb.items()))
        dom = max(kb, key=lambda k: kb[k][0] * kb[k][1])
        dom_ev_ms, dom_cnt = kb[dom]
        with torch.no_grad():
            dom_ms = kernel_time_ms(net, lr, dom)                       # back-to-back launches, no events in between
        timing = "50 back-to-back launches between two HIP events (lft_kernel_time)"
        if dom_ms is None:
            dom_ms, timing = dom_ev_ms, "HIP event after every kernel (lft_forward_profiled)"
        peak = PEAK_TFLOPS[args.precision]
        bpt = bytes_per_token(S, 4 if args.precision == "fp32" else 2)

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
            "ms_per_step": dt / args.steps * 1e3, "per_rank_ms": ranks_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            **({"rehearsal": f"{world} ranks share one GPU over gloo: not a scaling measurement"} if rehearsal else {}),
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"LFT {A}x{A} angRes {S}xSR inference, batch={args.batch} per GPU, {H}x{W} LR patches", "name": args.config,
                       "global_batch": world * args.batch, "parallelism": f"dp{world} (independent shards)",
                       "streams_per_gpu": args.streams, "hip_graph": not args.no_graph, "steps_in_flight": (args.inflight if not args.no_graph else 1),
                       "algorithmic_gflop_per_patch": flops_patch / 1e9},
            "roofline": dict(roof(dom, dom_ms), kernel=dom, **traffic_from_profile(dom, args), launch_ms=dom_ms, launch_ms_method=timing,
                             launch_ms_with_events=dom_ev_ms, launches_per_forward=dom_cnt, gpu_ms_per_forward=total_ms,
                             algorithmic_per_launch={"gflop": fpt[dom] * ntok / 1e9, "gbyte": bpt[dom] * ntok / 1e9},
                             # a fully fused SpaTrans / AngTrans block would read x and write x once
                             compulsory_gbyte_fused_block=2 * 64 * (4 if args.precision == "fp32" else 2) * ntok / 1e9,
                             kernels={k: dict(ms=round(ms, 4), n=c, **{kk: (round(vv, 3) if isinstance(vv, float) else vv)
                                                                         for kk, vv in roof(k, ms).items() if kk in ("bound", "achieved", "frac")})
                                      for k, (ms, c) in kb.items()}),
        }
        tr = result["roofline"].get("traffic")
        if tr:
            result["roofline"]["wasted_traffic_ratio_vs_kernel_contract"] = tr / (bpt[dom] * ntok)
        outs = None
        extras = world == 1 and not args.no_extras
        if extras and not args.no_graph:
            note("timing the strictly sequential path (one step in flight) ...")
            step1, owner1 = make_step(net, lr, args, 1)
            dt1, _ = timed_protocol(step1, lr, args, torch.cuda.synchronize)
            del step1, owner1
            # the same, with the batch split over two HIP streams INSIDE the captured step (module option `streams`): the two
            # half-batches' kernels fill each other's partial rounds the way two steps in flight do, without a second step
            net2 = make_net(args.precision, dev, streams=2)
            step1s, owner1s = make_step(net2, lr, args, 1)
            dt1s, _ = timed_protocol(step1s, lr, args, torch.cuda.synchronize)
            del step1s, owner1s, net2
            best = min(dt1, dt1s)
            result["latency_path"] = {"value": args.batch * args.steps / best, "unit": "patches/s", "ms_per_step": best / args.steps * 1e3,
                                      "steps_in_flight": 1, "batch_split_over_streams": 2 if dt1s <= dt1 else 1,
                                      "unsplit": {"value": args.batch * args.steps / dt1, "ms_per_step": dt1 / args.steps * 1e3},
                                      "split2": {"value": args.batch * args.steps / dt1s, "ms_per_step": dt1s / args.steps * 1e3},
                                      "note": "one captured forward replayed strictly one after the other: ms_per_step is the latency of a batch"}
        if extras and not args.no_graph:
            # cross-checks of the protocol: the same step over a region of more than a second (must agree with `value`, which is
            # timed behind SETTLE_SECONDS of load), and the round-3 protocol's burst window (100 settle steps, then --steps)
            note("timing a long region (sustained rate) and the burst window ...")
            n_long = max(args.steps, 3000)
            step2, owner2 = make_step(net, lr, args, args.inflight)
            with torch.no_grad():
                settle(step2, lr)
                t0 = time.perf_counter()
                for _ in range(n_long):
                    step2(lr)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t0
            result["sustained_path"] = {"value": args.batch * n_long / dt2, "unit": "patches/s", "steps": n_long, "ms_per_step": dt2 / n_long * 1e3,
                                        "note": "same step as the headline, timed over a region of more than a second"}
            time.sleep(0.5)                                              # let the card idle, as at the start of a process
            with torch.no_grad():
                for _ in range(100 + args.warmup):
                    step2(lr)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step2(lr)
                torch.cuda.synchronize()
                dt3 = time.perf_counter() - t0
            result["burst_path"] = {"value": args.batch * args.steps / dt3, "unit": "patches/s", "steps": args.steps, "ms_per_step": dt3 / args.steps * 1e3,
                                    "note": "round-3 protocol: 100 settle steps from idle, --warmup, --steps -- the window between clock ramp-up and power settling; NOT the headline"}
            del step2, owner2
        if extras and args.precision == "bf16":
            note("timing the fp16 and exact-fp32 parity paths ...")
            result["parity_path"], outs = parity_path(args, dev, lr, net)
        if extras and args.config == "cfg2":
            note("timing the training step (BASELINE configs[2] shape) ...")
            result["train"] = train_object(dev, "fp32", with_roofline=True)
            result["train"]["bf16x3"] = {k: v for k, v in train_object(dev, "bf16x3", with_roofline=True).items()
                                          if k in ("ms_per_step", "patches_per_s", "algorithmic_tflops", "roofline")}
            result["train"]["bf16x6"] = {k: v for k, v in train_object(dev, "bf16x6").items() if k in ("ms_per_step", "patches_per_s", "algorithmic_tflops")}
        if world == 1 and not args.no_cpu_baseline:
            note("timing the CPU oracle on host cores ...")
            result["cpu_baseline"], ref = cpu_baseline(args.cpu_seconds, lr[:1].cpu())
            if outs is not None:                                        # the oracle as the checker of all three paths (patch 0)
                result["parity_path"].update(error_figures(outs["fp16"], ref))
                result["parity_path"]["exact_fp32"].update(error_figures(outs["fp32"], ref))
                head = error_figures(outs["headline"], ref)
                result["parity_path"]["headline_path_rel_max_err_vs_oracle"] = head["rel_max_err_vs_oracle"]
                result["parity_path"]["headline_path_elementwise_rel_err_on_ref_ge_0.05"] = head["elementwise_rel_err_on_ref_ge_0.05"]
                # `value` is the configuration BASELINE names (bf16); north_star's tolerance is 1e-3: say which it is
                result["meets_tolerance"] = head["rel_max_err_vs_oracle"] <= 1e-3
                ok16 = result["parity_path"]["rel_max_err_vs_oracle"] <= 1e-3
                result["value_within_tolerance"] = (result["value"] if result["meets_tolerance"]
                                                    else result["parity_path"]["value"] if ok16 else result["parity_path"]["exact_fp32"]["value"])
                result["value_within_tolerance_path"] = (args.precision if result["meets_tolerance"] else "fp16" if ok16 else "fp32")
            if extras and args.config == "cfg2":
                # north_star's second parity form, "PSNR within 0.01 dB": a small 5x5 2x model trained here by the repo's own trainer,
                # held-out synthetic scenes through lft_amd.evaluate in each precision and through the CPU oracle (the checker);
                # per view |PSNR(path, HR) - PSNR(oracle, HR)| (tests/psnr_util.py; tests/test_gpu_psnr.py gates the same figure)
                note("PSNR-delta check against the CPU oracle on trained weights ...")
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import psnr_util as PU
                pd = {}
                for fam, fmax in (("hard_scenes", 0.25), ("smooth_scenes", 0.06)):
                    sd_t, hist = PU.train_small_model(dev, fmax=fmax)
                    r = PU.psnr_delta(dev, sd_t, PU.held_out_scenes(n=1, size=32, fmax=fmax))
                    pd[fam] = {"train_loss_first_last": [hist[0], hist[-1]], **{p: {k: round(v, 5) if isinstance(v, float) else v for k, v in d.items()} for p, d in r.items()}}
                result["psnr_delta_db"] = dict(pd, tolerance_db=0.01,
                                               note="max over views of |PSNR(path, HR) - PSNR(oracle, HR)|; A5 2x model trained in this run (20 epochs, synthetic light fields), "
                                                    "one held-out 5x5x32x32 scene per family through LFdivide / network / LFintegrate")
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
