"""GPU tests of the drop-in plugin surface: model.LFT.get_model(args).forward(lr) against the
fixtures captured from the real reference (tests/golden), plus batch properties at BASELINE sizes."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from lft_amd import _lib
from lft_amd.params import deterministic_state, synthetic_lr
from oracle import lft_oracle as O

pytestmark = pytest.mark.gpu
BF16_END_TO_END = 2.5e-3      # max|err| / max|ref| of the bf16 throughput path (operand rounding; see tests/diag_precision_study.py)
GOLDEN = ["tiny_a5_s2_b2_6x6", "small_a5_s4_b1_8x8", "small_a9_s4_b1_8x8", "rect_a5_s2_b1_8x6", "wide_a2_s2_b1_6x12", "cfg1_a5_s2_b1_32x32", "cfg2_a5_s4_b1_32x32"]


def make_net(A, s, wseed, flavor, precision):
    from model import LFT                      # the plugin entry point the reference drivers import
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s), precision=precision)
    sd = deterministic_state(64, s, seed=wseed, flavor=flavor)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to("cuda:0").eval()


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("precision", ["fp32", "fp16", "bf16"])
def test_forward_matches_reference_fixture(name, precision, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    A, s, B, h, w, wseed, iseed = [int(v) for v in g["meta"]]
    net = make_net(A, s, wseed, str(g["flavor"]), precision)
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=iseed)).to("cuda:0")
    with torch.no_grad():
        out = net(lr).cpu()
    ref = torch.from_numpy(g["out"])
    rel = float((out - ref).abs().max() / ref.abs().max())
    # "1e-3 relative" is applied in the max norm, max|err| / max|ref| (the output lives in [0, 1]: an absolute bound).  The
    # element-wise relative error is reported beside it on the pixels that are not dark (|ref| >= 0.05) and gated loosely.
    m = ref.abs() >= 0.05
    ew = (out - ref).abs()[m] / ref.abs()[m]
    print(f"{name} [{precision}] rel max err {rel:.3e}  element-wise rel err on |ref|>=0.05: max {float(ew.max()):.3e} mean {float(ew.mean()):.3e} "
          f"({int(m.sum())} px)  psnr(ours, reference) {O.psnr(out, ref):.2f} dB")
    assert out.shape == ref.shape
    # element-wise gates at the observed level (round 4, seven fixtures: fp32 5.6e-6, fp16 3.6e-3, bf16 3.2e-2 at the worst pixel)
    assert float(ew.max()) <= {"fp32": 2e-5, "fp16": 6e-3, "bf16": 4.5e-2}[precision]
    # north_star: 1e-3 relative -- met by the fp32 path (observed 3e-7) and by the fp16 path (11 significant bits, ~2e-4).  The bf16
    # path rounds every MFMA operand and every inter-kernel tensor to 8 significant bits: 1.5e-3 .. 1.9e-3 observed, which does NOT
    # meet 1e-3; its gate only pins that level.
    assert rel <= (BF16_END_TO_END if precision == "bf16" else 1e-3)
    assert O.psnr(out, ref) >= {"fp32": 100.0, "fp16": 70.0, "bf16": 55.0}[precision]


def test_batch_independence_cfg2():
    """BASELINE configs[1] shape (A5, 4x, B=4, 32x32): patches never interact, so each output must equal the
    B=1 result bit-for-bit (same kernels, same tiles), and a permuted batch must give permuted outputs."""
    A, s, B, h, w = 5, 4, 4, 32, 32
    net = make_net(A, s, 1, "default", "bf16")
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).to("cuda:0")
    with torch.no_grad():
        full = net(lr)
        single = torch.cat([net(lr[i:i + 1]) for i in range(B)])
        perm = net(lr[[2, 0, 3, 1]])
    assert torch.equal(full, single)
    assert torch.equal(perm, full[[2, 0, 3, 1]])
    assert not torch.isnan(full).any()


@pytest.mark.parametrize("A,s,B,h,w", [(5, 4, 3, 64, 64), (9, 4, 2, 32, 32)], ids=["cfg4_64x64", "cfg5_9x9"])
def test_full_size_properties_cfg4_cfg5(A, s, B, h, w):
    """BASELINE configs[3] (64x64 LR views) and configs[4] (9x9 views) at full view size, where the CPU oracle is too slow
    to be the checker: size-independent properties instead -- batch independence and permutation equivariance
    (bit-exact), the bf16 and fp16 paths agreeing with the fp32 path to their accuracy (fp16: north_star's 1e-3), and the network reducing to its bicubic skip when
    the last convolution's weights are zero (the skip path is checked against the oracle at small sizes)."""
    net = make_net(A, s, 1, "default", "bf16")
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=4)).to("cuda:0")
    perm = list(range(B))[::-1]
    with torch.no_grad():
        full = net(lr)
        single = torch.cat([net(lr[i:i + 1]) for i in range(B)])
        rev = net(lr[perm])
    assert full.shape == (B, 1, A * h * s, A * w * s) and not torch.isnan(full).any()
    assert torch.equal(full, single) and torch.equal(rev, full[perm])
    ref32 = make_net(A, s, 1, "default", "fp32")
    with torch.no_grad():
        y32 = ref32(lr[:1])
    rel = float((full[:1] - y32).abs().max() / y32.abs().max())
    print(f"A{A} {h}x{w}: bf16 vs fp32 path rel max {rel:.2e}")
    assert rel <= BF16_END_TO_END
    with torch.no_grad():
        y16 = make_net(A, s, 1, "default", "fp16")(lr[:1])
    rel16 = float((y16 - y32).abs().max() / y32.abs().max())
    print(f"A{A} {h}x{w}: fp16 vs fp32 path rel max {rel16:.2e}")
    assert rel16 <= 1e-3                                   # north_star tolerance at BASELINE's full sizes
    with torch.no_grad():
        dict(ref32.named_parameters())["upsampling.3.weight"].zero_()
        skip_only = ref32(lr[:1])
    from lft_amd import _lib
    bic = torch.empty_like(skip_only)
    _lib.check(_lib.lib().lft_bicubic_fwd(lr[:1].contiguous().data_ptr(), bic.data_ptr(), 1, A, h, w, s,
                                          torch.cuda.current_stream().cuda_stream), "lft_bicubic_fwd")
    torch.cuda.synchronize()
    assert torch.equal(skip_only, bic)


def test_loss_and_errors():
    from model import LFT
    a = torch.rand(2, 1, 8, 8, device="cuda:0")
    b = torch.rand(2, 1, 8, 8, device="cuda:0")
    assert abs(float(LFT.get_loss(None)(a, b)) - float((a - b).abs().mean())) < 1e-7
    net = make_net(5, 2, 1, "default", "fp32")
    with pytest.raises(_lib.LftError):
        net(torch.zeros(1, 1, 40, 40))                      # CPU tensor: no fallback
    with pytest.raises(ValueError):
        with torch.no_grad():
            net(torch.zeros(1, 1, 41, 40, device="cuda:0"))   # not divisible by angRes
    net12 = make_net(12, 2, 1, "default", "fp32")
    with pytest.raises(_lib.LftError):
        with torch.no_grad():
            net12(torch.zeros(1, 1, 48, 48, device="cuda:0"))  # 144 views: beyond this build (<= 128)


def test_streams_and_graph_match_single_stream():
    """Splitting the batch over HIP streams and replaying a captured HIP graph must not change a single bit:
    every patch goes through the same kernels with the same tiles."""
    from lft_amd.module import GraphedForward
    A, s, B, h, w = 5, 4, 4, 32, 32
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).to("cuda:0")
    outs = {}
    for streams in (1, 2, 3):
        from model import LFT
        net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s), precision="bf16", streams=streams)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, s, seed=1).items()})
        net = net.to("cuda:0").eval()
        with torch.no_grad():
            outs[streams] = net(lr).clone()
            if streams == 2:
                g = GraphedForward(net, lr)
                outs["graph"] = g(lr).clone()
                lr2 = lr.flip(0).contiguous()
                outs["graph_flipped"] = g(lr2).clone().flip(0)
    torch.cuda.synchronize()
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[1], outs[3])
    assert torch.equal(outs[1], outs["graph"]) and torch.equal(outs[1], outs["graph_flipped"])


def test_pipelined_forward_matches_plain_forward():
    """PipelinedForward (several captured forwards in flight on their own streams / buffers) returns, for every step, exactly
    what the plain forward returns for that step's input."""
    from lft_amd.module import PipelinedForward
    A, s, B, h, w = 5, 4, 2, 16, 16
    net = make_net(A, s, 1, "default", "bf16")
    ins = [torch.from_numpy(synthetic_lr(B, A, h, w, seed=10 + i)).to("cuda:0") for i in range(5)]
    with torch.no_grad():
        ref = [net(x).clone() for x in ins]
        pipe = PipelinedForward(net, ins[0], depth=2)
        outs = []
        for i, x in enumerate(ins):
            y = pipe(x)
            if i % 2 == 1 or i == len(ins) - 1:       # results stay valid for `depth` calls: collect every second call
                pipe.sync()
                torch.cuda.synchronize()
                outs.append((i, y.clone()))
                if i % 2 == 1:
                    outs.append((i - 1, pipe.graphs[(i - 1) % 2].static_out.clone()))
    for i, y in outs:
        assert torch.equal(y, ref[i]), i
    assert sorted(i for i, _ in outs) == [0, 1, 2, 3, 4]


def test_graph_refuses_replay_after_weights_change():
    """A captured forward has the packed-weight buffer's address baked in; once the weights are re-packed, dropped or merely
    changed in place (optimizer step, load_state_dict) a replay must fail loudly instead of reading freed memory or silently
    using the old weights -- also when NO eager forward ran in between (nothing has re-packed yet: only the parameters'
    version counters tell)."""
    from lft_amd.module import GraphedForward, PipelinedForward
    A, s, B, h, w = 2, 2, 1, 8, 8
    net = make_net(A, s, 1, "default", "bf16")
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).to("cuda:0")
    with torch.no_grad():
        g = GraphedForward(net, lr)
        a = g(lr).clone()
        assert torch.equal(a, net(lr))
        assert torch.equal(g(lr), a)                  # an eager forward with unchanged weights does not invalidate the graph
        next(net.parameters()).mul_(1.5)              # in-place change, no forward afterwards: net._packed is still the captured buffer
        with pytest.raises(RuntimeError):
            g(lr)
        b = net(lr)                                   # re-packs into a new buffer
        assert not torch.equal(a, b)
        with pytest.raises(RuntimeError):
            g(lr)
        g2 = GraphedForward(net, lr)
        assert torch.equal(g2(lr), b)
        pipe = PipelinedForward(net, lr, depth=2)
        pipe(lr); pipe.sync()
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        net.load_state_dict(sd)                       # same values, but copied in place: the versions moved
        with pytest.raises(RuntimeError):
            g2(lr)
        with pytest.raises(RuntimeError):
            pipe(lr)


def test_fp16_overflow_is_a_loud_error():
    """The fp16 path's range ends at 65504.  Scale one FFN weight up until an activation leaves it: from then on the forward
    must raise (the kernels' sticky status word, read after every eager fp16 forward) -- at no gain may it return inf / NaN or a
    wrong image silently.  The exact-fp32 path of the same weights stays finite and is the checker."""
    from lft_amd.module import GraphedForward
    A, s, B, h, w = 5, 2, 1, 8, 8
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).to("cuda:0")
    n16 = make_net(A, s, 1, "default", "fp16")
    n32 = make_net(A, s, 1, "default", "fp32")
    assert n16.check_finite and not n32.check_finite
    key = "altblock.1.spa_trans.feed_forward.1.weight"          # FFN hidden layer: the largest activations of the network
    base = {k: v.clone() for k, v in n16.state_dict().items()}
    tripped_at, gain = None, 1.0
    with torch.no_grad():
        for _ in range(14):
            for net in (n16, n32):
                sd = {k: v.clone() for k, v in base.items()}
                sd[key] = sd[key] * gain
                net.load_state_dict(sd)
            ref = n32(lr)
            assert bool(torch.isfinite(ref).all())
            try:
                out = n16(lr)
            except _lib.LftError as e:
                assert "fp16 range" in str(e) and "65504" in str(e)
                tripped_at = gain
                break
            assert bool(torch.isfinite(out).all()), gain
            rel = float((out - ref).abs().max() / ref.abs().max())
            print(f"gain {gain:g}: fp16 vs fp32 rel max err {rel:.2e}")
            assert rel <= 2e-2, (gain, rel)                        # not overflowed: still the right image
            gain *= 4.0
    assert tripped_at is not None and tripped_at > 1.0, "the weights were never large enough to leave the fp16 range"
    print(f"fp16 overflow reported at gain {tripped_at:g}")
    with torch.no_grad():
        # the status word was cleared by the failed check: a captured graph of the same (bad) weights replays silently but its
        # check() raises; the C-ABI entry points say the same thing
        n16.check_finite = False
        g = GraphedForward(n16, lr)
        g(lr)
        with pytest.raises(_lib.LftError):
            g.check()
        n16.check_status()                                       # cleared again: no new forward, no error
        # the fp32 and bf16 paths take the same weights without overflow (range 3e38)
        nb = make_net(A, s, 1, "default", "bf16")
        sd = {k: v.clone() for k, v in base.items()}
        sd[key] = sd[key] * tripped_at
        nb.load_state_dict(sd)
        assert bool(torch.isfinite(nb(lr)).all())
        nb.check_status()
        # good weights again: the fp16 path works and reports nothing
        n16.load_state_dict(base)
        n16.check_finite = True
        out = n16(lr)
        assert bool(torch.isfinite(out).all())
        # non-finite INPUT data is reported too, in every precision
        bad = lr.clone()
        bad[0, 0, 3, 3] = float("nan")
        nb.load_state_dict(base)
        nb(bad)
        with pytest.raises(_lib.LftError):
            nb.check_status()


def test_status_word_through_the_c_abi():
    """lft_status_reset / lft_status_read on a caller-owned workspace: clear after a clean forward, LFT_STATUS_NONFINITE (1001)
    after a forward on an input holding inf, clear again after a reset."""
    import ctypes
    A, s, B, h, w = 2, 2, 1, 8, 8
    prec = _lib.PREC_F16
    sd = deterministic_state(64, s, seed=1)
    from lft_amd.params import param_table
    params = [torch.from_numpy(sd[n]).to("cuda:0") for n, _, _ in param_table(64, s)]
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    packed = torch.empty(_lib.packed_bytes(A, h, w, s, prec), dtype=torch.uint8, device="cuda:0")
    arr = (ctypes.c_void_p * len(params))(*[p.data_ptr() for p in params])
    _lib.check(L.lft_pack_weights(arr, len(params), packed.data_ptr(), A, h, w, s, prec, st), "pack")
    work = torch.empty(_lib.workspace_bytes(B, A, h, w, s, prec), dtype=torch.uint8, device="cuda:0")
    work.fill_(0xFF)                                              # garbage, as a fresh allocation may hold
    dims = (B, A, h, w, s, prec)
    _lib.check(L.lft_status_reset(work.data_ptr(), *dims, st), "reset")
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).to("cuda:0")
    out = torch.empty(B, 1, A * h * s, A * w * s, device="cuda:0")
    flags = ctypes.c_uint(7)
    _lib.check(L.lft_forward(packed.data_ptr(), lr.data_ptr(), out.data_ptr(), work.data_ptr(), *dims, st), "forward")
    assert L.lft_status_read(work.data_ptr(), *dims, st, ctypes.byref(flags)) == 0 and flags.value == 0
    lr[0, 0, 5, 5] = float("inf")
    _lib.check(L.lft_forward(packed.data_ptr(), lr.data_ptr(), out.data_ptr(), work.data_ptr(), *dims, st), "forward")
    assert L.lft_status_read(work.data_ptr(), *dims, st, ctypes.byref(flags)) == _lib.STATUS_NONFINITE and flags.value == 1
    assert b"non-finite" in L.lft_last_error()
    assert L.lft_status_read(work.data_ptr(), *dims, st, None) == _lib.STATUS_NONFINITE      # sticky until reset
    _lib.check(L.lft_status_reset(work.data_ptr(), *dims, st), "reset")
    assert L.lft_status_read(work.data_ptr(), *dims, st, ctypes.byref(flags)) == 0 and flags.value == 0
    assert L.lft_status_read(None, *dims, st, None) == -1


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` run plainly must start its two ranks itself (torch.distributed.run as a child of a parent
    that never touched the GPU), time with the barrier / max-over-ranks protocol and print ONE JSON line for the job.  On the
    one-GPU test box the ranks share cuda:0 over gloo (LFT_BENCH_ONE_GPU_REHEARSAL); on a node the same path uses RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LFT_BENCH_ONE_GPU_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 10 and j["config"]["global_batch"] == 8 and j["scaling"] == "weak"
    assert j["value"] > 0 and "roofline" in j and "rehearsal" in j
    assert abs(j["value"] - 2 * 4 * 10 / (j["ms_per_step"] * 10 / 1e3)) < 1e-6 * j["value"]
    assert len(j["per_rank_ms"]) == 2 and abs(max(j["per_rank_ms"]) - j["ms_per_step"]) < 1e-6 * j["ms_per_step"]
    assert j["config"]["name"] == "cfg2"
    # the training configuration by name: 2 ranks, bucketed gradient exchange (gloo through the host in this rehearsal)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "cfg3", "--batch", "1", "--steps", "4", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["name"] == "cfg3" and j["config"]["global_batch"] == 2 and "training" in j["metric"]
    ar = j["allreduce"]
    assert ar and sum(ar["buckets_bytes"]) == 4 * 1_114_240 and ar["ms_alone"] > 0 and ar["step_ms_with_exchange"] > 0
    assert len(j["per_rank_ms"]) == 2 and j["value"] > 0 and j["loss"] > 0
