"""Diagnostic (GPU box; test infrastructure, not collected by pytest): per-parameter gradient error of the HIP training
step against the fp32 and fp64 CPU oracle, and where ReLU pre-activations sit within rounding of a kink.
  ISEED=1 TOPN=30 python tests/diag_train_grad_report.py"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from lft_amd import _lib, train as T
from lft_amd.params import deterministic_state, param_table, synthetic_lr
from oracle import lft_oracle as O

cases = [(3, 2, 1, 7, 5)]
TOPN = int(os.environ.get('TOPN', '8'))
for (A, s, B, h, w) in cases:
    sd_np = deterministic_state(64, s, seed=1, flavor="stress")
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=int(os.environ.get('ISEED', '0'))))
    rng = np.random.Generator(np.random.PCG64([2, B, A, h, w, s]))
    hr = torch.from_numpy(rng.random((B, 1, A * h * s, A * w * s), dtype=np.float32))
    sd = O.state_from_numpy(sd_np)
    _, _, g32 = O.loss_and_grads(sd, lr, hr, A, s)
    _, _, g64 = O.loss_and_grads({k: v.double() for k, v in sd.items()}, lr.double(), hr.double(), A, s)
    names = [n for n, _, _ in param_table(64, s)]
    ps = [torch.from_numpy(sd_np[n]).cuda().contiguous() for n in names]
    out, tape = T.train_forward(ps, lr.cuda(), A, s)
    n = out.numel()
    dout = torch.empty_like(out); scr = torch.empty(1025, device="cuda")
    _lib.check(_lib.lib().lft_l1_loss(out.data_ptr(), hr.cuda().data_ptr(), n, dout.data_ptr(), 1.0 / n, scr[1024:].data_ptr(), scr.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), "l1")
    flat = T.train_backward(ps, lr.cuda(), tape, dout, A, s).cpu()
    off, rows = 0, []
    for name, p in zip(names, ps):
        k = p.numel(); got = flat[off:off + k].view(p.shape).double(); off += k
        r64 = g64[name]; sc = float(r64.abs().max())
        e = (got - r64).abs()
        rows.append((float(e.max()) / sc, float(e.pow(2).mean().sqrt() / r64.pow(2).mean().sqrt()),
                     float((g32[name].double() - r64).abs().max()) / sc, int((e > 1e-4 * sc).sum()), k, name))
    rows.sort(reverse=True)
    print(f"case {(A, s, B, h, w)}: worst (rel max vs fp64, rel rms, torch-fp32 rel max vs fp64, #elements > 1e-4, numel, name)")
    for r in rows[:TOPN]:
        print("   %.2e  %.2e  %.2e  %5d / %-7d %s" % r)
    print("   median rel max %.2e" % sorted(r[0] for r in rows)[len(rows) // 2])
    # Are the deviations ReLU kinks?  Recompute every FFN pre-activation in fp64 from OUR saved LayerNorm output and compare
    # its sign with our saved post-activation.
    V = A * A
    N = B * V * h * w
    for l in range(4):
        for blk, C, H in (("ang", 64, 128), ("spa", 128, 256)):
            m = T.tape_view(tape, f"{blk}{l}.m", B, A, h, w, s, (N, C)).cpu().double()
            hd = T.tape_view(tape, f"{blk}{l}.hdn", B, A, h, w, s, (N, H)).cpu()
            W1 = sd[f"altblock.{l}.{blk}_trans.feed_forward.1.weight"].double()
            z = m @ W1.t()
            mism = (z > 0) != (hd > 0)
            za = z.abs().flatten()
            k5 = torch.topk(za, 3, largest=False)
            print(f"   {blk}{l}: smallest |z64| {[float(v) for v in k5.values]} at unit {[int(i) % H for i in k5.indices]} token {[int(i) // H for i in k5.indices]}")
            if int(mism.sum()):
                idx = mism.nonzero()
                print(f"   {blk}{l}: {int(mism.sum())} sign mismatches; |z64| there: {[float(z[i, j].abs()) for i, j in idx[:5]]} at {idx[:5].tolist()}")
    print("   smallest |z| overall checked; done")
