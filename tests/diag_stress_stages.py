#!/usr/bin/env python3
"""Repeat every stage of the HIP path many times on one input and report run-to-run differences and
errors against the oracle: localises races / nondeterminism.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from lft_amd import _lib
from lft_amd.params import deterministic_state, synthetic_lr
from oracle import lft_oracle as O
import gpu_util as G

def run(A, s, B, h, w, prec, reps):
    sd_np = deterministic_state(64, s, seed=1, flavor="stress")
    sd = O.state_from_numpy(sd_np)
    lr_c = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0))
    taps = {}
    ref_out = O.forward(sd, lr_c, A, s, taps)
    pk = G.Packed(sd_np, A, h, w, s, prec, B)
    lr = lr_c.to(G.DEV)
    L = _lib.lib()
    def stage(name, fn, ref):
        outs = []
        for _ in range(reps):
            if POISON:
                pk.work.fill_(0x7f)      # stale workspace data cannot help
            o = fn()
            torch.cuda.synchronize()
            outs.append(o.clone())
        base = outs[0]
        ndiff = sum(int(not torch.equal(base, o)) for o in outs[1:])
        worst = max(float((o.float() - base.float()).abs().max()) for o in outs)
        err = G.rel_max(G.from_act(base) if base.dim() == 5 else base.cpu(), ref)
        print(f"  {name:14s} runs differing from run0: {ndiff}/{reps-1}  max run-to-run |d|={worst:.3e}  rel max err vs oracle={err:.3e}", flush=True)
    def f_init():
        act = pk.new_act()
        _lib.check(L.lft_init_features_fwd(pk.buf.data_ptr(), lr.data_ptr(), act.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "init")
        return act
    def f_ang():
        act = pk.new_act()
        _lib.check(L.lft_ang_block_fwd(pk.buf.data_ptr(), 1, xin_ang.data_ptr(), act.data_ptr(), *pk.dims(), G.stream()), "ang")
        return act
    def f_spa():
        act = pk.new_act()
        _lib.check(L.lft_spa_block_fwd(pk.buf.data_ptr(), 1, xin_spa.data_ptr(), None, act.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "spa")
        return act
    def f_up():
        out = torch.empty(B, 1, A*h*s, A*w*s, device=G.DEV)
        _lib.check(L.lft_upsample_fwd(pk.buf.data_ptr(), xin_up.data_ptr(), lr.data_ptr(), out.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "up")
        return out
    def f_fwd():
        out = torch.empty(B, 1, A*h*s, A*w*s, device=G.DEV)
        _lib.check(L.lft_forward(pk.buf.data_ptr(), lr.data_ptr(), out.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "fwd")
        return out
    print(f"A{A} s{s} B{B} {h}x{w} {prec}")
    xin_ang = G.to_act(taps["spa0"], prec); xin_spa = G.to_act(taps["ang1"], prec); xin_up = G.to_act(taps["body"], prec)
    stage("init_features", f_init, taps["feat"])
    stage("ang_block1", f_ang, O.ang_block(sd, 1, G.from_act(xin_ang)))
    stage("spa_block1", f_spa, O.spa_block(sd, 1, G.from_act(xin_spa)))
    stage("upsample", f_up, O.upsample(sd, O.views_to_mosaic(G.from_act(xin_up), A), s) + taps["skip"])
    stage("forward", f_fwd, ref_out)

POISON = True
if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    POISON = reps <= 20
    precs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bf16", "fp32"]
    for prec in precs:
        if reps <= 500:
            run(5, 2, 2, 6, 6, prec, reps)
            run(5, 4, 1, 32, 32, prec, reps)
        run(5, 4, 4, 32, 32, prec, reps)
