#!/usr/bin/env python3
"""Cold-start determinism probe: in a fresh process run the forward stage by stage (per-stage C-ABI entry
points, fresh output buffer per stage), many times, and report which stage first differs from the majority.
GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from lft_amd import _lib
from lft_amd.params import deterministic_state, synthetic_lr
import gpu_util as G
A, s, B, h, w = 5, 4, 4, 32, 32
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pk = G.Packed(deterministic_state(64, s, seed=1, flavor="stress"), A, h, w, s, prec, B)
lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).to(G.DEV)
L = _lib.lib()
def staged():
    outs = {}
    feat = pk.new_act()
    _lib.check(L.lft_init_features_fwd(pk.buf.data_ptr(), lr.data_ptr(), feat.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "init")
    outs["feat"] = feat
    cur = feat
    for l in range(4):
        a = pk.new_act()
        _lib.check(L.lft_ang_block_fwd(pk.buf.data_ptr(), l, cur.data_ptr(), a.data_ptr(), *pk.dims(), G.stream()), "ang")
        outs[f"ang{l}"] = a
        b = pk.new_act()
        _lib.check(L.lft_spa_block_fwd(pk.buf.data_ptr(), l, a.data_ptr(), feat.data_ptr() if l == 3 else None, b.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "spa")
        outs[f"spa{l}"] = b
        cur = b
    out = torch.empty(B, 1, A*h*s, A*w*s, device=G.DEV)
    _lib.check(L.lft_upsample_fwd(pk.buf.data_ptr(), cur.data_ptr(), lr.data_ptr(), out.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "up")
    outs["out"] = out
    torch.cuda.synchronize()
    return outs
runs = [staged() for _ in range(reps)]
ref = runs[-1]
nbad = 0
for i, r in enumerate(runs):
    first = None
    for k in ref:
        if not torch.equal(r[k], ref[k]):
            first = k; break
    if first is not None:
        nbad += 1
        d = (r[first].float() - ref[first].float()).abs()
        idx = torch.nonzero(d > 0).cpu().numpy()
        toks = np.unique(idx[:, :4], axis=0) if idx.shape[1] == 5 else idx[:3]
        print(f"run {i}: first differing stage {first}: {len(idx)} elements, max {float(d.max()):.4f}, tokens {len(toks)} e.g. {toks[:4].tolist()} .. {toks[-2:].tolist()}", flush=True)
print(f"{nbad} of {reps} runs differ from the last run")
