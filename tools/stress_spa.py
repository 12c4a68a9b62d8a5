#!/usr/bin/env python3
"""Pattern of rare run-to-run differences in the spatial block (k_spa1 / k_spa_attn / k_spa2).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from lft_amd import _lib
from lft_amd.params import deterministic_state, synthetic_lr
import gpu_util as G
A, s, B, h, w = 5, 4, 4, 32, 32
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
sd_np = deterministic_state(64, s, seed=1, flavor="stress")
pk = G.Packed(sd_np, A, h, w, s, prec, B)
L = _lib.lib()
N = B * A * A * h * w
esz = 2 if prec == "bf16" else 4
dt = torch.bfloat16 if prec == "bf16" else torch.float32
xin = (torch.randn(B, A * A, h, w, 64, device=G.DEV) * 0.3).to(dt)
def al(x): return (x + 255) // 256 * 256
off64, off128 = al(N * 64 * esz), al(N * 128 * esz)
def wsview(i):   # tok, q, k, v, o follow the four [N,64] buffers
    o = 4 * off64 + i * off128
    return pk.work[o: o + N * 128 * esz].view(dt).view(B, A * A, h, w, 128)
names = ["tok", "q", "k", "v", "o"]
base = None
nbad = 0
for it in range(reps):
    act = pk.new_act()
    _lib.check(L.lft_spa_block_fwd(pk.buf.data_ptr(), 1, xin.data_ptr(), None, act.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "spa")
    torch.cuda.synchronize()
    if base is None:
        base = act.clone(); base_ws = [wsview(i).clone() for i in range(5)]; continue
    if not torch.equal(act, base):
        nbad += 1
        d = (act.float() - base.float()).abs().cpu().numpy()
        idx = np.argwhere(d > 0)
        toks = np.unique(idx[:, :4], axis=0)
        print(f"run {it}: out: {len(idx)} differ max {d.max():.4f}; tokens {len(toks)}: {toks[:10].tolist()}", flush=True)
        for i, nm in enumerate(names):
            dd = (wsview(i).float() - base_ws[i].float()).abs().cpu().numpy()
            ii = np.argwhere(dd > 0)
            if len(ii):
                tk = np.unique(ii[:, :4], axis=0); ch = np.unique(ii[:, 4])
                print(f"   {nm}: {len(ii)} differ max {dd.max():.4f} tokens {len(tk)} {tk.tolist()[:10]} channels {len(ch)} {ch.tolist()[:20]}", flush=True)
            else:
                print(f"   {nm}: identical", flush=True)
print(f"{nbad} bad runs of {reps - 1}")
