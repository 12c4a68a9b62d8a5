#!/usr/bin/env python3
"""Find the pattern of rare run-to-run differences in init_features (conv64 kernels).  GPU box only."""
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
lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).to(G.DEV)
L = _lib.lib()
base = None
nbad = 0
N = B * A * A * h * w
esz = 2 if prec == "bf16" else 4
dt = torch.bfloat16 if prec == "bf16" else torch.float32
def wsview(i):   # x0, feat, xa, xb are the first four 256-aligned [N,64] buffers of the workspace
    nbytes = (N * 64 * esz + 255) // 256 * 256
    return pk.work[i * nbytes: i * nbytes + N * 64 * esz].view(dt).view(B, A * A, h, w, 64)
base_ws = None
for it in range(reps):
    act = pk.new_act()
    _lib.check(L.lft_init_features_fwd(pk.buf.data_ptr(), lr.data_ptr(), act.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "init")
    torch.cuda.synchronize()
    if base is None:
        base = act.clone(); base_ws = [wsview(i).clone() for i in (0, 2, 3)]; continue
    if not torch.equal(act, base):
        nbad += 1
        d = (act.float() - base.float()).abs().cpu().numpy()      # [B,V,h,w,C]
        idx = np.argwhere(d > 0)
        toks = np.unique(idx[:, :4], axis=0)
        chans = np.unique(idx[:, 4])
        print(f"run {it}: {len(idx)} elements differ, max {d.max():.4f}; tokens {len(toks)}: first {toks[:6].tolist()} last {toks[-3:].tolist()}; "
              f"channels {len(chans)}: {chans[:16].tolist()}{'...' if len(chans) > 16 else ''}", flush=True)
        for nm, bw, i in (("x0", base_ws[0], 0), ("conv1(xa)", base_ws[1], 2), ("conv2(xb)", base_ws[2], 3)):
            dd = (wsview(i).float() - bw.float()).abs().cpu().numpy()
            ii = np.argwhere(dd > 0)
            if len(ii):
                tk = np.unique(ii[:, :4], axis=0); ch = np.unique(ii[:, 4])
                print(f"   {nm}: {len(ii)} differ max {dd.max():.4f} tokens {tk.tolist()[:12]} channels {ch.tolist()}", flush=True)
            else:
                print(f"   {nm}: identical", flush=True)
        lin = np.unique(((idx[:,0]*25 + idx[:,1])*1024 + idx[:,2]*32 + idx[:,3]))
        print("   linear tokens (b*V+v)*1024+p:", lin[:40].tolist(), flush=True)
print(f"{nbad} bad runs of {reps - 1}")
