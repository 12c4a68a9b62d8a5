#!/usr/bin/env python3
"""Loop ONE k_conv64 launch on fixed inputs; report run-to-run differences per variant.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from lft_amd import _lib
from lft_amd.params import deterministic_state
import gpu_util as G
A, s, B, h, w = 5, 4, 4, 32, 32
prec = "bf16"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
sd_np = deterministic_state(64, s, seed=1, flavor="stress")
pk = G.Packed(sd_np, A, h, w, s, prec, B)
L = _lib.lib()
dt = torch.bfloat16
torch.manual_seed(0)
xin = (torch.randn(B, A * A, h, w, 64, device=G.DEV) * 0.3).to(dt)
res = (torch.randn(B, A * A, h, w, 64, device=G.DEV) * 0.3).to(dt)
def trial(name, which, with_res, extra, fresh_out):
    out = pk.new_act()
    base, nbad, pat = None, 0, []
    for it in range(reps):
        if fresh_out:
            out = pk.new_act()
        _lib.check(L.lft_debug_conv64(pk.buf.data_ptr(), which, with_res, xin.data_ptr(), res.data_ptr(), out.data_ptr(),
                                      *pk.dims(), extra, G.stream()), "conv")
        torch.cuda.synchronize()
        if base is None:
            base = out.clone(); continue
        if not torch.equal(out, base):
            nbad += 1
            d = (out.float() - base.float()).abs().cpu().numpy()
            idx = np.argwhere(d > 0)
            toks = np.unique(idx[:, :4], axis=0)
            if len(pat) < 3:
                pat.append((len(idx), float(d.max()), toks[:3].tolist(), toks[-1].tolist()))
    print(f"{name:44s}: {nbad} bad of {reps - 1}  {pat}", flush=True)
trial("RES=1 stream2 (as conv_init[4])", 2, 1, 0, True)
trial("RES=0 stream2", 2, 0, 0, True)
trial("RES=1 stream0", 0, 1, 0, True)
trial("RES=0 stream0", 0, 0, 0, True)
trial("RES=1 stream2, 1 WG/CU (LDS +40K)", 2, 1, 40960, True)
trial("RES=1 stream2, same out buffer", 2, 1, 0, False)
trial("RES=0 stream2, 1 WG/CU (LDS +40K)", 2, 0, 40960, True)
