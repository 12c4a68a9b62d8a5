#!/usr/bin/env python3
"""Design study (CPU, not a test): which bf16 roundings of the throughput path cost how much end-to-end error.

TEST INFRASTRUCTURE: restates the forward with a rounding hook at every GEMM operand and every inter-kernel
tensor, checks itself against the oracle with all hooks off, then switches groups of hooks on.  The numbers
decide the precision policy of the HIP kernels.
usage: python tests/diag_precision_study.py [A s h w]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lft_amd.params import deterministic_state, synthetic_lr   # noqa: E402
from oracle import lft_oracle as O                              # noqa: E402

POLICY = {}          # site class -> 'bf16' | 'x2' | 'exact'


def rnd(x, cls):
    m = POLICY.get(cls, "exact")
    if m == "exact":
        return x
    if m == "fp16":
        return x.to(torch.float16).float()
    hi = x.to(torch.bfloat16).float()
    if m == "bf16":
        return hi
    lo = (x - hi).to(torch.bfloat16).float()      # split-bf16: hi + lo carries 16 mantissa bits
    return hi + lo


def lin(a, w, blk):
    """a [.., K] activations, w [N, K] weights; operand rounding by block class."""
    return rnd(a, blk + ".act") @ rnd(w, blk + ".w").t()


def conv3(x, wgt, blk):
    B, C, V, h, w = x.shape
    return F.conv3d(rnd(x, blk + ".act"), rnd(wgt, blk + ".w"), padding=(0, 1, 1))


def mha(q_in, v_in, w_in, w_out, mask, blk):
    L, N, E = q_in.shape
    d = E // 8
    wq, wk, wv = w_in[:E], w_in[E:2 * E], w_in[2 * E:]
    q = rnd(lin(q_in, wq, blk), blk + ".qkv").reshape(L, N, 8, d).permute(1, 2, 0, 3)
    k = rnd(lin(q_in, wk, blk), blk + ".qkv").reshape(L, N, 8, d).permute(1, 2, 0, 3)
    v = rnd(lin(v_in, wv, blk), blk + ".qkv").reshape(L, N, 8, d).permute(1, 2, 0, 3)
    s = (q @ k.transpose(-1, -2)) / (d ** 0.5)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    # kernels normalise after P.V with the fp32 row sum; P itself is an MFMA operand
    o = rnd(p, blk + ".p") @ v
    o = o.permute(2, 0, 1, 3).reshape(L, N, E)
    return lin(rnd(o, blk + ".o"), w_out, blk)


def ffn(t, lw, lb, w1, w2, blk):
    n = F.layer_norm(t, (t.shape[-1],), lw, lb, 1e-5)
    return lin(F.relu(lin(n, w1, blk)), w2, blk)


def forward(sd, lr, A, s):
    skip = O.bicubic_skip(lr, A, s)
    v = O.mosaic_to_views(lr, A)
    f0 = rnd(F.conv3d(v, sd["conv_init0.0.weight"], padding=(0, 1, 1)), "store.x")
    f = f0
    for i in (0, 2, 4):
        f = F.leaky_relu(conv3(f, sd[f"conv_init.{i}.weight"], "conv"), 0.2)
        if i != 4:
            f = rnd(f, "store.x")
    x = rnd(f + f0, "store.x")
    B, C, V, h, w = x.shape
    mask = O.window_mask(h, w)
    y = x
    for l in range(4):
        p = f"altblock.{l}.ang_trans."
        t = y.permute(2, 0, 3, 4, 1).reshape(V, B * h * w, C)
        pe = O.angular_pe(V, C).view(V, 1, C)
        n = F.layer_norm(t + pe, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
        t = mha(n, t, sd[p + "attention.in_proj_weight"], sd[p + "attention.out_proj.weight"], None, "ang") + t
        t = ffn(t, sd[p + "feed_forward.0.weight"], sd[p + "feed_forward.0.bias"], sd[p + "feed_forward.1.weight"],
                sd[p + "feed_forward.4.weight"], "ang") + t
        y = rnd(t.reshape(V, B, h, w, C).permute(1, 4, 0, 2, 3), "store.x")
        p = f"altblock.{l}.spa_trans."
        img = rnd(y, "spa.act").permute(0, 2, 1, 3, 4).reshape(B * V, C, h, w)
        t = F.unfold(img, kernel_size=3, padding=1).permute(2, 0, 1) @ rnd(sd[p + "MLP.weight"], "spa.w").t()
        pe = O.spa_tokens(O.spatial_pe(h, w, C).view(1, C, 1, h, w), sd[p + "MLP.weight"])
        t = rnd(t, "store.tok")
        n = F.layer_norm(t + rnd(pe, "store.tok"), (2 * C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
        t = mha(n, t, sd[p + "attention.in_proj_weight"], sd[p + "attention.out_proj.weight"], mask, "spa") + t
        t = ffn(t, sd[p + "feed_forward.0.weight"], sd[p + "feed_forward.0.bias"], sd[p + "feed_forward.1.weight"],
                sd[p + "feed_forward.4.weight"], "spa") + t
        t = lin(t, sd[p + "linear.0.weight"].reshape(C, 2 * C), "spa")
        y = t.reshape(h, w, B, V, C).permute(2, 4, 3, 0, 1)
        if l == 3:
            y = y + x
        y = rnd(y, "store.x")
    m = O.views_to_mosaic(y, A)
    u = F.conv2d(rnd(m, "up.act"), rnd(sd["upsampling.0.weight"], "up.w"))
    u = F.pixel_shuffle(F.leaky_relu(u, 0.2), s)
    r = F.conv2d(rnd(u, "up.act"), rnd(sd["upsampling.3.weight"], "up.w"), padding=1)
    return r + skip


ALL = ["store.x", "store.tok", "conv.act", "conv.w", "ang.act", "ang.w", "ang.qkv", "ang.p", "ang.o",
       "spa.act", "spa.w", "spa.qkv", "spa.p", "spa.o", "up.act", "up.w"]


def main():
    A, s, h, w = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (5, 4, 32, 32)
    flavor = sys.argv[5] if len(sys.argv) > 5 else "default"
    torch.set_num_threads(8)
    sd = O.state_from_numpy(deterministic_state(64, s, seed=1, flavor=flavor))
    lr = torch.from_numpy(synthetic_lr(1, A, h, w, seed=0))
    with torch.no_grad():
        ref = O.forward(sd, lr, A, s)
        POLICY.clear()
        base = forward(sd, lr, A, s)
        print(f"self-check vs oracle (all exact): {float((base - ref).abs().max() / ref.abs().max()):.2e}")

        def run(name, pol):
            POLICY.clear()
            POLICY.update(pol)
            out = forward(sd, lr, A, s)
            e = (out - ref).abs()
            print(f"{name:58s} max {float(e.max() / ref.abs().max()):.2e}  rms {float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}", flush=True)

        allb = {k: "bf16" for k in ALL}
        run("everything bf16 (round-1 throughput path)", allb)
        for k in ALL:
            run(f"only {k} bf16", {k: "bf16"})
        for grp in ("store", "conv", "ang", "spa", "up"):
            run(f"all bf16 except {grp}.* exact", {k: v for k, v in allb.items() if not k.startswith(grp)})
        run("everything fp16 (operands and storage)", {k: "fp16" for k in ALL})
        run("all bf16, weights x2 (hi+lo)", {k: ("x2" if k.endswith(".w") else "bf16") for k in ALL})
        run("all bf16, activations x2, storage x2", {k: ("bf16" if k.endswith(".w") else "x2") for k in ALL})
        run("all bf16, store.x exact", {k: v for k, v in allb.items() if k != "store.x"})
        run("all bf16, store.x + store.tok exact", {k: v for k, v in allb.items() if not k.startswith("store")})
        run("all bf16, up.* x2", {k: ("x2" if k.startswith("up") else "bf16") for k in ALL})
        run("all bf16, up.* x2, store.x exact", {k: ("x2" if k.startswith("up") else "bf16") for k in ALL if k != "store.x"})
        run("all bf16, up.* + conv.* x2, store.x exact", {k: ("x2" if k[:2] in ("up", "co") else "bf16") for k in ALL if k != "store.x"})


if __name__ == "__main__":
    main()
