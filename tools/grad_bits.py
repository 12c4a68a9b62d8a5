"""Bit-level regression check of the training kernels: output and flat gradients of one forward + backward pass (cfg 3 shape,
B = 8, both math modes) -> a file; run once per library (LFT_LIB_PATH=ab_so/liblft_ref.so for the reference build) and give the
second run the first one's file: it prints whether forward and gradients are bit-identical.  For changes that must not change
a single bit (re-ordered loads, re-used operands, new addressing): GPU box.

  LFT_LIB_PATH=$PWD/ab_so/liblft_ref.so python tools/grad_bits.py /tmp/ref.pt && python tools/grad_bits.py /tmp/new.pt /tmp/ref.pt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lft_amd import train as T
from lft_amd.params import deterministic_state, param_table, synthetic_lr
A, s, B, h, w = 5, 2, 8, 32, 32
dev = torch.device("cuda", 0)
sd = deterministic_state(64, s, seed=1, flavor="stress")
ps = [torch.from_numpy(sd[n]).to(dev).contiguous() for n, _, _ in param_table(64, s)]
lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).to(dev)
g = torch.Generator(device="cpu").manual_seed(5)
dout = torch.randn(B, 1, A * h * s, A * w * s, generator=g).to(dev) * 1e-3
res = {}
for math in ("fp32", "bf16x3"):
    out, tape = T.train_forward(ps, lr, A, s, math=math)
    res[math] = (out.cpu(), T.train_backward(ps, lr, tape, dout, A, s, math=math).cpu())
torch.save(res, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for math in res:
        print(math, "forward equal:", bool(torch.equal(res[math][0], ref[math][0])), " gradients equal:", bool(torch.equal(res[math][1], ref[math][1])),
              " max |diff|:", float((res[math][1] - ref[math][1]).abs().max()))
