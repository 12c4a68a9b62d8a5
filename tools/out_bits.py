#!/usr/bin/env python3
"""Bit-level regression of the inference output against another build: the forward of BASELINE configs[1] (and a ragged 2x shape that
takes the kernels' general paths) under each library given, each in its own process; prints whether the outputs are bit-identical.
  tools/out_bits.py ab_so/liblft_base0.so lft_amd/liblft_hip.so"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    from types import SimpleNamespace
    import torch
    from model import LFT
    from lft_amd.params import deterministic_state, synthetic_lr
    for A, s, B, h, w in ((5, 4, 2, 32, 32), (3, 2, 1, 13, 22), (2, 4, 1, 6, 12)):
        for prec in ("bf16", "fp16", "fp32"):
            net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s), precision=prec, streams=1).cuda().eval()
            net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, s, seed=1).items()})
            with torch.no_grad():
                out = net(torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).cuda())
            print(f"A{A} s{s} B{B} {h}x{w} {prec}", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        raise SystemExit(0)
    outs = []
    for lib in sys.argv[1:]:
        env = dict(os.environ, LFT_LIB_PATH=os.path.abspath(lib), LFT_AB_ANY_ABI="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, cwd=ROOT)
        if r.returncode:
            raise SystemExit(f"{lib} failed:\n{r.stderr[-2000:]}")
        outs.append([l for l in r.stdout.splitlines() if l.startswith("A")])
        print(lib)
        print("\n".join("   " + l for l in outs[-1]))
    same = all(o == outs[0] for o in outs[1:])
    print("bit-identical" if same else "DIFFERENT")
    raise SystemExit(0 if same else 1)
