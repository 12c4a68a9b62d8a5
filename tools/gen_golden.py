#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference model (build container only).

The reference (``/root/reference/model/LFT.py``) is loaded from its own path, filled with the
deterministic weights of ``lft_amd.params.deterministic_state`` and run on seeded inputs; forward
hooks capture per-stage activations.  Only data (inputs' seeds, expected outputs, sub-sampled
activations and their statistics) is written -- no reference source or bytecode leaves the
container.  The GPU box regenerates inputs and weights from the seeds recorded in each fixture.

Usage:  python tools/gen_golden.py            (needs /root/reference; writes tests/golden/)
"""
from __future__ import annotations

import importlib.util
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lft_amd.params import deterministic_state, synthetic_lr  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fixture_util import stats, sub_indices  # noqa: E402  (sub-sampling shared with the tests that read the fixtures)

REF_FILE = "/root/reference/model/LFT.py"
FULL_TAPS = ("feat", "ang0", "spa0", "spa3")  # kept whole on the tiny case only


def load_reference():
    spec = importlib.util.spec_from_file_location("_reference_lft", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def tap_record(t: torch.Tensor) -> dict:
    a = t.detach().contiguous().numpy().astype(np.float32).ravel()
    return {"sub": a[sub_indices(a.size)].copy(), "stats": stats(a)}


def run_case(ref, name, A, s, B, h, w, wseed=1, iseed=0, flavor="stress", full_taps=False):
    sd = deterministic_state(64, s, seed=wseed, flavor=flavor)
    net = ref.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s)).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=iseed))
    taps = {}

    def hook(key):
        def fn(_m, _i, o):
            taps[key] = o.detach().clone()
        return fn

    hs = [net.conv_init0.register_forward_hook(hook("conv0")),
          net.conv_init.register_forward_hook(hook("conv3")),
          net.upsampling.register_forward_hook(hook("res"))]
    for l, blk in enumerate(net.altblock):
        hs.append(blk.ang_trans.register_forward_hook(hook(f"ang{l}")))
        hs.append(blk.spa_trans.register_forward_hook(hook(f"spa{l}")))
    with torch.no_grad():
        out = net(lr)
        skip = ref.interpolate(lr, A, scale_factor=s, mode="bicubic")
    for x in hs:
        x.remove()
    taps["feat"] = taps["conv3"] + taps["conv0"]
    del taps["conv3"]
    rec = {"meta": np.array([A, s, B, h, w, wseed, iseed], dtype=np.int64),
           "flavor": np.array(flavor), "out": out.numpy()}
    taps["skip"] = skip
    for k, v in taps.items():
        r = tap_record(v)
        rec[f"tap_{k}_sub"] = r["sub"]
        rec[f"tap_{k}_stats"] = r["stats"]
        if full_taps and k in FULL_TAPS:
            rec[f"tap_{k}_full"] = v.numpy()
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: out {tuple(out.shape)} rms {float(out.pow(2).mean().sqrt()):.4f} -> {os.path.getsize(path) / 1024:.0f} KiB")


def misc(ref):
    """Input-independent pieces: window masks (incl. the h!=w quirk) and position encodings."""
    rec = {}
    for (h, w) in [(8, 8), (6, 4), (4, 8)]:
        m = ref.SpaTrans.gen_mask(h, w, 5).numpy()
        rec[f"mask_{h}x{w}"] = np.isfinite(m)          # True where attention is allowed
    pe = ref.PositionEncoding(temperature=10000)
    dummy = torch.zeros(1, 64, 25, 8, 6)
    rec["pe_spa_8x6"] = pe(dummy, dim=[3, 4], token_dim=64).numpy()   # [1,64,1,8,6]
    rec["pe_ang_25"] = pe(dummy, dim=[2], token_dim=64).numpy()       # [1,64,25,1,1]
    x = torch.from_numpy(synthetic_lr(1, 3, 7, 5, seed=3))
    for s in (2, 4):
        rec[f"bicubic_a3_7x5_s{s}"] = ref.interpolate(x, 3, scale_factor=s, mode="bicubic").numpy()
    # loss
    a = torch.from_numpy(synthetic_lr(1, 2, 4, 4, seed=5))
    b = torch.from_numpy(synthetic_lr(1, 2, 4, 4, seed=6))
    rec["l1_pair"] = np.stack([a.numpy(), b.numpy()])
    rec["l1_value"] = np.array(float(ref.get_loss(None)(a, b)))
    path = os.path.join(ROOT, "tests", "golden", "misc.npz")
    np.savez_compressed(path, **rec)
    print(f"misc -> {os.path.getsize(path) / 1024:.0f} KiB")


def load_reference_tiling():
    """LFdivide / ImageExtend / LFintegrate are pure torch, but utils/utils.py also imports skimage and the argparse
    singleton.  Compile just those three function definitions out of the file (no stand-in modules)."""
    import ast
    src = open("/root/reference/utils/utils.py").read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("LFdivide", "ImageExtend", "LFintegrate")]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "reference_utils_subset", "exec"), ns)
    return ns


def tiling(ns):
    rec = {}
    for name, (A, h0, w0, patch, stride, s) in {"a2_50x41": (2, 50, 41, 32, 16, 2), "a3_32x48": (3, 32, 48, 32, 16, 4),
                                                "a2_20x23_p8": (2, 20, 23, 8, 4, 2)}.items():
        rng = np.random.Generator(np.random.PCG64([11, A, h0, w0]))
        scene = torch.from_numpy(rng.random((A * h0, A * w0), dtype=np.float32))
        sub = ns["LFdivide"](scene, A, patch, stride)
        nu, nv = sub.shape[:2]
        srp = torch.from_numpy(rng.random((nu, nv, A * patch * s, A * patch * s), dtype=np.float32))
        out = ns["LFintegrate"](srp, A, patch * s, stride * s, h0 * s, w0 * s)
        rec[f"{name}_meta"] = np.array([A, h0, w0, patch, stride, s, nu, nv], dtype=np.int64)
        rec[f"{name}_divide"] = sub.numpy()
        rec[f"{name}_integrate"] = out.numpy()
    path = os.path.join(ROOT, "tests", "golden", "tiling.npz")
    np.savez_compressed(path, **rec)
    print(f"tiling -> {os.path.getsize(path) / 1024:.0f} KiB")


def train_case(ref, name, A, s, B, h, w, wseed=1, iseed=0, tseed=2, flavor="stress", adam_steps=2):
    """One optimisation step of the reference loop (train.py:74-107): forward, L1 loss, backward, torch.optim.Adam with
    the reference's hyper-parameters.  Records the loss, every gradient (sub-sample + statistics; full for the small
    tensors) and the weights after `adam_steps` steps on the same batch."""
    from types import SimpleNamespace
    net = ref.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s))
    sd = deterministic_state(64, s, seed=wseed, flavor=flavor)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.train()
    lr_in = torch.from_numpy(synthetic_lr(B, A, h, w, seed=iseed))
    rng = np.random.Generator(np.random.PCG64([tseed, B, A, h, w, s]))
    hr = torch.from_numpy(rng.random((B, 1, A * h * s, A * w * s), dtype=np.float32))
    crit = ref.get_loss(None)
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-4, betas=(0.9, 0.999), eps=1e-08, weight_decay=0)
    rec = {"meta": np.array([A, s, B, h, w, wseed, iseed, tseed, adam_steps], dtype=np.int64), "flavor": np.array(flavor), "hr": hr.numpy()}
    losses = []
    for step in range(adam_steps):
        opt.zero_grad()
        loss = crit(net(lr_in), hr)
        loss.backward()
        losses.append(float(loss))
        if step == 0:
            for k, p in net.named_parameters():
                g = p.grad.detach().contiguous().numpy().ravel()
                rec[f"grad_{k}_sub"] = g[sub_indices(g.size)]
                rec[f"grad_{k}_stats"] = stats(g)
                if g.size <= 1024:
                    rec[f"grad_{k}_full"] = g.copy()
        opt.step()
    rec["losses"] = np.array(losses, dtype=np.float64)
    for k, p in net.named_parameters():
        v = p.detach().contiguous().numpy().ravel()
        rec[f"post_{k}_sub"] = v[sub_indices(v.size)]
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: losses {losses} -> {os.path.getsize(path) / 1024:.0f} KiB")


KINK_TAU = 5e-4           # pre-activations closer to 0 than this are listed unit by unit in the kink fixtures
KINK_CHUNK = 4096         # flat units per entry of the popcount table


def kink_tags(net):
    """(tag, module whose OUTPUT is the pre-activation of a ReLU / LeakyReLU), in the layouts the reference computes them:
    conv{0,2,4}: [B,64,V,h,w]; ang{l}: [V, B*h*w, 128]; spa{l}: [h*w, B*V, 256]; up: [B, 64 s^2, A*h, A*w]."""
    tags = [(f"conv{i}", net.conv_init[i]) for i in (0, 2, 4)]
    for l, blk in enumerate(net.altblock):
        tags.append((f"ang{l}", blk.ang_trans.feed_forward[1]))
        tags.append((f"spa{l}", blk.spa_trans.feed_forward[1]))
    tags.append(("up", net.upsampling[0]))
    return tags


def train_kink_case(ref, name, A, s, B, h, w, wseed=1, iseed=0, tseed=2, flavor="stress"):
    """The reference's gradients on an UNSCREENED input, together with what an independent implementation needs to be held
    to 1e-3 on it although the network is piecewise linear: for every ReLU / LeakyReLU of the network the branch the reference
    took at every unit -- as a SHA-256 of the sign bitmap with the near-zero units (|z| < KINK_TAU) masked out, a popcount per
    4096 units to localise a mismatch, and the near-zero units themselves (flat index in the reference's layout, signed z).
    Data only: no reference code."""
    import hashlib
    net = ref.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s))
    sd = deterministic_state(64, s, seed=wseed, flavor=flavor)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.train()
    lr_in = torch.from_numpy(synthetic_lr(B, A, h, w, seed=iseed))
    rng = np.random.Generator(np.random.PCG64([tseed, B, A, h, w, s]))
    hr = torch.from_numpy(rng.random((B, 1, A * h * s, A * w * s), dtype=np.float32))
    pre = {}
    hooks = []
    for tag, mod in kink_tags(net):
        hooks.append(mod.register_forward_hook(lambda _m, _i, o, tag=tag: pre.__setitem__(tag, o.detach().clone())))   # before the in-place activation
    crit = ref.get_loss(None)
    out = net(lr_in)
    loss = crit(out, hr)
    loss.backward()
    for x in hooks:
        x.remove()
    # hr is not stored: the tests regenerate it from tseed with the generator above
    rec = {"meta": np.array([A, s, B, h, w, wseed, iseed, tseed, 1], dtype=np.int64), "flavor": np.array(flavor), "hr_sum": np.array(float(hr.double().sum())),
           "losses": np.array([float(loss)], dtype=np.float64), "kink_tau": np.array(KINK_TAU), "kink_chunk": np.array(KINK_CHUNK)}
    rec["l1_min_abs_diff"] = np.array(float((out.detach() - hr).abs().min()))          # the loss has a kink too (sign of sr - hr)
    # ... and it is aligned like the others: the pixels with |sr - hr| < KINK_TAU one by one (flat index, signed difference), the
    # sign of every other pixel as a SHA-256 of the bitmap (round 4: at 4x three pixels sit within 2e-6 of the kink, and an
    # implementation with 1e-6 of forward error takes the other sign there -- 2.6e-3 of a LayerNorm gradient's scale)
    dd = (out.detach() - hr).contiguous().numpy().ravel()
    near = np.nonzero(np.abs(dd) < KINK_TAU)[0]
    bits = dd > 0
    bits[near] = False
    rec["l1_near_idx"] = near.astype(np.int32)
    rec["l1_near_d"] = dd[near].astype(np.float32)
    rec["l1_sha256"] = np.frombuffer(hashlib.sha256(np.packbits(bits).tobytes()).digest(), dtype=np.uint8).copy()
    for k, p in net.named_parameters():
        g = p.grad.detach().contiguous().numpy().ravel()
        rec[f"grad_{k}_sub"] = g[sub_indices(g.size)]
        rec[f"grad_{k}_stats"] = stats(g)
        if g.size <= 1024:
            rec[f"grad_{k}_full"] = g.copy()
    n_near = 0
    for tag, z in pre.items():
        z = z.contiguous().numpy().ravel()
        near = np.nonzero(np.abs(z) < KINK_TAU)[0]
        bits = z > 0
        bits[near] = False
        pad = (-bits.size) % KINK_CHUNK
        rec[f"kink_{tag}_shape"] = np.array(pre[tag].shape, dtype=np.int64)
        rec[f"kink_{tag}_near_idx"] = near.astype(np.int32)
        rec[f"kink_{tag}_near_z"] = z[near].astype(np.float32)
        rec[f"kink_{tag}_sha256"] = np.frombuffer(hashlib.sha256(np.packbits(bits).tobytes()).digest(), dtype=np.uint8).copy()
        rec[f"kink_{tag}_popcount"] = np.concatenate([bits, np.zeros(pad, dtype=bool)]).reshape(-1, KINK_CHUNK).sum(1).astype(np.uint16)
        n_near += near.size
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: loss {float(loss):.6f}, {n_near} units within {KINK_TAU} of a kink -> {os.path.getsize(path) / 1024:.0f} KiB")


# The other shape families of the luck-free gradient check (round 4), each >= 5 k tokens on an unscreened input: a 4x case (1 024
# up-sampler channels), 9 x 9 views (81-view angular attention), and h < w (the empty windows of LFT.py:155 pass no gradient).
KINK_SHAPE_CASES = [("train_kink_a5_s4_b1_16x16_seed0", 5, 4, 1, 16, 16),
                    ("train_kink_a9_s2_b1_8x8_seed0", 9, 2, 1, 8, 8),
                    ("train_kink_a3_s2_b1_16x40_seed0", 3, 2, 1, 16, 40)]


def main():
    torch.set_num_threads(8)
    ref = load_reference()
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    if "--train-only" in sys.argv:     # add the training fixtures without rewriting the others
        train_case(ref, "train_a3_s2_b2_6x6", 3, 2, 2, 6, 6)
        train_case(ref, "train_a2_s4_b1_8x5", 2, 4, 1, 8, 5)
        return
    if "--train-kink-only" in sys.argv:
        # the same shape on UNSCREENED input seeds, with the reference's branch decisions (tests/test_gpu_train.py:
        # test_gradients_on_unscreened_inputs_with_aligned_kinks)
        for iseed in (0, 1, 2):
            train_kink_case(ref, f"train_kink_a5_s2_b2_16x16_seed{iseed}", 5, 2, 2, 16, 16, iseed=iseed)
        return
    if "--train-kink-shapes-only" in sys.argv:
        for args in KINK_SHAPE_CASES:
            train_kink_case(ref, *args)
        return
    if "--train-kink-all" in sys.argv:
        for iseed in (0, 1, 2):
            train_kink_case(ref, f"train_kink_a5_s2_b2_16x16_seed{iseed}", 5, 2, 2, 16, 16, iseed=iseed)
        for args in KINK_SHAPE_CASES:
            train_kink_case(ref, *args)
        return
    if "--wide-only" in sys.argv:      # add the h < w fixture without rewriting the others
        run_case(ref, "wide_a2_s2_b1_6x12", 2, 2, 1, 6, 12, full_taps=True)
        return
    misc(ref)
    tiling(load_reference_tiling())
    run_case(ref, "tiny_a5_s2_b2_6x6", 5, 2, 2, 6, 6, full_taps=True)       # per-stage activations in full
    run_case(ref, "small_a5_s4_b1_8x8", 5, 4, 1, 8, 8)
    run_case(ref, "small_a9_s4_b1_8x8", 9, 4, 1, 8, 8)
    run_case(ref, "rect_a5_s2_b1_8x6", 5, 2, 1, 8, 6)                       # h > w: still the correct window rule
    # h < w: LFT.py:155 clamps the window's column range with h, so queries with x - 2 >= h have an EMPTY window; under the
    # torch that runs here (>= 2.5) F.scaled_dot_product_attention returns 0 for those rows (older torch: NaN)
    run_case(ref, "wide_a2_s2_b1_6x12", 2, 2, 1, 6, 12, full_taps=True)
    run_case(ref, "cfg1_a5_s2_b1_32x32", 5, 2, 1, 32, 32, flavor="default")   # BASELINE configs[0]
    train_case(ref, "train_a3_s2_b2_6x6", 3, 2, 2, 6, 6)
    train_case(ref, "train_a2_s4_b1_8x5", 2, 4, 1, 8, 5)
    run_case(ref, "cfg2_a5_s4_b1_32x32", 5, 4, 1, 32, 32, flavor="default")   # one patch of configs[1]
    for iseed in (0, 1, 2):
        train_kink_case(ref, f"train_kink_a5_s2_b2_16x16_seed{iseed}", 5, 2, 2, 16, 16, iseed=iseed)
    for args in KINK_SHAPE_CASES:
        train_kink_case(ref, *args)


if __name__ == "__main__":
    main()
