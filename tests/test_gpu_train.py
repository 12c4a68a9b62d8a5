"""GPU parity of the training step (reference train.py:89-107 through PyTorch autograd): the fp32 forward-with-tape and
backward kernels, called through the C ABI, against (a) autograd over the CPU oracle on the same seeded inputs and
(b) fixtures produced by the REAL reference network + torch.optim.Adam (tests/golden/train_*.npz).

Tolerance (BASELINE.json north_star: 1e-3 relative fp32): every saved activation and every one of the 78 gradients
within 1e-3 * max|ref| of that tensor (observed ~1e-6 / ~1e-5)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from lft_amd import _lib, train as T
from lft_amd.params import deterministic_state, param_table, synthetic_lr
from oracle import lft_oracle as O
from fixture_util import KINK_TAGS, kink_compare, sub_indices

import gpu_util as G

pytestmark = pytest.mark.gpu
TOL = 1e-3

CASES = [(3, 2, 2, 6, 6), (2, 4, 1, 8, 5), (5, 2, 1, 8, 8), (2, 2, 1, 5, 9),      # h < w: empty windows pass no gradient
         (9, 2, 1, 4, 4), (3, 2, 1, 7, 5), (5, 2, 1, 16, 16)]                    # 81 views; ragged 32-token tiles (315 tokens); 6400 tokens
# The network is piecewise linear (ReLU, LeakyReLU, |.|): when a pre-activation lies within fp32 rounding of 0, two
# correct fp32 implementations take different branches and -- at small token counts, where one token is 1/300 of the
# batch -- whole gradient tensors move by ~1e-3 (torch-fp32 against torch-fp64 shows the same; tests/diag_train_grad_report.py).
# So gradients are checked two ways, neither of which depends on where an input happens to fall:
#   * exactly, on every small case: the oracle's autograd is told to take the branches OUR forward took (O.branch_masks, from our
#     tape) and to start from OUR d loss / d out -- every gradient, tolerance 1e-3 (observed ~1e-5);
#   * end to end against the REAL reference's autograd with the reference's branch decisions carried in the fixture
#     (test_gradients_on_unscreened_inputs_with_aligned_kinks): every shape family -- 2x, 4x, 9 x 9 views, h < w -- at >= 5 k
#     tokens on unscreened inputs.
# (Rounds 1-3 also compared the small cases end to end with the untouched oracle, which needed a hand-kept list of inputs that
# do not sit on a kink; that test and its lists are gone.)


def make_inputs(A, s, B, h, w):
    sd_np = deterministic_state(64, s, seed=1, flavor="stress")
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0))
    rng = np.random.Generator(np.random.PCG64([2, B, A, h, w, s]))
    hr = torch.from_numpy(rng.random((B, 1, A * h * s, A * w * s), dtype=np.float32))
    return sd_np, lr, hr


MATHS = ["fp32", "bf16x3", "bf16x6"]   # exact fp32 MFMA / split-bf16 products / fp32-class six-product weight gradients: same tolerance for all


def build_case(param):
    (A, s, B, h, w), math = param
    sd_np, lr, hr = make_inputs(A, s, B, h, w)
    sd = O.state_from_numpy(sd_np)
    taps = {}
    out_ref = O.forward(sd, lr, A, s, taps)
    loss_ref, _, grads_ref = O.loss_and_grads(sd, lr, hr, A, s)
    names = [n for n, _, _ in param_table(64, s)]
    ps = [torch.from_numpy(sd_np[n]).to(G.DEV).contiguous() for n in names]
    lr_d, hr_d = lr.to(G.DEV), hr.to(G.DEV)
    out, tape = T.train_forward(ps, lr_d, A, s, math=math)
    n = out.numel()
    dout = torch.empty_like(out)
    scratch = torch.empty(1025, device=G.DEV)
    _lib.check(_lib.lib().lft_l1_loss(out.data_ptr(), hr_d.data_ptr(), n, dout.data_ptr(), 1.0 / n, scratch[1024:].data_ptr(),
                                      scratch.data_ptr(), G.stream()), "lft_l1_loss")
    flat = T.train_backward(ps, lr_d, tape, dout, A, s, math=math)
    torch.cuda.synchronize()
    return dict(math=math, sd=sd, A=A, s=s, B=B, h=h, w=w, names=names, ps=ps, lr=lr_d, hr=hr_d, out=out, tape=tape, flat=flat, loss=float(scratch[1024]),
                taps=taps, out_ref=out_ref, loss_ref=float(loss_ref), grads_ref=grads_ref, dout=dout)


_CASE_ID = lambda cm: "A%d_s%d_B%d_%dx%d" % cm[0] + "_" + cm[1]      # noqa: E731


@pytest.fixture(scope="module", params=[(c, m) for c in CASES for m in MATHS], ids=_CASE_ID)
def case(request):
    return build_case(request.param)


# the per-block backward tests: three shapes (2x, 4x with h != w, 5 x 5 views) x the three math modes -- every backward kernel runs in each
BLOCK_CASES = [(3, 2, 2, 6, 6), (2, 4, 1, 8, 5), (5, 2, 1, 8, 8)]


@pytest.fixture(scope="module", params=[(c, m) for c in BLOCK_CASES for m in MATHS], ids=_CASE_ID)
def block_case(request):
    return build_case(request.param)


def test_forward_tape_matches_oracle(case):
    A, s, B, h, w = case["A"], case["s"], case["B"], case["h"], case["w"]
    V = A * A
    for name, key in [("feat", "feat")] + [(f"ang{l}.y", f"ang{l}") for l in range(4)] + [(f"spa{l}.y", f"spa{l}") for l in range(3)] + [("body", "body")]:
        got = T.tape_view(case["tape"], name, B, A, h, w, s, (B, V, h, w, 64)).cpu().permute(0, 4, 1, 2, 3)
        ref = case["taps"][key]
        assert not torch.isnan(got).any(), name
        assert G.rel_max(got, ref) <= TOL, f"{name}: " + G.err_report(got.contiguous(), ref)
    got, ref = case["out"].cpu(), case["out_ref"]
    print(f"train forward [{case['math']}]: " + G.err_report(got, ref))
    assert G.rel_max(got, ref) <= TOL
    assert abs(case["loss"] - case["loss_ref"]) <= 1e-5 * max(1.0, abs(case["loss_ref"]))


def our_branches(case):
    """ReLU / LeakyReLU decisions of OUR forward, from the tape, in the layouts of the oracle's pre-activations."""
    A, s, B, h, w = case["A"], case["s"], case["B"], case["h"], case["w"]
    V, ss = A * A, s * s
    tv = lambda name, C: T.tape_view(case["tape"], name, B, A, h, w, s, (B, V, h, w, C)).cpu() > 0   # noqa: E731
    m = {}
    for i, name in zip((0, 2, 4), ("c1", "c2", "c3")):
        m[f"conv{i}"] = tv(name, 64).permute(0, 4, 1, 2, 3)                                        # [B,64,V,h,w]
    for l in range(4):
        m[f"ang{l}"] = tv(f"ang{l}.hdn", 128).permute(1, 0, 2, 3, 4).reshape(V, B * h * w, 128)     # 'a (b h w) c'
        m[f"spa{l}"] = tv(f"spa{l}.hdn", 256).permute(2, 3, 0, 1, 4).reshape(h * w, B * V, 256)     # '(h w) (b a) c'
    m["up"] = O.views_to_mosaic(tv("act", 64 * ss).permute(0, 4, 1, 2, 3), A)                       # [B,64ss,A*h,A*w]
    return m


def compare_all(case, grads_ref, what):
    off, worst = 0, (0.0, "")
    for name, p in zip(case["names"], case["ps"]):
        k = p.numel()
        got = case["flat"][off:off + k].cpu().view(p.shape)
        ref = grads_ref[name]
        off += k
        assert not torch.isnan(got).any(), name
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        rel = err / max(scale, 1e-12)
        if rel > worst[0]:
            worst = (rel, name)
        assert err <= TOL * scale + 1e-10, f"{name}: " + G.err_report(got, ref)
    assert off == case["flat"].numel()
    print(f"{what} [{case['math']}]: worst gradient rel err {worst[0]:.2e} ({worst[1]})")


def test_all_78_gradients_exact_given_our_branches(case):
    O.branch_masks = our_branches(case)
    try:
        ref = O.param_grads(case["sd"], case["lr"].cpu(), case["A"], case["s"], case["dout"].cpu())
    finally:
        O.branch_masks = None
    compare_all(case, ref, "backward given our branches")


BLOCKS = [("upsample", _lib.BLOCK_UPSAMPLE, 0), ("spa", _lib.BLOCK_SPA, 1), ("spa", _lib.BLOCK_SPA, 3), ("ang", _lib.BLOCK_ANG, 0),
          ("ang", _lib.BLOCK_ANG, 2), ("init", _lib.BLOCK_INIT, 0)]


@pytest.mark.parametrize("kind,block,layer", BLOCKS, ids=[f"{k}{l}" for k, _, l in BLOCKS])
def test_block_backward_matches_oracle_autograd(block_case, kind, block, layer):
    case = block_case
    """lft_train_block_backward (the `_bwd` counterpart of the per-stage forward entry points): ONE block's backward kernels against
    autograd over the oracle's function of that block alone -- its input is our tape's activation, its incoming gradient a random
    tensor, its ReLU / LeakyReLU branches ours (O.branch_masks), so nothing depends on the rest of the network or on a kink."""
    A, s, B, h, w = case["A"], case["s"], case["B"], case["h"], case["w"]
    V = A * A
    gen = torch.Generator().manual_seed(11 + 7 * block + layer)
    tv = lambda name: T.tape_view(case["tape"], name, B, A, h, w, s, (B, V, h, w, 64)).cpu().permute(0, 4, 1, 2, 3).contiguous()   # noqa: E731
    sd = {k: v.clone().requires_grad_(True) for k, v in case["sd"].items()}
    O.branch_masks = our_branches(case)
    try:
        if kind == "upsample":
            x = tv("body").requires_grad_(True)
            y = O.upsample(sd, O.views_to_mosaic(x, A), s)                                  # the bicubic skip has no parameters and no input gradient
            prefix = "upsampling."
        elif kind == "spa":
            x = tv(f"ang{layer}.y").requires_grad_(True)
            y = O.spa_block(sd, layer, x)
            prefix = f"altblock.{layer}.spa_trans."
        elif kind == "ang":
            x = (tv("feat") if layer == 0 else tv(f"spa{layer - 1}.y")).requires_grad_(True)
            y = O.ang_block(sd, layer, x)
            prefix = f"altblock.{layer}.ang_trans."
        else:
            x = None
            y = O.init_features(sd, O.mosaic_to_views(case["lr"].cpu(), A))
            prefix = "conv_init"
        d_out = torch.randn(y.shape, generator=gen) / y.numel() ** 0.5
        y.backward(d_out)
    finally:
        O.branch_masks = None
    # our side: the same incoming gradient in the kernels' layout
    d_out_dev = (d_out if kind == "upsample" else d_out.permute(0, 2, 3, 4, 1)).contiguous().to(G.DEV)
    flat = torch.full((T.grad_floats(s),), float("nan"), device=G.DEV)
    d_in = T.block_backward(case["ps"], case["lr"], case["tape"], block, layer, d_out_dev, A, s, flat, math=case["math"])
    torch.cuda.synchronize()
    if x is not None:
        got, ref = d_in.cpu().permute(0, 4, 1, 2, 3), x.grad
        assert G.rel_max(got, ref) <= TOL, f"{kind}{layer} d_in: " + G.err_report(got.contiguous(), ref)
    off, checked = 0, 0
    for name, p in zip(case["names"], case["ps"]):
        k = p.numel()
        got = flat[off:off + k].cpu().view(p.shape)
        off += k
        if name.startswith(prefix):
            ref = sd[name].grad
            assert not torch.isnan(got).any(), name
            assert float((got - ref).abs().max()) <= TOL * float(ref.abs().max()) + 1e-10, f"{name}: " + G.err_report(got, ref)
            checked += 1
        else:
            assert bool(torch.isnan(got).all()), f"{name}: a gradient outside the block was written"
    assert checked == {"upsample": 2, "spa": 10, "ang": 8, "init": 4}[kind]


def test_backward_is_deterministic(case):
    again = T.train_backward(case["ps"], case["lr"], case["tape"], case["dout"], case["A"], case["s"], math=case["math"])
    torch.cuda.synchronize()
    assert torch.equal(again, case["flat"])


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("name", ["train_a3_s2_b2_6x6", "train_a2_s4_b1_8x5"])
def test_train_step_matches_reference_fixture(name, math, golden_dir):
    """TrainStep (flat buffers, C-ABI loss / backward / Adam) against the reference network + torch.optim.Adam."""
    from model import LFT
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    A, s, B, h, w, wseed, iseed, tseed, steps = [int(v) for v in g["meta"]]
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s))
    sd = deterministic_state(64, s, seed=wseed, flavor=str(g["flavor"]))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(G.DEV).train()
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=iseed)).to(G.DEV)
    hr = torch.from_numpy(g["hr"]).to(G.DEV)
    ts = T.TrainStep(net, lr=2e-4, math=math)
    losses = []
    for step in range(steps):
        losses.append(float(ts.step(lr, hr)))
        if step == 0:
            for k, p in net.named_parameters():
                got = p.grad.detach().cpu().numpy().ravel()
                ref = g[f"grad_{k}_sub"]
                scale = max(float(np.abs(ref).max()), 1e-12)
                sub = got[sub_indices(got.size)]
                if math != "bf16x3":
                    assert np.abs(sub - ref).max() <= TOL * scale + 1e-10, k
                else:       # branch flips against the reference are certain here (see the note at KINK_FLIP_LIMIT below): bound the rms instead
                    assert np.sqrt(np.mean((sub - ref) ** 2)) <= 3e-2 * max(np.sqrt(np.mean(ref ** 2)), 1e-12), k
    assert np.allclose(losses, g["losses"], rtol=0, atol=1e-5 if math != "bf16x3" else 1e-4), (losses, g["losses"])
    for k, p in net.state_dict().items():
        got = p.cpu().numpy().ravel()
        ref = g[f"post_{k}_sub"]
        # Adam moves a weight by about +-lr per step whatever the gradient's size, so where |g| is at rounding level the two
        # implementations may step in opposite directions: bound the worst case by the step size, the mean tightly
        assert np.abs(got[sub_indices(got.size)] - ref).max() <= (1.05 if math != "bf16x3" else 4.0) * steps * 2e-4, k
        assert np.mean(np.abs(got[sub_indices(got.size)] - ref)) <= 2e-5, k
    # the inference path sees the updated weights
    with torch.no_grad():
        y = net.eval()(lr)
    assert float((y - hr).abs().mean()) < losses[0]


# Largest |z| of the reference at which a branch of ours may differ from the reference's, and how many units may (of 10.3 M).
# fp32: the two implementations differ by summation order only (forward error ~5e-7 absolute); split-bf16 products carry
# 2^-17 relative rounding per operand, so more units sit within its noise of a kink.
# Observed (three seeds): fp32 4 .. 7 units with |z| <= 7.5e-7; split-bf16 84 .. 96 units with |z| <= 2.2e-5.
KINK_FLIP_LIMIT = {"fp32": (5e-6, 24), "bf16x3": (1e-4, 300), "bf16x6": (5e-6, 24)}     # bf16x6: the forward IS the exact-fp32 one
# After the alignment the two modes are held to the SAME 1e-3 gate of north_star; what is left is kernel arithmetic, observed
# 3.4e-6 (fp32) and 4.7e-5 (split-bf16) of each tensor's scale -- a second, tighter bound pins that level.
KINK_ALIGNED_LEVEL = {"fp32": 5e-5, "bf16x3": 3e-4, "bf16x6": 5e-5}



KINK_FIXTURES = [f"train_kink_a5_s2_b2_16x16_seed{i}" for i in (0, 1, 2)] + [
    "train_kink_a5_s4_b1_16x16_seed0",      # 4x: 1 024 up-sampler channels per token (6 400 tokens)
    "train_kink_a9_s2_b1_8x8_seed0",        # 9 x 9 views: the 81-view angular attention (5 184 tokens)
    "train_kink_a3_s2_b1_16x40_seed0"]      # h < w: queries with x - 2 >= h have an empty window and pass no gradient (5 760 tokens)


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("fixture", KINK_FIXTURES)
def test_gradients_on_unscreened_inputs_with_aligned_kinks(fixture, math, golden_dir):
    """All 78 gradients against the REAL reference network's autograd on UNSCREENED inputs (12 800 tokens on three seeds of the
    2x shape; one input each of a 4x, a 9 x 9-view and an h < w shape at >= 5 k tokens), both math modes held to 1e-3, with no dependence on luck at the ReLU / LeakyReLU kinks: the fixture (tools/gen_golden.py:
    train_kink_case) carries the reference's branch decision at every unit.  (1) Away from the kinks (|z| >= 5e-4) our forward
    must take the reference's branch at EVERY one of the 10.3 M units (hash of the sign bitmap).  (2) Among the listed near-zero
    units our decision may differ only where the reference's own |z| is at rounding level, and only at a few units.  (3) Exactly
    those units are set to the reference's branch in our tape (a change of < 1e-12 to the saved activation), after which all
    gradients must agree to 1e-3 of each tensor's scale -- whatever the compiler flags did to the last bit of the forward."""
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    seed = fixture
    A, s, B, h, w, wseed, iseed, tseed, _ = [int(v) for v in g["meta"]]
    V, ss = A * A, s * s
    sd_np = deterministic_state(64, s, seed=wseed, flavor=str(g["flavor"]))
    names = [n for n, _, _ in param_table(64, s)]
    ps = [torch.from_numpy(sd_np[n]).to(G.DEV).contiguous() for n in names]
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=iseed)).to(G.DEV)
    hr = torch.from_numpy(np.random.Generator(np.random.PCG64([tseed, B, A, h, w, s])).random((B, 1, A * h * s, A * w * s), dtype=np.float32))
    assert abs(float(hr.double().sum()) - float(g["hr_sum"])) < 1e-6 * float(g["hr_sum"])
    hr = hr.to(G.DEV)
    out, tape = T.train_forward(ps, lr, A, s, math=math)
    # saved activation of each tag, and the rearrangement of its [B,V,h,w,C] tape layout into the reference's layout
    to_ref = {}
    for i, nm in zip((0, 2, 4), ("c1", "c2", "c3")):
        to_ref[f"conv{i}"] = (nm, 64, lambda t: t.permute(0, 4, 1, 2, 3))
    for l in range(4):
        to_ref[f"ang{l}"] = (f"ang{l}.hdn", 128, lambda t: t.permute(1, 0, 2, 3, 4).reshape(V, B * h * w, 128))
        to_ref[f"spa{l}"] = (f"spa{l}.hdn", 256, lambda t: t.permute(2, 3, 0, 1, 4).reshape(h * w, B * V, 256))
    to_ref["up"] = ("act", 64 * ss, lambda t: O.views_to_mosaic(t.permute(0, 4, 1, 2, 3), A))
    limit, max_flips = KINK_FLIP_LIMIT[math]
    units = sum(int(np.prod(g[f"kink_{tag}_shape"])) for tag in KINK_TAGS)
    max_flips = max(8, int(round(max_flips * units / 25.4e6)))              # the limits were set on the 25.4 M units of the 2x shape (B = 2)
    nflip, worst_z = 0, 0.0
    for tag in KINK_TAGS:
        nm, C, rearr = to_ref[tag]
        tv = T.tape_view(tape, nm, B, A, h, w, s, (B, V, h, w, C))
        pos = rearr(tv.cpu() > 0).contiguous().numpy().ravel()
        assert pos.size == int(np.prod(g[f"kink_{tag}_shape"])), tag
        ok, where, flips = kink_compare(g, tag, pos)
        assert ok, f"[{math}] a unit at |z| >= 5e-4 took the other branch -- " + where
        if flips:
            where_in_tape = rearr(torch.arange(tv.numel()).view(B, V, h, w, C)).contiguous().view(-1)
            idx = torch.tensor([i for i, _ in flips], dtype=torch.long)
            zref = torch.tensor([z for _, z in flips])
            slope_side = 0.0 if ("ang" in tag or "spa" in tag) else -1e-12            # ReLU saves 0, LeakyReLU a negative value
            vals = torch.where(zref > 0, torch.full_like(zref, 1e-12), torch.full_like(zref, slope_side))
            tv.view(-1)[where_in_tape[idx].to(G.DEV)] = vals.to(G.DEV)
            nflip += len(flips)
            worst_z = max(worst_z, float(zref.abs().max()))
    print(f"{seed} [{math}]: {nflip} of {units / 1e6:.1f} M units took the other branch, all with reference |z| <= {worst_z:.2e}")
    assert worst_z < limit and nflip <= max_flips, (nflip, worst_z)
    n = out.numel()
    dout = torch.empty_like(out)
    scratch = torch.empty(1025, device=G.DEV)
    _lib.check(_lib.lib().lft_l1_loss(out.data_ptr(), hr.data_ptr(), n, dout.data_ptr(), 1.0 / n, scratch[1024:].data_ptr(),
                                      scratch.data_ptr(), G.stream()), "lft_l1_loss")
    # the loss's own kink, sign(sr - hr): every pixel away from it must have the reference's sign (hash of the bitmap); at the listed
    # near-zero pixels ours may differ only where the reference's |sr - hr| is at rounding level, and d loss / d out takes the
    # reference's sign there
    pos = (dout > 0).cpu().numpy().ravel()
    near, dref = g["l1_near_idx"].astype(np.int64), g["l1_near_d"]
    mine_near = pos[near].copy()
    pos[near] = False
    import hashlib
    assert hashlib.sha256(np.packbits(pos).tobytes()).digest() == g["l1_sha256"].tobytes(), f"[{math}] sign(sr - hr) differs from the reference away from the kink"
    l1_flips = [(int(i), float(d)) for i, d, m in zip(near, dref, mine_near) if bool(m) != (d > 0)]
    if l1_flips:
        assert max(abs(d) for _, d in l1_flips) < limit and len(l1_flips) <= 16, l1_flips
        idx = torch.tensor([i for i, _ in l1_flips], dtype=torch.long, device=G.DEV)
        val = torch.tensor([(1.0 if d > 0 else -1.0) / n for _, d in l1_flips], device=G.DEV)
        dout.view(-1)[idx] = val
    print(f"{seed} [{math}]: {len(l1_flips)} of {n} output pixels on the other side of the L1 kink" + (f", all with reference |sr - hr| <= {max(abs(d) for _, d in l1_flips):.1e}" if l1_flips else ""))
    flat = T.train_backward(ps, lr, tape, dout, A, s, math=math).cpu().numpy()
    assert abs(float(scratch[1024]) - float(g["losses"][0])) <= (1e-5 if math != "bf16x3" else 1e-4)
    off, worst = 0, (0.0, "")
    for name, p in zip(names, ps):
        got = flat[off:off + p.numel()]
        off += p.numel()
        ref = g[f"grad_{name}_sub"]
        scale = max(float(np.abs(ref).max()), 1e-12)
        rel = float(np.abs(got[sub_indices(got.size)] - ref).max()) / scale
        worst = max(worst, (rel, name))
        assert rel <= TOL, (name, rel, math)
    print(f"{seed} [{math}]: gradients vs the reference after aligning {nflip} units: worst rel err {worst[0]:.2e} ({worst[1]})")
    assert worst[0] <= KINK_ALIGNED_LEVEL[math], worst


def test_autograd_surface_like_reference_train_py():
    """net(data) -> criterion -> loss.backward() -> torch.optim.Adam, exactly the calls of train.py:89-107."""
    from model import LFT
    A, s, B, h, w = 3, 2, 2, 6, 6
    sd_np, lr, hr = make_inputs(A, s, B, h, w)
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    net = net.to(G.DEV)
    net.apply(LFT.weights_init)
    crit = LFT.get_loss(None).to(G.DEV)
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-4, betas=(0.9, 0.999), eps=1e-08, weight_decay=0)
    opt.zero_grad()
    out = net(lr.to(G.DEV))
    assert out.requires_grad
    loss = crit(out, hr.to(G.DEV))
    loss.backward()
    _, _, grads_ref = O.loss_and_grads(O.state_from_numpy(sd_np), lr, hr, A, s)
    for k, p in net.named_parameters():
        ref = grads_ref[k]
        assert float((p.grad.cpu() - ref).abs().max()) <= TOL * float(ref.abs().max()) + 1e-10, k
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    opt.step()
    assert all(not torch.equal(before[k], p) for k, p in net.named_parameters())


def test_two_rank_dp_step_equals_single_rank(tmp_path):
    """Two ranks (sharing the one GPU of the test box, gloo) with half the batch each == one rank with the whole batch:
    the flat-gradient sum + 1/world inside Adam reproduces the global-batch step."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(__file__), "dp_train_worker.py")
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, worker, "2", one], check=True, env=env, timeout=300)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29611", worker, "2", two], check=True, env=env, timeout=300)
    a, b = torch.load(one), torch.load(two)
    # Adam normalises the step (~lr per weight per step), so compare displacements: 2 steps of 2e-4
    d = (a["flat"] - b["flat"]).abs()
    print(f"single vs 2-rank DP after 2 steps: max |dw| {float(d.max()):.2e}, mean {float(d.mean()):.2e}")
    assert float(d.max()) <= 1.05 * 2 * 2e-4 and float(d.mean()) <= 2e-5


def test_two_rank_dp_from_scratch_keeps_replicas_identical(tmp_path):
    """No load_state_dict: each rank constructs its own randomly initialised network.  TrainStep broadcasts rank 0's
    weights, so after two data-parallel Adam steps both replicas hold bit-identical parameters."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(__file__), "dp_train_worker.py")
    out = str(tmp_path / "scratch.pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29613", worker, "2", out, "scratch"], check=True, env=env, timeout=300)
    a, b = torch.load(out + ".rank0"), torch.load(out + ".rank1")
    assert torch.equal(a["flat"], b["flat"])


def test_rccl_backend_runs_the_collectives_on_device(tmp_path):
    """init_process_group("nccl") (= RCCL on ROCm; world size 1 is what a one-GPU box allows) and the data-parallel
    collectives on DEVICE tensors: the timing all-reduce of bench.py, the parameter broadcast, the flat-gradient sum and
    a whole TrainStep.step.  A child process, so the process group does not leak into the other tests."""
    import subprocess
    import sys
    code = r"""
import os, sys
from types import SimpleNamespace
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from lft_amd import dp, train as T
from lft_amd.params import deterministic_state, synthetic_lr
from model import LFT
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
assert dp.barrier_max_seconds(1.5, dev) == 1.5                       # all_reduce(MAX) on a device tensor through RCCL
g = torch.arange(1000, dtype=torch.float32, device=dev)
ref = g.clone()
assert dp.sum_gradients_(g, force=True) == 1.0 and torch.equal(g, ref)   # ncclAllReduce(SUM), one rank: identity
dp.broadcast_(g, force=True); assert torch.equal(g, ref)
A, s, B, h, w = 3, 2, 2, 6, 6
net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s))
net.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, s, seed=1, flavor="stress").items()})
ts = T.TrainStep(net.to(dev).train(), lr=2e-4)
lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).to(dev)
hr = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).random((B, 1, A * h * s, A * w * s), dtype=np.float32)).to(dev)
l0 = float(ts.step(lr, hr)); l1 = float(ts.step(lr, hr))
assert np.isfinite(l0) and l1 < l0, (l0, l1)
# bucketed exchange: the backward pass reports three gradient buckets; each one's ncclAllReduce is started while the next
# bucket's kernels are enqueued (graph mode: one captured graph per bucket).  Must not change a bit against the plain step.
def run(force, graph):
    os.environ["LFT_DP_FORCE_COLLECTIVES"] = "1" if force else "0"
    n = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s))
    n.load_state_dict({k: torch.from_numpy(v) for k, v in deterministic_state(64, s, seed=1, flavor="stress").items()})
    t = T.TrainStep(n.to(dev).train(), lr=2e-4, graph=graph)
    losses = [float(t.step(lr, hr)) for _ in range(3)]
    if force and graph:
        assert len(next(iter(t._graphs.values()))["graphs"]) == 3
    return losses, t.flat_params.clone(), t.flat_grads.clone()
plain = run(False, True)
for force, graph in ((True, True), (True, False)):
    got = run(force, graph)
    assert got[0] == plain[0], (force, graph, got[0], plain[0])
    assert torch.equal(got[1], plain[1]) and torch.equal(got[2], plain[2]), (force, graph)
spans = sorted(T.grad_bucket(s, b) for b in range(3))
assert spans[0][0] == 0 and all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(2)) and spans[2][0] + spans[2][1] == T.grad_floats(s)
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK")
"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", LFT_DP_FORCE_COLLECTIVES="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_fit_trains_and_writes_reference_format_checkpoints(tmp_path):
    """lft_amd.trainer.fit (the reference's epoch loop, train.py:86-110) on a small synthetic set: the loss goes down, the
    per-epoch .pth files have the reference's name / keys and reload into a fresh model that reproduces the outputs."""
    from model import LFT
    from lft_amd import trainer
    A, s = 2, 2
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s)).to(G.DEV)
    src = trainer.SyntheticPatchSource(16, A, s, patch=8, seed=1, device=G.DEV)
    logs = []
    hist = trainer.fit(net, src, epochs=6, batch_size=4, lr=1e-3, n_steps=2, gamma=0.5, ckpt_dir=str(tmp_path), log=logs.append)
    print(hist)
    assert len(hist) == 6 and hist[-1] < 0.8 * hist[0]
    assert "lr 0.00025" in logs[-1]                                   # 1e-3 * 0.5 ** (5 // 2)
    path = os.path.join(str(tmp_path), "LFT_2x2_2x_epoch_06_model.pth")
    assert os.path.exists(path)
    fresh = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s)).to(G.DEV).eval()
    assert trainer.load_checkpoint(fresh, path) == 6
    lr, _ = src.get([0, 1])
    with torch.no_grad():
        assert torch.equal(fresh(lr), net.eval()(lr))


def test_fit_batch_metrics_are_the_reference_loops_cal_metrics(tmp_path):
    """fit(batch_metrics=True) = the reference's per-batch cal_metrics(args, label, out) of train.py:121-124 on the step's own output
    (before the update), averaged over the epoch's batches: replayed here step by step against the scikit-image-pinned oracle."""
    from model import LFT
    from lft_amd import trainer
    from lft_amd.train import TrainStep
    from oracle import metrics_oracle as MO
    A, s = 2, 2
    torch.manual_seed(3)
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s)).to(G.DEV)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    src = trainer.SyntheticPatchSource(8, A, s, patch=16, seed=2, device=G.DEV)
    logs = []
    hist = trainer.fit(net, src, epochs=2, batch_size=4, lr=5e-4, use_augmentation=False, log=logs.append, batch_metrics=True)
    assert len(trainer.fit.last_metrics) == 2 and "psnr is" in logs[0] and "ssim is" in logs[0]
    # replay: same weights, same batches, the oracle's metrics of every step's output
    net2 = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s)).to(G.DEV)
    net2.load_state_dict(state)
    ts = TrainStep(net2, lr=5e-4)
    for epoch in range(2):
        ts.lr = trainer.step_lr(5e-4, epoch)
        acc = []
        for ix in trainer.epoch_batches(len(src), 4, epoch, 0, 0, 1):
            a, b = src.get(ix)
            loss = float(ts.step(a, b))
            _, _, pm, sm = MO.cal_metrics(b.cpu().numpy(), ts.last_out.cpu().numpy(), A)
            acc.append((loss, pm, sm))
        m = np.mean(np.array(acc), axis=0)
        assert abs(m[0] - hist[epoch]) < 1e-6
        assert abs(m[1] - trainer.fit.last_metrics[epoch][0]) < 2e-4 and abs(m[2] - trainer.fit.last_metrics[epoch][1]) < 5e-6, (m, trainer.fit.last_metrics)


@pytest.mark.parametrize("hw", [(32, 32), (33, 35), (26, 32)], ids=["32x32", "33x35_ragged", "26x32_rows_across_workgroups"])
@pytest.mark.parametrize("math", MATHS)
def test_full_size_backward_properties_cfg3(math, hw):
    """BASELINE configs[2] shape (A5, 2x, 32x32 LR views), where autograd over the CPU oracle takes minutes: properties
    that hold at any size instead.  For a fixed forward the backward pass is linear in d loss / d out, and patches never
    interact, so (i) grads(a*g1 + g2) = a*grads(g1) + grads(g2), (ii) the gradient of a 3-patch batch is the sum of the
    per-patch gradients; and the forward-with-tape must agree with the fused inference kernels.  Three patches are 76 800 tokens:
    the batch runs the ring-fed GEMM kernel (k_linr, above 65 536 tokens), the single patches the direct one (k_lin) -- (ii) and
    the forward comparison hold the two against each other and against the inference path.  33 x 35 views: 86 625 tokens, not
    a multiple of 32 -- partial tiles, waves past the end that only serve the ring, 32-token tiles that straddle image rows.
    26 x 32 views: the one-image-row-per-wave 3x3 paths (lane shifts in the GEMM, ten values per lane in the weight gradient) with
    workgroups that straddle two images (832 tokens per image = 6.5 workgroups)."""
    A, s = 5, 2
    h, w = hw
    B = 3 if h * w >= 1024 else 4                                        # > 65 536 tokens: the batch runs k_linr
    tol = {"fp32": 2e-5, "bf16x3": 1e-4, "bf16x6": 2e-5}[math]
    sd_np = deterministic_state(64, s, seed=1, flavor="stress")
    names = [n for n, _, _ in param_table(64, s)]
    ps = [torch.from_numpy(sd_np[n]).to(G.DEV).contiguous() for n in names]
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0)).to(G.DEV)
    g = torch.Generator(device="cpu").manual_seed(5)
    g1 = torch.randn(B, 1, A * h * s, A * w * s, generator=g).to(G.DEV) * 1e-3
    g2 = torch.randn(B, 1, A * h * s, A * w * s, generator=g).to(G.DEV) * 1e-3
    out, tape = T.train_forward(ps, lr, A, s, math=math)
    ga = T.train_backward(ps, lr, tape, g1, A, s, math=math).clone()
    gb = T.train_backward(ps, lr, tape, g2, A, s, math=math).clone()
    gc = T.train_backward(ps, lr, tape, 0.5 * g1 + g2, A, s, math=math).clone()
    scale = float(gc.abs().max())
    assert float((gc - (0.5 * ga + gb)).abs().max()) <= tol * scale
    per, outs = [], []
    for i in range(B):
        o1, tp = T.train_forward(ps, lr[i:i + 1], A, s, math=math)
        outs.append(o1.clone())
        per.append(T.train_backward(ps, lr[i:i + 1], tp, g1[i:i + 1].contiguous(), A, s, math=math).clone())
    assert float((ga - sum(per)).abs().max()) <= tol * float(ga.abs().max())
    # the two forms of the GEMM add an accumulator's products in the same order: the forward is not just close, it is the same
    assert torch.equal(out, torch.cat(outs))
    # forward-with-tape (unfused fp32 kernels) vs the fused fp32 inference kernels
    from model import LFT
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s), precision="fp32")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    with torch.no_grad():
        y = net.to(G.DEV).eval()(lr)
    assert float((y - out).abs().max() / y.abs().max()) <= (1e-5 if math != "bf16x3" else 1e-4)


@pytest.mark.parametrize("math", MATHS)
def test_ring_gemm_passes_are_bit_reproducible_at_size(math):
    """76 800 tokens (three 5x5 patches of 32x32): every Linear / conv runs the ring-fed k_linr -- asm LDS-DMA, one barrier per
    k-step, LDS slots re-used behind barriers, a tile scratch overlaid on a ring slot.  A missing wait or barrier there shows as
    a rare, timing-dependent difference: 12 forward + backward passes on fresh tapes must agree bit for bit (also with
    whatever other kernels the previous pass left in flight)."""
    A, s, B, h, w = 5, 2, 3, 32, 32
    sd_np = deterministic_state(64, s, seed=1, flavor="stress")
    names = [n for n, _, _ in param_table(64, s)]
    ps = [torch.from_numpy(sd_np[n]).to(G.DEV).contiguous() for n in names]
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=3)).to(G.DEV)
    g = torch.Generator(device="cpu").manual_seed(9)
    dout = torch.randn(B, 1, A * h * s, A * w * s, generator=g).to(G.DEV) * 1e-3
    ref_out = ref_grad = None
    for rep in range(12):
        out, tape = T.train_forward(ps, lr, A, s, math=math)
        grad = T.train_backward(ps, lr, tape, dout, A, s, math=math).clone()
        if rep == 0:
            ref_out, ref_grad = out.clone(), grad
            assert bool(torch.isfinite(ref_out).all()) and bool(torch.isfinite(ref_grad).all())
        else:
            assert torch.equal(out, ref_out), f"forward differs in pass {rep}"
            assert torch.equal(grad, ref_grad), f"backward differs in pass {rep}"
        del tape


def test_training_is_bitwise_reproducible():
    """Two independent runs of 12 Adam steps (graph-captured forward + backward, weight gradients on the side stream,
    table-driven reduction) end in bit-identical weights: no atomics, no order-dependent sums, no races."""
    from model import LFT
    A, s, B, h, w = 3, 2, 2, 8, 8
    sd_np, lr, hr = make_inputs(A, s, B, h, w)
    finals = []
    for run in range(2):
        net = LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s))
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        ts = T.TrainStep(net.to(G.DEV), lr=1e-3, math="bf16x3")
        losses = [float(ts.step(lr.to(G.DEV), hr.to(G.DEV))) for _ in range(12)]
        torch.cuda.synchronize()
        finals.append((ts.flat_params.clone(), losses))
    assert torch.equal(finals[0][0], finals[1][0])
    assert finals[0][1] == finals[1][1] and finals[0][1][-1] < finals[0][1][0]
