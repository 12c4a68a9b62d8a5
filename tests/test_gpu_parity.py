"""GPU parity tests (run on the MI355X box with -m gpu): every stage of the HIP path, called through
the C ABI of liblft_hip.so, against the CPU oracle on the same seeded inputs, and the whole forward
against the fixtures captured from the real reference.

Tolerances (BASELINE.json north_star: 1e-3 relative fp32):
  fp32 path : max|err| <= 1e-4 * max|ref| per stage (observed ~1e-6), <= 1e-3 * max|ref| required end to end
  fp16 path : rms err <= 2e-3 * rms(ref) per stage; <= 1e-3 * max|ref| required end to end (observed ~2e-4: the bf16 kernels with
              IEEE-half operands and tensors -- the fast path that meets north_star)
  bf16 path : rms err <= 1e-2 * rms(ref) per stage; end to end bounded at 2.5e-3 * max|ref| (observed 1.5e-3 .. 1.9e-3: bf16 operand
              rounding, tests/diag_precision_study.py -- this path does NOT meet the 1e-3 of north_star; the fp32 and fp16 paths do)
"""
import os

import numpy as np
import pytest
import torch

from lft_amd import _lib
from lft_amd.params import deterministic_state, synthetic_lr
from oracle import lft_oracle as O

import gpu_util as G

pytestmark = pytest.mark.gpu

FP32_STAGE_TOL = 1e-4
BF16_STAGE_RMS = 1e-2
FP16_STAGE_RMS = 2e-3
END_TO_END = {"fp32": 1e-3, "fp16": 1e-3, "bf16": 2.5e-3}
ALL_PRECS = ["fp32", "fp16", "bf16"]


def check(got, ref, prec, what):
    msg = f"{what} [{prec}]: " + G.err_report(got, ref)
    assert not torch.isnan(got).any(), msg
    if prec == "fp32":
        assert G.rel_max(got, ref) <= FP32_STAGE_TOL, msg
    else:
        assert G.rel_rms(got, ref) <= (FP16_STAGE_RMS if prec == "fp16" else BF16_STAGE_RMS), msg
    print(msg)


@pytest.mark.parametrize("prec", ALL_PRECS)
def test_mfma_fragment_layout(prec):
    """C = A B and D = W2 C with asymmetric small-integer data (exact in bf16 and fp32)."""
    rng = np.random.default_rng(0)
    Am = torch.from_numpy(rng.integers(-3, 4, size=(32, 16)).astype(np.float32))
    Bm = torch.from_numpy(rng.integers(-3, 4, size=(16, 32)).astype(np.float32))
    W2 = torch.from_numpy(rng.integers(-2, 3, size=(32, 32)).astype(np.float32))
    C = torch.zeros(32, 32, device=G.DEV)
    D = torch.zeros(32, 32, device=G.DEV)
    a, b, w2 = Am.to(G.DEV), Bm.to(G.DEV), W2.to(G.DEV)
    _lib.check(_lib.lib().lft_mfma_selftest(a.data_ptr(), b.data_ptr(), w2.data_ptr(), C.data_ptr(), D.data_ptr(),
                                            G.PRECS[prec], G.stream()), "selftest")
    torch.cuda.synchronize()
    Cref = Am @ Bm
    assert torch.equal(C.cpu(), Cref), "C = A*B layout wrong"
    if prec == "fp32":      # |C| can exceed bf16's exact-integer range, so D is only exact in fp32
        assert torch.equal(D.cpu(), W2 @ Cref), "acc-order operand re-use wrong"
    else:
        assert G.rel_max(D.cpu(), W2 @ Cref.to(G.ACT_DTYPE[prec]).float()) < 1e-6


@pytest.mark.parametrize("A,h,w,s", [(3, 7, 5, 2), (5, 8, 8, 4), (2, 4, 9, 4)])
def test_bicubic(A, h, w, s):
    lr = torch.from_numpy(synthetic_lr(2, A, h, w, seed=3))
    out = torch.empty(2, 1, A * h * s, A * w * s, device=G.DEV)
    x = lr.to(G.DEV)
    _lib.check(_lib.lib().lft_bicubic_fwd(x.data_ptr(), out.data_ptr(), 2, A, h, w, s, G.stream()), "bicubic")
    torch.cuda.synchronize()
    ref = O.bicubic_skip(lr, A, s)
    assert (out.cpu() - ref).abs().max() <= 2e-6, G.err_report(out.cpu(), ref)


CASES = [(5, 2, 2, 6, 6), (5, 4, 1, 8, 8), (3, 2, 1, 9, 7), (5, 2, 1, 32, 32), (9, 4, 1, 8, 8), (6, 2, 1, 6, 5),   # 81 views: 3 column tiles; 36: 2
         (2, 2, 1, 64, 64),                   # 64-wide views: two column tiles per row, wide-tile LDS path
         (2, 2, 1, 6, 12), (2, 2, 1, 36, 64)]  # h < w: queries with x - 2 >= h have an EMPTY window (LFT.py:155) -> attention output 0


@pytest.fixture(scope="module", params=CASES, ids=lambda c: "A%d_s%d_B%d_%dx%d" % c)
def case(request):
    A, s, B, h, w = request.param
    sd_np = deterministic_state(64, s, seed=1, flavor="stress")
    sd = O.state_from_numpy(sd_np)
    lr = torch.from_numpy(synthetic_lr(B, A, h, w, seed=0))
    taps = {}
    out = O.forward(sd, lr, A, s, taps)
    packs = {p: G.Packed(sd_np, A, h, w, s, p, B) for p in ALL_PRECS}
    return dict(A=A, s=s, B=B, h=h, w=w, sd=sd, lr=lr, taps=taps, out=out, packs=packs)


@pytest.mark.parametrize("prec", ALL_PRECS)
def test_init_features(case, prec):
    pk = case["packs"][prec]
    lr = case["lr"].to(G.DEV)
    act = pk.new_act()
    _lib.check(_lib.lib().lft_init_features_fwd(pk.buf.data_ptr(), lr.data_ptr(), act.data_ptr(), pk.work.data_ptr(),
                                                *pk.dims(), G.stream()), "init_features")
    torch.cuda.synchronize()
    check(G.from_act(act), case["taps"]["feat"], prec, "init_features")


@pytest.mark.parametrize("prec", ALL_PRECS)
@pytest.mark.parametrize("layer", [0, 3])
def test_ang_block(case, prec, layer):
    pk = case["packs"][prec]
    x = case["taps"]["feat"] if layer == 0 else case["taps"][f"spa{layer - 1}"]
    xin = G.to_act(x, prec)
    ref = O.ang_block(case["sd"], layer, G.from_act(xin))       # oracle sees the same (possibly bf16-rounded) input
    act = pk.new_act()
    _lib.check(_lib.lib().lft_ang_block_fwd(pk.buf.data_ptr(), layer, xin.data_ptr(), act.data_ptr(), *pk.dims(), G.stream()),
               "ang_block")
    torch.cuda.synchronize()
    check(G.from_act(act), ref, prec, f"ang_block{layer}")


@pytest.mark.parametrize("prec", ALL_PRECS)
@pytest.mark.parametrize("layer,with_skip", [(0, False), (3, True)])
def test_spa_block(case, prec, layer, with_skip):
    pk = case["packs"][prec]
    xin = G.to_act(case["taps"][f"ang{layer}"], prec)
    skip = G.to_act(case["taps"]["feat"], prec) if with_skip else None
    ref = O.spa_block(case["sd"], layer, G.from_act(xin))
    if with_skip:
        ref = ref + G.from_act(skip)
    act = pk.new_act()
    _lib.check(_lib.lib().lft_spa_block_fwd(pk.buf.data_ptr(), layer, xin.data_ptr(), skip.data_ptr() if with_skip else None,
                                            act.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()), "spa_block")
    torch.cuda.synchronize()
    check(G.from_act(act), ref, prec, f"spa_block{layer}")


@pytest.mark.parametrize("prec", ALL_PRECS)
def test_upsample(case, prec):
    pk = case["packs"][prec]
    xin = G.to_act(case["taps"]["body"], prec)
    lr = case["lr"].to(G.DEV)
    A, s, B, h, w = case["A"], case["s"], case["B"], case["h"], case["w"]
    out = torch.empty(B, 1, A * h * s, A * w * s, device=G.DEV)
    _lib.check(_lib.lib().lft_upsample_fwd(pk.buf.data_ptr(), xin.data_ptr(), lr.data_ptr(), out.data_ptr(), pk.work.data_ptr(),
                                           *pk.dims(), G.stream()), "upsample")
    torch.cuda.synchronize()
    ref = O.upsample(case["sd"], O.views_to_mosaic(G.from_act(xin), A), s)
    skip = case["taps"]["skip"]
    check(out.cpu() - skip, ref, prec, "upsample(residual branch)")


@pytest.mark.parametrize("prec", ALL_PRECS)
def test_forward_vs_oracle(case, prec):
    pk = case["packs"][prec]
    lr = case["lr"].to(G.DEV)
    A, s, B, h, w = case["A"], case["s"], case["B"], case["h"], case["w"]
    out = torch.empty(B, 1, A * h * s, A * w * s, device=G.DEV)
    _lib.check(_lib.lib().lft_forward(pk.buf.data_ptr(), lr.data_ptr(), out.data_ptr(), pk.work.data_ptr(), *pk.dims(), G.stream()),
               "forward")
    torch.cuda.synchronize()
    got, ref = out.cpu(), case["out"]
    msg = f"forward [{prec}] " + G.err_report(got, ref) + f" psnr={O.psnr(got, ref):.2f}dB"
    print(msg)
    assert G.rel_max(got, ref) <= END_TO_END[prec], msg
    res_got, res_ref = got - case["taps"]["skip"], case["taps"]["res"]
    assert G.rel_rms(res_got, res_ref) <= {"fp32": 1e-4, "fp16": 4e-3, "bf16": 2e-2}[prec], "residual branch: " + G.err_report(res_got, res_ref)
