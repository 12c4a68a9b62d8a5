"""CPU-side tests: plugin surface, state-dict compatibility, C-ABI exports, loud failure without a GPU."""
import ctypes
import os
import re
from types import SimpleNamespace

import pytest
import torch

from lft_amd import _lib
from lft_amd.params import deterministic_state, num_params, param_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_param_counts():
    assert num_params(64, 2) == 1114240 and num_params(64, 4) == 1163392
    assert len(param_table(64, 2)) == 78


def test_plugin_surface_and_state_dict_roundtrip():
    from model import LFT
    for s in (2, 4):
        net = LFT.get_model(SimpleNamespace(channels=64, angRes=5, scale_factor=s))
        sd = net.state_dict()
        assert list(sd.keys()) == [n for n, _, _ in param_table(64, s)]
        assert all(tuple(sd[n].shape) == sh for n, sh, _ in param_table(64, s))
        ref = {k: torch.from_numpy(v) for k, v in deterministic_state(64, s, seed=3).items()}
        net.load_state_dict(ref)                                   # bare keys (reference test.py:45-50)
        assert torch.equal(net.state_dict()["altblock.2.ang_trans.attention.in_proj_weight"],
                           ref["altblock.2.ang_trans.attention.in_proj_weight"])
        net.apply(LFT.weights_init)                                # reference train.py:37
    assert isinstance(LFT.get_loss(None), torch.nn.Module)


def test_default_init_ranges():
    from model import LFT
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=5, scale_factor=2))
    sd = net.state_dict()
    assert float(sd["conv_init0.0.weight"].abs().max()) <= 1 / 3 + 1e-6
    assert float(sd["altblock.0.spa_trans.attention.in_proj_weight"].abs().max()) <= 1 / 128 ** 0.5 + 1e-6
    assert torch.all(sd["altblock.1.ang_trans.norm.weight"] == 1) and torch.all(sd["altblock.1.ang_trans.norm.bias"] == 0)


def test_no_cpu_fallback():
    from model import LFT
    net = LFT.get_model(SimpleNamespace(channels=64, angRes=5, scale_factor=2))
    with pytest.raises(_lib.LftError):
        net(torch.zeros(1, 1, 40, 40))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "lft_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(lft_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert _lib.lib().lft_version() == 1


def test_size_queries_and_argument_errors():
    assert _lib.packed_bytes(5, 32, 32, 4, _lib.PREC_BF16) > 2_000_000
    assert _lib.workspace_bytes(4, 5, 32, 32, 4, _lib.PREC_BF16) > _lib.workspace_bytes(1, 5, 32, 32, 4, _lib.PREC_BF16)
    with pytest.raises(_lib.LftError, match="scale factor"):
        _lib.packed_bytes(5, 32, 32, 3, _lib.PREC_BF16)
    with pytest.raises(_lib.LftError, match="prec"):
        _lib.packed_bytes(5, 32, 32, 2, 7)
    with pytest.raises(_lib.LftError, match="not implemented"):
        _lib.packed_bytes(12, 32, 32, 2, _lib.PREC_F32)
    assert _lib.packed_bytes(9, 32, 32, 4, _lib.PREC_BF16) > 2_000_000      # 9x9 = 81 views is supported
