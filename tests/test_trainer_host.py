"""CPU tests of the training driver's host logic (lft_amd/trainer.py): checkpoint format, schedule, sharded sampling,
augmentation, synthetic source."""
import os
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from lft_amd import trainer
from lft_amd.params import param_table


def make_net(A=2, s=2):
    from model import LFT
    return LFT.get_model(SimpleNamespace(channels=64, angRes=A, scale_factor=s))


def test_checkpoint_roundtrip_reference_format(tmp_path):
    net = make_net()
    path = str(tmp_path / trainer.checkpoint_name("LFT", 2, 2, 7))
    assert os.path.basename(path) == "LFT_2x2_2x_epoch_07_model.pth"                 # reference train.py:99-100
    trainer.save_checkpoint(net, path, 7)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"epoch", "state_dict"} and ck["epoch"] == 7
    assert list(ck["state_dict"]) == [n for n, _, _ in param_table(64, 2)]           # the reference's 78 keys, in order
    other = make_net()
    assert trainer.load_checkpoint(other, path) == 7
    for (k, a), (_, b) in zip(net.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k
    # DataParallel-style keys (reference train.py:45-49)
    pref = {"epoch": 3, "state_dict": OrderedDict(("module." + k, v) for k, v in ck["state_dict"].items())}
    p2 = str(tmp_path / "prefixed.pth")
    torch.save(pref, p2)
    third = make_net()
    assert trainer.load_checkpoint(third, p2) == 3
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), third.state_dict().values()))
    bad = {"epoch": 1, "state_dict": OrderedDict(list(ck["state_dict"].items())[:-1])}
    torch.save(bad, p2)
    with pytest.raises(KeyError):
        trainer.load_checkpoint(third, p2)


def test_step_lr_matches_torch_steplr():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=2e-4)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=15, gamma=0.5)
    for epoch in range(50):
        assert abs(opt.param_groups[0]["lr"] - trainer.step_lr(2e-4, epoch, 15, 0.5)) < 1e-12
        opt.step()
        sch.step()


def test_epoch_batches_partition_every_global_batch():
    n, gb, world = 37, 8, 4
    per_rank = [trainer.epoch_batches(n, gb, 3, 0, r, world) for r in range(world)]
    assert all(len(b) == 5 for b in per_rank)                                        # ceil(37 / 8)
    seen = []
    for b in range(5):
        batch = np.concatenate([per_rank[r][b] for r in range(world)])
        assert len(batch) == gb and all(len(per_rank[r][b]) == gb // world for r in range(world))
        seen.append(batch)
    flat = np.concatenate(seen)
    assert sorted(flat[:n].tolist()) == list(range(n))                               # one pass over the data, then wrap-around
    assert not np.array_equal(flat[:n], np.concatenate([np.concatenate([trainer.epoch_batches(n, gb, 4, 0, r, world)[b] for r in range(world)])
                                                         for b in range(5)])[:n])    # reshuffled per epoch
    with pytest.raises(ValueError):
        trainer.epoch_batches(n, 6, 0, 0, 0, 4)


def test_augment_is_the_reference_mosaic_flip():
    A, p, s = 2, 3, 2
    lr = torch.arange(2 * (A * p) ** 2, dtype=torch.float32).reshape(2, 1, A * p, A * p)
    hr = torch.arange(2 * (A * p * s) ** 2, dtype=torch.float32).reshape(2, 1, A * p * s, A * p * s)
    seen = set()
    for seed in range(40):
        a, b = trainer.augment(lr, hr, np.random.Generator(np.random.PCG64(seed)))
        assert a.shape == lr.shape and b.shape == hr.shape and a.is_contiguous()
        for i in range(2):
            # same multiset of values, and LR / HR get the same transform: compare against the 8 dihedral images
            cands = {}
            for fl in (0, 1):
                for fu in (0, 1):
                    for tr in (0, 1):
                        x, y = lr[i], hr[i]
                        if fl: x, y = x.flip(-1), y.flip(-1)
                        if fu: x, y = x.flip(-2), y.flip(-2)
                        if tr: x, y = x.transpose(-1, -2), y.transpose(-1, -2)
                        cands[(fl, fu, tr)] = (x, y)
            hit = [k for k, (x, y) in cands.items() if torch.equal(x, a[i]) and torch.equal(y, b[i])]
            assert len(hit) == 1
            seen.add(hit[0])
    assert len(seen) == 8


def test_synthetic_source_shapes_and_determinism():
    s1 = trainer.SyntheticPatchSource(3, 2, 2, patch=8, seed=5)
    s2 = trainer.SyntheticPatchSource(3, 2, 2, patch=8, seed=5)
    assert len(s1) == 3
    a, b = s1.get([2, 0])
    assert a.shape == (2, 1, 16, 16) and b.shape == (2, 1, 32, 32)
    assert torch.equal(a, s2.get([2, 0])[0]) and 0.0 <= float(b.min()) and float(b.max()) <= 1.0
    # LR is the box-downsampled HR, per view
    v = b[0, 0, :16, :16].reshape(8, 2, 8, 2).mean(dim=(1, 3))
    assert torch.allclose(v, a[0, 0, :8, :8], atol=1e-6)
