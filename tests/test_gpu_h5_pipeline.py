"""GPU: the reference's file-based loops on this framework, end to end from .h5 files written by the real h5py (tests/golden/h5/):
test.py's loop over a test tree (lft_amd.evaluate.test_sets) and a training epoch fed from a training tree (H5PatchSource ->
trainer.fit)."""
import os
import shutil
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from lft_amd import datasets, evaluate, trainer
from model import LFT as MODEL

pytestmark = pytest.mark.gpu
DIR = os.path.join(os.path.dirname(__file__), "golden", "h5")
EXP = np.load(os.path.join(DIR, "expected.npz"))


def test_test_sets_over_h5_tree_equals_the_in_memory_evaluation(tmp_path):
    base = tmp_path / "data_for_test" / "SR_2x2_2x"
    for dset, names in {"SetA": ["s1.h5", "s2.h5"], "SetB": ["s1.h5"]}.items():
        os.makedirs(base / dset)
        for n in names:
            shutil.copy(os.path.join(DIR, "scene_a2_2x.h5"), base / dset / n)
    args = SimpleNamespace(path_for_test=str(tmp_path / "data_for_test") + "/", angRes=2, scale_factor=2, channels=64, num_workers=0,
                           patch_size_for_test=32, stride_for_test=16)
    torch.manual_seed(0)
    net = MODEL.get_model(args).cuda()
    net.apply(MODEL.weights_init)
    lines = []
    res = evaluate.test_sets(net, args, log=lines.append)
    assert sorted(res) == ["SetA", "SetB"] and len(lines) == 2 and all(l.startswith("Test on Set") for l in lines)
    # the same scene given as arrays (what h5py read, transposed as utils_datasets.py:88-89 does)
    lr = torch.from_numpy(EXP["scene_a2_2x.h5:Lr_SAI_y"].T.copy())
    hr = torch.from_numpy(EXP["scene_a2_2x.h5:Hr_SAI_y"].T.copy())
    assert tuple(lr.shape) == (80, 72) and tuple(hr.shape) == (160, 144)
    p, s, sr = evaluate.test_scene(net, lr, hr)
    assert tuple(sr.shape) == (160, 144)
    for name in res:
        assert abs(res[name][0] - p) < 1e-4 and abs(res[name][1] - s) < 1e-6, (res, p, s)
    assert 5.0 < p < 60.0 and 0.0 < s < 1.0


def test_training_epoch_from_an_h5_training_tree(tmp_path):
    base = tmp_path / "data_for_train" / "SR_5x5_2x"
    os.makedirs(base / "A")
    os.makedirs(base / "B")
    for i, (d, fixture) in enumerate([("A", "train_000001.h5"), ("A", "chunked_gzip.h5"), ("B", "latest.h5"), ("B", "train_000001.h5")]):
        shutil.copy(os.path.join(DIR, fixture), base / d / ("%06d.h5" % (i + 1)))
    src = datasets.H5PatchSource(str(tmp_path / "data_for_train") + "/", 5, 2, cache=True)
    assert len(src) == 4
    args = SimpleNamespace(angRes=5, scale_factor=2, channels=64)
    torch.manual_seed(0)
    net = MODEL.get_model(args).cuda()
    net.apply(MODEL.weights_init)
    hist = trainer.fit(net, src, epochs=3, batch_size=2, lr=5e-4, ckpt_dir=str(tmp_path / "ckpt"), model_name="LFT", log=lambda *_: None)
    assert len(hist) == 3 and all(np.isfinite(hist)) and hist[-1] < hist[0], hist            # (the four files hold the same pair)
    assert sorted(os.listdir(tmp_path / "ckpt")) == ["LFT_5x5_2x_epoch_%02d_model.pth" % e for e in (1, 2, 3)]
