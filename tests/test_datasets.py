"""CPU: lft_amd/datasets.py (the reference's loaders, utils/utils_datasets.py, on lft_amd.h5lite) -- SURVEY.md section 8 f-3.
Pinned by tests/golden/h5/: files written by the real h5py, the arrays h5py read back from them, and the outputs of the reference's
own `augmentation` / `flip_SAI` function bodies on those arrays (tools/gen_golden_h5.py)."""
import os
import random
import shutil
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from lft_amd import datasets, h5lite

DIR = os.path.join(os.path.dirname(__file__), "golden", "h5")
EXP = np.load(os.path.join(DIR, "expected.npz"))


def make_tree(root, kind, layout):
    """<root>/data_for_<kind>/SR_5x5_2x/<dataset>/<file>.h5 from the fixture files; layout = {dataset: [(fixture, name), ...]}"""
    base = os.path.join(root, f"data_for_{kind}", "SR_5x5_2x")
    for dset, files in layout.items():
        os.makedirs(os.path.join(base, dset))
        for fixture, name in files:
            shutil.copy(os.path.join(DIR, fixture), os.path.join(base, dset, name))
    return os.path.join(root, f"data_for_{kind}") + "/"


def test_augmentation_and_flip_match_the_reference_function_bodies():
    lr, hr = EXP["scene_rect.h5:Lr_SAI_y"], EXP["scene_rect.h5:Hr_SAI_y"]
    combos = set()
    for seed in range(8):
        random.seed(seed)
        d, l = datasets.augmentation(lr, hr)
        assert np.array_equal(d, EXP[f"aug:{seed}:data"]) and np.array_equal(l, EXP[f"aug:{seed}:label"])
        assert random.random() == float(EXP[f"aug:{seed}:draws_after"])               # exactly three draws consumed, as the reference
        combos.add((d.shape, float(d[0, 0]), float(d[-1, 0])))
    assert len(combos) >= 6                                                             # the seeds exercise (nearly) all flip combinations
    assert np.array_equal(datasets.flip_SAI(EXP["train_000001.h5:Lr_SAI_y"], 5), EXP["flip:2d"])
    hr5 = EXP["train_000001.h5:Hr_SAI_y"]
    assert np.array_equal(datasets.flip_SAI(np.stack([hr5, 1 - hr5], axis=-1), 5), EXP["flip:3d"])
    with pytest.raises(ValueError):
        datasets.flip_SAI(np.zeros((7, 10), dtype=np.float32), 5)


def test_train_loader_tree_samples_and_augmentation(tmp_path):
    root = make_tree(str(tmp_path), "train", {"SetA": [("train_000001.h5", "000001.h5"), ("chunked_gzip.h5", "000002.h5")],
                                              "SetB": [("latest.h5", "000001.h5")]})
    args = SimpleNamespace(path_for_train=root, angRes=5, scale_factor=2, data_name="ALL")
    ds = datasets.TrainSetDataLoader(args)
    assert ds.dataset_dir == root + "SR_5x5_2x/" and sorted(ds.data_list) == ["SetA", "SetB"] and len(ds) == ds.item_num == 3
    assert sorted(ds.file_list) == ["SetA/000001.h5", "SetA/000002.h5", "SetB/000001.h5"]
    lr0, hr0 = EXP["train_000001.h5:Lr_SAI_y"], EXP["train_000001.h5:Hr_SAI_y"]      # all three files hold the same pair, in three on-disk formats
    for i in range(3):
        random.seed(5)
        d, l = ds[i]
        random.seed(5)
        ed, el = datasets.augmentation(lr0, hr0)
        assert d.dtype == torch.float32 and tuple(d.shape) == (1, 40, 40) and tuple(l.shape) == (1, 80, 80)   # ToTensor of a float32 [H, W] array
        assert np.array_equal(d[0].numpy(), ed) and np.array_equal(l[0].numpy(), el)
    one = datasets.TrainSetDataLoader(SimpleNamespace(path_for_train=root, angRes=5, scale_factor=2, data_name="SetB"))
    assert one.data_list == ["SetB"] and one.file_list == ["SetB/000001.h5"]
    # through torch's DataLoader, as train.py:86-87 uses it
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, num_workers=0)
    random.seed(0)
    data, label = next(iter(loader))
    assert tuple(data.shape) == (3, 1, 40, 40) and tuple(label.shape) == (3, 1, 80, 80)


def test_test_loaders_transpose_and_multi(tmp_path):
    root = make_tree(str(tmp_path), "test", {"Scenes": [("scene_rect.h5", "rect.h5")], "Patches": [("train_000001.h5", "a.h5"), ("latest.h5", "b.h5")]})
    args = SimpleNamespace(path_for_test=root, angRes=5, scale_factor=2, num_workers=0)
    names, loaders, total = datasets.MultiTestSetDataLoader(args)
    assert sorted(names) == ["Patches", "Scenes"] and total == 3 and len(loaders) == 2
    by = dict(zip(names, loaders))
    (lr, hr), = list(by["Scenes"])
    assert tuple(lr.shape) == (1, 1, 30, 50) and tuple(hr.shape) == (1, 1, 60, 100)                     # stored [50, 30] / [100, 60]: transposed on load
    assert np.array_equal(lr[0, 0].numpy(), EXP["scene_rect.h5:Lr_SAI_y"].T) and np.array_equal(hr[0, 0].numpy(), EXP["scene_rect.h5:Hr_SAI_y"].T)
    assert len(list(by["Patches"])) == 2
    single = datasets.TestSetDataLoader(args, "Scenes")
    assert single.data_list == ["Scenes"] and len(single) == 1


def test_patch_source_feeds_the_trainer_protocol(tmp_path):
    root = make_tree(str(tmp_path), "train", {"B": [("latest.h5", "000001.h5")], "A": [("train_000001.h5", "000002.h5"), ("chunked_gzip.h5", "000001.h5")]})
    src = datasets.H5PatchSource(root, 5, 2, cache=True)
    assert len(src) == 3 and src.file_list == ["A/000001.h5", "A/000002.h5", "B/000001.h5"]           # sorted: identical on every rank
    lr, hr = src.get([2, 0])
    assert tuple(lr.shape) == (2, 1, 40, 40) and tuple(hr.shape) == (2, 1, 80, 80) and lr.dtype == torch.float32
    assert np.array_equal(lr[0, 0].numpy(), EXP["latest.h5:Lr_SAI_y"]) and np.array_equal(hr[1, 0].numpy(), EXP["chunked_gzip.h5:Hr_SAI_y"])
    assert src.get([2])[0].data_ptr() != lr.data_ptr()


def test_missing_dataset_is_named(tmp_path):
    root = make_tree(str(tmp_path), "train", {"X": [("many.h5", "000001.h5")]})
    ds = datasets.TrainSetDataLoader(SimpleNamespace(path_for_train=root, angRes=5, scale_factor=2, data_name="ALL"))
    with pytest.raises(h5lite.H5Error, match="Lr_SAI_y"):
        ds[0]
