"""The reference's data loaders (utils/utils_datasets.py) on top of lft_amd.h5lite: same class names, constructor arguments,
directory convention and sample format, so that the reference's train.py / test.py loops run unchanged on this framework's data path.

    <path_for_train>/SR_{A}x{A}_{s}x/<dataset>/<000001>.h5     Lr_SAI_y [A*32, A*32], Hr_SAI_y [A*32*s, A*32*s]  float32
    <path_for_test>/SR_{A}x{A}_{s}x/<dataset>/<scene>.h5        Lr_SAI_y [A*w, A*h], Hr_SAI_y [A*w*s, A*h*s]      (transposed on load)

(MATLAB writes column-major, so h5py -- and h5lite -- see every array transposed; the reference undoes that for the test scenes only,
utils_datasets.py:88-89, and trains on transposed patches, where the random transposition of `augmentation` makes it immaterial.)
Samples are ``[1, H, W]`` float32 tensors, what ``torchvision.transforms.ToTensor`` makes of a float32 ``[H, W]`` array (no scaling:
ToTensor divides only integer images by 255); torchvision is not needed.

`augmentation` draws from Python's ``random`` in the reference's order (three draws per sample), so a seeded run selects the same
flips; `lft_amd.trainer.augment` is the on-device form of the same three operations for whole batches.  `H5PatchSource` feeds
`lft_amd.trainer.fit` straight from the training tree.
"""
from __future__ import annotations

import os
import random
from typing import List, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import h5lite


def to_tensor(a: np.ndarray) -> torch.Tensor:
    """torchvision's ToTensor for the arrays of these files: [H, W] (or [H, W, C]) -> [C, H, W]; integer images / 255."""
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
    return t.float().div(255) if t.dtype == torch.uint8 else t


def _tree(root: str, angRes: int, scale: int) -> str:
    return root + "SR_" + str(angRes) + "x" + str(angRes) + "_" + str(scale) + "x/"        # utils_datasets.py:17-18 (the path option ends in '/')


def _files(dataset_dir: str, data_list: Sequence[str]) -> List[str]:
    out: List[str] = []
    for data_name in data_list:                                                               # utils_datasets.py:26-31: os.listdir order
        out.extend(data_name + "/" + f for f in os.listdir(dataset_dir + data_name))
    return out


def read_pair(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """(Lr_SAI_y, Hr_SAI_y) as stored -- np.array(hf.get(...)) of utils_datasets.py:36-38."""
    with h5lite.File(path, "r") as hf:
        lr, hr = hf.get("Lr_SAI_y"), hf.get("Hr_SAI_y")
        if lr is None or hr is None:
            raise h5lite.H5Error(f"{path}: needs the datasets Lr_SAI_y and Hr_SAI_y, has {hf.keys()}")
        return np.array(lr), np.array(hr)


class TrainSetDataLoader(Dataset):
    """utils_datasets.py:14-48.  args: path_for_train, angRes, scale_factor, data_name ('ALL' or one dataset directory)."""

    def __init__(self, args):
        super().__init__()
        self.dataset_dir = _tree(args.path_for_train, args.angRes, args.scale_factor)
        self.data_list = os.listdir(self.dataset_dir) if args.data_name == "ALL" else [args.data_name]
        self.file_list = _files(self.dataset_dir, self.data_list)
        self.item_num = len(self.file_list)

    def __getitem__(self, index):
        data, label = read_pair(self.dataset_dir + self.file_list[index])
        data, label = augmentation(data, label)
        return to_tensor(data.copy()), to_tensor(label.copy())

    def __len__(self):
        return self.item_num


class TestSetDataLoader(Dataset):
    """utils_datasets.py:68-100: the scenes of ONE test dataset, transposed back to row-major on load."""

    def __init__(self, args, data_name="ALL"):
        super().__init__()
        self.dataset_dir = _tree(args.path_for_test, args.angRes, args.scale_factor)
        self.data_list = [data_name]
        self.file_list = _files(self.dataset_dir, self.data_list)
        self.item_num = len(self.file_list)

    def __getitem__(self, index):
        lr, hr = read_pair(self.dataset_dir + self.file_list[index])
        return to_tensor(np.transpose(lr, (1, 0)).copy()), to_tensor(np.transpose(hr, (1, 0)).copy())

    def __len__(self):
        return self.item_num


def MultiTestSetDataLoader(args):
    """utils_datasets.py:51-65: (dataset names, one batch-1 DataLoader per test dataset, total number of scenes)."""
    dataset_dir = _tree(args.path_for_test, args.angRes, args.scale_factor)
    data_list = os.listdir(dataset_dir)
    test_Loaders, length_of_tests = [], 0
    for data_name in data_list:
        test_Dataset = TestSetDataLoader(args, data_name)
        length_of_tests += len(test_Dataset)
        test_Loaders.append(DataLoader(dataset=test_Dataset, num_workers=getattr(args, "num_workers", 0), batch_size=1, shuffle=False))
    return data_list, test_Loaders, length_of_tests


def flip_SAI(data, angRes):
    """utils_datasets.py:103-113: the view grid AND every view reversed along both axes -- together that is the whole mosaic
    reversed along both axes (row index u*h + y -> (A-1-u)*h + (h-1-y) = H-1-row); returns [H, W, C] like the reference."""
    a = np.asarray(data)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.shape[0] % angRes or a.shape[1] % angRes:
        raise ValueError(f"flip_SAI: a {a.shape[0]} x {a.shape[1]} mosaic is not a grid of {angRes} x {angRes} views")
    return np.ascontiguousarray(a[::-1, ::-1, :])


def augmentation(data, label):
    """utils_datasets.py:116-126: three coin flips from ``random`` -- mirror columns, mirror rows, transpose (each acts on the view
    grid and on the views at once, which keeps the mosaic a light field)."""
    if random.random() < 0.5:
        data, label = data[:, ::-1], label[:, ::-1]
    if random.random() < 0.5:
        data, label = data[::-1, :], label[::-1, :]
    if random.random() < 0.5:
        data, label = data.transpose(1, 0), label.transpose(1, 0)
    return data, label


class H5PatchSource:
    """Patch source of lft_amd.trainer.fit over the reference's training tree: ``get(indices) -> (lr [n,1,A*p,A*p], hr [n,1,A*p*s,
    A*p*s])`` float32, samples as stored (the trainer augments on the device).  Files are read on demand; ``cache=True`` keeps them
    (a 5x5 2x sample is 0.13 MB)."""

    def __init__(self, path_for_train: str, angRes: int, scale: int, data_name: str = "ALL", cache: bool = False):
        self.dataset_dir = _tree(path_for_train, angRes, scale)
        data_list = sorted(os.listdir(self.dataset_dir)) if data_name == "ALL" else [data_name]
        self.file_list = []
        for d in data_list:                                      # sorted: every rank must see the same order
            self.file_list.extend(d + "/" + f for f in sorted(os.listdir(self.dataset_dir + d)))
        self._cache = {} if cache else None

    def __len__(self):
        return len(self.file_list)

    def _pair(self, i: int):
        if self._cache is not None and i in self._cache:
            return self._cache[i]
        p = read_pair(self.dataset_dir + self.file_list[i])
        if self._cache is not None:
            self._cache[i] = p
        return p

    def get(self, indices: Sequence[int]):
        pairs = [self._pair(int(i)) for i in indices]
        lr = torch.from_numpy(np.stack([p[0] for p in pairs]).astype(np.float32, copy=False))[:, None]
        hr = torch.from_numpy(np.stack([p[1] for p in pairs]).astype(np.float32, copy=False))[:, None]
        return lr, hr
