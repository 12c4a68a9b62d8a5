#!/opt/conda/bin/python3.9
"""Generate tests/golden/h5/*.h5 with the REAL h5py / libhdf5 (build container only).

The reference's data files are written by MATLAB (Generate_Data_for_Training.m:72-78, Generate_Data_for_Test.m:70-76:
``h5create(file, '/Lr_SAI_y', size(X), 'Datatype', 'single'); h5write(...)`` -- float32, contiguous, dimensions reversed
because MATLAB is column-major) and read with h5py (utils/utils_datasets.py:36-38, 85-87).  Neither MATLAB nor a data file
is in the image, and the interpreter the framework runs on has no h5py; the image's /opt/conda Python 3.9 has h5py 3.3.0 on
libhdf5 1.10.6.  This script runs under THAT interpreter and writes small files in the reference's format -- the two
datasets, float32 -- in every on-disk variant lft_amd/h5lite.py claims to read, plus `expected.npz` = what
``np.array(hf.get(name))`` returns for each of them (h5py reading its own files): data only.

    /opt/conda/bin/python3.9 tools/gen_golden_h5.py        (writes tests/golden/h5/)"""
import os

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "h5")


def sai_pair(rng, A, p, s):
    """A training sample as the MATLAB script lays it out: Hr_SAI_y [A*p*s, A*p*s], Lr_SAI_y [A*p, A*p] (view-major
    mosaics), values in [0, 1] -- stored transposed (column-major writer)."""
    hr = rng.random((A * p * s, A * p * s), dtype=np.float32)
    lr = hr.reshape(A, p, s, A, p, s).mean(axis=(2, 5)).reshape(A * p, A * p).astype(np.float32)
    return lr.T.copy(), hr.T.copy()


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(7)
    expected = {}

    def record(fname):
        with h5py.File(os.path.join(OUT, fname), "r") as hf:
            def visit(name, obj):
                if isinstance(obj, h5py.Dataset):
                    expected[fname + ":" + name] = np.array(obj)
            hf.visititems(visit)

    # 1. the reference's training sample as MATLAB writes it: earliest format, contiguous float32, two root datasets
    lr, hr = sai_pair(rng, 5, 8, 2)
    with h5py.File(os.path.join(OUT, "train_000001.h5"), "w", libver="earliest") as hf:
        hf.create_dataset("Lr_SAI_y", data=lr)
        hf.create_dataset("Hr_SAI_y", data=hr)
    record("train_000001.h5")
    # 2. a second sample (the loaders list a directory), non-square (a test scene: H != W), Hr written first as Generate_Data_for_Test.m does
    lr2 = rng.random((5 * 6, 5 * 10), dtype=np.float32)
    hr2 = rng.random((5 * 12, 5 * 20), dtype=np.float32)
    with h5py.File(os.path.join(OUT, "scene_rect.h5"), "w", libver="earliest") as hf:
        hf.create_dataset("Hr_SAI_y", data=hr2.T.copy())
        hf.create_dataset("Lr_SAI_y", data=lr2.T.copy())
    record("scene_rect.h5")
    # 3. chunked + deflate + shuffle + fletcher32 (what a re-packed / compressed copy of the data looks like), version-1 chunk B-tree,
    #    chunks that do not divide the shape
    with h5py.File(os.path.join(OUT, "chunked_gzip.h5"), "w", libver="earliest") as hf:
        hf.create_dataset("Lr_SAI_y", data=lr, chunks=(16, 12), compression="gzip", compression_opts=4, shuffle=True)
        hf.create_dataset("Hr_SAI_y", data=hr, chunks=(32, 32), compression="gzip", fletcher32=True)
        hf.create_dataset("plain_chunks", data=np.arange(7 * 9, dtype=np.float64).reshape(7, 9), chunks=(4, 4))
        hf.create_dataset("many_chunks", data=rng.random((40, 44), dtype=np.float32), chunks=(4, 4))     # 110 chunks: a two-level chunk B-tree
    record("chunked_gzip.h5")
    # 4. libver latest: superblock 3, version-2 object headers, link messages, layout version 4 (single chunk, implicit / fixed-array index)
    with h5py.File(os.path.join(OUT, "latest.h5"), "w", libver="latest") as hf:
        hf.create_dataset("Lr_SAI_y", data=lr)
        hf.create_dataset("Hr_SAI_y", data=hr, chunks=hr.shape, compression="gzip")                  # single chunk, filtered
        hf.create_dataset("fa", data=np.arange(20 * 6, dtype=np.int32).reshape(20, 6), chunks=(8, 4))   # fixed array index
        hf.create_dataset("fa_gz", data=rng.random((20, 6)), chunks=(8, 4), compression="gzip", shuffle=True)
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)                                                 # allocated early, no filter: implicit index
        dcpl.set_chunk((8, 4))
        dcpl.set_alloc_time(h5py.h5d.ALLOC_TIME_EARLY)
        did = h5py.h5d.create(hf.id, b"implicit", h5py.h5t.NATIVE_INT32, h5py.h5s.create_simple((20, 6)), dcpl=dcpl)
        did.write(h5py.h5s.ALL, h5py.h5s.ALL, np.arange(100, 220, dtype=np.int32).reshape(20, 6))
        g = hf.create_group("sub")
        g.create_dataset("x", data=np.arange(5, dtype=np.int16))
        g.create_dataset("scalar", data=np.float32(2.5))
    record("latest.h5")
    # 5. many datasets in one old-style group (B-tree with several symbol nodes), nested groups, integer / big-endian / float64 types,
    #    a compact dataset, an allocated-late dataset that was never written (reads as zeros)
    with h5py.File(os.path.join(OUT, "many.h5"), "w", libver="earliest") as hf:
        for i in range(200):                                                                             # enough for a two-level group B-tree
            hf.create_dataset(f"d{i:03d}", data=np.full((3,), i, dtype=np.float32))
        g = hf.create_group("a/b")
        g.create_dataset("be", data=np.arange(6, dtype=">f4").reshape(2, 3))
        g.create_dataset("u8", data=np.arange(10, dtype=np.uint8))
        g.create_dataset("i64", data=np.arange(-3, 3, dtype=np.int64))
        hf.create_dataset("empty", shape=(4, 5), dtype="f4")
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_layout(h5py.h5d.COMPACT)
        sid = h5py.h5s.create_simple((2, 2))
        did = h5py.h5d.create(hf.id, b"compact", h5py.h5t.NATIVE_FLOAT, sid, dcpl=dcpl)
        did.write(h5py.h5s.ALL, h5py.h5s.ALL, np.array([[1, 2], [3, 4]], dtype=np.float32))
    record("many.h5")
    # 6. a user block in front of the superblock (superblock at offset 512)
    with h5py.File(os.path.join(OUT, "userblock.h5"), "w", libver="earliest", userblock_size=512) as hf:
        hf.create_dataset("Lr_SAI_y", data=lr[:10, :10].copy())
    record("userblock.h5")
    # 6a. outside the reader's subset, to be refused by name: a dataset whose datatype is a shared (committed) one
    with h5py.File(os.path.join(OUT, "unsupported_shared_dtype.h5"), "w", libver="earliest") as hf:
        hf["mytype"] = np.dtype("f4")
        hf.create_dataset("x", data=np.arange(4, dtype="f4"), dtype=hf["mytype"])
    # 6b. a test scene large enough for the reference's 32 / 16 patch tiling (test.py:75-104): 2 x 2 views of 40 x 36 (LR), 2x,
    #     smooth content so that metrics mean something; written as Generate_Data_for_Test.m does (transposed, Hr first)
    yy, xx = np.meshgrid(np.linspace(0, 1, 80), np.linspace(0, 1, 72), indexing="ij")
    view_hr = 0.5 + 0.25 * np.sin(9 * yy + 4 * xx) * np.cos(7 * xx) + 0.1 * np.sin(31 * yy * xx)
    hr3 = np.zeros((2 * 80, 2 * 72), dtype=np.float32)
    for u in range(2):
        for v in range(2):
            hr3[u * 80:(u + 1) * 80, v * 72:(v + 1) * 72] = np.roll(view_hr, (u, 2 * v), axis=(0, 1))      # a little disparity between the views
    lr3 = hr3.reshape(2, 40, 2, 2, 36, 2).mean(axis=(2, 5)).reshape(80, 72).astype(np.float32)
    with h5py.File(os.path.join(OUT, "scene_a2_2x.h5"), "w", libver="earliest") as hf:
        hf.create_dataset("Hr_SAI_y", data=hr3.T.copy())
        hf.create_dataset("Lr_SAI_y", data=lr3.T.copy())
    record("scene_a2_2x.h5")
    # 7. the reference's own augmentation / flip_SAI (utils/utils_datasets.py:103-126: pure numpy + random) executed on the training
    #    sample: only those two function definitions are compiled out of the file (its imports -- torch, torchvision, skimage -- are
    #    not available to this interpreter); inputs are the arrays above, the `random` seed is the key
    import ast
    import random
    tree = ast.parse(open("/root/reference/utils/utils_datasets.py").read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("augmentation", "flip_SAI")]
    ns = {"random": random, "np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "reference_utils_datasets_subset", "exec"), ns)
    for seed in range(8):                                    # 8 seeds: all 2^3 flip combinations occur (checked below)
        random.seed(seed)
        d, l = ns["augmentation"](lr2.T.copy(), hr2.T.copy())                   # the non-square pair: a transposition shows in the shape
        expected[f"aug:{seed}:data"], expected[f"aug:{seed}:label"] = np.ascontiguousarray(d), np.ascontiguousarray(l)
        expected[f"aug:{seed}:draws_after"] = np.float64(random.random())       # the stream position after the call (three draws consumed)
    expected["flip:2d"] = np.ascontiguousarray(ns["flip_SAI"](lr, 5))
    expected["flip:3d"] = np.ascontiguousarray(ns["flip_SAI"](np.stack([hr, 1 - hr], axis=-1), 5))
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
    for f in sorted(os.listdir(OUT)):
        print(f"{f:22s} {os.path.getsize(os.path.join(OUT, f)):8d} bytes")
    print("h5py", h5py.__version__, "libhdf5", h5py.version.hdf5_version)


if __name__ == "__main__":
    main()
