"""CPU: lft_amd/h5lite.py (the .h5 reader behind the reference's data loaders) against files written by the REAL
h5py 3.3.0 / libhdf5 1.10.6 (tests/golden/h5/*.h5, tools/gen_golden_h5.py, run under the image's /opt/conda Python 3.9) and against
what h5py's own ``np.array(hf.get(name))`` returned for each dataset of each file (tests/golden/h5/expected.npz)."""
import os

import numpy as np
import pytest

from lft_amd import h5lite

DIR = os.path.join(os.path.dirname(__file__), "golden", "h5")
EXP = np.load(os.path.join(DIR, "expected.npz"))
KEYS = sorted(k for k in EXP.files if ".h5:" in k)                    # (the aug: / flip: records belong to tests/test_datasets.py)


FILES = sorted({k.split(":")[0] for k in KEYS})


@pytest.mark.parametrize("fname", FILES)
def test_every_dataset_of_every_fixture_reads_as_h5py_read_it(fname):
    with h5lite.File(os.path.join(DIR, fname), "r") as hf:
        for key in (k for k in KEYS if k.startswith(fname + ":")):
            name = key.split(":", 1)[1]
            ds = hf.get(name)
            assert ds is not None, (fname, name, hf.keys())
            a, e = np.array(ds), EXP[key]
            assert a.shape == e.shape and a.dtype.itemsize == e.dtype.itemsize and a.dtype.kind == e.dtype.kind, key
            assert np.array_equal(a, e), key                               # bit-exact: bytes are moved, nothing is computed


def test_fixture_set_covers_the_formats_the_reader_claims():
    assert set(FILES) == {"train_000001.h5", "scene_rect.h5", "scene_a2_2x.h5", "chunked_gzip.h5", "latest.h5", "many.h5", "userblock.h5"}
    # superblock versions: 0 (earliest) and 3 (latest); the user-block file has its signature at offset 512
    sig = h5lite.SIGNATURE
    assert open(os.path.join(DIR, "train_000001.h5"), "rb").read(9) == sig + b"\x00"
    assert open(os.path.join(DIR, "latest.h5"), "rb").read(9) == sig + b"\x03"
    raw = open(os.path.join(DIR, "userblock.h5"), "rb").read(1024)
    assert raw[:8] != sig and raw[512:520] == sig


def test_h5py_spelling_keys_get_and_missing_names():
    with h5lite.File(os.path.join(DIR, "train_000001.h5")) as hf:
        assert sorted(hf.keys()) == ["Hr_SAI_y", "Lr_SAI_y"]
        assert "Lr_SAI_y" in hf and "nope" not in hf
        assert hf.get("nope") is None
        with pytest.raises(KeyError):
            hf["nope"]
        ds = hf["Lr_SAI_y"]
        assert ds.shape == (40, 40) and ds.dtype == np.float32 and ds.ndim == 2 and ds.size == 1600
        assert np.array_equal(ds[3:5, ::2], EXP["train_000001.h5:Lr_SAI_y"][3:5, ::2])
    with h5lite.File(os.path.join(DIR, "many.h5")) as hf:
        assert len(hf.keys()) == 203                                            # 200 datasets + group a + empty + compact
        assert hf["a"].keys() == ["b"] and sorted(hf["a/b"].keys()) == ["be", "i64", "u8"]
        assert np.array(hf.get("/a/b/be")).dtype == np.dtype(">f4")
        assert not np.array(hf["empty"]).any()


def test_errors_are_named_not_garbage(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(h5lite.H5Error, match="signature"):
        h5lite.File(str(p))
    p.write_bytes(b"")
    with pytest.raises(h5lite.H5Error, match="empty"):
        h5lite.File(str(p))
    with pytest.raises(h5lite.H5Error, match="read-only"):
        h5lite.File(os.path.join(DIR, "many.h5"), "w")
    # a feature outside the subset (the dataset's datatype is a shared, committed one): refused by name, not misread
    with h5lite.File(os.path.join(DIR, "unsupported_shared_dtype.h5")) as hf:
        assert sorted(hf.keys()) == ["mytype", "x"]
        with pytest.raises(h5lite.H5Error, match="shared header message"):
            hf["x"]
    # a truncated file: the header parses, the data is beyond the end
    raw = open(os.path.join(DIR, "train_000001.h5"), "rb").read()
    p.write_bytes(raw[:len(raw) // 2])
    with pytest.raises(h5lite.H5Error):
        with h5lite.File(str(p)) as hf:
            np.array(hf["Hr_SAI_y"]), np.array(hf["Lr_SAI_y"])


def _read_everything(path):
    with h5lite.File(path) as hf:
        def walk(g, depth=0):
            for k in g.keys():
                o = g[k]
                if isinstance(o, h5lite.Group):
                    if depth < 8:
                        walk(o, depth + 1)
                else:
                    np.array(o)
        walk(hf)


def test_damaged_files_end_in_h5error_never_in_a_hang_or_a_stray_exception(tmp_path):
    """Byte flips and truncations of the fixtures (seeded, a few hundred variants): reading everything either works or raises H5Error --
    no IndexError / zlib.error from the middle of the parser, no endless walk through a B-tree or continuation chain that points at
    itself, no attempt to allocate what a damaged shape field asks for."""
    import time
    rng = np.random.default_rng(12345)
    p = str(tmp_path / "damaged.h5")
    t0, n_err, n_ok = time.time(), 0, 0
    for fname in ("train_000001.h5", "chunked_gzip.h5", "latest.h5", "userblock.h5"):
        raw = open(os.path.join(DIR, fname), "rb").read()
        head = min(len(raw), 4096)                                      # the metadata lives in front: damage there is what reaches the parser
        for trial in range(60):
            b = bytearray(raw)
            if trial % 6 == 5:
                b = b[:int(rng.integers(8, len(raw)))]
            else:
                for _ in range(int(rng.integers(1, 5))):
                    b[int(rng.integers(0, min(head, len(b))))] = int(rng.integers(0, 256))
            open(p, "wb").write(bytes(b))
            try:
                _read_everything(p)
                n_ok += 1
            except h5lite.H5Error:
                n_err += 1
    assert n_err > 20 and n_ok + n_err == 240, (n_ok, n_err)
    assert time.time() - t0 < 60.0
