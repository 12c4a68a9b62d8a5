"""A minimal read-only HDF5 reader (numpy + zlib only) for the data files of the reference: one flat file, a few numeric
datasets in the root group or in sub-groups -- what MATLAB's ``h5create`` / ``h5write`` (Generate_Data_for_Training.m:72-78,
Generate_Data_for_Test.m:70-76) and h5py's ``create_dataset`` produce.  The reference reads them with
``np.array(h5py.File(name, 'r').get('Lr_SAI_y'))`` (utils/utils_datasets.py:36-38, 85-87); the interpreter this framework runs
on has no h5py, so the bytes are parsed here, following the HDF5 File Format Specification (version 3.0):

  * superblock versions 0 / 1 (MATLAB, h5py default) and 2 / 3 (``libver='latest'``), at offset 0 or 512 * 2^n (user block:
    addresses are relative to the superblock's base-address field);
  * groups: symbol-table groups (version-1 B-tree + local heap + SNOD nodes) and compact new-style groups (link messages);
    densely stored links (fractal heap: groups with more than 8 links written with libver latest) are refused by name;
  * object headers version 1 and 2, with continuation blocks;
  * dataspace versions 1 / 2 (simple and scalar), datatypes: fixed-point and floating-point of 1 / 2 / 4 / 8 bytes, either
    byte order;
  * data layout versions 1-3: compact, contiguous, chunked (version-1 B-tree); version 4: single-chunk, implicit and
    fixed-array chunk indices (what libver latest writes for fixed-size datasets); unallocated storage reads as the fill value
    (zero);
  * filters: deflate, shuffle, fletcher32 (checksum stripped, not verified).
Anything else raises ``H5Error`` naming the feature.  ``File(path).get(name)`` returns an object whose ``np.array(...)``
/ ``[()]`` is the dataset in C order with the stored dtype -- the two forms the reference uses.

Pinned by files written with the real h5py / libhdf5 (tests/golden/h5/, tools/gen_golden_h5.py; tests/test_h5lite.py)."""
from __future__ import annotations

import mmap
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5Error(Exception):
    pass


def _guard(fn, what: str):
    """A damaged file must end in H5Error, not in an index / value / zlib error from the middle of the parser (or a recursion
    without end through a B-tree that points at itself)."""
    try:
        return fn()
    except H5Error:
        raise
    except (IndexError, ValueError, OverflowError, KeyError, TypeError, RecursionError, MemoryError, zlib.error, UnicodeDecodeError) as e:
        raise H5Error(f"{what}: damaged or unsupported file ({type(e).__name__}: {e})") from e


class _Buf:
    """Little-endian cursor over the file image."""

    def __init__(self, data, pos: int = 0):
        self.d, self.p = data, pos

    def u(self, n: int) -> int:
        v = int.from_bytes(self.d[self.p:self.p + n], "little")
        self.p += n
        return v

    def raw(self, n: int) -> bytes:
        b = bytes(self.d[self.p:self.p + n])
        self.p += n
        return b

    def skip(self, n: int) -> None:
        self.p += n

    def align(self, a: int, base: int = 0) -> None:
        self.p = base + ((self.p - base + a - 1) // a) * a


class Dataset:
    def __init__(self, f: "File", name: str, shape: Tuple[int, ...], dtype: np.dtype, layout: dict, filters: List[Tuple[int, List[int]]]):
        self._f, self.name, self.shape, self.dtype, self._layout, self._filters = f, name, tuple(shape), dtype, layout, filters

    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1

    def read(self) -> np.ndarray:
        return _guard(lambda: self._f._read_dataset(self), self.name)

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a.astype(dtype) if dtype is not None and np.dtype(dtype) != a.dtype else a

    def __getitem__(self, key):
        return self.read()[key]

    def __repr__(self):
        return f"<h5lite.Dataset {self.name!r} shape {self.shape} dtype {self.dtype}>"


class Group:
    def __init__(self, f: "File", name: str, links: Dict[str, int]):
        self._f, self.name, self._links = f, name, links

    def keys(self):
        return list(self._links)

    def __contains__(self, k):
        return k in self._links

    def get(self, path: str, default=None):
        try:
            return self[path]
        except KeyError:
            return default

    def __getitem__(self, path: str):
        node = self
        parts = [p for p in path.split("/") if p]
        for i, part in enumerate(parts):
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(path)
            name = "/" + "/".join(parts[:i + 1])
            addr = node._links[part]
            node = _guard(lambda: self._f._object(addr, name), name)
        return node


class File(Group):
    """``with File(path) as hf: a = np.array(hf.get('Lr_SAI_y'))`` -- h5py's spelling, read-only."""

    def __init__(self, path: str, mode: str = "r"):
        if mode != "r":
            raise H5Error("h5lite is read-only")
        self._fh = open(path, "rb")
        try:
            self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._fh.close()
            raise H5Error(f"{path}: empty file")
        self._cache: Dict[int, object] = {}
        try:
            root = _guard(self._superblock, path)
            Group.__init__(self, self, "/", {})
            rootobj = _guard(lambda: self._object(root, "/"), path)
            if not isinstance(rootobj, Group):
                raise H5Error(f"{path}: the root object is not a group")
            self._links = rootobj._links
        except Exception:
            self.close()
            raise

    # ---- life cycle ----
    def close(self):
        if getattr(self, "_mm", None) is not None:
            self._mm.close()
            self._mm = None
        if getattr(self, "_fh", None) is not None:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- superblock ----
    def _superblock(self) -> int:
        d, off = self._mm, 0
        while off + 8 <= len(d) and d[off:off + 8] != SIGNATURE:
            off = 512 if off == 0 else off * 2
        if off + 8 > len(d):
            raise H5Error("not an HDF5 file (no signature)")
        b = _Buf(d, off + 8)
        ver = b.u(1)
        if ver in (0, 1):
            b.skip(4)                                   # free-space version, root-entry version, reserved, shared-header version
            self.O, self.L = b.u(1), b.u(1)
            b.skip(1)
            b.skip(4)                                   # group leaf / internal node K
            b.skip(4)                                   # consistency flags
            if ver == 1:
                b.skip(4)                               # indexed-storage internal node K, reserved
            self.base = b.u(self.O)
            b.skip(3 * self.O)                          # free-space info, end of file, driver info
            b.skip(self.O)                              # root symbol-table entry: link name offset
            root = b.u(self.O)
        elif ver in (2, 3):
            self.O, self.L = b.u(1), b.u(1)
            b.skip(1)
            self.base = b.u(self.O)
            b.skip(2 * self.O)                          # superblock extension, end of file
            root = b.u(self.O)
        else:
            raise H5Error(f"superblock version {ver} not supported")
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise H5Error(f"offset / length sizes {self.O} / {self.L} not supported")
        self.undef = (1 << (8 * self.O)) - 1
        return root

    def _at(self, addr: int) -> _Buf:
        if addr == self.undef or addr + self.base >= len(self._mm):
            raise H5Error(f"address {addr:#x} outside the file")
        return _Buf(self._mm, addr + self.base)

    # ---- object headers ----
    def _messages(self, addr: int) -> List[Tuple[int, bytes]]:
        b = self._at(addr)
        msgs: List[Tuple[int, bytes]] = []
        if bytes(self._mm[b.p:b.p + 4]) == b"OHDR":
            b.skip(4)
            if b.u(1) != 2:
                raise H5Error("object header version")
            flags = b.u(1)
            if flags & 0x20:
                b.skip(16)
            if flags & 0x10:
                b.skip(4)
            size = b.u(1 << (flags & 3))
            blocks = [(b.p, size)]
            nblocks = 0
            while blocks:
                nblocks += 1
                if nblocks > 4096:
                    raise H5Error("object header continuation chain without end")
                p, n = blocks.pop(0)
                if n < 0 or p + n > len(self._mm):
                    raise H5Error(f"object header at {addr:#x}: a message block of {n} bytes at {p:#x} lies outside the file")
                c = _Buf(self._mm, p)
                end = p + n
                while c.p + 4 <= end:
                    t, sz, mfl = c.u(1), c.u(2), c.u(1)
                    if flags & 0x04:
                        c.skip(2)
                    body = c.raw(sz)
                    self._no_shared(t, mfl, addr)
                    if t == 0x10:
                        q = _Buf(body)
                        a, ln = q.u(self.O), q.u(self.L)
                        if bytes(self._mm[a + self.base:a + self.base + 4]) != b"OCHK":
                            raise H5Error("object header continuation without OCHK")
                        blocks.append((a + self.base + 4, ln - 8))           # signature in front, checksum behind
                    elif t != 0:
                        msgs.append((t, body))
            return msgs
        ver = b.u(1)
        if ver != 1:
            raise H5Error(f"object header version {ver} at {addr:#x}")
        b.skip(1)
        nmsg = b.u(2)
        b.skip(4)
        size = b.u(4)
        b.align(8, addr + self.base)
        blocks = [(b.p, size)]
        nblocks = 0
        while blocks and len(msgs) < nmsg + 64:
            nblocks += 1
            if nblocks > 4096:
                raise H5Error("object header continuation chain without end")
            p, n = blocks.pop(0)
            if n < 0 or p + n > len(self._mm):
                raise H5Error(f"object header at {addr:#x}: a message block of {n} bytes at {p:#x} lies outside the file")
            c = _Buf(self._mm, p)
            while c.p + 8 <= p + n:
                t, sz, mfl = c.u(2), c.u(2), c.u(1)
                c.skip(3)
                body = c.raw(sz)
                self._no_shared(t, mfl, addr)
                if t == 0x10:
                    q = _Buf(body)
                    a, ln = q.u(self.O), q.u(self.L)
                    blocks.append((a + self.base, ln))
                elif t != 0:
                    msgs.append((t, body))
        return msgs

    @staticmethod
    def _no_shared(t: int, mflags: int, addr: int) -> None:
        # a shared message's body is a reference to the real message (committed datatypes, shared-message heaps): not in the subset
        if mflags & 0x02 and t in (0x0001, 0x0003, 0x0005, 0x0008, 0x000B):
            raise H5Error(f"object header at {addr:#x}: shared header message (type {t:#x}) not supported")

    def _object(self, addr: int, name: str):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._messages(addr)
        types = {t for t, _ in msgs}
        if 0x0001 in types and 0x0003 in types and 0x0008 in types:
            obj = self._dataset(msgs, name)
        else:
            obj = Group(self, name, self._group_links(msgs, name))
        self._cache[addr] = obj
        return obj

    # ---- groups ----
    def _group_links(self, msgs, name: str) -> Dict[str, int]:
        links: Dict[str, int] = {}
        for t, body in msgs:
            b = _Buf(body)
            if t == 0x0011:                                            # symbol table: B-tree + local heap
                btree, heap = b.u(self.O), b.u(self.O)
                h = self._at(heap)
                if h.raw(4) != b"HEAP":
                    raise H5Error("local heap signature")
                h.skip(4)
                h.skip(2 * self.L)
                seg = h.u(self.O) + self.base
                self._walk_group_btree(btree, seg, links)
            elif t == 0x0006:                                          # link message (compact new-style group)
                if b.u(1) != 1:
                    raise H5Error("link message version")
                fl = b.u(1)
                ltype = b.u(1) if fl & 0x08 else 0
                if fl & 0x04:
                    b.skip(8)
                if fl & 0x10:
                    b.skip(1)
                n = b.u(1 << (fl & 3))
                nm = b.raw(n).decode("utf-8")
                if ltype == 0:
                    links[nm] = b.u(self.O)                            # (soft / external links are not followed: absent from keys())
            elif t == 0x0002:                                          # link info: dense storage?
                if b.u(1) != 0:
                    raise H5Error("link info version")
                fl = b.u(1)
                if fl & 1:
                    b.skip(8)
                if b.u(self.O) != self.undef:
                    raise H5Error(f"group {name}: densely stored links (fractal heap) are not supported")
        return links

    def _walk_group_btree(self, addr: int, heap_seg: int, links: Dict[str, int]) -> None:
        b = self._at(addr)
        sig = b.raw(4)
        if sig == b"SNOD":
            b.skip(2)
            n = b.u(2)
            for _ in range(n):
                name_off, obj = b.u(self.O), b.u(self.O)
                b.skip(24)
                s = heap_seg + name_off
                e = self._mm.find(b"\0", s)
                links[bytes(self._mm[s:e]).decode("utf-8")] = obj
            return
        if sig != b"TREE":
            raise H5Error("group B-tree signature")
        if b.u(1) != 0:
            raise H5Error("group B-tree node type")
        b.skip(1)
        n = b.u(2)
        b.skip(2 * self.O)
        b.skip(self.L)                                                 # key 0
        for _ in range(n):
            child = b.u(self.O)
            b.skip(self.L)
            self._walk_group_btree(child, heap_seg, links)

    # ---- datasets ----
    def _dataset(self, msgs, name: str) -> Dataset:
        shape: Tuple[int, ...] = ()
        dtype = None
        layout: Optional[dict] = None
        filters: List[Tuple[int, List[int]]] = []
        for t, body in msgs:
            b = _Buf(body)
            if t == 0x0001:
                ver, rank, fl = b.u(1), b.u(1), b.u(1)
                if ver == 1:
                    b.skip(5)
                elif ver == 2:
                    if b.u(1) == 2:
                        raise H5Error(f"{name}: null dataspace")
                else:
                    raise H5Error(f"{name}: dataspace version {ver}")
                shape = tuple(b.u(self.L) for _ in range(rank))
            elif t == 0x0003:
                cv = b.u(1)
                cls, bits = cv & 15, b.u(3)
                size = b.u(4)
                order = ">" if bits & 1 else "<"
                if cls == 0 and size in (1, 2, 4, 8):
                    dtype = np.dtype(f"{order}{'i' if bits & 8 else 'u'}{size}")
                elif cls == 1 and size in (2, 4, 8):
                    dtype = np.dtype(f"{order}f{size}")
                else:
                    raise H5Error(f"{name}: datatype class {cls} of {size} bytes not supported")
            elif t == 0x0008:
                layout = self._layout(b, name)
            elif t == 0x000B:
                ver, nf = b.u(1), b.u(1)
                if ver == 1:
                    b.skip(6)
                elif ver != 2:
                    raise H5Error(f"{name}: filter pipeline version {ver}")
                for _ in range(nf):
                    fid = b.u(2)
                    nlen = b.u(2) if ver == 1 or fid >= 256 else 0
                    b.skip(2)
                    nv = b.u(2)
                    b.skip((nlen + 7) // 8 * 8 if ver == 1 else nlen)
                    vals = [b.u(4) for _ in range(nv)]
                    if ver == 1 and nv % 2:
                        b.skip(4)
                    filters.append((fid, vals))
        if dtype is None or layout is None:
            raise H5Error(f"{name}: incomplete dataset header")
        for fid, _ in filters:
            if fid not in (1, 2, 3):
                raise H5Error(f"{name}: filter {fid} not supported (deflate, shuffle, fletcher32 are)")
        return Dataset(self, name, shape, dtype, layout, filters)

    def _layout(self, b: _Buf, name: str) -> dict:
        ver = b.u(1)
        if ver in (1, 2):
            nd, cls = b.u(1), b.u(1)
            b.skip(5)
            addr = b.u(self.O) if cls != 0 else None
            dims = [b.u(4) for _ in range(nd)]
            if cls == 2:
                return {"cls": 2, "btree": addr, "chunk": dims[:-1] if len(dims) > 1 else dims, "index": "btree1"}
            if cls == 1:
                return {"cls": 1, "addr": addr, "size": None}
            size = b.u(4)
            return {"cls": 0, "data": b.raw(size)}
        if ver == 3:
            cls = b.u(1)
            if cls == 0:
                size = b.u(2)
                return {"cls": 0, "data": b.raw(size)}
            if cls == 1:
                return {"cls": 1, "addr": b.u(self.O), "size": b.u(self.L)}
            if cls == 2:
                nd = b.u(1)
                addr = b.u(self.O)
                dims = [b.u(4) for _ in range(nd)]
                return {"cls": 2, "btree": addr, "chunk": dims[:-1], "index": "btree1"}
            raise H5Error(f"{name}: layout class {cls}")
        if ver == 4:
            cls = b.u(1)
            if cls == 0:
                size = b.u(2)
                return {"cls": 0, "data": b.raw(size)}
            if cls == 1:
                return {"cls": 1, "addr": b.u(self.O), "size": b.u(self.L)}
            if cls != 2:
                raise H5Error(f"{name}: layout class {cls} (virtual datasets are not supported)")
            fl, nd, enc = b.u(1), b.u(1), b.u(1)
            dims = [b.u(enc) for _ in range(nd)]
            itype = b.u(1)
            lay = {"cls": 2, "chunk": dims[:-1], "flags": fl}
            if itype == 1:                                             # single chunk
                if fl & 2:
                    lay["fsize"], lay["fmask"] = b.u(self.L), b.u(4)
                lay["index"], lay["addr"] = "single", b.u(self.O)
            elif itype == 2:                                           # implicit: chunks back to back, no filters
                lay["index"], lay["addr"] = "implicit", b.u(self.O)
            elif itype == 3:                                           # fixed array
                lay["page_bits"] = b.u(1)
                lay["index"], lay["addr"] = "farray", b.u(self.O)
            else:
                raise H5Error(f"{name}: chunk index type {itype} (extensible array / version-2 B-tree: resizable datasets) not supported")
            return lay
        raise H5Error(f"{name}: data layout version {ver}")

    def _unfilter(self, raw: bytes, ds: Dataset, mask: int) -> bytes:
        for i in range(len(ds._filters) - 1, -1, -1):
            if mask & (1 << i):
                continue
            fid = ds._filters[i][0]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = ds.dtype.itemsize
                n = len(raw) // es
                a = np.frombuffer(raw, dtype=np.uint8, count=n * es).reshape(es, n).T
                raw = np.ascontiguousarray(a).tobytes() + raw[n * es:]
            elif fid == 3:
                raw = raw[:-4]
        return raw

    def _chunks_btree1(self, addr: int, nd: int, out: List[Tuple[Tuple[int, ...], int, int, int]]) -> None:
        if addr == self.undef:
            return
        b = self._at(addr)
        if b.raw(4) != b"TREE" or b.u(1) != 1:
            raise H5Error("chunk B-tree node")
        level, n = b.u(1), b.u(2)
        b.skip(2 * self.O)
        for _ in range(n):
            size, mask = b.u(4), b.u(4)
            offs = tuple(b.u(8) for _ in range(nd + 1))[:nd]
            child = b.u(self.O)
            if level:
                self._chunks_btree1(child, nd, out)
            else:
                out.append((offs, child, size, mask))

    def _read_dataset(self, ds: Dataset) -> np.ndarray:
        lay, n, es = ds._layout, ds.size, ds.dtype.itemsize
        if n * es > max(1 << 20, 4096 * len(self._mm)):                # (deflate cannot expand by more than ~1000x)
            raise H5Error(f"{ds.name}: shape {ds.shape} is not plausible for a file of {len(self._mm)} bytes (damaged header?)")
        if lay["cls"] == 0:
            return np.frombuffer(lay["data"], dtype=ds.dtype, count=n).reshape(ds.shape).copy()
        if lay["cls"] == 1:
            if lay["addr"] == self.undef:                              # never written: fill value
                return np.zeros(ds.shape, dtype=ds.dtype)
            p = lay["addr"] + self.base
            if p + n * es > len(self._mm):
                raise H5Error(f"{ds.name}: data beyond the end of the file (truncated?)")
            return np.frombuffer(self._mm, dtype=ds.dtype, count=n, offset=p).reshape(ds.shape).copy()
        chunk = tuple(lay["chunk"])
        nd = len(ds.shape)
        if len(chunk) != nd:
            raise H5Error(f"{ds.name}: chunk rank {len(chunk)} for a dataset of rank {nd}")
        csize = int(np.prod(chunk)) * es
        grid = [(-(-s // c)) for s, c in zip(ds.shape, chunk)]
        entries: List[Tuple[Tuple[int, ...], int, int, int]] = []
        if lay["index"] == "btree1":
            self._chunks_btree1(lay["btree"], nd, entries)
        elif lay["index"] == "single":
            if lay["addr"] != self.undef:
                entries.append(((0,) * nd, lay["addr"], lay.get("fsize", csize), lay.get("fmask", 0)))
        elif lay["index"] == "implicit":
            if lay["addr"] != self.undef:
                for i, idx in enumerate(np.ndindex(*grid)):
                    entries.append((tuple(k * c for k, c in zip(idx, chunk)), lay["addr"] + i * csize, csize, 0))
        elif lay["index"] == "farray":
            self._chunks_farray(lay, ds, grid, chunk, csize, entries)
        out = np.zeros(ds.shape, dtype=ds.dtype)
        for offs, addr, size, mask in entries:
            p = addr + self.base
            raw = bytes(self._mm[p:p + size])
            if ds._filters:
                raw = self._unfilter(raw, ds, mask)
            if len(raw) < csize:
                raise H5Error(f"{ds.name}: short chunk at {offs}")
            c = np.frombuffer(raw, dtype=ds.dtype, count=csize // es).reshape(chunk)
            sl = tuple(slice(o, min(o + k, s)) for o, k, s in zip(offs, chunk, ds.shape))
            out[sl] = c[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def _chunks_farray(self, lay, ds, grid, chunk, csize, entries) -> None:
        if lay["addr"] == self.undef:
            return
        b = self._at(lay["addr"])
        if b.raw(4) != b"FAHD":
            raise H5Error("fixed array header")
        b.skip(1)
        client = b.u(1)                                                # 0: unfiltered chunks, 1: filtered
        esz, page_bits = b.u(1), b.u(1)
        nent = b.u(self.L)
        dblk = b.u(self.O)
        if dblk == self.undef:
            return
        d = self._at(dblk)
        if d.raw(4) != b"FADB":
            raise H5Error("fixed array data block")
        d.skip(2)
        d.skip(self.O)                                                 # header address
        if nent > (1 << page_bits):
            raise H5Error(f"{ds.name}: paged fixed-array chunk index ({nent} chunks) not supported")
        for i, idx in enumerate(np.ndindex(*grid)):
            if i >= nent:
                break
            addr = d.u(self.O)
            size, mask = csize, 0
            if client == 1:
                size = d.u(esz - self.O - 4)
                mask = d.u(4)
            if addr != self.undef:
                entries.append((tuple(k * c for k, c in zip(idx, chunk)), addr, size, mask))
