"""Reader/writer for the one faiss file layout WISE produces for flat search:
`faiss.write_index(IndexIDMap(IndexFlatIP(d)))` (reference: src/index/feature_search_index.py:47-52,84,96).

Layout restated from faiss's published index_write.cpp / index_read.cpp (faiss 1.7.x, little endian):

    u32  'IxMp'                                   IndexIDMap fourcc
    header: i32 d | i64 ntotal | i64 dummy(1<<20) | i64 dummy(1<<20) | u8 is_trained | i32 metric_type(0 = IP)
    u32  'IxFI'                                   IndexFlatIP fourcc
    header (same fields)
    u64  n_floats (= ntotal*d) | f32[n_floats]    the rows (1.7.2 writes xb as a float vector; newer versions
                                                  write the same bytes as a uint8 `codes` vector sized in 4-byte units)
    u64  ntotal | i64[ntotal]                     id_map

The IVF file the reference writes for index_type 'IndexIVFFlat' (feature_search_index.py:53-76,84) is restated the
same way (write_ivf_flat_ip / read_ivf_flat_ip below):

    u32  'IwFl'                                   IndexIVFFlat fourcc
    header (d, ntotal, ...) | u64 nlist | u64 nprobe
    u32  'IxFI' + header(d, nlist) + u64 n_floats + f32[nlist*d]      the coarse quantizer (centroids)
    u8   direct-map type (0 = none) | u64 0                           empty direct map
    u32  'ilar' | u64 nlist | u64 code_size (= 4*d)
    u32  'full' | u64 nlist | u64 sizes[nlist]        (or 'sprs' | u64 2*m | (list, size) pairs when most lists are empty)
    per non-empty list: u8 codes[size*code_size] (the fp32 rows) | i64 ids[size]

faiss is not in the container, so these layouts are UNPINNED against a real faiss binary; the round
trip is pinned by tests/test_feature_store_index_io.py.  The rows are memory-mapped on read so a
158 GiB index (docs/Search-Index-Evaluation.md:109) streams to the GPU without a host copy.
"""
from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

_DUMMY = 1 << 20


def _fourcc(s: str) -> int:
    b = s.encode("ascii")
    return b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24)


def _header(d: int, ntotal: int, metric: int = 0) -> bytes:
    return struct.pack("<iqqq?i", d, ntotal, _DUMMY, _DUMMY, True, metric)


_HDR_SIZE = struct.calcsize("<iqqq?i")  # 4 + 8*3 + 1 + 4 = 33


def write_idmap_flat_ip(path, X: np.ndarray, ids: np.ndarray) -> None:
    X = np.ascontiguousarray(X, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    n, d = X.shape
    assert ids.shape == (n,)
    with open(path, "wb") as f:
        f.write(struct.pack("<I", _fourcc("IxMp")))
        f.write(_header(d, n))
        f.write(struct.pack("<I", _fourcc("IxFI")))
        f.write(_header(d, n))
        f.write(struct.pack("<Q", n * d))
        X.tofile(f)
        f.write(struct.pack("<Q", n))
        ids.tofile(f)


def _read_header(buf, off):
    d, ntotal, _, _, trained, metric = struct.unpack_from("<iqqq?i", buf, off)
    off += _HDR_SIZE
    if metric > 1:
        off += 4  # metric_arg
    return d, ntotal, metric, off


def read_idmap_flat_ip(path, mmap: bool = True):
    """-> (X [n,d] float32 (memmap view), ids int64[n]).  Raises RuntimeError like faiss on a missing file."""
    p = Path(path)
    if not p.exists():
        raise RuntimeError(f"Error: 'f' failed: could not open {p} for reading: No such file or directory")
    with open(p, "rb") as f:
        head = f.read(4 + _HDR_SIZE + 4 + 4 + _HDR_SIZE + 4 + 8)
    (cc,) = struct.unpack_from("<I", head, 0)
    if cc == _fourcc("IxFI"):  # a bare IndexFlatIP: ids are the positions
        d, n, metric, off = _read_header(head, 4)
        (cnt,) = struct.unpack_from("<Q", head, off)
        off += 8
        X = np.memmap(p, dtype=np.float32, mode="r", offset=off, shape=(n, d)) if mmap else \
            np.fromfile(p, dtype=np.float32, count=n * d, offset=off).reshape(n, d)
        return X, np.arange(n, dtype=np.int64)
    if cc != _fourcc("IxMp"):
        raise RuntimeError(f"{p}: index type 0x{cc:08x} is not IndexIDMap/IndexFlatIP; only flat IP indexes are "
                           f"supported by the MI355X search path")
    d, n, metric, off = _read_header(head, 4)
    (cc2,) = struct.unpack_from("<I", head, off)
    if cc2 != _fourcc("IxFI"):
        raise RuntimeError(f"{p}: IndexIDMap wraps index type 0x{cc2:08x}, expected IndexFlatIP")
    d2, n2, metric2, off = _read_header(head, off + 4)
    (cnt,) = struct.unpack_from("<Q", head, off)
    off += 8
    if cnt != n2 * d2 or d2 != d:
        raise RuntimeError(f"{p}: inconsistent flat payload ({cnt} values for {n2} x {d2})")
    X = np.memmap(p, dtype=np.float32, mode="r", offset=off, shape=(n2, d2)) if mmap else \
        np.fromfile(p, dtype=np.float32, count=n2 * d2, offset=off).reshape(n2, d2)
    off += cnt * 4
    with open(p, "rb") as f:
        f.seek(off)
        (nid,) = struct.unpack("<Q", f.read(8))
        ids = np.fromfile(f, dtype=np.int64, count=nid)
    if nid != n2:
        raise RuntimeError(f"{p}: id_map has {nid} entries for {n2} rows")
    return X, ids


def write_ivf_flat_ip(path, centroids: np.ndarray, X: np.ndarray, ids: np.ndarray, list_off: np.ndarray,
                      nprobe: int = 1) -> None:
    """X / ids hold the lists back to back; list l is rows list_off[l] .. list_off[l+1]-1."""
    centroids = np.ascontiguousarray(centroids, dtype=np.float32)
    X = np.ascontiguousarray(X, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    list_off = np.ascontiguousarray(list_off, dtype=np.int64)
    nlist, d = centroids.shape
    n = X.shape[0]
    assert X.shape == (n, d) and ids.shape == (n,) and list_off.shape == (nlist + 1,) and list_off[-1] == n
    sizes = (list_off[1:] - list_off[:-1]).astype(np.uint64)
    with open(path, "wb") as f:
        f.write(struct.pack("<I", _fourcc("IwFl")))
        f.write(_header(d, n))
        f.write(struct.pack("<QQ", nlist, nprobe))
        f.write(struct.pack("<I", _fourcc("IxFI")))
        f.write(_header(d, nlist))
        f.write(struct.pack("<Q", nlist * d))
        centroids.tofile(f)
        f.write(struct.pack("<BQ", 0, 0))
        f.write(struct.pack("<IQQ", _fourcc("ilar"), nlist, 4 * d))
        nonzero = np.flatnonzero(sizes)
        if len(nonzero) > nlist // 2:
            f.write(struct.pack("<IQ", _fourcc("full"), nlist))
            sizes.tofile(f)
        else:
            f.write(struct.pack("<IQ", _fourcc("sprs"), 2 * len(nonzero)))
            np.stack([nonzero.astype(np.uint64), sizes[nonzero]], axis=1).tofile(f)
        for l in nonzero:
            a, b = int(list_off[l]), int(list_off[l + 1])
            X[a:b].tofile(f)
            ids[a:b].tofile(f)


def read_ivf_flat_ip(path):
    """-> dict(centroids [nlist,d], X [n,d], ids [n], list_off [nlist+1], nprobe).  The lists are returned back to
    back in list order, which is the layout the search kernel wants."""
    p = Path(path)
    if not p.exists():
        raise RuntimeError(f"Error: 'f' failed: could not open {p} for reading: No such file or directory")
    with open(p, "rb") as f:
        (cc,) = struct.unpack("<I", f.read(4))
        if cc != _fourcc("IwFl"):
            raise RuntimeError(f"{p}: index type 0x{cc:08x} is not IndexIVFFlat")
        hdr = f.read(_HDR_SIZE + 4)
        d, n, metric, off = _read_header(hdr, 0)
        f.seek(4 + off)
        nlist, nprobe = struct.unpack("<QQ", f.read(16))
        (cq,) = struct.unpack("<I", f.read(4))
        if cq != _fourcc("IxFI"):
            raise RuntimeError(f"{p}: coarse quantizer type 0x{cq:08x}, expected IndexFlatIP")
        pos = f.tell()
        hdr = f.read(_HDR_SIZE + 4)
        dq, nq_, _, off = _read_header(hdr, 0)
        f.seek(pos + off)
        (cnt,) = struct.unpack("<Q", f.read(8))
        if dq != d or nq_ != nlist or cnt != nlist * d:
            raise RuntimeError(f"{p}: inconsistent quantizer ({cnt} values for {nq_} x {dq})")
        centroids = np.fromfile(f, dtype=np.float32, count=cnt).reshape(nlist, d)
        (dm_type,) = struct.unpack("<B", f.read(1))
        (dm_n,) = struct.unpack("<Q", f.read(8))
        f.seek(8 * dm_n, 1)
        if dm_type == 2:  # hashtable pairs
            (npairs,) = struct.unpack("<Q", f.read(8))
            f.seek(16 * npairs, 1)
        il, nl2, code_size = struct.unpack("<IQQ", f.read(20))
        if il != _fourcc("ilar") or nl2 != nlist or code_size != 4 * d:
            raise RuntimeError(f"{p}: unexpected inverted lists (type 0x{il:08x}, code size {code_size})")
        (lt, vn) = struct.unpack("<IQ", f.read(12))
        sizes = np.zeros(nlist, dtype=np.int64)
        if lt == _fourcc("full"):
            sizes[:] = np.fromfile(f, dtype=np.uint64, count=vn).astype(np.int64)
        elif lt == _fourcc("sprs"):
            pairs = np.fromfile(f, dtype=np.uint64, count=vn).reshape(-1, 2).astype(np.int64)
            sizes[pairs[:, 0]] = pairs[:, 1]
        else:
            raise RuntimeError(f"{p}: unknown list layout 0x{lt:08x}")
        list_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        if list_off[-1] != n:
            raise RuntimeError(f"{p}: lists hold {list_off[-1]} rows, header says {n}")
        X = np.empty((n, d), dtype=np.float32)
        ids = np.empty((n,), dtype=np.int64)
        for l in np.flatnonzero(sizes):
            a, b = int(list_off[l]), int(list_off[l + 1])
            X[a:b] = np.fromfile(f, dtype=np.float32, count=(b - a) * d).reshape(b - a, d)
            ids[a:b] = np.fromfile(f, dtype=np.int64, count=b - a)
    return {"centroids": centroids, "X": X, "ids": ids, "list_off": list_off, "nprobe": int(nprobe)}


def index_fourcc(path) -> str:
    with open(path, "rb") as f:
        return f.read(4).decode("ascii", errors="replace")
