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

faiss is not in the container, so this layout is UNPINNED against a real faiss binary; the round
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
