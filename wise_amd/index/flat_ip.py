"""faiss-shaped flat inner-product index whose rows live in HBM and whose search is the HIP scan.

Stands where `faiss.IndexIDMap(faiss.IndexFlatIP(d))` stands in the reference
(src/index/feature_search_index.py:47-52).  The attributes and methods are the ones WISE touches on
`SearchIndex.index`: `d`, `ntotal`, `add_with_ids`, `search`,
`reconstruct_batch`; it deliberately has no `nprobe`/`direct_map` (api/routes.py:899-909 guards
both with hasattr/try).
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch

from .. import _lib


class FlatIPIndex:
    def __init__(self, d: int, device: str = "cuda", shadow=None):
        """shadow: keep a reduced-precision copy of the rows for the first stage of the exact two-stage search (same
        results as the fp32 scan, by construction): True / "int8" = an int8 copy with a scale per row (+25 % memory, a
        quarter of the bytes per query or per pass of 128 queries: wise_ip_topk_shadow8_f32; d % 16 == 0, else bf16);
        "bf16" = a bf16 copy (+50 %: wise_ip_topk_shadow_f32); False = the fp32 scan only.
        None = WISE_FLAT_SHADOW (0 / bf16 / int8; default int8)."""
        if d < 4 or d % 4 != 0 or d > 2048:
            raise ValueError(f"FlatIPIndex: d={d} must be a multiple of 4 in [4, 2048]")
        self.d = int(d)
        self.device = torch.device(device)
        if shadow is None:
            env = os.environ.get("WISE_FLAT_SHADOW", "1")
            shadow = False if env == "0" else ("bf16" if env == "bf16" else True)
        self.shadow = bool(shadow) and d % 8 == 0 and d <= 1024
        self.shadow8 = self.shadow and shadow != "bf16" and d % 16 == 0     # single queries over the int8 copy
        self._Xq: Optional[torch.Tensor] = None      # [N,d] int8 on device, its row scales [N] and error norms [4]
        self._scales8: Optional[torch.Tensor] = None
        self._norms8: Optional[torch.Tensor] = None
        self._Xb: Optional[torch.Tensor] = None      # [N,d] bf16 bits (int16) on device
        self._norms: Optional[torch.Tensor] = None   # device [2]: largest row norm, largest bf16 rounding-residual norm
        self._sws: Optional[torch.Tensor] = None
        # two-stage search counters of THIS index: device [2] int32 the kernels add to (answered from the shadow, handed
        # to the fp32 scan); read back without ever blocking the search path (pinned snapshot + event)
        self._counters: Optional[torch.Tensor] = None
        self._snap = self._snap_event = None
        self._shadow_calls = 0
        self.shadow_certified = self.shadow_fallback = 0
        self._chunks: List[torch.Tensor] = []
        self._id_chunks: List[torch.Tensor] = []
        self._X: Optional[torch.Tensor] = None  # [N,d] fp32 on device
        self._ids: Optional[torch.Tensor] = None  # [N] int64 on device, None => id_base + row
        self.id_base = 0
        self._n = 0
        self._ws: Optional[torch.Tensor] = None
        self.is_trained = True

    # -- construction ---------------------------------------------------------------------------
    @property
    def ntotal(self) -> int:
        return self._n

    def reserve(self, n: int) -> None:
        """Room for n rows in HBM up front: later add_with_ids calls copy into their slice of ONE [n,d] tensor instead of
        queueing chunks that a final torch.cat would have to duplicate (an index larger than ~40 % of HBM could not be
        loaded through chunks; load_index knows the row count from the file header)."""
        if self._n or self._chunks:
            raise ValueError("reserve: the index already holds rows")
        self._rX = torch.empty(int(n), self.d, dtype=torch.float32, device=self.device)
        self._rids = torch.empty(int(n), dtype=torch.int64, device=self.device)
        self._rfill = 0

    def add_with_ids(self, x, ids) -> None:
        """x [n,d] float32, ids [n] int64 (feature_search_index.py:81)."""
        if not torch.is_tensor(x):
            x = np.ascontiguousarray(x, dtype=np.float32)
            if not x.flags.writeable:   # e.g. a read-only memory map of an index file: torch wants a writable buffer
                x = x.copy()
            x = torch.from_numpy(x)
        ids = torch.as_tensor(np.ascontiguousarray(ids, dtype=np.int64)) if not torch.is_tensor(ids) else ids
        if x.dim() != 2 or x.shape[1] != self.d:
            raise ValueError(f"add_with_ids: expected [n,{self.d}], got {tuple(x.shape)}")
        if ids.shape != (x.shape[0],):
            raise ValueError("add_with_ids: ids must have one entry per row")
        rX = getattr(self, "_rX", None)
        if rX is not None and self._rfill + x.shape[0] <= rX.shape[0]:
            a, b = self._rfill, self._rfill + x.shape[0]
            rX[a:b].copy_(x)                       # host -> its slice of the reserved tensor, no intermediate
            self._rids[a:b].copy_(ids)
            self._rfill = b
            self._X, self._ids = rX[:b], self._rids[:b]
            self._Xb = self._Xq = None
            self._n = b
            return
        if rX is not None:                          # more rows than reserved: fall back to chunks from here on
            self._rX = None
        self._chunks.append(x.to(self.device, torch.float32))
        self._id_chunks.append(ids.to(self.device, torch.int64))
        self._n += x.shape[0]

    def adopt(self, X: torch.Tensor, ids: Optional[torch.Tensor] = None, id_base: int = 0) -> "FlatIPIndex":
        """Take ownership of rows already resident in HBM (no copy). ids None => id = id_base + row."""
        if X.dtype != torch.float32 or X.dim() != 2 or X.shape[1] != self.d or not X.is_contiguous():
            raise ValueError("adopt: X must be contiguous float32 [N,d]")
        if ids is not None and (ids.dtype != torch.int64 or ids.shape != (X.shape[0],)):
            raise ValueError("adopt: ids must be int64 [N]")
        self._chunks, self._id_chunks = [], []
        self._X, self._ids, self.id_base, self._n = X, ids, int(id_base), X.shape[0]
        self._Xb = self._Xq = None
        return self

    def _finalize(self):
        if self._chunks:
            parts = ([self._X] if self._X is not None else []) + self._chunks
            idp = ([self._ids] if self._ids is not None else []) + self._id_chunks
            if self._X is not None and self._ids is None:
                idp = [torch.arange(self._X.shape[0], device=self.device, dtype=torch.int64) + self.id_base] + \
                    self._id_chunks
            self._X = torch.cat(parts, dim=0).contiguous()
            self._ids = torch.cat(idp, dim=0).contiguous()
            self._chunks, self._id_chunks = [], []
            self._Xb = self._Xq = None
            self._rX = None
        if self._X is None:
            self._X = torch.empty(0, self.d, dtype=torch.float32, device=self.device)
            self._ids = torch.empty(0, dtype=torch.int64, device=self.device)

    # -- search ---------------------------------------------------------------------------------
    def search_device(self, q: torch.Tensor, k: int):
        """q [nq,d] fp32 on device -> (D [nq,k] fp32, I [nq,k] int64) on device, no host sync."""
        lib = _lib.lib()
        self._finalize()
        if q.dim() != 2 or q.shape[1] != self.d:
            raise ValueError(f"search: expected [nq,{self.d}], got {tuple(q.shape)}")
        q = q.to(self.device, torch.float32).contiguous()
        nq = q.shape[0]
        D = torch.empty(nq, k, dtype=torch.float32, device=self.device)
        I = torch.empty(nq, k, dtype=torch.int64, device=self.device)
        if nq == 0:
            return D, I
        # one query (the reference's shape) at any k <= 1024 — REST `end` = 20, evaluation k = 100 / 1000 —, or two and
        # more where the matrix-core passes apply (d = 256 / 512 / 768 / 1024: k <= 12 from two queries on, k <= 128 —
        # the evaluation's k = 100 — from three); a few queries outside those limits go one at a time when that beats the f32 batch kernels.  Small
        # indexes (under 2^18 rows the library answers with the f32 scan anyway) never build a shadow.
        two_stage = self.shadow and self._n >= (1 << 18) and self.d % 8 == 0 and self.d <= 1024 and (
            (nq == 1 and 1 <= k <= 1024) or
            (nq >= 2 and 1 <= k <= 12 and self.d in (256, 512)) or
            (nq >= 3 and 1 <= k <= 128 and self.d in (256, 512, 768, 1024)) or      # batched passes, k up to 128
            (2 <= nq <= 3 and 1 <= k <= 1024))
        if two_stage and lib.wise_ip_topk_shadow_workspace_bytes(self._n, self.d, nq, k) == 0:
            two_stage = False
        # the int8 copy answers one query at a time (0.9 ms each at 10M x 512): alone or in pairs — two queries cost what one
        # bf16 matrix-core pass costs (1.8 ms whatever it carries), three or more cost more (measured: 1093 q/s at nq = 4
        # one at a time against 2.09 k through a pass) —, and wherever no batched shape applies (k > 128, or k > 12 with two
        # queries).  A batch proper goes through the bf16 passes: with 128
        # queries in flight the int8 error band (~0.7 sigma of the score distribution at 10M rows against bf16's 0.25)
        # puts a candidate in most 32-row groups, and the hit path, not the bytes, then sets the pace (measured: 50 k
        # queries/s against 65 k at nq = 256, and k = 100 overflows its lists).
        batched_shape = ((nq >= 2 and 1 <= k <= 12 and self.d in (256, 512)) or
                         (nq >= 3 and 1 <= k <= 128 and self.d in (256, 512, 768, 1024)))
        # k > 256 (the retrieval evaluation's --topk 1000) goes over the bf16 copy: at k = 1000 on 10M iid rows the int8
        # band already keeps 13 k of the 16 k rows the re-scoring list holds — any clustering would overflow it into the
        # fp32 scan — and the two are equally fast there (the lists, not the bytes, set the time)
        use8 = (self.shadow and self.shadow8 and self._n >= (1 << 18) and 1 <= k <= 256 and
                (nq <= 2 or (not batched_shape and nq <= 64)))
        if use8 and self._ensure_shadow8(lib):
            need = lib.wise_ip_topk_shadow_workspace_bytes(self._n, self.d, nq, k)
            if self._sws is None or self._sws.numel() < need:
                self._sws = torch.empty(need, dtype=torch.uint8, device=self.device)
            rc = lib.wise_ip_topk_shadow8_f32(self._X.data_ptr(), self._Xq.data_ptr(), self._scales8.data_ptr(),
                                              self._norms8.data_ptr(), self._n, self.d, q.data_ptr(), nq, k,
                                              _lib.ptr(self._ids), self.id_base, D.data_ptr(), I.data_ptr(),
                                              self._counters.data_ptr(), self._sws.data_ptr(), self._sws.numel(),
                                              _lib.stream_ptr())
            _lib.check(rc, "wise_ip_topk_shadow8_f32")
            self._shadow_calls += 1
            self._review_shadow()
            return D, I
        if two_stage and self._ensure_shadow(lib):
            need = lib.wise_ip_topk_shadow_workspace_bytes(self._n, self.d, nq, k)
            if self._sws is None or self._sws.numel() < need:
                self._sws = torch.empty(need, dtype=torch.uint8, device=self.device)
            rc = lib.wise_ip_topk_shadow_f32(self._X.data_ptr(), self._Xb.data_ptr(), self._norms.data_ptr(), self._n,
                                             self.d, q.data_ptr(), nq, k, _lib.ptr(self._ids), self.id_base, D.data_ptr(),
                                             I.data_ptr(), self._counters.data_ptr(), self._sws.data_ptr(),
                                             self._sws.numel(), _lib.stream_ptr())
            _lib.check(rc, "wise_ip_topk_shadow_f32")
            self._shadow_calls += 1
            self._review_shadow()
            return D, I
        need = lib.wise_ip_topk_workspace_bytes(self._n, self.d, nq, k)
        if need == 0:
            raise ValueError(f"search: unsupported shape N={self._n} d={self.d} nq={nq} k={k}")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        rc = lib.wise_ip_topk_f32(self._X.data_ptr(), self._n, self.d, q.data_ptr(), nq, k, _lib.ptr(self._ids),
                                  self.id_base, D.data_ptr(), I.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                  _lib.stream_ptr())
        _lib.check(rc, "wise_ip_topk_f32")
        return D, I

    def _ensure_shadow(self, lib) -> bool:
        """Build the bf16 copy if it is not there; False (and the shadow switched off) when HBM has no room for it."""
        if self._Xb is not None and self._Xb.shape[0] == self._n:
            return True
        need = self._n * self.d * 2
        free, _ = torch.cuda.mem_get_info(self.device)
        if free < need + (2 << 30):          # keep 2 GiB for workspaces and the caller
            self.shadow = False
            return False
        self._Xb = torch.empty(self._n, self.d, dtype=torch.int16, device=self.device)
        self._norms = torch.zeros(2, dtype=torch.float32, device=self.device)
        if self._counters is None:
            self._counters = torch.zeros(2, dtype=torch.int32, device=self.device)
        rc = lib.wise_ip_shadow_bf16(self._X.data_ptr(), self._n, self.d, self._Xb.data_ptr(), self._norms.data_ptr(),
                                     _lib.stream_ptr())
        _lib.check(rc, "wise_ip_shadow_bf16")
        return True

    def _ensure_shadow8(self, lib) -> bool:
        """Build the int8 copy (rows, scales, error norms) if it is not there; False when HBM has no room for it."""
        if self._Xq is not None and self._Xq.shape[0] == self._n:
            return True
        need = self._n * (self.d + 4)
        free, _ = torch.cuda.mem_get_info(self.device)
        if free < need + (2 << 30):
            self.shadow8 = False
            return False
        self._Xq = torch.empty(self._n, self.d, dtype=torch.int8, device=self.device)
        self._scales8 = torch.empty(self._n, dtype=torch.float32, device=self.device)
        self._norms8 = torch.zeros(4, dtype=torch.float32, device=self.device)
        if self._counters is None:
            self._counters = torch.zeros(2, dtype=torch.int32, device=self.device)
        rc = lib.wise_ip_shadow_i8(self._X.data_ptr(), self._n, self.d, self._Xq.data_ptr(), self._scales8.data_ptr(),
                                   self._norms8.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "wise_ip_shadow_i8")
        return True

    def _review_shadow(self) -> None:
        """Never blocks: every 64 two-stage searches a snapshot of this index's counters is copied to pinned memory
        behind an event; a later call that finds the event complete reads it.  If the fp32 scan had to answer most
        queries of a snapshot (more rows within the bf16 error of the k-th score than the candidate list holds), stop
        paying for the first stage."""
        if self._snap_event is not None and self._snap_event.query():
            done, handed = int(self._snap[0]), int(self._snap[1])
            self._snap_event = None
            d_done, d_handed = done - self.shadow_certified, handed - self.shadow_fallback
            self.shadow_certified, self.shadow_fallback = done, handed
            if d_handed > d_done:
                self.shadow = self.shadow8 = False
                self._Xb = self._Xq = self._scales8 = self._sws = None
                return
        if self._snap_event is None and self._shadow_calls % 64 == 0 and self._counters is not None:
            if self._snap is None:
                self._snap = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._snap.copy_(self._counters, non_blocking=True)
            self._snap_event = torch.cuda.Event()
            self._snap_event.record()

    def shadow_counts(self):
        """(answered from the shadow, recomputed by the fp32 scan) over the two-stage searches of THIS index so far.
        Synchronises (a diagnostic, not part of the search path)."""
        if self._counters is None:
            return 0, 0
        c = self._counters.cpu()
        return int(c[0]), int(c[1])

    def search(self, x, k: int):
        """faiss signature: x np.ndarray [nq,d] float32 -> (D, I) numpy (feature_search_index.py:113)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2:
            raise ValueError("search: x must be 2-D")
        D, I = self.search_device(torch.from_numpy(x).to(self.device), int(k))
        return D.cpu().numpy(), I.cpu().numpy()

    def reconstruct_batch(self, ids) -> np.ndarray:
        """rows stored under the given external ids (api/routes.py:1078)."""
        lib = _lib.lib()
        self._finalize()
        qi = torch.as_tensor(np.ascontiguousarray(ids, dtype=np.int64)).to(self.device)
        out = torch.empty(qi.numel(), self.d, dtype=torch.float32, device=self.device)
        rc = lib.wise_reconstruct_batch(self._X.data_ptr(), self._n, self.d, _lib.ptr(self._ids), self.id_base,
                                        qi.data_ptr(), qi.numel(), out.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "wise_reconstruct_batch")
        return out.cpu().numpy()
