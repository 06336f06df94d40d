"""faiss-shaped inverted-file index (inner product) whose lists live in HBM and whose search is two HIP scans.

Stands where `faiss.IndexIVFFlat(IndexFlatIP(d), d, nlist, METRIC_INNER_PRODUCT)` stands in the reference
(src/index/feature_search_index.py:53-76: nlist = 3 or 10 * round(sqrt(N)), trained on min(N, 100 * nlist) rows;
api/routes.py:899-909 then sets `parallel_mode`, `nprobe` and calls `make_direct_map(True)`).

  train(x)            spherical k-means (10 Lloyd iterations, assignment by inner product — what faiss's Clustering
                      does for an inner-product IVF), run on the GPU as dense products; deterministic (seeded
                      sample for the initial centroids, empty cells re-seeded from the fullest cell)
  add_with_ids(x,ids) rows are assigned to the centroid of largest inner product and kept grouped by list
  search(q, k)        stage 1: `nprobe` best centroids per query = wise_ip_topk_f32 over the centroid table (nprobe
                      <= 64) or wise_ip_scores_f32 + wise_select_topk_f32 (the reference's nprobe = 1024);
                      stage 2: wise_ivf_scan_f32 over the probed lists (the flat scan kernel run per list segment)
An approximate index cannot be pinned value-for-value against faiss (its k-means starts from faiss's own random
permutation); what IS exact and tested: given the same centroids and lists, the result equals the brute-force top-k
restricted to the probed lists (oracle/ivf_ref.py), and nprobe = nlist reproduces the flat index.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch

from .. import _lib
from .flat_ip import FlatIPIndex


def reference_nlist(feature_count: int) -> int:
    """The cell count the reference picks (feature_search_index.py:55-58)."""
    return (3 if feature_count < 200000 else 10) * round(math.sqrt(feature_count))


class _DirectMap:
    """The two attributes api/routes.py:1317 reads."""
    NoMap, Array, Hashtable = 0, 1, 2

    def __init__(self):
        self.type = self.NoMap


class IVFFlatIPIndex:
    def __init__(self, d: int, nlist: int, device: str = "cuda"):
        if d < 4 or d % 4 != 0 or d > 2048:
            raise ValueError(f"IVFFlatIPIndex: d={d} must be a multiple of 4 in [4, 2048]")
        if nlist < 1:
            raise ValueError("IVFFlatIPIndex: nlist must be positive")
        self.d, self.nlist = int(d), int(nlist)
        self.device = torch.device(device)
        self.nprobe = 1           # faiss default; the REST layer sets it (routes.py:902)
        self.parallel_mode = 0    # accepted and ignored (routes.py:901)
        self.direct_map = _DirectMap()
        self.is_trained = False
        self.niter = 10
        self.seed = 1234
        self.centroids: Optional[torch.Tensor] = None    # [nlist, d] fp32, unit rows
        self._quantizer: Optional[FlatIPIndex] = None
        self._pending: List[tuple] = []                  # (x, ids, assign) chunks not yet merged into the lists
        self._X: Optional[torch.Tensor] = None           # [N, d] rows grouped by list
        self._ids: Optional[torch.Tensor] = None         # [N] external ids, same order
        self._list_off: Optional[torch.Tensor] = None    # [nlist + 1] int64
        self._n = 0
        self._ws: Optional[torch.Tensor] = None

    @property
    def ntotal(self) -> int:
        return self._n

    def _need_gpu(self):
        _lib.lib()  # raises without a gfx950 device: there is no CPU path

    # -- training -------------------------------------------------------------------------------
    # Everything numeric below is this library's own kernels (csrc/ivf_build.hip, wise_ip_scores_f32): torch only allocates,
    # concatenates and draws the seeding permutation on the host.
    @staticmethod
    def _assign(x: torch.Tensor, centroids: torch.Tensor, chunk: int = 4096) -> torch.Tensor:
        """nearest centroid of every row by inner product: the exact-f32 score kernel (the coarse stage's own) + wise_ivf_argmax"""
        lib = _lib.lib()
        c = centroids.contiguous()
        out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
        scores = torch.empty(min(chunk, max(x.shape[0], 1)), c.shape[0], dtype=torch.float32, device=x.device)
        st = _lib.stream_ptr()
        for s in range(0, x.shape[0], chunk):
            q = x[s:s + chunk]                       # (a slice of whole rows of a contiguous tensor: contiguous)
            _lib.check(lib.wise_ip_scores_f32(c.data_ptr(), c.shape[0], c.shape[1], q.data_ptr(), q.shape[0],
                                              scores.data_ptr(), st), "wise_ip_scores_f32")
            _lib.check(lib.wise_ivf_argmax(scores.data_ptr(), q.shape[0], c.shape[0], out[s:].data_ptr(), st), "wise_ivf_argmax")
        return out

    def _group(self, assign: torch.Tensor):
        """(order, list_off, counts): the rows grouped by list, stable (wise_ivf_group: a radix sort on the device)"""
        lib = _lib.lib()
        n = assign.shape[0]
        order = torch.empty(n, dtype=torch.int64, device=self.device)
        list_off = torch.empty(self.nlist + 1, dtype=torch.int64, device=self.device)
        counts = torch.empty(self.nlist, dtype=torch.int64, device=self.device)
        ws = torch.empty(lib.wise_ivf_group_workspace_bytes(n, self.nlist), dtype=torch.uint8, device=self.device)
        _lib.check(lib.wise_ivf_group(assign.data_ptr(), n, self.nlist, order.data_ptr(), list_off.data_ptr(), counts.data_ptr(),
                                      ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "wise_ivf_group")
        return order, list_off, counts

    def _gather_rows(self, x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        out = torch.empty(idx.shape[0], x.shape[1], dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().wise_ivf_gather_rows(x.data_ptr(), idx.data_ptr(), idx.shape[0], x.shape[1], out.data_ptr(),
                                                   _lib.stream_ptr()), "wise_ivf_gather_rows")
        return out

    def train(self, x) -> None:
        self._need_gpu()
        lib = _lib.lib()
        x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)) if not torch.is_tensor(x) else x
        x = x.to(self.device, torch.float32).contiguous()
        if x.dim() != 2 or x.shape[1] != self.d:
            raise ValueError(f"train: expected [n,{self.d}], got {tuple(x.shape)}")
        n = x.shape[0]
        if n < self.nlist:
            raise ValueError(f"train: {n} training vectors for {self.nlist} cells")
        if self.d % 4:
            raise ValueError("train: d must be a multiple of 4")
        g = torch.Generator(device="cpu").manual_seed(self.seed)
        perm = torch.randperm(n, generator=g)[: self.nlist].to(self.device)
        st = _lib.stream_ptr()
        c = self._gather_rows(x, perm)
        _lib.check(lib.wise_ivf_normalize_rows(c.data_ptr(), self.nlist, self.d, c.data_ptr(), st), "wise_ivf_normalize_rows")
        sums = torch.empty(self.nlist, self.d, dtype=torch.float32, device=self.device)
        for _ in range(self.niter):
            a = self._assign(x, c)
            order, list_off, counts = self._group(a)
            _lib.check(lib.wise_ivf_list_sums(x.data_ptr(), order.data_ptr(), list_off.data_ptr(), self.nlist, self.d,
                                              sums.data_ptr(), st), "wise_ivf_list_sums")
            cnt = counts.cpu().numpy()                       # nlist numbers: which cells are empty is decided on the host
            empty = np.flatnonzero(cnt == 0)
            if empty.size:
                # re-seed every empty cell with a slightly perturbed copy of the fullest cells' sums (ties: the lower cell first)
                donors = np.argsort(-cnt, kind="stable")[: empty.size]
                e_d = torch.from_numpy(empty.astype(np.int64)).to(self.device)
                d_d = torch.from_numpy(donors.astype(np.int64)).to(self.device)
                _lib.check(lib.wise_ivf_reseed(sums.data_ptr(), e_d.data_ptr(), d_d.data_ptr(), int(empty.size), self.d, st),
                           "wise_ivf_reseed")
            _lib.check(lib.wise_ivf_normalize_rows(sums.data_ptr(), self.nlist, self.d, c.data_ptr(), st),
                       "wise_ivf_normalize_rows")    # spherical: unit centroids
        self.centroids = c.contiguous()
        self._quantizer = FlatIPIndex(self.d, device=str(self.device)).adopt(self.centroids, None, id_base=0)
        self.is_trained = True

    def set_centroids(self, centroids) -> None:
        """Install a trained quantizer (file load, tests)."""
        c = torch.as_tensor(np.ascontiguousarray(centroids, dtype=np.float32)) if not torch.is_tensor(centroids) \
            else centroids
        if c.shape != (self.nlist, self.d):
            raise ValueError(f"set_centroids: expected [{self.nlist},{self.d}]")
        self.centroids = c.to(self.device, torch.float32).contiguous()
        self._quantizer = FlatIPIndex(self.d, device=str(self.device)).adopt(self.centroids, None, id_base=0)
        self.is_trained = True

    # -- construction ---------------------------------------------------------------------------
    def add_with_ids(self, x, ids) -> None:
        if not self.is_trained:
            raise RuntimeError("IVFFlatIPIndex: train() before add_with_ids()")
        x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)) if not torch.is_tensor(x) else x
        ids = torch.as_tensor(np.ascontiguousarray(ids, dtype=np.int64)) if not torch.is_tensor(ids) else ids
        if x.dim() != 2 or x.shape[1] != self.d:
            raise ValueError(f"add_with_ids: expected [n,{self.d}], got {tuple(x.shape)}")
        if ids.shape != (x.shape[0],):
            raise ValueError("add_with_ids: ids must have one entry per row")
        x = x.to(self.device, torch.float32).contiguous()
        ids = ids.to(self.device, torch.int64).contiguous()
        self._pending.append((x, ids, self._assign(x, self.centroids)))
        self._n += x.shape[0]

    def adopt_lists(self, X: torch.Tensor, ids: torch.Tensor, list_off: torch.Tensor) -> "IVFFlatIPIndex":
        """Take rows that are already grouped by list (file load)."""
        self._pending = []
        self._X = X.to(self.device, torch.float32).contiguous()
        self._ids = ids.to(self.device, torch.int64).contiguous()
        self._list_off = list_off.to(self.device, torch.int64).contiguous()
        self._n = self._X.shape[0]
        return self

    def _finalize(self):
        if self._pending:
            lib = _lib.lib()
            st = _lib.stream_ptr()
            xs = ([self._X] if self._X is not None and self._X.shape[0] else []) + [p[0] for p in self._pending]
            iss = ([self._ids] if self._ids is not None and self._ids.shape[0] else []) + [p[1] for p in self._pending]
            old_assign = []
            if self._X is not None and self._X.shape[0]:
                oa = torch.empty(self._X.shape[0], dtype=torch.int64, device=self.device)     # the rows already grouped: list c, list_off[c] .. [c + 1]
                _lib.check(lib.wise_ivf_expand_lists(self._list_off.data_ptr(), self.nlist, oa.data_ptr(), st), "wise_ivf_expand_lists")
                old_assign = [oa]
            a = torch.cat(old_assign + [p[2] for p in self._pending]).contiguous()
            order, list_off, _ = self._group(a)          # stable: rows of a list keep their order of insertion
            allx, allids = torch.cat(xs).contiguous(), torch.cat(iss).contiguous()
            self._X = self._gather_rows(allx, order)
            ids = torch.empty_like(allids)
            _lib.check(lib.wise_ivf_gather_i64(allids.data_ptr(), order.data_ptr(), order.shape[0], ids.data_ptr(), st), "wise_ivf_gather_i64")
            self._ids = ids
            self._list_off = list_off
            self._pending = []
        if self._X is None:
            self._X = torch.empty(0, self.d, dtype=torch.float32, device=self.device)
            self._ids = torch.empty(0, dtype=torch.int64, device=self.device)
            self._list_off = torch.zeros(self.nlist + 1, dtype=torch.int64, device=self.device)

    # -- search ---------------------------------------------------------------------------------
    def probes_device(self, q: torch.Tensor, nprobe: int) -> torch.Tensor:
        """[nq, nprobe] int64 list numbers: the nprobe centroids of largest inner product (-1 padding when
        nprobe > nlist).  Few probes: the flat top-k kernel over the centroid table.  Many probes (threshold lists
        stop filtering when k is a sizeable fraction of nlist): all centroid scores in exact fp32 on the matrix
        cores (wise_ip_scores_f32), then the radix-select kernel; the probes then come in list order, which the
        list scan does not care about."""
        if nprobe <= 64:
            _, I = self._quantizer.search_device(q, nprobe)
            return I
        lib = _lib.lib()
        scores = torch.empty(q.shape[0], self.nlist, dtype=torch.float32, device=self.device)
        rc = lib.wise_ip_scores_f32(self.centroids.data_ptr(), self.nlist, self.d, q.data_ptr(), q.shape[0],
                                    scores.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "wise_ip_scores_f32")
        out = torch.empty(q.shape[0], nprobe, dtype=torch.int64, device=self.device)
        rc = lib.wise_select_topk_f32(scores.data_ptr(), q.shape[0], self.nlist, nprobe, out.data_ptr(),
                                      _lib.stream_ptr())
        _lib.check(rc, "wise_select_topk_f32")
        return out

    def search_device(self, q: torch.Tensor, k: int):
        lib = _lib.lib()
        if not self.is_trained:
            raise RuntimeError("IVFFlatIPIndex: not trained")
        self._finalize()
        if q.dim() != 2 or q.shape[1] != self.d:
            raise ValueError(f"search: expected [nq,{self.d}], got {tuple(q.shape)}")
        q = q.to(self.device, torch.float32).contiguous()
        nq = q.shape[0]
        D = torch.empty(nq, k, dtype=torch.float32, device=self.device)
        I = torch.empty(nq, k, dtype=torch.int64, device=self.device)
        if nq == 0:
            return D, I
        nprobe = max(1, min(int(self.nprobe), self.nlist, 2048))
        probes = self.probes_device(q, nprobe).contiguous()
        need = lib.wise_ivf_scan_workspace_bytes(nq, nprobe, k)
        if need == 0:
            raise ValueError(f"search: unsupported shape nq={nq} nprobe={nprobe} k={k}")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        rc = lib.wise_ivf_scan_f32(self._X.data_ptr(), self._n, self.d, self._list_off.data_ptr(), self.nlist,
                                   self._ids.data_ptr(), q.data_ptr(), nq, probes.data_ptr(), nprobe, k, D.data_ptr(),
                                   I.data_ptr(), self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "wise_ivf_scan_f32")
        return D, I

    def search(self, x, k: int):
        """faiss signature: x np.ndarray [nq,d] float32 -> (D, I) numpy."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2:
            raise ValueError("search: x must be 2-D")
        D, I = self.search_device(torch.from_numpy(x).to(self.device), int(k))
        return D.cpu().numpy(), I.cpu().numpy()

    # -- the rest of the surface the REST layer touches -------------------------------------------
    def make_direct_map(self, enable: bool = True) -> None:
        """routes.py:904-909: afterwards reconstruct works by id.  Ids are looked up in the stored id array."""
        self.direct_map.type = _DirectMap.Hashtable if enable else _DirectMap.NoMap

    def reconstruct_batch(self, ids) -> np.ndarray:
        lib = _lib.lib()
        self._finalize()
        qi = torch.as_tensor(np.ascontiguousarray(ids, dtype=np.int64)).to(self.device)
        out = torch.empty(qi.numel(), self.d, dtype=torch.float32, device=self.device)
        rc = lib.wise_reconstruct_batch(self._X.data_ptr(), self._n, self.d, self._ids.data_ptr(), 0, qi.data_ptr(),
                                        qi.numel(), out.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "wise_reconstruct_batch")
        return out.cpu().numpy()

    def lists_host(self):
        """(centroids [nlist,d], X [N,d], ids [N], list_off [nlist+1]) as numpy (file save, tests)."""
        self._finalize()
        return (self.centroids.cpu().numpy(), self._X.cpu().numpy(), self._ids.cpu().numpy(),
                self._list_off.cpu().numpy())
