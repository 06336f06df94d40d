"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

CPU restatement of what `faiss.IndexIVFFlat(IndexFlatIP(d), d, nlist, METRIC_INNER_PRODUCT)` computes at search time
(the index the reference builds at /root/reference/src/index/feature_search_index.py:53-76 and queries at :113 with
`nprobe` set at api/routes.py:899-902), for GIVEN centroids and inverted lists:
  1. the `nprobe` centroids of largest inner product with the query are the probed lists;
  2. the result is the exact top-k (descending score) among the rows of those lists, padded with (-3.4028235e38, -1).
faiss is not installed and its k-means is seeded by its own random generator, so the quantizer itself cannot be
pinned against faiss offline; this oracle pins everything that is deterministic given the quantizer.  Ties: the row
that comes first in the list-ordered storage wins (the kernel's rule).
"""
from __future__ import annotations

import numpy as np

NEG = np.float32(-3.4028234663852886e38)


def coarse_probes(centroids: np.ndarray, Q: np.ndarray, nprobe: int) -> np.ndarray:
    S = (Q.astype(np.float32) @ centroids.astype(np.float32).T).astype(np.float32)
    nlist = centroids.shape[0]
    out = np.full((Q.shape[0], nprobe), -1, dtype=np.int64)
    for q in range(Q.shape[0]):
        order = np.lexsort((np.arange(nlist), -S[q].astype(np.float64)))[:nprobe]
        out[q, :len(order)] = order
    return out


def ivf_search(X: np.ndarray, ids: np.ndarray, list_off: np.ndarray, Q: np.ndarray, probes: np.ndarray, k: int):
    """X [N,d] rows grouped by list, probes [nq,nprobe] (entries < 0 skipped) -> (D [nq,k], I [nq,k])."""
    nq = Q.shape[0]
    D = np.full((nq, k), NEG, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        rows = np.concatenate([np.arange(list_off[l], list_off[l + 1]) for l in probes[q] if l >= 0] +
                              [np.zeros(0, dtype=np.int64)]).astype(np.int64)
        if rows.size == 0:
            continue
        s = (X[rows].astype(np.float32) @ Q[q].astype(np.float32)).astype(np.float32)
        order = np.lexsort((rows, -s.astype(np.float64)))[:k]
        D[q, :len(order)] = s[order]
        I[q, :len(order)] = ids[rows[order]]
    return D, I


def assign(X: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    return (X.astype(np.float32) @ centroids.astype(np.float32).T).argmax(axis=1).astype(np.int64)
