"""ORACLE (test infrastructure — never imported by the product path under wise_amd/).

CPU restatement of the search the reference issues at
/root/reference/src/index/feature_search_index.py:113 (`self.index.search(query_features, topk)`)
and /root/reference/api/routes.py:1407, on the index built at feature_search_index.py:47-52,80-82
(`faiss.IndexIDMap(faiss.IndexFlatIP(d))` + `add_with_ids`).

faiss (faiss-gpu==1.7.2 / faiss-cpu, /root/reference/torch-faiss-requirements.txt:6) is not in the
container.  Its published IndexFlatIP/IndexIDMap semantics are restated:
  * scores = x @ X^T in fp32; the k largest per query, sorted descending;
  * labels are the int64 ids given to add_with_ids;
  * fewer than k vectors: tail is (distance -3.4028235e38, label -1)
    (relied on by /root/reference/search.py:142-143 and api/routes.py:1411);
  * tie order is unspecified upstream; this restatement (and the HIP path) fix "lower row first".
PINNING: exact by construction for the integer part (ids, order, padding); the reference holds no
golden vectors for this call (its only numeric pin, tests/test-kinetics-6.sh:124-142, needs network
weights), so parity with faiss binaries is UNPINNED offline.
"""
from __future__ import annotations

import numpy as np

NEG = np.float32(-3.4028234663852886e38)


def ip_topk(X: np.ndarray, Q: np.ndarray, k: int, ids: np.ndarray | None = None, id_base: int = 0,
            scores: np.ndarray | None = None):
    """X [N,d] fp32, Q [nq,d] fp32 -> (D [nq,k] fp32 desc, I [nq,k] int64), faiss padding."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    N = X.shape[0]
    nq = Q.shape[0]
    D = np.full((nq, k), NEG, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if N == 0:
        return D, I
    S = (Q @ X.T).astype(np.float32) if scores is None else scores
    kk = min(k, N)
    rows = np.arange(N)
    for q in range(nq):
        # stable: score descending, then row ascending
        order = np.lexsort((rows, -S[q].astype(np.float64)))[:kk]
        D[q, :kk] = S[q, order]
        I[q, :kk] = (ids[order] if ids is not None else order + id_base)
    return D, I


def merge_topk(Ds: np.ndarray, Is: np.ndarray, k: int):
    """[parts,nq,k] partial lists -> [nq,k]; ties: lower part first, then list order; -1 ids are padding."""
    parts, nq, kk = Ds.shape
    D = np.full((nq, k), NEG, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        d = Ds[:, q, :].reshape(-1)
        i = Is[:, q, :].reshape(-1)
        slot = np.arange(d.size)
        valid = i >= 0
        d, i, slot = d[valid], i[valid], slot[valid]
        order = np.lexsort((slot, -d.astype(np.float64)))[:k]
        D[q, : order.size] = d[order]
        I[q, : order.size] = i[order]
    return D, I


def recall_at_k(I_test: np.ndarray, I_ref: np.ndarray) -> float:
    hits = 0
    total = 0
    for a, b in zip(I_test, I_ref):
        bs = set(int(x) for x in b if x >= 0)
        hits += len(bs & set(int(x) for x in a if x >= 0))
        total += len(bs)
    return hits / max(total, 1)
