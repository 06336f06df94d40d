"""IndexIVFFlat without a GPU: the oracle's own invariants, the cell-count rule of the reference
and the IVF file layout round trip."""
import numpy as np
import pytest

from oracle import ip_topk_ref, ivf_ref
from wise_amd.index import faiss_io
from wise_amd.index.ivf_flat import IVFFlatIPIndex, reference_nlist


def make_lists(N, d, nlist, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    c = X[rng.permutation(N)[:nlist]].copy()
    a = ivf_ref.assign(X, c)
    order = np.argsort(a, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(a, minlength=nlist))]).astype(np.int64)
    ids = (np.arange(N, dtype=np.int64) * 2 + 5)[order]
    return c, X[order], ids, off


def test_oracle_probing_every_list_is_the_flat_search():
    c, Xs, ids, off = make_lists(3000, 32, 20, 0)
    Q = Xs[:6] + 0.01
    probes = ivf_ref.coarse_probes(c, Q, 20)
    assert sorted(probes[0].tolist()) == list(range(20))
    D, I = ivf_ref.ivf_search(Xs, ids, off, Q, probes, 10)
    Df, If = ip_topk_ref.ip_topk(Xs, Q, 10, ids=ids)
    assert np.array_equal(I, If) and np.allclose(D, Df, atol=1e-6)
    # one probe: results come from exactly that list, padded when it is short
    probes1 = ivf_ref.coarse_probes(c, Q, 1)
    D1, I1 = ivf_ref.ivf_search(Xs, ids, off, Q, probes1, 500)
    for q in range(6):
        l = probes1[q, 0]
        n = off[l + 1] - off[l]
        assert set(I1[q, :n]) == set(ids[off[l]:off[l + 1]]) and (I1[q, n:] == -1).all() and (D1[q, n:] < -3e38).all()
    skipped = ivf_ref.ivf_search(Xs, ids, off, Q[:1], np.array([[-1, -1]]), 3)
    assert (skipped[1] == -1).all()


def test_reference_cell_count_rule():
    # feature_search_index.py:55-58
    assert reference_nlist(10000) == 3 * 100 and reference_nlist(199999) == 3 * round(199999 ** 0.5)
    assert reference_nlist(200000) == 10 * round(200000 ** 0.5) and reference_nlist(10_000_000) == 31620


@pytest.mark.parametrize("nlist,fill", [(12, "full"), (40, "sparse")])
def test_ivf_file_round_trip(tmp_path, nlist, fill):
    c, Xs, ids, off = make_lists(500, 16, nlist, 1)
    if fill == "sparse":      # most lists empty -> the 'sprs' size table
        keep = off[5]
        Xs, ids = Xs[:keep], ids[:keep]
        off = np.minimum(off, keep)
    fn = tmp_path / "image-IndexIVFFlat.faiss"
    faiss_io.write_ivf_flat_ip(fn, c, Xs, ids, off, nprobe=7)
    raw = fn.read_bytes()
    assert raw[:4] == b"IwFl" and (b"full" in raw[:4000]) == (fill == "full")
    f = faiss_io.read_ivf_flat_ip(fn)
    assert np.array_equal(f["centroids"], c) and np.array_equal(f["X"], Xs) and np.array_equal(f["ids"], ids)
    assert np.array_equal(f["list_off"], off) and f["nprobe"] == 7
    with pytest.raises(RuntimeError):
        faiss_io.read_ivf_flat_ip(tmp_path / "missing.faiss")
    flat = tmp_path / "flat.faiss"
    faiss_io.write_idmap_flat_ip(flat, Xs, ids)
    with pytest.raises(RuntimeError):
        faiss_io.read_ivf_flat_ip(flat)


def test_index_object_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    idx = IVFFlatIPIndex(16, 4)
    with pytest.raises(RuntimeError):
        idx.train(np.zeros((10, 16), dtype=np.float32))
    with pytest.raises(ValueError):
        IVFFlatIPIndex(10, 4)
