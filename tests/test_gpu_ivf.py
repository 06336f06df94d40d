"""-m gpu: IndexIVFFlat — the inverted-list scan through the C ABI against the CPU oracle, and the
faiss-shaped index object: training, adding, nprobe semantics, file round trip, the SearchIndex surface."""
import numpy as np
import pytest
import torch

from oracle import ip_topk_ref, ivf_ref
from wise_amd.index import faiss_io
from wise_amd.index.ivf_flat import IVFFlatIPIndex, reference_nlist

pytestmark = pytest.mark.gpu


def unit_rows(n, d, seed):
    x = np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def check_against_oracle(D, I, Do, Io, tol=2e-5):
    assert D.shape == Do.shape and I.dtype == np.int64
    assert np.allclose(D, Do, atol=tol)
    # ids must agree wherever the oracle's neighbouring scores are further apart than the tolerance
    gap_ok = np.ones_like(Io, dtype=bool)
    gap_ok[:, 1:] &= (Do[:, :-1] - Do[:, 1:]) > tol
    gap_ok[:, :-1] &= (Do[:, :-1] - Do[:, 1:]) > tol
    assert np.array_equal(I[gap_ok], Io[gap_ok])


@pytest.mark.parametrize("N,d,nlist,nprobe,nq,k", [(20000, 128, 64, 8, 5, 10), (5000, 512, 37, 37, 3, 10),
                                                   (30000, 64, 200, 1, 9, 5), (3000, 768, 16, 4, 2, 100),
                                                   (1000, 32, 50, 60, 4, 7), (30000, 64, 400, 128, 6, 10),
                                                   (8000, 128, 90, 90, 3, 20)])
def test_list_scan_equals_oracle_on_the_same_lists(N, d, nlist, nprobe, nq, k):
    X = unit_rows(N, d, N + d)
    Q = unit_rows(nq, d, 7)
    idx = IVFFlatIPIndex(d, nlist)
    idx.train(X[: max(nlist * 20, min(N, 4000))])
    ids = (np.arange(N, dtype=np.int64) * 3 + 11)
    for s in range(0, N, 7000):
        idx.add_with_ids(X[s:s + 7000], ids[s:s + 7000])
    idx.nprobe = nprobe
    D, I = idx.search(Q, k)
    c, Xs, ids_s, off = idx.lists_host()
    assert off[-1] == N and sorted(ids_s.tolist()) == sorted(ids.tolist())
    assert np.array_equal(ivf_ref.assign(Xs, c), np.repeat(np.arange(nlist), off[1:] - off[:-1]))   # lists are right
    probes = idx.probes_device(torch.from_numpy(Q).cuda(), min(nprobe, nlist)).cpu().numpy()
    # stage 1 is the flat kernel over the centroids
    want_probes = ivf_ref.coarse_probes(c, Q, min(nprobe, nlist))
    # (more than 64 probes come from the radix-select kernel, in list order rather than score order)
    same = np.mean([len(set(probes[q]) & set(want_probes[q])) / probes.shape[1] for q in range(nq)])
    assert same > 0.98
    Do, Io = ivf_ref.ivf_search(Xs, ids_s, off, Q, probes, k)
    check_against_oracle(D, I, Do, Io)
    if nprobe >= nlist:   # probing every list is the exhaustive search
        Df, If = ip_topk_ref.ip_topk(X, Q, k, ids=ids)
        check_against_oracle(D, I, Df, If)


def test_recall_grows_with_nprobe_and_padding():
    N, d, nlist = 40000, 64, reference_nlist(40000)
    assert nlist == 3 * 200
    # embeddings are clustered; structureless unit vectors would make any IVF look bad
    rng = np.random.default_rng(1)
    centres = unit_rows(300, d, 3)
    X = centres[rng.integers(0, 300, N)] + 0.35 * unit_rows(N, d, 1)
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    Q = X[:64] + 0.05 * unit_rows(64, d, 2)
    idx = IVFFlatIPIndex(d, nlist)
    idx.train(X[:20000])
    idx.add_with_ids(X, np.arange(N, dtype=np.int64))
    _, If = ip_topk_ref.ip_topk(X, Q, 10)
    recalls = []
    for nprobe in (1, 8, 64, nlist):
        idx.nprobe = nprobe
        _, I = idx.search(Q, 10)
        recalls.append(np.mean([len(set(I[q]) & set(If[q])) / 10 for q in range(64)]))
    assert recalls == sorted(recalls) and recalls[-1] == 1.0 and recalls[1] > 0.8, recalls
    # a probe that yields fewer than k rows is padded like faiss: (-3.4e38, -1)
    idx.nprobe = 1
    D, I = idx.search(Q[:2], 2048)
    assert (I[:, -1] == -1).all() and (D[:, -1] < -3e38).all()
    sizes = np.diff(idx.lists_host()[3])
    first_pad = (I[0] == -1).argmax()
    assert first_pad in sizes


def test_surface_the_rest_layer_touches_and_file_round_trip(tmp_path):
    N, d, nlist = 6000, 32, 40
    X = unit_rows(N, d, 5)
    ids = np.arange(N, dtype=np.int64) + 1
    idx = IVFFlatIPIndex(d, nlist)
    with pytest.raises(RuntimeError):
        idx.add_with_ids(X, ids)                      # not trained
    idx.train(X)
    idx.add_with_ids(X, ids)
    assert idx.ntotal == N and idx.d == d and idx.is_trained and hasattr(idx, "nprobe")
    idx.parallel_mode = 1                             # routes.py:901
    idx.nprobe = 32                                   # routes.py:902
    idx.make_direct_map(True)                         # routes.py:907
    assert idx.direct_map.type != idx.direct_map.NoMap
    rec = idx.reconstruct_batch([5, 17, 6000])        # routes.py:1078
    assert np.array_equal(rec, X[[4, 16, 5999]])
    D, I = idx.search(X[:3], 5)
    assert (I[:, 0] == ids[:3]).all()
    fn = tmp_path / "video-IndexIVFFlat.faiss"
    c, Xs, ids_s, off = idx.lists_host()
    faiss_io.write_ivf_flat_ip(fn, c, Xs, ids_s, off, nprobe=idx.nprobe)
    assert faiss_io.index_fourcc(fn) == "IwFl"
    f = faiss_io.read_ivf_flat_ip(fn)
    assert np.array_equal(f["centroids"], c) and np.array_equal(f["X"], Xs) and np.array_equal(f["ids"], ids_s)
    assert np.array_equal(f["list_off"], off) and f["nprobe"] == 32
    again = IVFFlatIPIndex(d, nlist)
    again.set_centroids(f["centroids"])
    again.adopt_lists(torch.from_numpy(f["X"]), torch.from_numpy(f["ids"]), torch.from_numpy(f["list_off"]))
    again.nprobe = 32
    D2, I2 = again.search(X[:3], 5)
    assert np.array_equal(I, I2) and np.array_equal(D, D2)


def test_search_index_builds_and_loads_an_ivf_index(tmp_path):
    from wise_amd.feature.store.feature_store_factory import FeatureStoreFactory, FeatureStoreType
    from wise_amd.index.search_index_factory import SearchIndexFactory

    fdir, idir = tmp_path / "features", tmp_path / "index"
    fdir.mkdir()
    X = unit_rows(3000, 512, 9)
    st = FeatureStoreFactory.create_store(FeatureStoreType.WEBDATASET, "video", str(fdir))
    st.enable_write(2048, 20 * 1024 * 1024)
    for i in range(X.shape[0]):
        st.add(i + 1, X[i:i + 1])
    st.close()
    si = SearchIndexFactory("video", "mlfoundations/open_clip/ViT-B-32/seeded-0", {"features_dir": fdir,
                                                                                  "index_dir": idir})
    si.create_index("IndexIVFFlat")
    assert si.get_index_filename("IndexIVFFlat").exists()
    assert si.load_index("IndexIVFFlat") is True
    assert si.index.nlist == reference_nlist(3000) and si.index.ntotal == 3000
    si.index.nprobe = si.index.nlist
    D, I = si.index.search(X[:4], 3)
    assert (I[:, 0] == np.arange(4) + 1).all() and np.allclose(D[:, 0], 1.0, atol=1e-5)
    dist, ids = si.search("video", "dog", topk=5)
    assert dist.shape == (5,) and ids.shape == (5,) and (ids >= 1).all()


@pytest.mark.parametrize("rows,n,k", [(1, 31620, 1024), (5, 1000, 65), (3, 70, 70), (2, 50, 64), (4, 4097, 1), (2, 100000, 2048)])
def test_select_topk_kernel(rows, n, k):
    from wise_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(rows * n + k)
    S = rng.standard_normal((rows, n)).astype(np.float32)
    S[0, : n // 3] = np.round(S[0, : n // 3], 1)          # many exact ties
    S[-1, -1] = np.float32(-0.0); S[-1, 0] = np.float32(0.0)
    Sd = torch.from_numpy(S).cuda()
    out = torch.empty(rows, k, dtype=torch.int64, device="cuda")
    _lib.check(lib.wise_select_topk_f32(Sd.data_ptr(), rows, n, k, out.data_ptr(), _lib.stream_ptr()), "select")
    got = out.cpu().numpy()
    kk = min(k, n)
    for r in range(rows):
        order = np.lexsort((np.arange(n), -S[r].astype(np.float64)))[:kk]    # score descending, lower index on ties
        # -0.0 and +0.0 are different keys to the kernel (+0.0 is larger), numpy treats them as equal: compare values
        assert np.array_equal(np.sort(S[r, got[r, :kk]])[::-1], S[r, order])
        assert (np.diff(got[r, :kk]) > 0).all()                              # ascending index order, no duplicates
        if not (S[r] == 0).any():
            assert set(got[r, :kk]) == set(order)
        assert (got[r, kk:] == -1).all()


@pytest.mark.parametrize("N,d,nq", [(31620, 512, 1), (31620, 512, 256), (1000, 768, 40), (129, 64, 33), (5, 4, 1), (300, 100, 7)])
def test_dense_scores_kernel(N, d, nq):
    """wise_ip_scores_f32 (the coarse stage at the reference's nprobe = 1024, config.py:19): exact-f32 scores on the
    matrix cores.  One-hot queries must return the rows' elements bit for bit (and catch a transposed tile);
    general queries agree with a float64 product to f32 accumulation accuracy."""
    from wise_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(N + d + nq)
    X = rng.standard_normal((N, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    cols = rng.integers(0, d, size=nq)
    hot = min(nq, 5)
    Q[:hot] = 0
    Q[np.arange(hot), cols[:hot]] = 1.0
    Xd, Qd = torch.from_numpy(X).cuda(), torch.from_numpy(Q).cuda()
    S = torch.full((nq, N), float("nan"), device="cuda")
    _lib.check(lib.wise_ip_scores_f32(Xd.data_ptr(), N, d, Qd.data_ptr(), nq, S.data_ptr(), _lib.stream_ptr()), "scores")
    got = S.cpu().numpy()
    assert np.isfinite(got).all()
    for i in range(hot):
        assert np.array_equal(got[i], X[:, cols[i]])
    ref = Q.astype(np.float64) @ X.astype(np.float64).T
    bound = 1e-6 * (np.abs(Q).astype(np.float64) @ np.abs(X).astype(np.float64).T) + 1e-12
    assert (np.abs(got - ref) <= bound).all()


def test_coarse_stage_at_the_reference_nprobe():
    """nprobe = 1024 (config.py:19): probes = the 1024 best centroids by exact-f32 score (wise_ip_scores_f32 +
    wise_select_topk_f32), against numpy on the same centroids."""
    nlist, d, nq = 4000, 128, 6
    cent = unit_rows(nlist, d, 5)
    idx = IVFFlatIPIndex(d, nlist)
    idx.set_centroids(cent)
    Q = torch.from_numpy(unit_rows(nq, d, 6)).cuda()
    probes = idx.probes_device(Q, 1024).cpu().numpy()
    S = (Q.cpu().numpy().astype(np.float64) @ cent.astype(np.float64).T)
    for q in range(nq):
        want = set(np.argsort(-S[q])[:1024])
        got = set(probes[q])
        edge = np.sort(S[q])[-1024]
        # only centroids within f32 rounding of the 1024th score may differ
        assert all(abs(S[q, c] - edge) < 1e-6 for c in want ^ got)
        assert len(got) == 1024
