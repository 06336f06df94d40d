"""-m gpu: HP-2 parity — HIP scan+top-k (through the C ABI) against the CPU oracle and golden vectors."""
import numpy as np
import pytest
import torch

from oracle import ip_topk_ref
from wise_amd.index.flat_ip import FlatIPIndex

pytestmark = pytest.mark.gpu


def unit_rows(n, d, seed):
    x = np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def check_against_oracle(X, Q, k, ids, D, I, tol=2e-5):
    Dr, Ir = ip_topk_ref.ip_topk(X, Q, k, ids=ids)
    assert D.shape == Dr.shape and I.shape == Ir.shape and I.dtype == np.int64 and D.dtype == np.float32
    # padding is exact
    assert np.array_equal(I == -1, Ir == -1)
    assert np.all(D[I == -1] == ip_topk_ref.NEG)
    valid = Ir != -1
    assert np.allclose(D[valid], Dr[valid], atol=tol, rtol=0), np.abs(D[valid] - Dr[valid]).max()
    # descending
    assert np.all(np.diff(D, axis=1) <= 0)
    # ids: identical wherever the oracle's neighbouring scores are separated by more than fp noise
    S = (Q @ X.T).astype(np.float32) if X.shape[0] else None
    for q in range(Q.shape[0]):
        for j in range(k):
            if Ir[q, j] < 0 or I[q, j] == Ir[q, j]:
                continue
            # a swap is only acceptable between near-tied scores
            row = int(np.where(ids == I[q, j])[0][0]) if ids is not None else int(I[q, j])
            assert abs(S[q, row] - Dr[q, j]) <= tol, (q, j, I[q, j], Ir[q, j])
    assert ip_topk_ref.recall_at_k(I, Ir) >= 0.99


def test_golden_vectors(golden_dir):
    g = np.load(golden_dir / "ip_topk.npz")
    X = unit_rows(4096, 512, 2)
    Q = unit_rows(8, 512, 3)
    ids = np.arange(4096, dtype=np.int64) + 1
    idx = FlatIPIndex(512)
    for s in range(0, 4096, 512):  # batches of 512 like feature_search_index.py:80-82
        idx.add_with_ids(X[s:s + 512], ids[s:s + 512])
    assert idx.ntotal == 4096 and idx.d == 512
    for k in (1, 10, 100):
        D, I = idx.search(Q, k)
        assert np.array_equal(I, g[f"I{k}"])
        assert np.allclose(D, g[f"D{k}"], atol=2e-6, rtol=0)
    # tie case: identical rows, lower row first
    Xt = X[:1024].copy()
    Xt[[5, 17, 900]] = Q[0]
    it = FlatIPIndex(512)
    it.add_with_ids(Xt, ids[:1024])
    D, I = it.search(Q[:2], 5)
    assert np.array_equal(I, g["Itie"])
    assert list(I[0, :3]) == [6, 18, 901]
    # N < k: faiss padding
    ish = FlatIPIndex(512)
    ish.add_with_ids(X[:7], ids[:7])
    D, I = ish.search(Q[:2], 10)
    assert np.array_equal(I, g["Ishort"]) and np.all(I[:, 7:] == -1)
    assert np.all(D[:, 7:] == ip_topk_ref.NEG)
    assert np.allclose(D[:, :7], g["Dshort"][:, :7], atol=2e-6)


@pytest.mark.parametrize("N,d,nq,k", [
    (1, 512, 1, 1), (3, 512, 2, 10), (4097, 512, 1, 10), (5000, 768, 3, 10), (5000, 1024, 5, 100),
    (20000, 512, 8, 1000), (3001, 100, 2, 7), (2000, 4, 1, 3), (777, 2048, 2, 10), (70000, 512, 1, 2048),
    (50000, 64, 9, 33),
])
def test_shapes_against_oracle(N, d, nq, k):
    X = unit_rows(N, d, 100 + N % 97)
    Q = unit_rows(nq, d, 7)
    ids = (np.arange(N, dtype=np.int64) * 3 + 11)
    idx = FlatIPIndex(d)
    idx.add_with_ids(X, ids)
    D, I = idx.search(Q, k)
    check_against_oracle(X, Q, k, ids, D, I)


def test_empty_index_and_all_ties():
    idx = FlatIPIndex(512)
    D, I = idx.search(unit_rows(2, 512, 1), 5)
    assert np.all(I == -1) and np.all(D == ip_topk_ref.NEG)
    # every row identical (8 black frames give 8 identical embeddings): ids come back in row order
    row = unit_rows(1, 512, 9)
    X = np.repeat(row, 300, axis=0)
    idx = FlatIPIndex(512)
    idx.add_with_ids(X, np.arange(300, dtype=np.int64) + 1)
    D, I = idx.search(row, 10)
    assert list(I[0]) == list(range(1, 11))


def test_adopt_without_ids_and_reconstruct():
    X = unit_rows(1000, 512, 4)
    Xd = torch.from_numpy(X).cuda()
    idx = FlatIPIndex(512).adopt(Xd, None, id_base=1)
    D, I = idx.search(X[123:124], 3)
    assert I[0, 0] == 124 and abs(D[0, 0] - 1.0) < 1e-5
    rec = idx.reconstruct_batch([124, 1, 1000])
    assert np.array_equal(rec, X[[123, 0, 999]])
    idx2 = FlatIPIndex(512)
    idx2.add_with_ids(X, np.arange(1000, dtype=np.int64) * 2 + 5)
    rec = idx2.reconstruct_batch([5, 2003])
    assert np.array_equal(rec, X[[0, 999]])


def test_reserved_rows_are_filled_in_place():
    """FlatIPIndex.reserve (what load_index uses): slices land in one preallocated [N,d] tensor — no chunk list, no
    concatenated copy — and the index searches like one built in a single add; rows beyond the reservation still work."""
    N, d, k = 3000, 64, 5
    X = unit_rows(N, d, 21)
    ids = np.arange(N, dtype=np.int64) * 7 + 2
    Q = unit_rows(3, d, 22)
    one = FlatIPIndex(d)
    one.add_with_ids(X, ids)
    res = FlatIPIndex(d)
    res.reserve(N)
    base = res._rX.data_ptr()
    for s0 in range(0, N, 1024):
        res.add_with_ids(X[s0:s0 + 1024], ids[s0:s0 + 1024])
    assert res.ntotal == N and not res._chunks and res._X.data_ptr() == base
    Da, Ia = one.search(Q, k)
    Db, Ib = res.search(Q, k)
    assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)
    assert np.array_equal(res.reconstruct_batch(ids[[5, 2999]]), X[[5, 2999]])
    over = FlatIPIndex(d)
    over.reserve(2000)
    over.add_with_ids(X[:1500], ids[:1500])
    over.add_with_ids(X[1500:], ids[1500:])      # does not fit the reservation: chunk path
    Dc, Ic = over.search(Q, k)
    assert np.array_equal(Ia, Ic) and np.array_equal(Da, Dc)


def test_full_size_properties():
    """BASELINE cfg-3 shape (10M x 512) is too big for the oracle: check size-independent properties."""
    N, d = 2_000_000, 512
    g = torch.Generator(device="cuda").manual_seed(2)
    X = torch.randn(N, d, generator=g, device="cuda")
    X /= X.norm(dim=1, keepdim=True)
    idx = FlatIPIndex(d).adopt(X, None, id_base=1)
    # planted neighbours: query = a database row => that row is rank 1 with score 1
    rows = torch.tensor([0, 1, 999_999, N - 1, 123_457], device="cuda")
    D, I = idx.search_device(X[rows], 10)
    assert torch.equal(I[:, 0].cpu(), rows.cpu() + 1)
    assert torch.allclose(D[:, 0], torch.ones(5, device="cuda"), atol=1e-5)
    assert bool((D[:, :-1] >= D[:, 1:]).all())
    # against torch's own matmul+topk on the device (fp32)
    Dt, It = torch.topk(X[rows] @ X.T, 10, dim=1)
    assert torch.allclose(D, Dt, atol=2e-5)
    assert (I == It + 1).float().mean().item() >= 0.99
    # linearity: scores for 2q are 2x, same ids
    D2, I2 = idx.search_device(2.0 * X[rows], 10)
    assert torch.equal(I2, I) and torch.allclose(D2, 2 * D, atol=1e-5)


@pytest.mark.parametrize("N,d,nq,k", [
    (4097, 512, 8, 10), (5000, 512, 32, 10), (33333, 512, 70, 10), (31, 512, 9, 5), (1000, 64, 33, 16),
    (20000, 256, 64, 1), (50000, 512, 40, 16),
])
def test_batched_mfma_path_against_oracle(N, d, nq, k):
    """nq >= 8, k <= 16, d % 32 == 0, d <= 512 runs on the fp32 matrix cores (ip_topk_mfma.hip): same contract."""
    X = unit_rows(N, d, 300 + N % 89)
    Q = unit_rows(nq, d, 17)
    ids = (np.arange(N, dtype=np.int64) * 2 + 7)
    idx = FlatIPIndex(d)
    idx.add_with_ids(X, ids)
    D, I = idx.search(Q, k)
    check_against_oracle(X, Q, k, ids, D, I)


def test_batched_path_matches_single_query_path_and_ties():
    """A query's result must not depend on whether it travelled alone (VALU kernel) or in a batch (MFMA kernel);
    identical rows come back lowest row first in both."""
    N, d = 30000, 512
    X = unit_rows(N, d, 5)
    X[[100, 7, 29999, 15000]] = X[3]
    idx = FlatIPIndex(d)
    idx.add_with_ids(X, np.arange(N, dtype=np.int64) + 1)
    Q = np.concatenate([X[3:4], unit_rows(19, d, 6)], axis=0)
    Db, Ib = idx.search(Q, 10)
    assert list(Ib[0, :5]) == [4, 8, 101, 15001, 30000]
    for q in range(0, 20, 7):
        Ds, Is = idx.search(Q[q:q + 1], 10)
        assert np.array_equal(Is[0], Ib[q]), q
        assert np.allclose(Ds[0], Db[q], atol=2e-6)


def test_batched_split_candidates_resolve_near_ties():
    """The batched scan ranks candidates by split-bf16 scores (error <= 1.6e-5) and re-scores the best 16 in f32:
    a cluster of 14 rows whose scores differ by ~1e-5 must still come back in exact f32 order."""
    N, d, k = 40000, 512, 10
    X = unit_rows(N, d, 41)
    q = unit_rows(1, d, 42)[0]
    rng = np.random.default_rng(43)
    cluster = rng.choice(N, size=14, replace=False)
    for c in cluster:
        v = q + 3e-3 * rng.standard_normal(d).astype(np.float32)
        X[c] = v / np.linalg.norm(v)
    Q = np.concatenate([q[None], unit_rows(15, d, 44)], axis=0)
    ids = np.arange(N, dtype=np.int64) + 1
    idx = FlatIPIndex(d)
    idx.add_with_ids(X, ids)
    D, I = idx.search(Q, k)
    check_against_oracle(X, Q, k, ids, D, I)
    assert set(I[0]) <= set(cluster + 1)
    Ds, Is = idx.search(Q[:1], k)          # the single-query (VALU) path agrees
    assert np.array_equal(Is[0], I[0])


THRESHOLD_FORM_MIN_ROWS = 1 << 18   # csrc/ip_topk.hip COLLECT_MIN_ROWS: smaller indexes go to the fp32 scan directly


def counts_since(idx, before):
    """(answered from the shadow, handed to the fp32 scan) since `before`, from THIS index's own counters."""
    now = idx.shadow_counts()
    return now[0] - before[0], now[1] - before[1]


@pytest.mark.parametrize("N,d,k", [(300000, 512, 10), (262144, 256, 1), (270001, 768, 16), (400003, 1024, 5),
                                   (50000, 512, 10), (4097, 512, 1), (40, 512, 10), (64, 64, 16), (65, 128, 3),
                                   (300017, 128, 10), (262151, 272, 7), (270011, 528, 3)])
@pytest.mark.parametrize("kind", ["int8", "bf16"])
def test_single_query_two_stage_search_is_exact(N, d, k, kind):
    """nq = 1, k <= 16 on an index with a bf16 shadow — the reference's call shape (feature_search_index.py:113).
    From 2^18 rows on: sample -> threshold -> every row that could belong to the top-k collected from the bf16 rows ->
    exact fp32 scores; smaller indexes: the fp32 scan itself.  Either way the result is the oracle's and the f32 path's."""
    X = unit_rows(N, d, 500 + N % 97)
    ids = np.arange(N, dtype=np.int64) * 3 + 11
    idx = FlatIPIndex(d, shadow=kind)      # int8: 16 / 32 / 64 lanes per row (d <= 256 / 512 / 1024), ragged last groups
    idx.add_with_ids(X, ids)
    ref = FlatIPIndex(d, shadow=False)
    ref.add_with_ids(X, ids)
    before = idx.shadow_counts()
    for seed in range(4):
        Q = unit_rows(1, d, 900 + seed)
        if seed == 3:
            Q = (X[N // 3] + 0.02 * Q[0])[None]          # a query with a planted neighbour
        D, I = idx.search(Q, k)
        check_against_oracle(X, Q, k, ids, D, I)
        Dr, Ir = ref.search(Q, k)
        assert np.array_equal(I, Ir) and np.allclose(D, Dr, atol=2e-6)
    assert counts_since(idx, before) == ((4, 0) if N >= THRESHOLD_FORM_MIN_ROWS else (0, 0))


@pytest.mark.parametrize("d", [64, 512, 768])
def test_int8_shadow_rows_obey_their_definition(d):
    """wise_ip_shadow_i8: |c| <= 127, scale_r = max|x_r| / 127, every component within half a step of scale_r c, the error
    norms the search's bound is computed from are upper bounds of what they stand for (tests/test_shadow_bound_cpu.py has
    the bound itself)."""
    from wise_amd import _lib
    lib = _lib.lib()
    N = 5003
    rng = np.random.default_rng(5 + d)
    X = rng.standard_normal((N, d)).astype(np.float32) * np.exp(rng.uniform(-3, 3, size=(N, 1))).astype(np.float32)
    X[17] = 0.0                                                     # an all-zero row: scale 0, codes 0
    X[18, 3] = 1000.0
    Xd = torch.from_numpy(X).cuda()
    Xq = torch.empty(N, d, dtype=torch.int8, device="cuda")
    sc = torch.empty(N, device="cuda")
    norms = torch.zeros(4, device="cuda")
    _lib.check(lib.wise_ip_shadow_i8(Xd.data_ptr(), N, d, Xq.data_ptr(), sc.data_ptr(), norms.data_ptr(), _lib.stream_ptr()), "i8")
    torch.cuda.synchronize()
    c, s, nm = Xq.cpu().numpy().astype(np.float64), sc.cpu().numpy().astype(np.float64), norms.cpu().numpy().astype(np.float64)
    assert np.abs(c).max() <= 127 and np.all(c[17] == 0) and s[17] == 0
    mx = np.abs(X).max(axis=1).astype(np.float64)
    assert np.allclose(s, mx / 127, rtol=1e-6, atol=0)
    assert np.all(np.abs(c).max(axis=1)[mx > 0] == 127)             # the largest component sits on the last code
    back = s[:, None] * c
    assert np.all(np.abs(X - back) <= s[:, None] * 0.5 * (1 + 1e-4) + 1e-30)      # (x * (127 / max) rounds in fp32: ties move by 3e-5 of a step)
    assert nm[0] >= np.linalg.norm(back, axis=1).max() and nm[0] <= np.linalg.norm(back, axis=1).max() * 1.0001
    rho = np.linalg.norm(X - back, axis=1).max()
    assert rho <= nm[2] <= rho * 1.0001
    assert np.isclose(nm[1], nm[2] + np.sqrt(d) * 1.6e-5 * nm[0], rtol=1e-5)


@pytest.mark.parametrize("N,d,k", [(300000, 256, 1000), (600000, 512, 1000), (262144, 1024, 300)])
def test_int8_entry_point_beyond_the_k_the_index_class_sends_it(N, d, k):
    """FlatIPIndex keeps k > 256 on the bf16 copy (the int8 band fills most of the re-scoring list at k = 1000 on 10M rows); the
    entry point itself serves any k <= 1024: same ids and scores as the fp32 scan."""
    import wise_amd.torch_ops  # noqa: F401  (registers torch.ops.wise_hip)
    X = torch.from_numpy(unit_rows(N, d, 1700 + k)).cuda()
    Q = torch.from_numpy(unit_rows(2, d, 1701)).cuda()
    Xq, scales, norms = torch.ops.wise_hip.ip_shadow_i8(X)
    counters = torch.zeros(2, dtype=torch.int32, device="cuda")
    for q in range(2):
        D8, I8 = torch.ops.wise_hip.ip_topk_shadow8(X, Xq, scales, norms, Q[q:q + 1], k, None, 3, counters)
        Df, If = torch.ops.wise_hip.ip_topk(X, Q[q:q + 1], k, None, 3)
        assert torch.equal(I8, If) and torch.allclose(D8, Df, atol=2e-6)
    assert counters.tolist() == [2, 0]


def clustered_rows(n_items, per_item, d, seed, spread=0.03):
    """`per_item` near-duplicates of each of `n_items` directions, stored back to back — 2-fps frames of the same
    shot (extract-features.py:292-297,353); cosine between duplicates >= 1 - spread^2 (0.9991 at 0.03)."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n_items, 1, d), dtype=np.float32)
    base /= np.linalg.norm(base, axis=2, keepdims=True)
    X = base + (spread / np.sqrt(d)) * rng.standard_normal((n_items, per_item, d), dtype=np.float32)
    X /= np.linalg.norm(X, axis=2, keepdims=True)
    return X.reshape(n_items * per_item, d)


def test_single_query_search_on_clustered_rows_needs_no_fallback():
    """The data the reference actually indexes: runs of near-identical rows.  Here 20 near-duplicates per item
    (cosine >= 0.999) and, for one query, a run of 300 rows all within 1e-3 of each other's score — far more near-ties
    than a fixed-size candidate list holds and closer together than the bf16 error.  The threshold form collects them
    all: exact ids, nothing handed to the fp32 scan."""
    d, k = 512, 10
    X = clustered_rows(20000, 20, d, 41)
    N = X.shape[0]
    rng = np.random.default_rng(42)
    q_flat = unit_rows(1, d, 43)[0]
    run0 = 123456                                          # 300 consecutive rows scoring 0.3 +- 1e-3 against q_flat
    for r in range(run0, run0 + 300):
        v = rng.standard_normal(d).astype(np.float32)
        v -= (v @ q_flat) * q_flat
        v /= np.linalg.norm(v)
        s = 0.3 + 1e-3 * rng.uniform(-1, 1)
        X[r] = s * q_flat + np.sqrt(1 - s * s) * v
    ids = np.arange(N, dtype=np.int64) + 1
    idx = FlatIPIndex(d, shadow=True)
    idx.add_with_ids(X, ids)
    ref = FlatIPIndex(d, shadow=False)
    ref.add_with_ids(X, ids)
    before = idx.shadow_counts()
    queries = [q_flat, X[777] + 0.01 * unit_rows(1, d, 44)[0], X[N - 5], unit_rows(1, d, 45)[0]]
    for q in queries:
        D, I = idx.search(q[None], k)
        check_against_oracle(X, q[None], k, ids, D, I)
        Dr, Ir = ref.search(q[None], k)
        assert np.array_equal(I, Ir) and np.allclose(D, Dr, atol=2e-6)
    assert counts_since(idx, before) == (len(queries), 0)
    D, I = idx.search(q_flat[None], k)
    assert np.all((I[0] > run0) & (I[0] <= run0 + 300))   # the whole top-10 comes out of the flat run


def test_single_query_search_hands_over_when_the_candidate_list_overflows():
    """20,000 rows scoring within 2.5e-3 of the query's best: more rows inside the bf16 error band of the k-th score than
    the re-scoring list holds (16384), so the gate is raised and the fp32 scan queued behind gives the answer — the
    same bits the f32 path gives.  (The 12 best are 4e-5 apart so that the oracle's order is unambiguous.)"""
    N, d, k = 300000, 512, 10
    X = unit_rows(N, d, 61)
    q = unit_rows(1, d, 62)[0]
    rng = np.random.default_rng(63)
    for n, c in enumerate(rng.choice(N, size=20000, replace=False)):
        v = rng.standard_normal(d).astype(np.float32)
        v -= (v @ q) * q
        v /= np.linalg.norm(v)
        sc = 1.0 - 4e-5 * n if n < 12 else rng.uniform(0.9975, 0.9990)
        X[c] = sc * q + np.sqrt(1 - sc * sc) * v
    ids = np.arange(N, dtype=np.int64) + 1
    idx = FlatIPIndex(d, shadow=True)
    idx.add_with_ids(X, ids)
    before = idx.shadow_counts()
    D, I = idx.search(q[None], k)
    assert counts_since(idx, before) == (0, 1)
    check_against_oracle(X, q[None], k, ids, D, I)
    ref = FlatIPIndex(d, shadow=False)
    ref.add_with_ids(X, ids)
    Dr, Ir = ref.search(q[None], k)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)     # the fallback IS the f32 path
    # an ordinary query on the same index is answered from the shadow, and a second index keeps its own counters
    other = FlatIPIndex(d, shadow=True)
    other.add_with_ids(X[:270000], ids[:270000])
    b_idx, b_other = idx.shadow_counts(), other.shadow_counts()
    idx.search(unit_rows(1, d, 64), k)
    other.search(unit_rows(2, d, 65)[:1], k)
    other.search(q[None], k)                                   # 18,000 of the close rows are in this slice too: overflow
    assert counts_since(idx, b_idx) == (1, 0)
    assert counts_since(other, b_other) == (1, 1)


@pytest.mark.parametrize("N,d,k", [(300000, 512, 17), (300000, 512, 20), (400003, 512, 100), (300000, 256, 1000),
                                   (270001, 768, 100), (600000, 512, 1000), (262144, 1024, 128)])
@pytest.mark.parametrize("kind", ["int8", "bf16"])
def test_single_query_two_stage_search_at_the_k_the_reference_sends(N, d, k, kind):
    """nq = 1 at the k WISE's server and evaluations actually use: REST `end` defaults to 20 (api/routes.py:1171,1407),
    the index evaluation runs k = 100 (docs/Search-Index-Evaluation.md:109), the retrieval evaluation --topk 1000
    (docs/Retrieval-Evaluation.md:39).  Same threshold form, radix selections instead of k rounds of a maximum: the
    oracle's ids, the f32 scan's ids and scores, everything answered from the shadow."""
    X = unit_rows(N, d, 700 + N % 89 + k)
    ids = np.arange(N, dtype=np.int64) * 2 + 5
    idx = FlatIPIndex(d, shadow=kind)
    idx.add_with_ids(X, ids)
    ref = FlatIPIndex(d, shadow=False)
    ref.add_with_ids(X, ids)
    before = idx.shadow_counts()
    for seed in range(3):
        Q = unit_rows(1, d, 950 + seed)
        if seed == 2:
            Q = (X[N // 5] + 0.05 * Q[0])[None]
        D, I = idx.search(Q, k)
        check_against_oracle(X, Q, k, ids, D, I)
        Dr, Ir = ref.search(Q, k)
        assert np.array_equal(I, Ir) and np.allclose(D, Dr, atol=2e-6)
        assert np.all(np.diff(D[0]) <= 0)
    assert counts_since(idx, before) == (3, 0)


@pytest.mark.parametrize("k", [20, 100, 1000])
@pytest.mark.parametrize("kind", ["int8", "bf16"])
def test_general_k_on_clustered_rows_and_on_overflow(k, kind):
    """The same k on the data the reference indexes (runs of 20 near-duplicates, cosine >= 0.999): the threshold form
    keeps every row that could matter — for k = 1000 several thousand survive the refinement, more than the one-block
    finish re-scores itself, so the multi-block kernels behind it answer — and on an adversarial index (20,000 rows
    inside the bf16 error band of the k-th score) the lists overflow and the f32 scan queued behind gives the answer."""
    d = 512
    X = clustered_rows(20000, 20, d, 141)
    N = X.shape[0]
    ids = np.arange(N, dtype=np.int64) + 1
    idx = FlatIPIndex(d, shadow=kind)
    idx.add_with_ids(X, ids)
    ref = FlatIPIndex(d, shadow=False)
    ref.add_with_ids(X, ids)
    before = idx.shadow_counts()
    queries = [X[4321] + 0.01 * unit_rows(1, d, 144)[0], unit_rows(1, d, 145)[0]]
    for q in queries:
        D, I = idx.search(q[None], k)
        check_against_oracle(X, q[None], k, ids, D, I)
        Dr, Ir = ref.search(q[None], k)
        assert np.array_equal(I, Ir) and np.allclose(D, Dr, atol=2e-6)
    assert counts_since(idx, before) == (len(queries), 0)
    # overflow: the k + 2 best rows 2.5e-5 apart (an unambiguous order for the oracle), and right under the last of them
    # 20,000 rows within 1.5e-3: all inside the bf16 error band of the k-th score, more than the re-scoring list holds
    N2 = 300000
    X2 = unit_rows(N2, d, 161)
    q = unit_rows(1, d, 162)[0]
    rng = np.random.default_rng(163)
    kth = 0.9999 - 2.5e-5 * (k + 1)
    for n, c in enumerate(rng.choice(N2, size=20000 + k + 2, replace=False)):
        v = rng.standard_normal(d).astype(np.float32)
        v -= (v @ q) * q
        v /= np.linalg.norm(v)
        sc = 0.9999 - 2.5e-5 * n if n < k + 2 else rng.uniform(kth - 1.5e-3, kth - 3e-4)
        X2[c] = sc * q + np.sqrt(1 - sc * sc) * v
    ids2 = np.arange(N2, dtype=np.int64) + 1
    idx2 = FlatIPIndex(d, shadow=kind)
    idx2.add_with_ids(X2, ids2)
    ref2 = FlatIPIndex(d, shadow=False)
    ref2.add_with_ids(X2, ids2)
    b2 = idx2.shadow_counts()
    D, I = idx2.search(q[None], k)
    assert counts_since(idx2, b2) == (0, 1)
    Dr, Ir = ref2.search(q[None], k)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)     # the fallback IS the f32 path
    check_against_oracle(X2, q[None], k, ids2, D, I)


@pytest.mark.parametrize("N,d,k,nq", [(300000, 512, 20, 40), (300000, 512, 100, 130), (262144, 256, 128, 9),
                                      (270001, 768, 100, 70), (300000, 512, 17, 3)])
@pytest.mark.parametrize("kind", ["int8", "bf16"])
def test_batched_two_stage_search_up_to_k128(N, d, k, nq, kind):
    """Batches of queries at the evaluation's k (100: docs/Search-Index-Evaluation.md:109) and the REST default (20):
    the matrix-core passes over the bf16 shadow with the selections of the general-k path; ids and scores are the f32
    scan's and the oracle's, everything answered from the shadow (the gated fallback for k > 12 is the f32 VALU scan)."""
    X = unit_rows(N, d, 800 + k + nq)
    ids = np.arange(N, dtype=np.int64) + 7
    idx = FlatIPIndex(d, shadow=kind)
    idx.add_with_ids(X, ids)
    ref = FlatIPIndex(d, shadow=False)
    ref.add_with_ids(X, ids)
    Q = unit_rows(nq, d, 990 + k)
    Q[0] = X[N // 7] + 0.03 * Q[0]
    Q[0] /= np.linalg.norm(Q[0])
    before = idx.shadow_counts()
    D, I = idx.search(Q, k)
    assert counts_since(idx, before) == (nq, 0)
    check_against_oracle(X, Q, k, ids, D, I, tol=2e-5)
    Dr, Ir = ref.search(Q, k)
    assert np.array_equal(I, Ir) and np.allclose(D, Dr, atol=2e-5)


def test_small_index_with_near_ties_in_one_block():
    """N ~ 100 with more than 16 rows within 1e-3 of the query.  (The old two-stage path kept 16 rows per
    scan block and could report such a query exact; small indexes now never enter the two-stage path.)"""
    N, d, k = 100, 512, 10
    X = unit_rows(N, d, 51)
    q = unit_rows(1, d, 52)[0]
    rng = np.random.default_rng(53)
    for c in range(3, 3 + 24):
        v = q + 1e-3 * rng.standard_normal(d).astype(np.float32)
        X[c] = v / np.linalg.norm(v)
    ids = np.arange(N, dtype=np.int64) + 1
    idx = FlatIPIndex(d, shadow=True)
    idx.add_with_ids(X, ids)
    D, I = idx.search(q[None], k)
    Dr, Ir = ip_topk_ref.ip_topk(X, q[None], k, ids=ids)
    assert np.array_equal(I, Ir) and np.allclose(D, Dr, atol=2e-6)


@pytest.mark.parametrize("kind", ["int8", "bf16"])
def test_batched_two_stage_search_on_ordinary_and_on_overflowing_queries(kind):
    """A batch of 40 queries on an index with a bf16 shadow (threshold form on the matrix cores): ordinary queries are
    answered from the bf16 pass; one query with 70,000 rows inside the bf16 error band of its best scores overflows its
    list, and the pass is redone from the f32 rows.  Either way: the oracle's."""
    N, d, k = 300000, 512, 10
    X = unit_rows(N, d, 71)
    Q = unit_rows(40, d, 72)
    ids = np.arange(N, dtype=np.int64) + 5
    idx = FlatIPIndex(d, shadow=kind)
    idx.add_with_ids(X, ids)
    before = idx.shadow_counts()
    D, I = idx.search(Q, k)
    assert counts_since(idx, before) == (40, 0)
    check_against_oracle(X, Q, k, ids, D, I)
    rng = np.random.default_rng(73)
    for n, c in enumerate(rng.choice(N, size=70000, replace=False)):
        v = rng.standard_normal(d).astype(np.float32)
        v -= (v @ Q[7]) * Q[7]
        v /= np.linalg.norm(v)
        sc = 1.0 - 4e-5 * n if n < 12 else rng.uniform(0.9975, 0.9990)
        X[c] = sc * Q[7] + np.sqrt(1 - sc * sc) * v
    idx2 = FlatIPIndex(d, shadow=kind)
    idx2.add_with_ids(X, ids)
    D2, I2 = idx2.search(Q, k)
    answered, handed = idx2.shadow_counts()
    assert handed >= 1 and answered + handed == 40
    assert counts_since(idx, before) == (40, 0)               # the first index's counters did not move
    check_against_oracle(X, Q, k, ids, D2, I2)


@pytest.mark.parametrize("kind", ["int8", "bf16"])
def test_batched_search_on_clustered_rows_needs_no_fallback(kind):
    """Runs of 20 near-duplicates (cosine >= 0.999): the batched threshold form answers every query from the shadow and
    returns what the f32 path returns."""
    d, k = 512, 10
    X = clustered_rows(20000, 20, d, 47)
    N = X.shape[0]
    ids = np.arange(N, dtype=np.int64) + 1
    Q = unit_rows(70, d, 48)
    Q[::2] = X[np.random.default_rng(49).integers(0, N, 35)] + 0.02 * Q[::2]
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    idx = FlatIPIndex(d, shadow=kind)
    idx.add_with_ids(X, ids)
    ref = FlatIPIndex(d, shadow=False)
    ref.add_with_ids(X, ids)
    D, I = idx.search(Q, k)
    assert idx.shadow_counts() == (70, 0)
    Dr, Ir = ref.search(Q, k)
    check_against_oracle(X, Q, k, ids, D, I)
    assert np.allclose(D, Dr, atol=2e-6)
    assert (I == Ir).mean() > 0.99            # ids may swap only between scores equal to f32 rounding


def test_batched_two_stage_search_with_threshold_passes():
    """d = 256 rows, 70 queries (one full pass of 64 and one of 6)."""
    N, d, k = 600000, 256, 10
    X = unit_rows(N, d, 81)
    Q = unit_rows(70, d, 82)
    ids = np.arange(N, dtype=np.int64) + 1
    idx = FlatIPIndex(d, shadow=True)
    idx.add_with_ids(X, ids)
    D, I = idx.search(Q, k)
    certified, fallback = idx.shadow_counts()
    assert certified + fallback == 70
    check_against_oracle(X, Q, k, ids, D, I)


@pytest.mark.parametrize("d,nq", [(512, 65), (512, 128), (512, 129), (512, 200), (512, 257), (768, 65), (768, 130), (256, 300)])
@pytest.mark.parametrize("kind", ["int8", "bf16"])
def test_batched_pass_sizes_and_remainders(d, nq, kind):
    """passes of 128 queries at d <= 512 (64 at d = 768), a remainder of <= 64 through the 64-query kernel, single-query
    remainders; two collect ranges (N >= 4M rows) — ids and scores equal the f32 scan's for every query"""
    N, k = 4_300_000, 10
    g = torch.Generator(device="cuda").manual_seed(d + nq)
    X = torch.empty(N, d, device="cuda")
    for s0 in range(0, N, 1_000_000):
        n = min(1_000_000, N - s0)
        X[s0:s0 + n] = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(nq, d, device="cuda", generator=g), dim=1)
    ref = FlatIPIndex(d, shadow=False).adopt(X)
    D0, I0 = ref.search_device(Q, k)
    del ref
    idx = FlatIPIndex(d, shadow=kind).adopt(X)
    before = idx.shadow_counts()
    D, I = idx.search_device(Q, k)
    certified, fallback = counts_since(idx, before)
    assert certified + fallback == nq and fallback == 0
    assert torch.equal(I, I0) and torch.equal(D, D0)


def test_full_size_two_stage_against_the_f32_scan():
    """BASELINE cfg-3 size (10M x 512): the two-stage searches (one query; batches of 64 on the matrix cores, with
    the threshold pass and both row ranges) return exactly what the f32 scans of the same index return, and what
    torch's own f32 matmul + topk returns; scores scale linearly with the query."""
    N, d, k = 10_000_000, 512, 10
    g = torch.Generator(device="cuda").manual_seed(12)
    X = torch.randn(N, d, generator=g, device="cuda")
    X /= X.norm(dim=1, keepdim=True)
    Q = torch.randn(96, d, generator=g, device="cuda")
    Q /= Q.norm(dim=1, keepdim=True)
    Q[0] = X[N - 1]                       # a planted neighbour in the last row
    two = FlatIPIndex(d, shadow=True).adopt(X, None, id_base=1)
    f32 = FlatIPIndex(d, shadow=False).adopt(X, None, id_base=1)
    Db, Ib = two.search_device(Q, k)       # batched two-stage: 64 + 32
    Dr, Ir = f32.search_device(Q, k)       # split-bf16 candidates of the f32 rows + exact re-scoring
    assert torch.equal(Ib, Ir) and torch.allclose(Db, Dr, atol=2e-6)
    assert Ib[0, 0].item() == N and abs(Db[0, 0].item() - 1.0) < 1e-5
    for q in (0, 1, 17, 95):               # one query at a time: threshold form over the bf16 rows vs the f32 VALU scan
        D1, I1 = two.search_device(Q[q:q + 1], k)
        D0, I0 = f32.search_device(Q[q:q + 1], k)
        assert torch.equal(I1, I0) and torch.allclose(D1, D0, atol=2e-6)
        assert torch.equal(I1[0], Ib[q])
    assert two.shadow_counts() == (96 + 4, 0)   # iid rows: every query answered from the shadow
    Dt, It = torch.topk(Q[:8] @ X.T, k, dim=1)
    assert torch.equal(Ib[:8], It + 1) and torch.allclose(Db[:8], Dt, atol=2e-5)
    D2, I2 = two.search_device(2.0 * Q[:40], k)
    assert torch.equal(I2, Ib[:40]) and torch.allclose(D2, 2 * Db[:40], atol=2e-5)


def test_cfg4_shard_6p25m_x_768():
    """BASELINE cfg-4, search half: ONE rank's shard of the 50M x 768 index (6.25M rows, ids = global row + 1 of rank 3
    of 8).  Two-stage searches (one query; 32-query matrix-core passes for d = 768) against the f32 scan of the same
    shard and against torch's own f32 matmul + topk; scores scale linearly with the query; the per-shard lists of two
    'ranks' merged by wise_topk_merge equal the search over both halves at once (feature_search_index.py:113 sharded)."""
    from wise_amd.index.sharded import merge_device, shard_range
    N_total, world, rank = 50_000_000, 8, 3
    lo, hi = shard_range(N_total, rank, world)
    N, d, k = hi - lo, 768, 10
    assert N == 6_250_000
    g = torch.Generator(device="cuda").manual_seed(100 + rank)
    X = torch.empty(N, d, device="cuda")
    for s0 in range(0, N, 1_250_000):
        blk = torch.randn(1_250_000, d, generator=g, device="cuda")
        X[s0:s0 + 1_250_000] = blk / blk.norm(dim=1, keepdim=True)
    Q = torch.randn(40, d, generator=g, device="cuda")
    Q /= Q.norm(dim=1, keepdim=True)
    Q[0] = X[N - 1]
    Q[1] = torch.nn.functional.normalize(X[12345] + 0.05 * Q[1], dim=0)
    two = FlatIPIndex(d, shadow=True).adopt(X, None, id_base=lo + 1)
    f32 = FlatIPIndex(d, shadow=False).adopt(X, None, id_base=lo + 1)
    Db, Ib = two.search_device(Q, k)              # 32 + 8 queries: batched two-stage passes
    Dr, Ir = f32.search_device(Q[:8], k)
    assert torch.equal(Ib[:8], Ir) and torch.allclose(Db[:8], Dr, atol=2e-6)
    assert Ib[0, 0].item() == lo + N and abs(Db[0, 0].item() - 1.0) < 1e-5 and Ib[1, 0].item() == lo + 12345 + 1
    for q in (0, 1, 5, 39):                        # the reference's call shape, one query at a time
        D1, I1 = two.search_device(Q[q:q + 1], k)
        assert torch.equal(I1[0], Ib[q]) and torch.allclose(D1[0], Db[q], atol=2e-6)
    done, handed = two.shadow_counts()
    assert done + handed == 40 + 4 and handed == 0
    Dt, It = torch.topk(Q[:8] @ X.T, k, dim=1)
    assert torch.equal(Ib[:8], It + lo + 1) and torch.allclose(Db[:8], Dt, atol=2e-5)
    D2, I2 = two.search_device(3.0 * Q[:8], k)
    assert torch.equal(I2, Ib[:8]) and torch.allclose(D2, 3 * Db[:8], atol=6e-5)
    # two half shards searched separately and merged == the whole shard
    h = N // 2
    a = FlatIPIndex(d, shadow=False).adopt(X[:h], None, id_base=lo + 1)
    b = FlatIPIndex(d, shadow=False).adopt(X[h:], None, id_base=lo + h + 1)
    Da, Ia = a.search_device(Q[:8], k)
    Dbb, Ibb = b.search_device(Q[:8], k)
    Dm, Im = merge_device(torch.stack([Da, Dbb]), torch.stack([Ia, Ibb]), k)
    assert torch.equal(Im, Ir) and torch.allclose(Dm, Dr, atol=2e-6)


@pytest.mark.parametrize("N,d,nq,k", [(50000, 512, 4, 10), (33333, 512, 7, 16), (20000, 768, 3, 5), (70, 256, 6, 10),
                                      (40000, 1024, 2, 10), (300000, 512, 4, 10), (270000, 768, 3, 5), (280000, 256, 2, 16)])
def test_few_queries_through_the_two_stage_search(N, d, nq, k):
    """2-7 queries: one 64-query matrix-core pass over the bf16 rows where those kernels apply (k <= 12, d = 256 / 512),
    else one query at a time (2-3 queries) or the f32 batch kernels; each certified or recomputed from the f32 rows."""
    X = unit_rows(N, d, 600 + N % 53)
    Q = unit_rows(nq, d, 31)
    ids = np.arange(N, dtype=np.int64) * 2 + 3
    idx = FlatIPIndex(d, shadow=True)
    idx.add_with_ids(X, ids)
    D, I = idx.search(Q, k)
    certified, fallback = idx.shadow_counts()
    # indexes below 2^18 rows never enter the two-stage path (the f32 scans answer); above: 64-query passes (k <= 12,
    # d = 256 / 512), 32-query passes (3+ queries, d = 768 / 1024), or one query at a time (2-3 queries)
    two_stage = N >= THRESHOLD_FORM_MIN_ROWS and ((k <= 12 and d in (256, 512)) or (nq >= 3 and d in (768, 1024)) or nq <= 3)
    assert (certified, fallback) == ((nq, 0) if two_stage else (0, 0))
    check_against_oracle(X, Q, k, ids, D, I)
    ref = FlatIPIndex(d, shadow=False)
    ref.add_with_ids(X, ids)
    Dr, Ir = ref.search(Q, k)
    assert np.array_equal(I, Ir) and np.allclose(D, Dr, atol=2e-6)


@pytest.mark.parametrize("N,d,nq,k", [(280000, 768, 40, 10), (30000, 1024, 5, 16), (600000, 768, 33, 10), (300000, 1024, 5, 16)])
@pytest.mark.parametrize("kind", ["int8", "bf16"])
def test_batched_two_stage_search_for_wide_rows(N, d, nq, k, kind):
    """512 < d <= 1024 (768 is the ViT-L/14 dimension): 32 queries per pass over the bf16 rows on the matrix cores, the
    f32 VALU scan as the gated fallback."""
    X = unit_rows(N, d, 700 + N % 31)
    Q = unit_rows(nq, d, 33)
    ids = np.arange(N, dtype=np.int64) + 9
    idx = FlatIPIndex(d, shadow=kind)
    idx.add_with_ids(X, ids)
    D, I = idx.search(Q, k)
    certified, fallback = idx.shadow_counts()
    assert certified + fallback == (nq if N >= THRESHOLD_FORM_MIN_ROWS else 0)
    check_against_oracle(X, Q, k, ids, D, I)


def test_wide_row_batch_falls_back_to_the_f32_scan():
    N, d, k = 270000, 768, 10
    X = unit_rows(N, d, 91)
    Q = unit_rows(12, d, 92)
    rng = np.random.default_rng(93)
    for n, c in enumerate(rng.choice(N, size=70000, replace=False)):
        v = rng.standard_normal(d).astype(np.float32)
        v -= (v @ Q[5]) * Q[5]
        v /= np.linalg.norm(v)
        sc = 1.0 - 4e-5 * n if n < 12 else rng.uniform(0.9975, 0.9990)
        X[c] = sc * Q[5] + np.sqrt(1 - sc * sc) * v
    ids = np.arange(N, dtype=np.int64) + 1
    idx = FlatIPIndex(d, shadow=True)
    idx.add_with_ids(X, ids)
    D, I = idx.search(Q, k)
    certified, fallback = idx.shadow_counts()
    assert fallback >= 1 and certified + fallback == 12
    check_against_oracle(X, Q, k, ids, D, I)


def test_searches_beside_matrix_kernels_of_another_stream():
    """The two-stage searches (one query; a batch of 32) run while another stream keeps the matrix cores busy return
    what they return alone, bit for bit, every time.  (Built with packed f32 VALU math the candidate re-scoring kernel
    returned dot products off by up to 5e-2 in about one call of four — see wise_amd/build.py.)"""
    import ctypes

    from wise_amd import _lib
    lib = _lib.load_debug()   # the neighbour (a bare MFMA loop) lives in the debug twin; the searches are the product's
    lib.wise_debug_neighbour.restype = ctypes.c_int
    lib.wise_debug_neighbour.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
    other = torch.cuda.Stream()
    g = torch.Generator("cuda").manual_seed(1)
    X = torch.nn.functional.normalize(torch.randn(1_000_000, 512, device="cuda", generator=g), dim=1)
    idx = FlatIPIndex(512, shadow=True).adopt(X)
    src = torch.zeros(1024, dtype=torch.int32, device="cuda")
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    for q in (X[100:132] + 0.01, X[7:8] + 0.01):
        D0, I0 = idx.search_device(q, 10)
        torch.cuda.synchronize()
        for _ in range(12):
            for _ in range(4):
                lib.wise_debug_neighbour(3, 2048, 1024, 64, src.data_ptr(), sink.data_ptr(), other.cuda_stream)
            D1, I1 = idx.search_device(q, 10)
            for _ in range(2):
                lib.wise_debug_neighbour(3, 2048, 1024, 64, src.data_ptr(), sink.data_ptr(), other.cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(I1, I0) and torch.equal(D1, D0)
