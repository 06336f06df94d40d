"""CPU test (-m "not gpu"): the N>1 search path (row sharding, all-gather of per-shard top-k, merge)
with world_size 2 over gloo.  Local search and merge are played by the oracle here (no GPU in this
container); on the GPU box the same class runs the HIP kernels over RCCL (bench.py --gpus N)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, d, k, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ip_topk_ref
    from wise_amd.index.sharded import ShardedFlatIPIndex, shard_range

    X = np.random.default_rng(5).standard_normal((N, d)).astype(np.float32)  # every rank regenerates, keeps its slice
    Q = np.random.default_rng(6).standard_normal((5, d)).astype(np.float32)
    lo, hi = shard_range(N, rank, world)

    class Local:  # stands where FlatIPIndex stands on a GPU
        def __init__(self, dim, n):
            self.d, self.ntotal, self.device = dim, n, torch.device("cpu")

        def search_device(self, q, kk):
            raise RuntimeError("not used: local_search is injected")

    def local_search(q, kk):
        D, I = ip_topk_ref.ip_topk(X[lo:hi], q.numpy(), kk, id_base=lo + 1)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(Ds, Is, kk):
        D, I = ip_topk_ref.merge_topk(Ds.numpy(), Is.numpy(), kk)
        return torch.from_numpy(D), torch.from_numpy(I)

    idx = ShardedFlatIPIndex(Local(d, hi - lo), local_search=local_search, merge=merge)
    assert idx.world == world
    assert idx.ntotal == N
    D, I = idx.search_device(torch.from_numpy(Q), k)
    np.savez(Path(out_dir) / f"rank{rank}.npz", D=D.numpy(), I=I.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_world2(tmp_path):
    from oracle import ip_topk_ref

    N, d, k, world = 1001, 64, 10, 2
    mp.spawn(_worker, args=(world, _free_port(), N, d, k, str(tmp_path)), nprocs=world, join=True)
    X = np.random.default_rng(5).standard_normal((N, d)).astype(np.float32)
    Q = np.random.default_rng(6).standard_normal((5, d)).astype(np.float32)
    Dr, Ir = ip_topk_ref.ip_topk(X, Q, k, id_base=1)
    for r in range(world):
        g = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(g["I"], Ir), f"rank {r}: sharded ids differ from the unsharded search"
        assert np.array_equal(g["D"], Dr)


# ---------------------------------------------------------------------------------------------------------------------
# HP-1 over several ranks: file-level sharding (src/dataloader/dataset.py:334), per-rank vector ids and shard ranges
def _media_files(n_files, size=8):
    """Deterministic synthetic media: file `mid` -> its chunks (what the dataloader yields for it)."""
    from tests.test_extract_driver import Chunk

    out = []
    for mid in range(1, n_files + 1):
        g = torch.Generator().manual_seed(1000 + mid)
        chunks = []
        n_chunks = 2 + mid % 3
        for c in range(n_chunks):
            last = c == n_chunks - 1
            nf = 8 if not last else 3 + mid % 5
            ch = {"video": Chunk(torch.randn(nf, 3, size, size, generator=g), c * 4.0)}
            ns = 192000 if not last else (192000 if mid % 2 else 100000)
            ch["audio"] = Chunk(torch.randn(1, 1, ns, generator=g), c * 4.0) if mid != 3 else None
            chunks.append(ch)
        out.append((mid, chunks))
    return out


def _extract_worker(rank, world, port, n_files, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.test_extract_driver import FakeExtractor
    from wise_amd.extract import BatchedExtractionDriver, RankVectorIds, open_rank_stores, rank_files
    from wise_amd.feature.store.feature_store_factory import FeatureStoreType

    out_dir = Path(out_dir)
    dirs = {m: out_dir / m for m in ("video", "audio")}
    if rank == 0:
        for d in dirs.values():
            d.mkdir()
    dist.barrier()
    fx = FakeExtractor()
    ids = RankVectorIds(rank, world)
    stores = open_rank_stores(FeatureStoreType.WEBDATASET, dirs, rank, world, shard_maxcount=7)
    drv = BatchedExtractionDriver({"video": fx, "audio": fx}, stores, ids, video_batch=20, audio_batch=3)
    for mid, chunks in rank_files(_media_files(n_files), rank, world):
        for ch in chunks:
            drv.feed(mid, ch)
    drv.close()
    rows = [None] * world
    dist.all_gather_object(rows, ids.rows)      # control plane of the TEST only: the data path above has no collective
    if rank == 0:
        import pickle
        (out_dir / "rows.pkl").write_bytes(pickle.dumps(rows))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_sharded_extraction_world2(tmp_path):
    """Two ranks extract disjoint files into ONE store directory (own shard ranges, ids unique without communication);
    what a reader then sees — every (media, timestamp) with its vector — is what the reference's single loop produces
    (oracle/extract_loop_ref.py), only the ids differ (they are handed out per rank)."""
    import pickle

    from oracle.extract_loop_ref import reference_loop
    from tests.test_extract_driver import FakeExtractor, Recorder
    from wise_amd.extract import RANK_SHARD_STRIDE, rank_files
    from wise_amd.feature.store.feature_store_factory import FeatureStoreFactory, FeatureStoreType

    n_files, world = 7, 2
    assert rank_files(list(range(7)), 1, 2) == [1, 3, 5] and rank_files(list(range(7)), 0, 3) == [0, 3, 6]
    mp.spawn(_extract_worker, args=(world, _free_port(), n_files, str(tmp_path)), nprocs=world, join=True)
    rows = pickle.loads((tmp_path / "rows.pkl").read_bytes())
    all_rows = [r for per_rank in rows for r in per_rank]
    ids = [r[0] for r in all_rows]
    assert len(set(ids)) == len(ids) and min(ids) == 1
    for rank, per_rank in enumerate(rows):
        assert all((vid - 1) % world == rank for vid, *_ in per_rank)
        assert {mid for _, _, mid, _, _ in per_rank} <= set(range(rank + 1, n_files + 1, world))
    by_id = {vid: (m, mid, ts, end) for vid, m, mid, ts, end in all_rows}
    # the reference's loop over the same media, one process
    ref_dir = tmp_path / "ref"
    ref_stores = {}
    for m in ("video", "audio"):
        (ref_dir / m).mkdir(parents=True)
        st = FeatureStoreFactory.create_store(FeatureStoreType.WEBDATASET, m, str(ref_dir / m))
        st.enable_write(7, 20 * 1024 * 1024)
        ref_stores[m] = st
    rec = Recorder()
    fx = FakeExtractor()
    reference_loop(((mid, ch) for mid, chunks in _media_files(n_files) for ch in chunks), {"video": fx, "audio": fx},
                   ref_stores, rec)
    ref_by_key = {}
    for m in ("video", "audio"):
        rd = FeatureStoreFactory.load_store(m, ref_dir / m)
        rd.enable_read()
        for vid, vec in rd:
            mod, mid, ts, end = rec.rows[vid - 1]
            ref_by_key[(mod, mid, ts, end)] = vec
    assert len(ref_by_key) == len(all_rows)
    for m in ("video", "audio"):
        shards = sorted(p.name for p in (tmp_path / m).glob("*.tar"))
        assert f"{m}-000000.tar" in shards and f"{m}-{RANK_SHARD_STRIDE:06d}.tar" in shards
        rd = FeatureStoreFactory.load_store(m, tmp_path / m)       # one store, both ranks' shards
        rd.enable_read()
        n = 0
        for vid, vec in rd:
            assert by_id[vid][0] == m
            assert np.allclose(vec, ref_by_key[by_id[vid]], atol=1e-6)   # (the fake extractor is batch-dependent in the last bit)
            n += 1
        assert n == rd.feature_count == sum(1 for r in all_rows if r[1] == m)


# ---------------------------------------------------------------------------------------------------------------------
# HP-2 over several ranks THROUGH THE PLUGIN SURFACE (src/index/search_index_factory.py:4-21): SearchIndexFactory(...)
# .create_index / .load_index / .search under an initialised process group.  The rows of a rank live in a CPU stand-in
# for FlatIPIndex here (the oracle does the local scan and the merge); on the GPU box tests/test_gpu_sharded.py runs the
# same wiring with the HIP kernels and RCCL.
class _CpuFlat:
    """What FeatureSearchIndex.flat_index_factory must offer: reserve / add_with_ids / search_device / reconstruct_batch."""

    def __init__(self, d):
        self.d, self.device = int(d), torch.device("cpu")
        self._X, self._ids = [], []

    def reserve(self, n):
        self.reserved = int(n)

    def add_with_ids(self, x, ids):
        self._X.append(np.array(x, dtype=np.float32))
        self._ids.append(np.array(ids, dtype=np.int64))

    @property
    def ntotal(self):
        return sum(len(i) for i in self._ids)

    def _rows(self):
        if not self._ids:
            return np.zeros((0, self.d), np.float32), np.zeros((0,), np.int64)
        return np.concatenate(self._X), np.concatenate(self._ids)

    def search_device(self, q, k):
        from oracle import ip_topk_ref
        X, ids = self._rows()
        D, I = ip_topk_ref.ip_topk(X, q.numpy(), k, ids=ids)
        return torch.from_numpy(D), torch.from_numpy(I)

    def reconstruct_batch(self, want):
        X, ids = self._rows()
        out = np.full((len(want), self.d), np.nan, np.float32)
        for i, w in enumerate(want):
            hit = np.flatnonzero(ids == w)
            if len(hit):
                out[i] = X[hit[0]]
        return out

    @staticmethod
    def merge_lists(Ds, Is, k):
        from oracle import ip_topk_ref
        D, I = ip_topk_ref.merge_topk(Ds.numpy(), Is.numpy(), k)
        return torch.from_numpy(D), torch.from_numpy(I)


class _FakeTextTower:
    """extract_text_features: a deterministic unit vector per string (the text tower is not what this test is about)."""

    def __init__(self, d):
        self.d = d

    def extract_text_features(self, texts):
        import zlib
        out = np.stack([np.random.default_rng(zlib.crc32(t.encode())).standard_normal(self.d) for t in texts])
        return (out / np.linalg.norm(out, axis=1, keepdims=True)).astype(np.float32)


def _plugin_worker(rank, world, port, root, N, d):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wise_amd.index.feature_search_index as fsi
    from wise_amd.index.search_index_factory import SearchIndexFactory
    from wise_amd.index.sharded import ShardedFlatIPIndex, shard_range

    fsi.FeatureSearchIndex.flat_index_factory = _CpuFlat
    fsi.FeatureExtractorFactory = lambda fid: _FakeTextTower(d)
    root = Path(root)
    out = {}
    # (A) the single .faiss file a one-process create-index wrote: each rank takes rows shard_range(N, rank, world)
    si = SearchIndexFactory("video", "mlfoundations/open_clip/ViT-B-32/seeded-0",
                            {"features_dir": root / "features", "index_dir": root / "index_single"})
    assert si.load_index("IndexFlatIP") is True
    assert isinstance(si.index, ShardedFlatIPIndex)
    lo, hi = shard_range(N, rank, world)
    assert si.index.local.ntotal == hi - lo and si.index.local.reserved == hi - lo and si.index.ntotal == N
    out["A_dist"], out["A_ids"] = si.search("video", "dog", topk=7)
    Q = np.random.default_rng(6).standard_normal((3, d)).astype(np.float32)
    out["A_D"], out["A_I"] = si.index.search(Q, 10)                       # the REST call shape (api/routes.py:1407)
    out["A_rec"] = si.index.reconstruct_batch(np.array([1, N, 500, N + 5], dtype=np.int64))
    # (B) the sharded build: every rank reads its own tar files only and writes its own part; load takes the part
    si2 = SearchIndexFactory("video", "mlfoundations/open_clip/ViT-B-32/seeded-0",
                             {"features_dir": root / "features", "index_dir": root / "index_parts"})
    si2.create_index("IndexFlatIP")
    part = si2.get_index_part_filename("IndexFlatIP", rank, world)
    assert part.exists() and not si2.get_index_filename("IndexFlatIP").exists()
    dist.barrier()
    assert si2.load_index("IndexFlatIP") is True
    out["B_local"] = np.array([si2.index.local.ntotal])
    assert si2.index.ntotal == N
    out["B_dist"], out["B_ids"] = si2.search("video", "dog", topk=7)
    out["B_D"], out["B_I"] = si2.index.search(Q, 10)
    np.savez(root / f"plugin_rank{rank}.npz", **out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_index_through_the_plugin_surface_world2(tmp_path):
    from oracle import ip_topk_ref
    from wise_amd.feature.store.feature_store_factory import FeatureStoreFactory, FeatureStoreType
    from wise_amd.index.search_index_factory import SearchIndexFactory

    N, d, world = 1001, 32, 2
    X = np.random.default_rng(5).standard_normal((N, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    fdir = tmp_path / "features"
    fdir.mkdir()
    st = FeatureStoreFactory.create_store(FeatureStoreType.WEBDATASET, "video", str(fdir))
    st.enable_write(100, 20 * 1024 * 1024)                               # 11 tar files: ranks get 6 and 5 of them
    for i in range(N):
        st.add(i + 1, X[i:i + 1])
    st.close()
    single = SearchIndexFactory("video", "mlfoundations/open_clip/ViT-B-32/seeded-0",
                                {"features_dir": fdir, "index_dir": tmp_path / "index_single"})
    single.create_index("IndexFlatIP")                                   # no process group here: the one-file build
    mp.spawn(_plugin_worker, args=(world, _free_port(), str(tmp_path), N, d), nprocs=world, join=True)

    ids = np.arange(N, dtype=np.int64) + 1
    q = _FakeTextTower(d).extract_text_features(["This is a photo of a dog"])
    D1, I1 = ip_topk_ref.ip_topk(X, q, 7, ids=ids)
    Q = np.random.default_rng(6).standard_normal((3, d)).astype(np.float32)
    D3, I3 = ip_topk_ref.ip_topk(X, Q, 10, ids=ids)
    locals_ = []
    for r in range(world):
        g = np.load(tmp_path / f"plugin_rank{r}.npz")
        for tag in "AB":
            assert np.array_equal(g[f"{tag}_ids"], I1[0]) and np.array_equal(g[f"{tag}_dist"], D1[0]), (r, tag)
            assert np.array_equal(g[f"{tag}_I"], I3) and np.array_equal(g[f"{tag}_D"], D3), (r, tag)
        rec = g["A_rec"]
        assert np.array_equal(rec[0], X[0]) and np.array_equal(rec[1], X[N - 1]) and np.array_equal(rec[2], X[499])
        assert np.isnan(rec[3]).all()                                    # an id no rank holds
        locals_.append(int(g["B_local"][0]))
    assert locals_ == [501, 500]                                         # tar files 0,2,..,10 (5 x 100 + the 1-row tail) and 1,3,..,9
