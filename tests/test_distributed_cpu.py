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
