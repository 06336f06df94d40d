"""Worker of tests/test_gpu_sharded.py (not a test module): ONE rank on the `nccl` backend (= RCCL) on the GPU box.

Initialises the process group before any other GPU call, then drives the sharded flat index through the plugin surface
(SearchIndexFactory -> load_index -> ShardedFlatIPIndex with always_exchange: WISE_SHARDED_INDEX=1), so that
all_gather_into_tensor (RCCL) and wise_topk_merge really run, and compares with the unsharded FlatIPIndex and the oracle.
Prints one JSON line."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main(tmp):
    import numpy as np
    import torch
    import torch.distributed as dist

    os.environ["WISE_SHARDED_INDEX"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from oracle import ip_topk_ref
    from wise_amd.feature.store.feature_store_factory import FeatureStoreFactory, FeatureStoreType
    from wise_amd.index import faiss_io
    from wise_amd.index.flat_ip import FlatIPIndex
    from wise_amd.index.search_index_factory import SearchIndexFactory
    from wise_amd.index.sharded import ShardedFlatIPIndex

    tmp = Path(tmp)
    fid = "mlfoundations/open_clip/ViT-B-32/seeded-0"
    res = {}
    # (A) a 300k x 512 single-file index (above 2^18 rows: the two-stage search is what runs locally)
    N, d = 300_000, 512
    X = np.random.default_rng(2).standard_normal((N, d), dtype=np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    ids = np.arange(N, dtype=np.int64) * 3 + 7
    (tmp / "indexA").mkdir()
    (tmp / "features").mkdir()
    faiss_io.write_idmap_flat_ip(tmp / "indexA" / "video-IndexFlatIP.faiss", X, ids)
    si = SearchIndexFactory("video", fid, {"features_dir": tmp / "features", "index_dir": tmp / "indexA"})
    assert si.load_index("IndexFlatIP") is True
    assert isinstance(si.index, ShardedFlatIPIndex) and si.index.always_exchange and si.index.world == 1
    assert dist.get_backend() == "nccl"
    res["ntotal"] = si.index.ntotal                                    # all_reduce over RCCL
    plain = FlatIPIndex(d)
    plain.add_with_ids(X, ids)
    Q = np.random.default_rng(3).standard_normal((4, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    ok = True
    for nq, k in [(1, 10), (1, 100), (4, 10), (3, 128)]:
        D, I = si.index.search(Q[:nq], k)                              # all_gather_into_tensor + wise_topk_merge
        Dp, Ip = plain.search(Q[:nq], k)
        Dr, Ir = ip_topk_ref.ip_topk(X, Q[:nq], k, ids=ids)
        same = bool(np.array_equal(I, Ip) and np.array_equal(D, Dp) and np.array_equal(I, Ir)
                    and np.allclose(D, Dr, atol=2e-5))
        res[f"search_nq{nq}_k{k}"] = same
        ok &= same
    res["exchange_bytes"] = si.index.last_exchange_bytes
    rec = si.index.reconstruct_batch(np.array([7, 7 + 3 * 12345, 8], dtype=np.int64))
    res["reconstruct"] = bool(np.array_equal(rec[0], X[0]) and np.array_equal(rec[1], X[12345]) and np.isnan(rec[2]).all())
    dist_, ids_ = si.search("video", "dog", topk=5)                    # text tower -> collective search, first query only
    q = si.feature_extractor.extract_text_features(["This is a photo of a dog"])
    Dr, Ir = ip_topk_ref.ip_topk(X, q, 5, ids=ids)
    res["plugin_search"] = bool(np.array_equal(ids_, Ir[0]) and np.allclose(dist_, Dr[0], atol=2e-5))
    # (B) the sharded build: store -> part file -> load
    Xs = X[:5000]
    st = FeatureStoreFactory.create_store(FeatureStoreType.NUMPY, "audio", str(tmp / "features"))
    st.enable_write(1000, 0)
    for i in range(5000):
        st.add(i + 1, Xs[i:i + 1])
    st.close()
    si2 = SearchIndexFactory("audio", fid, {"features_dir": tmp / "features", "index_dir": tmp / "indexB"})
    si2.create_index("IndexFlatIP")
    res["part_file"] = si2.get_index_part_filename("IndexFlatIP", 0, 1).exists()
    assert si2.load_index("IndexFlatIP") is True
    D, I = si2.index.search(Q[:2], 10)
    Dr, Ir = ip_topk_ref.ip_topk(Xs, Q[:2], 10, ids=np.arange(5000, dtype=np.int64) + 1)
    res["part_search"] = bool(np.array_equal(I, Ir) and np.allclose(D, Dr, atol=2e-5))
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    res["ok"] = bool(ok and res["reconstruct"] and res["plugin_search"] and res["part_file"] and res["part_search"]
                     and res["ntotal"] == N)
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1])
