"""FeatureSearchIndex — the reference's plugin boundary for vector search (src/index/feature_search_index.py:13-114)
over an index resident in HBM and searched by the HIP scan+top-k kernels.

`__init__`, `get_index_filename`, `is_index_loaded` and `search` are restated VERBATIM from
src/index/feature_search_index.py:14-31, :87-88 and :100-114 — they ARE the boundary, and their quirks are part of
the contract a drop-in must keep (the prompt rules, the missing `f` prefix in the
`query_type` error message, the 1-D return of the first query only).  `create_index` and `load_index` are new:
they build / read the same `.faiss` files without faiss (wise_amd/index/faiss_io.py) into FlatIPIndex / IVFFlatIPIndex.

Same constructor contract (asserts on 'features_dir'/'index_dir'), prompts, file naming
(`{index_dir}/{media_type}-{index_type}.faiss`), skip-if-exists create, `load_index` that also
builds the FeatureExtractor, and the prompt quirks of `search`.  Both index types the
reference offers are built: `IndexFlatIP` (exhaustive, the hot path) and `IndexIVFFlat` (approximate; cell count
and training-sample size chosen as at feature_search_index.py:55-59, k-means and list scan on the GPU).

**One process per GPU** (the reference has no distributed path).  When `torch.distributed` is initialised
with more than one rank (or WISE_SHARDED_INDEX=1), the same two calls shard the flat index by rows:
  * `create_index('IndexFlatIP')`: rank r reads ONLY feature-store shard files r, r + W, ... and writes its own part,
    `{media_type}-IndexFlatIP.faiss.part-RRR-of-WWW` (same file layout, ids are the store's global ids) — no collective;
  * `load_index('IndexFlatIP')`: rank r loads its part file if the parts of this world size exist, otherwise rows
    `shard_range(N, r, W)` of the single `.faiss` file — memory-mapped, so a rank touches only its own rows' pages —
    and `self.index` is a `ShardedFlatIPIndex`: `search` / `reconstruct_batch` are then collective (every rank calls
    them with the same arguments and gets the global answer: one all-gather of per-shard top-k + `wise_topk_merge`).
IndexIVFFlat stays a single-GPU index (every rank loads the whole file; rank 0 alone builds it).
"""
import os
from pathlib import Path

import numpy as np

from ..feature.feature_extractor_factory import FeatureExtractorFactory
from ..feature.store.feature_store_factory import FeatureStoreFactory
from . import faiss_io
from .flat_ip import FlatIPIndex
from .ivf_flat import IVFFlatIPIndex, reference_nlist
from .search_index import SearchIndex
from .sharded import ShardedFlatIPIndex, shard_range


def _dist_rank_world():
    """(rank, world, sharded?) of the default process group; (0, 1, False) outside torch.distributed."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
        return rank, world, world > 1 or os.environ.get('WISE_SHARDED_INDEX') == '1'
    return 0, 1, False


class FeatureSearchIndex(SearchIndex):
    # the class that holds a rank's rows in HBM; CPU tests of the multi-rank wiring put a stand-in here
    flat_index_factory = FlatIPIndex

    def __init__(self, media_type, asset_id, asset):
        self.media_type = media_type
        self.feature_extractor_id = asset_id

        assert 'features_dir' in asset, "features_dir missing in assets"
        self.features_dir = Path(asset['features_dir'])

        assert 'index_dir' in asset, "index_dir missing in assets"
        self.index_dir = Path(asset['index_dir'])

        self.prompt = {
            'image': 'This is a photo of a ',
            'video': 'This is a photo of a ',
            'audio': 'this is the sound of '
        }

    def get_index_filename(self, index_type):
        return self.index_dir / (self.media_type + '-' + index_type + '.faiss')

    def get_index_part_filename(self, index_type, rank, world):
        """rank's part of a row-sharded flat index built by `world` processes (not in the reference)."""
        fn = self.get_index_filename(index_type)
        return fn.with_name(fn.name + '.part-%03d-of-%03d' % (rank, world))

    def create_index(self, index_type, overwrite=False):
        self.index_dir.mkdir(parents=True, exist_ok=True)
        index_fn = self.get_index_filename(index_type)
        rank, world, sharded = _dist_rank_world()
        if sharded and index_type == 'IndexFlatIP':
            index_fn = self.get_index_part_filename(index_type, rank, world)
        if index_fn.exists() and overwrite is False:
            print(f'{index_type} for {self.media_type} already exists')
            return
        if index_type not in ('IndexFlatIP', 'IndexIVFFlat'):
            raise NotImplementedError(f'{index_type}: IndexFlatIP and IndexIVFFlat are the index types WISE builds')
        self.index_type = index_type
        if sharded and index_type == 'IndexIVFFlat' and rank != 0:
            return                                  # k-means needs every row: one rank builds the one file

        feature_store = FeatureStoreFactory.load_store(self.media_type, self.features_dir)
        if sharded and index_type == 'IndexFlatIP':
            feature_store.enable_read(shard_shuffle=False, shard_slice=(rank, world))   # this rank's shard files only
        else:
            feature_store.enable_read(shard_shuffle=False)
        feature_count = feature_store.feature_count
        feature_dim = feature_store.feature_dim

        # the on-disk index is assembled on the host (I/O-bound: tar + unpickle per vector), 512 at a time
        X = np.empty((feature_count, feature_dim), dtype=np.float32)
        ids = np.empty((feature_count,), dtype=np.int64)
        n = 0
        print('Adding feature vectors to index')
        for feature_ids_batch, feature_vectors_batch in feature_store.iter_batch():
            m = len(feature_ids_batch)
            X[n:n + m] = feature_vectors_batch
            ids[n:n + m] = feature_ids_batch
            n += m
        if index_type == 'IndexIVFFlat':
            cell_count = reference_nlist(n)
            train_count = min(n, 100 * cell_count)
            # the reference trains on the first train_count vectors of a shard-shuffled pass (:62-69); a seeded
            # sample of the same size stands in for it
            sample = np.random.default_rng(1234).permutation(n)[:train_count]
            sample.sort()
            print(f'  training {index_type} index with {train_count} features with {cell_count} clusters ...')
            ivf = IVFFlatIPIndex(feature_dim, cell_count)
            ivf.train(X[sample])
            for s0 in range(0, n, 1 << 20):
                ivf.add_with_ids(X[s0:s0 + (1 << 20)], ids[s0:s0 + (1 << 20)])
            c, Xs, ids_s, off = ivf.lists_host()
            faiss_io.write_ivf_flat_ip(index_fn, c, Xs, ids_s, off, nprobe=ivf.nprobe)
        else:
            faiss_io.write_idmap_flat_ip(index_fn, X[:n], ids[:n])
        print(f'  saved index to {index_fn}')

    def is_index_loaded(self):
        return hasattr(self, 'index')

    def load_index(self, index_type):
        index_fn = self.get_index_filename(index_type)
        rank, world, sharded = _dist_rank_world()
        part_fn = self.get_index_part_filename(index_type, rank, world)
        if not index_fn.exists() and not (sharded and part_fn.exists()):
            print(f'  index {index_fn} does not exist')
            print(f'  use create-index.py script to create an index')
        # like the reference (App. B.3) a missing file raises from the reader, it does not return False
        if index_fn.exists() and faiss_io.index_fourcc(index_fn) == 'IwFl':
            import torch
            f = faiss_io.read_ivf_flat_ip(index_fn)
            index = IVFFlatIPIndex(f["centroids"].shape[1], f["centroids"].shape[0])
            index.set_centroids(f["centroids"])
            index.adopt_lists(torch.from_numpy(f["X"]), torch.from_numpy(f["ids"]), torch.from_numpy(f["list_off"]))
            index.nprobe = f["nprobe"]
        else:
            if sharded and part_fn.exists():         # built by this many ranks: a rank's part is its shard
                X, ids = faiss_io.read_idmap_flat_ip(part_fn)
                lo, hi = 0, X.shape[0]
            else:
                X, ids = faiss_io.read_idmap_flat_ip(index_fn)      # rows memory-mapped: only [lo, hi) is ever touched
                lo, hi = shard_range(X.shape[0], rank, world) if sharded else (0, X.shape[0])
            index = self.flat_index_factory(X.shape[1])
            index.reserve(hi - lo)                   # one [n,d] device tensor, filled slice by slice
            for s in range(lo, hi, 1 << 20):         # stream the memory-mapped rows into HBM
                e = min(s + (1 << 20), hi)
                index.add_with_ids(np.ascontiguousarray(X[s:e]), ids[s:e])
            if sharded:
                index = ShardedFlatIPIndex(index, merge=getattr(index, 'merge_lists', None),
                                           always_exchange=os.environ.get('WISE_SHARDED_INDEX') == '1')
        self.index = index
        self.feature_extractor = FeatureExtractorFactory(self.feature_extractor_id)
        return True

    def search(self, media_type, query, topk=5, query_type='text'):
        if query_type != 'text':
            raise ValueError('query_type={query_type} not implemented')

        if media_type == 'audio':
            if isinstance(query, str):
                media_query_text = [query]
            else:
                media_query_text = [(self.prompt[media_type] + x) for x in query]
        else:
            media_query_text = [(self.prompt[media_type] + query)]

        query_features = self.feature_extractor.extract_text_features(media_query_text)
        dist, ids = self.index.search(query_features, topk)
        return dist[0], ids[0]

    def search_batch(self, media_type, queries, topk=5, query_type='text'):
        """Not in the reference: what `search` returns for every string of `queries`, from ONE text-tower batch and ONE
        batched index search per 256 of them (wise_amd/search/batch_queries.py; the --queries-from loop of search.py:894-950
        calls `search` row by row)."""
        if query_type != 'text':
            raise ValueError('query_type={query_type} not implemented')
        from ..search.batch_queries import batched_text_search
        return batched_text_search(self, media_type, queries, topk)
