"""NumpySaveStore — reader and writer of the reference's npz shard format.

The CONTRACT (src/feature/store/numpy_save_store.py:9-116, and what other WISE code relies on): shard files
`{dir}/{name}-%06d.npz` with two arrays, `feature_id int32[n]` and `features float32[n, D]`; the method names
`enable_write / enable_read / add / save_current_shard / close`, one `[1, D]` row per `add()`, at most `shard_maxcount`
rows per shard, iteration yielding `(feature_id, vector[1, D])` in file-name order.  The format is pinned against shards
written by the reference class itself (tests/golden/store_ref, made by oracle/make_golden_store.py): this reader must
return what the reference's reader returned, this writer must produce array-identical files.

The implementation is this repo's own: rows are appended to per-shard Python lists and a shard is stacked and written
when it is full or the store is closed (the reference keeps a preallocated `[shard_maxcount, D]` buffer and trims the last
one).  Beyond the reference: `iter_batch()` (the reference's `create_index` calls it on whatever store it loads, which its
own npz store lacks: SURVEY App. B.9) and `first_shard` (several ranks writing disjoint shard ranges of one store).
"""
from __future__ import annotations

import random
from pathlib import Path

import numpy as np

from .feature_store import FeatureStore

_ID_DTYPE = np.int32          # the reference's shard_feature_id buffer
_VEC_DTYPE = np.float32       # ... and its shard_features buffer


class NumpySaveStore(FeatureStore):
    def __init__(self, store_name, store_data_dir):
        self.store_name = store_name
        self.store_data_dir = Path(store_data_dir)
        self._ids: list = []
        self._rows: list = []

    # ------------------------------------------------------------------ writing
    def enable_write(self, shard_maxcount, shard_maxsize, verbose=0, first_shard=0):
        """`shard_maxsize` is accepted and, as in the reference, not used by this store."""
        if int(shard_maxcount) < 1:
            raise ValueError(f'shard_maxcount must be positive, not {shard_maxcount}')
        self.shard_maxcount = int(shard_maxcount)
        self.shard_maxsize = shard_maxsize
        self.verbose = verbose
        self.current_shard_index = int(first_shard)
        self.feature_dim = None
        self._ids, self._rows = [], []

    @property
    def shard_feature_index(self):
        """rows waiting for the next shard file (the reference exposes its buffer cursor under this name)"""
        return len(self._ids)

    def add(self, id, features):
        features = np.asarray(features)
        if features.ndim != 2:
            raise ValueError(f'features must be [1, D], not {features.shape}')
        if features.shape[0] != 1:
            raise ValueError(f'cannot add {features.shape[0]} features, only one feature can be added at a time')
        if self.feature_dim is None:
            self.feature_dim = features.shape[1]
        elif features.shape[1] != self.feature_dim:
            raise ValueError(f'feature dimension cannot change and must be {self.feature_dim}')
        self._ids.append(id)
        self._rows.append(features[0].astype(_VEC_DTYPE))
        if len(self._ids) == self.shard_maxcount:
            self.save_current_shard()

    def save_current_shard(self):
        """Write the pending rows as the next shard file (nothing pending: nothing written)."""
        if not self._ids:
            return
        target = self.store_data_dir / ('%s-%06d' % (self.store_name, self.current_shard_index))
        np.savez(target, feature_id=np.asarray(self._ids, dtype=_ID_DTYPE), features=np.stack(self._rows))
        if self.verbose:
            print(f'saved {len(self._ids)} features to shard {target}')
        self.current_shard_index += 1
        self._ids, self._rows = [], []

    def close(self):
        self.save_current_shard()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reading
    def enable_read(self, shard_shuffle=False, shuffle_values=False, shuffle_bufsize=10000, shard_slice=None):
        """shard_slice (not in the reference): (rank, world) -> this reader sees shard files rank, rank + world, ... of the
        name-sorted list only — one process per GPU builds its own part of an index (index build)."""
        self.shard_shuffle = shard_shuffle
        self.shuffle_values = shuffle_values
        self.shuffle_bufsize = shuffle_bufsize
        shards = sorted(p.as_posix() for p in self.store_data_dir.glob(self.store_name + '-*.npz'))
        if shard_slice is not None:
            shards = shards[int(shard_slice[0])::int(shard_slice[1])]
        if shard_shuffle:
            random.shuffle(shards)
        self.npz_filename_list = shards
        self.feature_count = 0
        self.feature_dim = -1
        for fn in shards:
            with np.load(fn) as payload:
                self.feature_count += int(payload['feature_id'].shape[0])
                if self.feature_dim < 0:
                    shape = payload['features'].shape      # [n, D], or [n, 1, D] from writers that kept the row axis
                    if len(shape) not in (2, 3):
                        raise ValueError(f'unrecognized feature shape {shape[1:]}')
                    self.feature_dim = int(shape[-1])

    def _shard_arrays(self, fn):
        # A shard whose `features` kept a row axis ([n, 1, D]: not something the reference's writer can produce, its buffer is
        # [shard_maxcount, D], but other tools do) is read as [n, D] ON PURPOSE: iteration yields (1, D) rows for every
        # shard, where the reference's np.take(features, [i], 0) would hand such a file's rows on as (1, 1, D) and break
        # its own consumers (pinned by tests/test_host_api.py::test_numpy_save_store_row_axis_shards_are_normalised).
        with np.load(fn) as payload:
            return payload['feature_id'], payload['features'].reshape(-1, self.feature_dim)

    def __iter__(self):
        for fn in self.npz_filename_list:
            ids, feats = self._shard_arrays(fn)
            order = list(range(ids.shape[0]))
            if self.shuffle_values:
                random.shuffle(order)
            for i in order:
                yield ids[i], feats[i:i + 1]          # (1, D), as the reference yields

    def iter_batch(self, batch_size=512):
        """(ids list[<=batch], vectors [<=batch, D]) like WebdatasetStore.iter_batch."""
        for fn in self.npz_filename_list:
            ids, feats = self._shard_arrays(fn)
            for s in range(0, ids.shape[0], batch_size):
                yield ([int(v) for v in ids[s:s + batch_size]],
                       np.ascontiguousarray(feats[s:s + batch_size], dtype=np.float32))
