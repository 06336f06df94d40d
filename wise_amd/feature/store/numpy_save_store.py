"""NumpySaveStore — reads and writes the reference's npz shard format
(src/feature/store/numpy_save_store.py:9-116): `{dir}/{name}-%06d.npz` holding
`feature_id int32[n]` and `features float32[n,D]`; one [1,D] row per add(); shard roll-over at
shard_maxcount.  Unlike the reference it also offers iter_batch(), so an npz store can be indexed
(the reference's create_index calls iter_batch on whatever load_store returns: SURVEY App. B.9).
"""
import glob
import random
from pathlib import Path

import numpy as np

from .feature_store import FeatureStore


class NumpySaveStore(FeatureStore):
    def __init__(self, store_name, store_data_dir):
        self.store_name = store_name
        self.store_data_dir = Path(store_data_dir)

    def enable_write(self, shard_maxcount, shard_maxsize, verbose=0, first_shard=0):
        """first_shard (not in the reference; default = its behaviour): number of the first shard file this writer
        creates, so that several processes can write disjoint shard ranges of one store."""
        self.shard_maxcount = shard_maxcount
        self.shard_maxsize = shard_maxsize
        self.verbose = verbose
        self.current_shard_index = -1
        self.first_shard_index = int(first_shard)

    def enable_read(self, shard_shuffle=False, shuffle_values=False, shuffle_bufsize=10000):
        self.shard_shuffle = shard_shuffle
        self.shuffle_values = shuffle_values
        self.shuffle_bufsize = shuffle_bufsize
        pattern = self.store_data_dir / (self.store_name + '-*.npz')
        self.npz_filename_list = list(glob.iglob(pathname=pattern.as_posix(), recursive=False))
        if self.shard_shuffle:
            random.shuffle(self.npz_filename_list)
        else:
            self.npz_filename_list.sort()
        self.feature_count = 0
        self.feature_dim = -1
        for fn in self.npz_filename_list:
            payload = np.load(fn)
            self.feature_count += payload['feature_id'].shape[0]
            if self.feature_dim < 0:
                first = payload['features'][0]
                if first.ndim == 1:
                    self.feature_dim = first.shape[0]
                elif first.ndim == 2:
                    self.feature_dim = first.shape[1]
                else:
                    raise ValueError(f'unrecognized feature shape {first.shape}')

    def add(self, id, features):
        if self.current_shard_index == -1:
            self.feature_dim = features.shape[1]
            self.shard_features = np.ndarray((self.shard_maxcount, self.feature_dim), dtype=np.float32)
            self.shard_feature_id = np.ndarray((self.shard_maxcount), dtype=np.int32)
            self.shard_feature_index = 0
            self.current_shard_index = getattr(self, 'first_shard_index', 0)
        if self.feature_dim != features.shape[1]:
            raise ValueError(f'feature dimension cannot change and must be {self.feature_dim}')
        if features.shape[0] != 1:
            raise ValueError(f'cannot add {features.shape[0]} features, only one feature can be added at a time')
        if self.shard_feature_index == self.shard_maxcount:
            self.save_current_shard()
            self.add(id, features)
        else:
            self.shard_features[self.shard_feature_index] = features
            self.shard_feature_id[self.shard_feature_index] = id
            self.shard_feature_index += 1

    def save_current_shard(self):
        shard_id = '%s-%06d' % (self.store_name, self.current_shard_index)
        np.savez(self.store_data_dir / shard_id, feature_id=self.shard_feature_id, features=self.shard_features)
        if self.verbose:
            print(f'saved {self.shard_feature_index} features to shard {self.store_data_dir / shard_id}')
        self.current_shard_index += 1
        self.shard_feature_index = 0

    def __iter__(self):
        for fn in self.npz_filename_list:
            payload = np.load(fn)
            ids, feats = payload['feature_id'], payload['features']
            n = ids.shape[0]
            order = random.sample(range(0, n), n) if self.shuffle_values else range(0, n)
            for i in order:
                yield ids[i], np.take(feats, [i], 0)  # (1,D), not (D,)

    def iter_batch(self, batch_size=512):
        """(ids list[<=batch], vectors [<=batch, D]) like WebdatasetStore.iter_batch."""
        for fn in self.npz_filename_list:
            payload = np.load(fn)
            ids, feats = payload['feature_id'], payload['features'].reshape(-1, self.feature_dim)
            for s in range(0, ids.shape[0], batch_size):
                yield [int(v) for v in ids[s:s + batch_size]], np.ascontiguousarray(feats[s:s + batch_size],
                                                                                   dtype=np.float32)

    def close(self):
        if getattr(self, 'shard_feature_index', 0) != 0:
            n = self.shard_feature_index
            self.shard_feature_id = self.shard_feature_id[:n].copy()
            self.shard_features = self.shard_features[:n].copy()
            self.save_current_shard()
            self.shard_feature_index = 0

    def __del__(self):
        try:
            if getattr(self, 'shard_feature_index', 0) != 0:
                self.close()
        except Exception:
            pass
