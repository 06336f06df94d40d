"""WebdatasetStore — reads and writes the reference's tar shard format
(src/feature/store/webdataset_store.py:11-148) WITHOUT the webdataset package (not installed):

  {dir}/{name}-%06d.tar ; one member per vector named '%010d.features.pyd' whose bytes are
  pickle.dumps(np.ndarray[1,D] float32)  (webdataset encodes the `pyd` extension with pickle; the
  reference reads it back with np.load(BytesIO, allow_pickle=True), :72,113,129).

Shard roll-over follows webdataset.ShardWriter.write: before a record is written, a new shard starts when the current one
already holds `maxcount` records or `maxsize` or more bytes of MEMBER DATA (TarWriter.write returns the sum of the members'
sizes, headers and padding not counted).  One add() = one member, whatever the array's first dimension is: the reference's
test writes [3,4] arrays under ids 0 and 3 and reads them back whole (src/feature/store/test_feature_store.py:48-102).
"""
import glob
import io
import os
import pickle
import random
import tarfile
import time

import numpy as np

from .feature_store import FeatureStore


class WebdatasetStore(FeatureStore):
    def __init__(self, store_name, store_data_dir):
        self.store_name = store_name
        self.store_data_dir = store_data_dir
        self.EXTENSION = 'tar'
        self.store_data_filename = os.path.join(self.store_data_dir, self.store_name + '-%06d.' + self.EXTENSION)
        self.feature_count = -1
        self.feature_dim = -1
        self._tar = None

    # ---- write ----
    def enable_write(self, shard_maxcount, shard_maxsize, verbose=0, first_shard=0):
        """first_shard (not in the reference, default = its behaviour): number of the first tar this writer creates —
        one process per GPU writes its own shard range of the same store (wise_amd/extract.py:open_rank_stores)."""
        self.shard_maxcount = shard_maxcount
        self.shard_maxsize = shard_maxsize
        self.verbose = verbose
        self._shard = int(first_shard)
        self._count = 0
        self._size = 0
        self._tar = None

    def _next_shard(self):
        if self._tar is not None:
            self._tar.close()
        self._tar = tarfile.open(self.store_data_filename % self._shard, 'w')
        self._shard += 1
        self._count = 0
        self._size = 0

    def add(self, id, features):
        if not hasattr(self, 'shard_maxcount'):
            raise ValueError('enable_write() must be activated before invoking add() method')
        if self._tar is None or self._count >= self.shard_maxcount or self._size >= self.shard_maxsize:
            self._next_shard()
        data = pickle.dumps(features)
        ti = tarfile.TarInfo(('%010d' % id) + '.features.pyd')
        ti.size = len(data)
        ti.mtime = time.time()
        ti.mode = 0o444
        ti.uname = 'bigdata'
        ti.gname = 'bigdata'
        self._tar.addfile(ti, io.BytesIO(data))
        self._count += 1
        self._size += len(data)

    def close(self):
        if self._tar is not None:
            self._tar.close()
            self._tar = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- read ----
    def _shards(self):
        prefix = os.path.join(self.store_data_dir, self.store_name + '-')
        files = []
        for fn in glob.iglob(pathname=prefix + '*.tar', recursive=False):
            files.append((int(fn.split(prefix)[1].split('.tar')[0]), fn))
        files.sort()
        return [fn for _, fn in files]

    def enable_read(self, shard_shuffle=False, shuffle_values=False, shuffle_bufsize=10000, shard_slice=None):
        """shard_slice (not in the reference): (rank, world) -> this reader sees tar files rank, rank + world, ... of the
        number-sorted list only (feature_count counts those) — one process per GPU builds its own part of an index."""
        self.shard_shuffle = shard_shuffle
        self.shuffle_values = shuffle_values
        self.shuffle_bufsize = shuffle_bufsize
        self._files = self._shards()
        if shard_slice is not None:
            self._files = self._files[int(shard_slice[0])::int(shard_slice[1])]
        for _, vec in self._records(self._shards()[:1]):      # the store's dimension, also for a rank whose slice is empty
            self.feature_dim = vec.shape[1]
            break
        # the reference memoises member counts by file size; count exactly instead
        self.feature_count = 0
        for fn in self._files:
            with tarfile.open(fn) as f:
                self.feature_count += sum(1 for m in f if m.isreg())

    @staticmethod
    def _records(files):
        for fn in files:
            with tarfile.open(fn) as f:
                for m in f:
                    if not m.isreg() or not m.name.endswith('.features.pyd'):
                        continue
                    key = m.name[: -len('.features.pyd')]
                    vec = np.load(io.BytesIO(f.extractfile(m).read()), allow_pickle=True)
                    yield int(key), vec

    def _ordered(self):
        files = list(self._files)
        if self.shard_shuffle:
            random.shuffle(files)
        it = self._records(files)
        if not self.shuffle_values:
            yield from it
            return
        buf = []
        for rec in it:  # webdataset-style bounded shuffle buffer
            buf.append(rec)
            if len(buf) >= self.shuffle_bufsize:
                yield buf.pop(random.randrange(len(buf)))
        random.shuffle(buf)
        yield from buf

    def __iter__(self):
        yield from self._ordered()

    def iter_batch(self, batch_size=512):
        ids, vecs = [], []
        for fid, vec in self._ordered():
            ids.append(fid)
            vecs.append(vec.squeeze(axis=0))
            if len(ids) == batch_size:
                yield ids, np.stack(vecs)
                ids, vecs = [], []
        if ids:
            yield ids, np.stack(vecs)
