"""Batched HP-1 driver: the counterpart of the reference's per-chunk loop at extract-features.py:324-375.

The reference embeds one 4-second chunk per iteration (8 frames, or 1 audio segment), creating one
`vectors` row per embedding and adding it to the modality's FeatureStore right away.  On an MI355X a
forward wants hundreds of frames, so this driver re-batches: rows are allocated ids IMMEDIATELY and in
the reference's order (so ids, timestamps and per-store order are identical), the tensors are queued, and
the extractor runs once per `video_batch` frames / `audio_batch` segments.  Embeddings do not depend on
the batch a frame sits in (tests/test_gpu_vit.py::test_vit_b32_batch256_consistency), so the stores come
out identical to the reference loop's.

When an extractor offers `extract_*_features_async` (the HIP extractors do) a batch is only SUBMITTED when it is
full; its vectors go to the store when the next batch has been submitted (or at flush), so the GPU works on one
batch while the host writes the previous one.  Store order is unchanged: results are drained first in, first out.

`create_vector(modality, media_id, timestamp, end_timestamp) -> int` stands where VectorRepo.create
stands (extract-features.py:348-357,364-372); SQLite itself is outside this build.

Several GPUs (one process per GPU): extraction shards by FILES, the way the reference's DataLoader
shards them over its decode workers — `islice(files, worker_id, None, n_workers)` (src/dataloader/dataset.py:334) —
with no collective on the data path: `rank_files` gives a rank its files, `RankVectorIds` hands out vector ids that
are unique across ranks without talking to anyone (the n-th vector of rank r gets id n * world + r + 1), and
`open_rank_stores` opens each modality's store for writing in the rank's own shard range
(`{media_type}-{rank * RANK_SHARD_STRIDE + n:06d}.tar`), so all ranks write one store directory and a reader
(FeatureStoreFactory.load_store, create_index) sees one store.
"""
from __future__ import annotations

import itertools
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch


class BatchedExtractionDriver:
    def __init__(self, feature_extractors: Dict[str, object], feature_stores: Dict[str, object],
                 create_vector: Callable[[str, object, float, Optional[float]], int], *, video_batch: int = 256,
                 audio_batch: int = 128, video_frame_rate: float = 2.0, audio_segment_length: float = 4.0,
                 audio_samples_per_chunk: int = 192000):
        self.extractors = feature_extractors          # media_type -> extractor; dict order = the reference's loop order
        self.stores = feature_stores
        self.create_vector = create_vector
        self.batch = {"video": video_batch, "image": video_batch, "audio": audio_batch}
        self.video_frame_rate = video_frame_rate
        self.audio_segment_length = audio_segment_length
        self.audio_samples_per_chunk = audio_samples_per_chunk
        self._queue: Dict[str, List[Tuple[torch.Tensor, List[int]]]] = {m: [] for m in feature_extractors}
        self._rows: Dict[str, int] = {m: 0 for m in feature_extractors}
        self._pending: Dict[str, list] = {m: [] for m in feature_extractors}   # submitted, not yet written
        self.vectors_written = 0

    def feed(self, mid, chunks: Dict[str, object]) -> None:
        """One iteration of the reference's `for idx, (mid, chunks) in enumerate(av_data_loader)`."""
        for media_type in self.extractors:
            chunk = chunks.get(media_type)
            if chunk is None:
                continue
            tensor, pts = chunk.tensor, chunk.pts
            if media_type in ("image", "video"):
                ids = [self.create_vector(media_type, mid, pts + i * (1 / self.video_frame_rate), None)
                       for i in range(tensor.shape[0])]                      # extract-features.py:347-357
            elif media_type == "audio":
                if tensor.shape[2] < self.audio_samples_per_chunk:           # :336-338 malformed segments dropped
                    continue
                ids = [self.create_vector(media_type, mid, pts, pts + self.audio_segment_length)]  # :362-372
            else:
                raise ValueError(f"Unknown media_type {media_type}")
            self._queue[media_type].append((tensor, ids))
            self._rows[media_type] += tensor.shape[0]
            if self._rows[media_type] >= self.batch[media_type]:
                self._run(media_type)

    def _run(self, media_type: str) -> None:
        items = self._queue[media_type]
        if not items:
            return
        fx = self.extractors[media_type]
        if media_type == "audio":
            # segments may differ in length; the model sees each as the reference would (one forward per length)
            by_len: Dict[int, List[int]] = {}
            for n, (t, _) in enumerate(items):
                by_len.setdefault(t.shape[2], []).append(n)
            submit = getattr(fx, "extract_audio_features_async", None)
            groups = []
            for _, idxs in by_len.items():
                batch = torch.cat([items[n][0] for n in idxs], dim=0)
                groups.append((idxs, submit(batch) if submit else fx.extract_audio_features(batch)))
            self._pending[media_type].append(("audio", items, groups))
        else:
            batch = torch.cat([t for t, _ in items], dim=0)
            submit = getattr(fx, "extract_image_features_async", None)
            self._pending[media_type].append(("frames", items, submit(batch) if submit else fx.extract_image_features(batch)))
        self._queue[media_type] = []
        self._rows[media_type] = 0
        while len(self._pending[media_type]) > 1:      # keep one batch in flight, write the older ones
            self._drain_one(media_type)

    def _drain_one(self, media_type: str) -> None:
        kind, items, res = self._pending[media_type].pop(0)
        store = self.stores[media_type]
        if kind == "audio":
            feats: List[Optional[np.ndarray]] = [None] * len(items)
            for idxs, out in res:
                out = out.result() if hasattr(out, "result") else out
                for r, n in enumerate(idxs):
                    feats[n] = out[r:r + 1]
            for (t, ids), f in zip(items, feats):
                store.add(ids[0], f)                                         # whole segment, [1, D]
                self.vectors_written += 1
        else:
            out = res.result() if hasattr(res, "result") else res
            row = 0
            for t, ids in items:
                for i, vid in enumerate(ids):
                    store.add(vid, np.expand_dims(out[row + i], axis=0))
                    self.vectors_written += 1
                row += t.shape[0]

    def flush(self) -> None:
        for media_type in self.extractors:
            self._run(media_type)
            while self._pending[media_type]:
                self._drain_one(media_type)

    def close(self) -> None:
        self.flush()
        for store in self.stores.values():
            store.close()


# ---- one process per GPU: file-level sharding, no collective on the data path ---------------------------------------
RANK_SHARD_STRIDE = 100_000      # shard numbers of rank r: r * RANK_SHARD_STRIDE ... (a rank would need 2e8 vectors to run out)


def rank_files(files: Iterable, rank: int, world: int) -> List:
    """The files rank `rank` of `world` extracts: every world-th file starting at `rank`, exactly how the reference
    deals files to DataLoader workers (src/dataloader/dataset.py:322-336)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(itertools.islice(files, rank, None, world))


class RankVectorIds:
    """`create_vector` for one rank: ids n * world + rank + 1 (n = 0, 1, ... in this rank's creation order) — unique
    over all ranks, positive like SQLite's autoincrement (src/db/tables/__init__.py:35-38), no communication.  The rows
    (id, modality, media id, timestamps) are kept for whoever imports them into the `vectors` table afterwards."""

    def __init__(self, rank: int, world: int):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.rank, self.world, self.rows = rank, world, []

    def __call__(self, modality, media_id, timestamp, end_timestamp) -> int:
        vid = len(self.rows) * self.world + self.rank + 1
        self.rows.append((vid, modality, media_id, timestamp, end_timestamp))
        return vid


def open_rank_stores(store_type, features_dirs: Dict[str, object], rank: int, world: int, shard_maxcount: int = 2048,
                     shard_maxsize: int = 20 * 1024 * 1024) -> Dict[str, object]:
    """One writable store per modality for this rank, in the rank's own shard range of the shared directory
    (defaults: the reference's roll-over limits, extract-features.py:158,166)."""
    from .feature.store.feature_store_factory import FeatureStoreFactory

    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    stores = {}
    for media_type, d in features_dirs.items():
        st = FeatureStoreFactory.create_store(store_type, media_type, str(d))
        st.enable_write(shard_maxcount, shard_maxsize, first_shard=rank * RANK_SHARD_STRIDE)
        stores[media_type] = st
    return stores
