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
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

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
