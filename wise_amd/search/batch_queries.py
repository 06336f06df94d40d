"""Many text queries against one index in one go — the batched form of the reference's `--queries-from` loop.

/root/reference/search.py:894-950 reads a CSV of queries and, per row, calls process_query -> process_text_query (:121-159) ->
FeatureSearchIndex.search (src/index/feature_search_index.py:100-114): ONE text-tower forward of one prompt and ONE nq = 1
index search per row (3842 of them in docs/Retrieval-Evaluation.md:36-45, 0.307 s each).  Here the rows of one media type
become one text-tower batch and one batched index search (the matrix-core passes of csrc/ip_topk_mfma.hip take 128 queries
per pass over the rows): `batched_text_search` returns, per query, exactly what FeatureSearchIndex.search returns for it.

Prompt rule, restated from FeatureSearchIndex.search for a `str` query (what a CSV row is): audio queries go to the text
tower as they are, image / video queries behind the index's prompt ('This is a photo of a ').
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def prompts_for(search_index, media_type: str, queries: Sequence[str]) -> List[str]:
    if media_type == 'audio':
        return [q for q in queries]
    return [search_index.prompt[media_type] + q for q in queries]


def batched_text_search(search_index, media_type: str, queries: Sequence[str], topk: int = 5,
                        batch: int = 256) -> List[Tuple[np.ndarray, np.ndarray]]:
    """[(dist [topk] float32, ids [topk] int64)] — entry i equals search_index.search(media_type, queries[i], topk).

    `batch` queries at a time go through the text tower and the index (256 = two 128-query passes over the rows)."""
    if any(not isinstance(q, str) for q in queries):
        raise ValueError('queries must be strings (one CSV row each)')
    out: List[Tuple[np.ndarray, np.ndarray]] = []
    texts = prompts_for(search_index, media_type, queries)
    for s in range(0, len(texts), batch):
        feats = search_index.feature_extractor.extract_text_features(texts[s:s + batch])
        dist, ids = search_index.index.search(np.ascontiguousarray(feats, dtype=np.float32), topk)
        out.extend((dist[i], ids[i]) for i in range(dist.shape[0]))
    return out
