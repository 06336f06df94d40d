"""Result-merge rules of the reference CLI (search.py:192-445), checked on hand-worked cases that follow the rules
the reference states in its docstrings (:285-310, :372-396)."""
from wise_amd.search.merge import merge_modalities, merge_ranked_hits, segments_overlap


def test_segments_overlap_rules():
    assert segments_overlap(2.0, [1.0, 3.0]) and segments_overlap([1.0, 3.0], 3.0)        # point inside segment
    assert not segments_overlap(3.5, [1.0, 3.0])
    assert segments_overlap([0.0, 4.0], [2.0, 6.0])                                        # iou = 2/6
    assert not segments_overlap([0.0, 4.0], [3.97, 8.0])                                   # 0.03/8 < 0.01
    assert not segments_overlap([0.0, 4.0], [5.0, 8.0])                                    # disjoint -> negative
    assert segments_overlap([2.0], [2.0, 2.0]) is True                                     # single-element list = point


def test_merge_ranked_hits_video_frames():
    # 2 fps frames of one file: ranks 0,1,2 are 0.5 s apart, rank 3 is another file, rank 4 same file but far in time
    files = ['a.mp4', 'a.mp4', 'a.mp4', 'b.mp4', 'a.mp4', 'a.mp4']
    pts = [10.0, 10.5, 11.0, 3.0, 50.0, 11.5]
    scores = [0.9, 0.8, 0.7, 0.6, 0.5, 0.4]
    f, p, s, r = merge_ranked_hits(files, pts, scores, pts_tolerance=1.0, rank_tolerance=20)
    # chain 10.0-10.5-11.0-11.5 merges through pairwise-close members even though 10.0 and 11.5 are 1.5 s apart
    assert f == ['a.mp4', 'b.mp4', 'a.mp4']
    assert p == [[10.0, 11.5], [3.0], [50.0]]
    assert s == [0.9, 0.6, 0.5]                      # merged entry keeps the best-ranked member's score
    assert r == [[0, 1, 2, 5], [3], [4]]
    # rank tolerance: the same hits with tolerance 1 cannot pull in rank 5 (|2-5| > 1)
    f, p, s, r = merge_ranked_hits(files, pts, scores, pts_tolerance=1.0, rank_tolerance=1)
    assert p[0] == [10.0, 11.0] and r[0] == [0, 1, 2] and f == ['a.mp4', 'b.mp4', 'a.mp4', 'a.mp4']
    # image search: tolerances 0 merge nothing but exact duplicates
    f, p, s, r = merge_ranked_hits(['x', 'x'], [1.0, 1.0], [0.5, 0.4], 0, 0)
    assert f == ['x', 'x']                            # same time but rank distance 1 > 0


def test_merge_ranked_hits_audio_ranges_use_midpoints():
    files = ['a.mp4', 'a.mp4', 'a.mp4']
    pts = [[0.0, 4.0], [4.0, 8.0], [20.0, 24.0]]     # audio hits are [pts, pts + 4] ranges
    f, p, s, r = merge_ranked_hits(files, pts, [0.9, 0.8, 0.7], pts_tolerance=4.0, rank_tolerance=20)
    assert p == [[0.0, 8.0], [20.0, 24.0]] and r == [[0, 1], [2]] and s == [0.9, 0.7]


def test_merge_modalities_sums_scores_of_overlapping_hits():
    video = {'match_filename_list': ['a.mp4', 'b.mp4'], 'match_pts_list': [[10.0, 11.5], [3.0]],
             'match_score_list': [0.30, 0.28], 'search_time_sec': 0.1, 'query': ['cooking'], 'in': ['video']}
    audio = {'match_filename_list': ['b.mp4', 'a.mp4', 'a.mp4'], 'match_pts_list': [[0.0, 4.0], [8.0, 12.0], [40.0, 44.0]],
             'match_score_list': [0.50, 0.20, 0.90], 'search_time_sec': 0.2, 'query': ['music'], 'in': ['audio'],
             'not_in': ['image']}
    m = merge_modalities(video, audio)
    assert m['match_filename_list'] == ['b.mp4', 'a.mp4']            # 0.78 > 0.50, the 40-44 s audio hit has no video
    assert m['match_score_list'] == [0.28 + 0.50, 0.30 + 0.20]
    assert m['match_pts_list'] == [[0.0, 4.0], [8.0, 12.0]]          # hull of point 3.0 with [0,4]; [10,11.5] with [8,12]
    assert m['merged_rank_list'] == [[1, 0], [0, 1]]
    assert m['in'] == ['video', 'audio'] and m['not_in'] == ['image'] and abs(m['search_time_sec'] - 0.3) < 1e-12


# ------------------------------------------------------------------------------------------------------------------
# Held to the reference's own functions: tests/golden/merge_ref/cases.json holds inputs and the outputs (or the exception)
# of search.py's does_segment_overlap / merge_a_ranked_result_list / merge0 / merge1 / apply_subtract, produced by
# oracle/make_golden_merge.py from the reference's function bodies (ast-lifted; search.py itself is not importable).
import copy
import json
from pathlib import Path
from types import SimpleNamespace

import pytest

from wise_amd.search.merge import merge_each_query, merge_pair, subtract_hits

CASES = json.loads((Path(__file__).parent / "golden" / "merge_ref" / "cases.json").read_text())


def _same_or_raises(expected, fn, *a):
    if "raises" in expected:
        with pytest.raises(Exception) as e:
            fn(*a)
        assert type(e.value).__name__ == expected["raises"]
        return None
    return fn(*a)


def test_segments_overlap_equals_the_reference_on_408_pairs():
    assert len(CASES["overlap"]) >= 400
    for c in CASES["overlap"]:
        s1, s2 = copy.deepcopy(c["seg1"]), copy.deepcopy(c["seg2"])
        got = _same_or_raises(c, segments_overlap, s1, s2)
        if "out" in c:
            assert got is c["out"], c
        assert s1 == c["seg1"] and s2 == c["seg2"]          # the pure form leaves its arguments alone


def test_merge_ranked_hits_equals_the_reference_on_240_lists():
    kinds = set()
    for c in CASES["ranked"]:
        f, p, s, r = merge_ranked_hits(c["files"], c["pts"], c["scores"], c["pts_tolerance"], c["rank_tolerance"])
        ef, ep, es, er = c["out"]
        assert (f, p, s) == (ef, ep, es), c
        assert r == [sorted(x) for x in er], c                # the reference prints them sorted (search.py:586-599)
        kinds.add(c["kind"])
    assert kinds == {"video", "audio", "image"}


def test_merge0_then_merge1_flow_equals_the_reference():
    n_two = n_raise = 0
    for c in CASES["flow"]:
        args = SimpleNamespace(**c["args"])
        inp = copy.deepcopy(c["result"])
        m0 = merge_each_query(inp, args)
        assert inp == c["result"]                              # caller's lists untouched
        exp0 = c["merge0"]["out"]
        assert len(m0) == len(exp0)
        for got, exp in zip(m0, exp0):
            exp = dict(exp, merged_rank_list=[sorted(x) for x in exp["merged_rank_list"]])
            assert got == exp, c
        if "merge1" not in c:
            assert merge_pair(m0, args) is None or len(m0) == 2
            continue
        n_two += 1
        before = copy.deepcopy(m0)
        got1 = _same_or_raises(c["merge1"], merge_pair, m0, args)
        assert m0 == before
        if "raises" in c["merge1"]:
            n_raise += 1
            continue
        exp1 = c["merge1"]["out"]
        assert len(got1) == 1 and got1[0] == exp1[0], c
    assert n_two >= 60 and n_raise >= 1                        # the [t] -> [t, t] widening is exercised (ZeroDivisionError cases)


def test_subtract_hits_equals_the_reference():
    dropped = 0
    for c in CASES["subtract"]:
        got = subtract_hits(copy.deepcopy(c["search_result"]), copy.deepcopy(c["not_search_result"]))
        assert got == c["out"], c
        dropped += len(c["search_result"]["match_filename_list"]) - len(got["match_filename_list"])
    assert dropped > 50


def test_batched_text_search_equals_one_search_per_query():
    """wise_amd/search/batch_queries.py against FeatureSearchIndex.search called row by row (the reference's --queries-from loop,
    search.py:894-950), with a stand-in text tower and the oracle as the index: same prompts, same (dist, ids) per query."""
    import zlib

    import numpy as np

    from oracle import ip_topk_ref
    from wise_amd.index.feature_search_index import FeatureSearchIndex

    d = 32
    X = np.random.default_rng(1).standard_normal((500, d)).astype(np.float32)
    seen = []

    class Tower:
        def extract_text_features(self, texts):
            seen.append(list(texts))
            out = np.stack([np.random.default_rng(zlib.crc32(t.encode())).standard_normal(d) for t in texts])
            return (out / np.linalg.norm(out, axis=1, keepdims=True)).astype(np.float32)

    class Index:
        def search(self, q, k):
            return ip_topk_ref.ip_topk(X, q, k, ids=np.arange(500, dtype=np.int64) + 1)

    for media in ("video", "audio"):
        si = FeatureSearchIndex(media, "mlfoundations/open_clip/ViT-B-32/seeded-0", {"features_dir": ".", "index_dir": "."})
        si.index, si.feature_extractor = Index(), Tower()
        queries = [f"query number {i}" for i in range(300)]
        seen.clear()
        got = si.search_batch(media, queries, topk=7)
        assert [len(b) for b in seen] == [256, 44]                     # two text-tower batches, not 300 calls
        assert seen[0][0] == ("query number 0" if media == "audio" else "This is a photo of a query number 0")
        assert len(got) == 300
        for i in (0, 1, 255, 256, 299):
            dist, ids = si.search(media, queries[i], topk=7)
            assert np.array_equal(got[i][1], ids) and np.allclose(got[i][0], dist, atol=2e-5)   # (BLAS sums a batch in another order)
    import pytest
    with pytest.raises(ValueError):
        si.search_batch("audio", ["a"], query_type="image")
