"""Post-processing of ranked search hits, restating the rules of the reference CLI's merge step
(/root/reference/search.py): `merge_ranked_hits` = merge_a_ranked_result_list (:311-370, used by merge0
:253-283) and `merge_modalities` = merge1 (:397-445) with `segments_overlap` = does_segment_overlap (:192-230).
Pure host-side list logic, written independently of the reference's code; rules, in the reference's words:

merge_ranked_hits — hits i < j of ONE ranked list are merged when (1) same filename, (2) |i - j| <= rank
tolerance, (3) |pts_i - pts_j| <= time tolerance (ranges compare by their mid points).  Walking the list in rank
order, hit i collects every later, not yet consumed hit of the same file; pairs among those (including pairs that
do not involve i) that satisfy (2) and (3) put BOTH members into i's merge set; the merged entry keeps i's score,
spans [min, max] of the merged timestamps (a single [t] if only one), and consumes every member.

merge_modalities — for two result lists (e.g. video and audio): every pair with the same filename and
overlapping segments yields one hit whose score is the SUM and whose span is [min, max]; sorted by score, descending.
A point overlaps a segment when it lies inside it; two segments overlap when intersection / hull > 0.01.
"""
from __future__ import annotations

from itertools import combinations
from typing import List, Sequence, Tuple, Union

Pts = Union[float, List[float]]


def _as_points(p: Pts) -> List[float]:
    return list(p) if isinstance(p, list) else [p]


def _centre(p: Pts) -> float:
    return sum(p) / len(p) if isinstance(p, list) else p


def segments_overlap(a: Pts, b: Pts) -> bool:
    pa, pb = _as_points(a), _as_points(b)
    a_point, b_point = len(pa) == 1, len(pb) == 1
    a0, a1 = pa[0], pa[-1]
    b0, b1 = pb[0], pb[-1]
    if len(pa) > 2 or len(pb) > 2:
        raise AssertionError("a segment is one timestamp or a [start, end] pair")
    if a_point:
        return b0 <= a0 <= b1
    if b_point:
        return a0 <= b0 <= a1
    hull = max(a1, b1) - min(a0, b0)
    return (min(a1, b1) - max(a0, b0)) / hull > 0.01


def merge_ranked_hits(filenames: Sequence[str], pts: Sequence[Pts], scores: Sequence[float], pts_tolerance: float,
                      rank_tolerance: int) -> Tuple[List[str], List[List[float]], List[float], List[List[int]]]:
    n = len(filenames)
    consumed = [False] * n
    out_files, out_pts, out_scores, out_ranks = [], [], [], []
    for i in range(n):
        if consumed[i]:
            continue
        same_file = [i] + [j for j in range(i + 1, n) if not consumed[j] and filenames[j] == filenames[i]]
        members = {i}
        for u, v in combinations(same_file, 2):
            both_ranges = isinstance(pts[u], list) and isinstance(pts[v], list)
            gap = abs(_centre(pts[u]) - _centre(pts[v])) if both_ranges else abs(pts[u] - pts[v])
            if gap <= pts_tolerance and abs(u - v) <= rank_tolerance:
                members.update((u, v))
        stamps = sorted(t for m in members for t in _as_points(pts[m]))
        for m in members:
            consumed[m] = True
        out_files.append(filenames[i])
        out_pts.append([stamps[0], stamps[-1]] if len(stamps) > 1 else [stamps[0]])
        out_scores.append(scores[i])
        out_ranks.append(sorted(members))
    return out_files, out_pts, out_scores, out_ranks


def merge_modalities(first: dict, second: dict) -> dict:
    """first/second: {'match_filename_list', 'match_pts_list', 'match_score_list', ...} -> merged dict."""
    hits = []
    for i0, (f0, p0, s0) in enumerate(zip(first['match_filename_list'], first['match_pts_list'],
                                          first['match_score_list'])):
        for i1, (f1, p1, s1) in enumerate(zip(second['match_filename_list'], second['match_pts_list'],
                                              second['match_score_list'])):
            if f0 != f1 or not segments_overlap(_as_points(p0) if isinstance(p0, list) else p0,
                                                _as_points(p1) if isinstance(p1, list) else p1):
                continue
            stamps = sorted(_as_points(p0) + _as_points(p1))
            span = [stamps[0], stamps[-1]] if len(stamps) > 1 else [stamps[0]]
            hits.append((s0 + s1, f0, span, [i0, i1]))
    order = sorted(range(len(hits)), key=lambda t: hits[t][0], reverse=True)  # stable, like the reference's sort
    merged = {
        'match_filename_list': [hits[t][1] for t in order],
        'match_pts_list': [hits[t][2] for t in order],
        'match_score_list': [hits[t][0] for t in order],
        'merged_rank_list': [hits[t][3] for t in order],
    }
    for key in ('search_time_sec', 'query', 'in'):
        if key in first and key in second:
            merged[key] = first[key] + second[key]
    merged['not_in'] = list(first.get('not_in', [])) + list(second.get('not_in', []))
    return merged
