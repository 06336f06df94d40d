"""Post-processing of ranked search hits: the rules of the reference CLI's merge step (/root/reference/search.py),
written independently of its code and held to its OUTPUTS by tests/golden/merge_ref (made by oracle/make_golden_merge.py,
which runs the reference's own function bodies):

  segments_overlap      = does_segment_overlap (:192-230)
  merge_ranked_hits     = merge_a_ranked_result_list (:311-363)
  merge_each_query      = merge0 (:253-283)
  merge_modalities / merge_pair = merge1 (:393-445)
  hit_exists / subtract_hits    = result_exists (:180-190) / apply_subtract (:160-178), the --not-in filter

Rules, in the reference's words.  merge_ranked_hits — hits i < j of ONE ranked list are merged when (1) same filename,
(2) |i - j| <= rank tolerance, (3) |pts_i - pts_j| <= time tolerance (two ranges compare by their mid points).  Walking the
list in rank order, hit i collects every later, not yet consumed hit of the same file; pairs among those (including pairs
that do not involve i) that satisfy (2) and (3) put BOTH members into i's merge set; the merged entry keeps i's score, spans
[min, max] of the merged timestamps (a single [t] if only one), and consumes every member.
merge_modalities — for two result lists (e.g. video and audio): every pair with the same filename and overlapping
segments yields one hit whose score is the SUM and whose span is [min, max]; sorted by score, descending, stable.
A point overlaps a segment when it lies inside it; two segments overlap when intersection / hull > 0.01.

One behaviour of the reference is kept because its outputs depend on it: does_segment_overlap turns a one-element list
[t] into [t, t] IN PLACE.  In merge1 a merged single hit [t] is therefore a *point* in the first same-file pairing it
meets and the zero-length *range* [t, t] in every later one (which overlaps nothing: intersection 0; two such ranges at the
same instant divide by zero — ZeroDivisionError there, ZeroDivisionError here).  This module reproduces that on private
copies; the caller's lists are never modified.  `merged_rank_list` entries are sorted (the reference emits a set's
iteration order and sorts before printing, search.py:586-599).
"""
from __future__ import annotations

from itertools import combinations
from typing import List, Sequence, Tuple, Union

Pts = Union[float, List[float]]


def _as_points(p: Pts) -> List[float]:
    return list(p) if isinstance(p, list) else [p]


def _centre(p: Pts) -> float:
    return sum(p) / len(p) if isinstance(p, list) else p


def _overlap_widening(a: Pts, b: Pts) -> bool:
    """does_segment_overlap with its side effect: a one-element list argument leaves as [t, t]."""
    a_point = b_point = False
    if isinstance(a, float):
        a, a_point = [a, a], True
    if isinstance(b, float):
        b, b_point = [b, b], True
    if len(a) == 1:
        a.append(a[0])
        a_point = True
    if len(b) == 1:
        b.append(b[0])
        b_point = True
    assert len(a) == 2, f'segment1 must be defined using a list of length 2; received {a}'
    assert len(b) == 2, f'segment2 must be defined using a list of length 2; received {b}'
    if a_point:
        return b[0] <= a[0] <= b[1]
    if b_point:
        return a[0] <= b[0] <= a[1]
    hull = max(a + b) - min(a + b)
    return (min(a[1], b[1]) - max(a[0], b[0])) / hull > 0.01


def segments_overlap(a: Pts, b: Pts) -> bool:
    """Pure form: the arguments are left as they are."""
    return _overlap_widening(_as_points(a) if isinstance(a, list) else a, _as_points(b) if isinstance(b, list) else b)


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


def merge_each_query(result: List[dict], args) -> List[dict]:
    """merge0: every entry of `result` (one per --query/--in pair, 'in' holding ONE media type) is merged on its own with
    the tolerances `args.merge_tolerance_<media type>` / `args.merge_rank_tolerance`; images merge nothing (0 / 0).
    Returns new dicts (the reference rewrites its argument in place and returns it)."""
    out = []
    for entry in result:
        assert len(entry['in']) == 1, f'unexpected {entry["in"]}'
        media_type = entry['in'][0]
        if media_type == 'image':
            time_tol, rank_tol = 0, 0
        else:
            time_tol, rank_tol = getattr(args, 'merge_tolerance_' + media_type), getattr(args, 'merge_rank_tolerance')
        f, p, s, r = merge_ranked_hits(entry['match_filename_list'], entry['match_pts_list'], entry['match_score_list'],
                                       time_tol, rank_tol)
        merged = dict(entry)
        merged.update(match_filename_list=f, match_pts_list=p, match_score_list=s, merged_rank_list=r)
        out.append(merged)
    return out


def merge_modalities(first: dict, second: dict) -> dict:
    """first/second: two entries as merge_each_query returns them (pts are [t] or [start, end]) -> one merged dict."""
    pts0 = [_as_points(p) if isinstance(p, list) else p for p in first['match_pts_list']]    # private, widened below
    pts1 = [_as_points(p) if isinstance(p, list) else p for p in second['match_pts_list']]
    hits = []
    for i0, (f0, s0) in enumerate(zip(first['match_filename_list'], first['match_score_list'])):
        for i1, (f1, s1) in enumerate(zip(second['match_filename_list'], second['match_score_list'])):
            if f0 != f1 or not _overlap_widening(pts0[i0], pts1[i1]):
                continue
            stamps = sorted(_as_points(pts0[i0]) + _as_points(pts1[i1]))
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


def merge_pair(result: List[dict], args=None):
    """merge1's call shape: a list of exactly two entries -> a list of one; anything else prints and returns None."""
    if len(result) != 2:
        print('merge1() can be only applied if result contains two entries')
        return None
    return [merge_modalities(result[0], result[1])]


def hit_exists(filename: str, pts: Pts, results: dict) -> bool:
    """result_exists: does a hit of `results` lie in the same file at an overlapping time?  (raw hits: floats / ranges)"""
    for other_file, other_pts in zip(results['match_filename_list'], results['match_pts_list']):
        if filename == other_file and segments_overlap(pts, other_pts):
            return True
    return False


def subtract_hits(search_result: dict, not_search_result: dict) -> dict:
    """apply_subtract: the --in hits that no --not-in hit overlaps; 'query', 'in', 'not_in' (and 'query_id') carried over."""
    kept = {'match_filename_list': [], 'match_pts_list': [], 'match_score_list': [],
            'query': search_result['query'], 'in': search_result['in'], 'not_in': search_result['not_in']}
    if 'query_id' in search_result:
        kept['query_id'] = search_result['query_id']
    for f, p, s in zip(search_result['match_filename_list'], search_result['match_pts_list'],
                       search_result['match_score_list']):
        if not hit_exists(f, p, not_search_result):
            kept['match_filename_list'].append(f)
            kept['match_pts_list'].append(p)
            kept['match_score_list'].append(s)
    return kept
