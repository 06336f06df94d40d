"""Reference-generated fixtures for the result-merge rules of the CLI (test infrastructure; BUILD container only).

/root/reference/search.py cannot be imported as a module (it pulls src.dataloader -> torchaudio, not installed), but its
merge rules are PURE functions.  This script parses the file with `ast`, compiles ONLY these FunctionDef nodes

    apply_subtract (:160-178), result_exists (:180-190), does_segment_overlap (:192-230), merge0 (:253-283),
    merge_a_ranked_result_list (:311-363), merge1 (:393-445)

into an empty namespace that holds `itertools` and `math` (the two modules they use), runs them on seeded inputs and
writes inputs + outputs — data only, no source text — to tests/golden/merge_ref/cases.json.  tests/test_search_merge.py
holds wise_amd/search/merge.py to these outputs exactly.

What the cases cover: points (float), one-element lists, [start, end] ranges; video (floats on a 0.5-s grid), audio
([t, t+4] ranges) and image (tolerance 0) lists; rank / time tolerances incl. the CLI defaults (4 / 8 s, rank 20); equal scores;
the two-list flow merge0 -> merge1 of the reference's main() (:887-891), whose does_segment_overlap call APPENDS to
one-element lists in place (so a merged single hit [t] is a point for the first pairing it meets and the zero-length
range [t, t] afterwards — recorded as the reference behaves, exceptions (ZeroDivisionError) included); --not-in
subtraction on raw hit lists.

    python oracle/make_golden_merge.py        (needs /root/reference; never runs on the GPU box)
"""
import ast
import copy
import itertools
import json
import math
import random
from pathlib import Path
from types import SimpleNamespace

SRC = Path("/root/reference/search.py")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "merge_ref"
WANTED = ("apply_subtract", "result_exists", "does_segment_overlap", "merge0", "merge_a_ranked_result_list", "merge1")


def load_reference_functions():
    tree = ast.parse(SRC.read_text())
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(n.name for n in picked) == sorted(WANTED), [n.name for n in picked]
    ns = {"itertools": itertools, "math": math}
    exec(compile(ast.Module(body=picked, type_ignores=[]), str(SRC), "exec"), ns)
    return SimpleNamespace(**{name: ns[name] for name in WANTED})


def call(fn, *a):
    """-> {'out': ...} or {'raises': 'ExceptionName'}; arguments are deep-copied (the reference mutates them)."""
    try:
        return {"out": fn(*copy.deepcopy(a))}
    except Exception as e:  # noqa: BLE001 — the exception type IS the recorded behaviour
        return {"raises": type(e).__name__}


def rnd_pts(rng, kind):
    if kind == "video":
        return rng.randrange(0, 60) * 0.5
    if kind == "audio":
        t = float(rng.randrange(0, 12) * 4)
        return [t, t + 4.0]
    return float(rng.randrange(0, 3))        # image: pts is whatever the vector row holds; ties matter, not values


def rnd_hits(rng, kind, n, nfiles):
    files = [f"{'abcdefgh'[rng.randrange(nfiles)]}.mp4" for _ in range(n)]
    pts = [rnd_pts(rng, kind) for _ in range(n)]
    raw = sorted((round(rng.uniform(0.1, 0.4), 3 if rng.random() < 0.5 else 6) for _ in range(n)), reverse=True)
    if n > 3 and rng.random() < 0.3:
        raw[2] = raw[1]                       # equal scores
    return files, pts, raw


def main():
    ref = load_reference_functions()
    rng = random.Random(20250117)
    cases = {"overlap": [], "ranked": [], "flow": [], "subtract": []}

    # does_segment_overlap on every kind pair, incl. boundaries and the 0.01 threshold
    def seg(kind):
        a = rng.randrange(0, 40) * 0.25
        if kind == 0:
            return a
        if kind == 1:
            return [a]
        return [a, a + rng.choice([0.0, 0.01, 0.03, 0.25, 1.0, 4.0, 8.0])]
    for _ in range(400):
        s1, s2 = seg(rng.randrange(3)), seg(rng.randrange(3))
        cases["overlap"].append({"seg1": s1, "seg2": s2, **call(ref.does_segment_overlap, s1, s2)})
    for s1, s2 in [([0.0, 4.0], [3.97, 8.0]), ([0.0, 4.0], [3.9, 8.0]), ([0.0, 4.0], [4.0, 8.0]), ([1.0, 1.0], [1.0, 1.0]),
                   ([1.0], [1.0]), (1.0, 1.0), ([0.0, 100.0], [99.0, 100.0]), ([0.0, 100.0], [98.9, 100.0])]:
        cases["overlap"].append({"seg1": s1, "seg2": s2, **call(ref.does_segment_overlap, s1, s2)})

    # merge_a_ranked_result_list
    tolerances = [(4, 20), (8, 20), (0, 0), (1, 1), (2, 3), (0.5, 2), (100, 100), (4, 0)]
    for t in range(240):
        kind = ("video", "audio", "image")[t % 3]
        n = rng.choice([0, 1, 2, 3, 5, 8, 13, 20, 30, 45])
        files, pts, scores = rnd_hits(rng, kind, n, rng.choice([1, 2, 3, 6]))
        tol = tolerances[rng.randrange(len(tolerances))] if kind != "image" else (0, 0)
        cases["ranked"].append({"kind": kind, "files": files, "pts": pts, "scores": scores, "pts_tolerance": tol[0],
                                "rank_tolerance": tol[1],
                                **call(ref.merge_a_ranked_result_list, files, pts, scores, tol[0], tol[1])})

    # the CLI's flow: merge0 over one or two raw result lists, then merge1 when there are two
    for t in range(160):
        kinds = [("video",), ("audio",), ("image",), ("video", "audio"), ("audio", "video"), ("video", "video"),
                 ("audio", "audio")][t % 7]
        args = {"merge_tolerance_video": rng.choice([4, 1, 0]), "merge_tolerance_audio": rng.choice([8, 4, 0]),
                "merge_tolerance_metadata": 0, "merge_rank_tolerance": rng.choice([20, 2, 0])}
        result = []
        nfiles = rng.choice([1, 2, 4])
        for kind in kinds:
            files, pts, scores = rnd_hits(rng, kind, rng.choice([0, 1, 3, 6, 10, 16]), nfiles)
            entry = {"match_filename_list": files, "match_pts_list": pts, "match_score_list": scores,
                     "query": [f"q-{kind}"], "in": [kind], "search_time_sec": round(rng.uniform(0.01, 0.3), 4)}
            if rng.random() < 0.5:
                entry["not_in"] = [] if rng.random() < 0.5 else ["audio"]
            result.append(entry)
        rec = {"args": args, "result": result}
        try:
            m0 = ref.merge0(copy.deepcopy(result), SimpleNamespace(**args))
            rec["merge0"] = {"out": copy.deepcopy(m0)}
            if len(m0) == 2:
                try:
                    rec["merge1"] = {"out": ref.merge1(m0, SimpleNamespace(**args))}
                except Exception as e:  # noqa: BLE001
                    rec["merge1"] = {"raises": type(e).__name__}
        except Exception as e:  # noqa: BLE001
            rec["merge0"] = {"raises": type(e).__name__}
        cases["flow"].append(rec)

    # --not-in: apply_subtract on raw hit lists (process_query :100-106)
    for t in range(80):
        k1, k2 = [("video", "video"), ("video", "audio"), ("audio", "video"), ("audio", "audio")][t % 4]
        nfiles = rng.choice([1, 2, 3])
        f1, p1, s1 = rnd_hits(rng, k1, rng.choice([0, 2, 6, 12]), nfiles)
        f2, p2, s2 = rnd_hits(rng, k2, rng.choice([0, 2, 6, 12]), nfiles)
        sr = {"match_filename_list": f1, "match_pts_list": p1, "match_score_list": s1, "query": ["a"], "in": [k1], "not_in": []}
        if rng.random() < 0.3:
            sr["query_id"] = ["7"]
        nr = {"match_filename_list": f2, "match_pts_list": p2, "match_score_list": s2}
        cases["subtract"].append({"search_result": sr, "not_search_result": nr, **call(ref.apply_subtract, sr, nr)})

    OUT.mkdir(parents=True, exist_ok=True)
    (OUT / "cases.json").write_text(json.dumps(cases, separators=(",", ":")))
    n_exc = sum(1 for grp in cases.values() for c in grp
                if "raises" in c or "raises" in c.get("merge0", {}) or "raises" in c.get("merge1", {}))
    print({k: len(v) for k, v in cases.items()}, "cases with a recorded exception:", n_exc)


if __name__ == "__main__":
    main()
