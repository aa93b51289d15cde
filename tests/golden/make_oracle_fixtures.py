#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_small_corpus.json: the CPU oracle's answers on a seeded synthetic
corpus.  Run from the repository root:  python tests/golden/make_oracle_fixtures.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from nrtsearch_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

SPEC = {"n_docs": 20_000, "ranks": [1, 2, 5, 13, 40, 200, 1000], "n_segments": 3, "delete_fraction": 0.01, "seed": 1234}
CASES = [
    {"terms": [1, 13, 200], "k": 10, "threshold": 1000},
    {"terms": [2, 5, 40, 1000], "k": 25, "threshold": 1000},
    {"terms": [1, 2, 5, 13, 40], "k": 100, "threshold": 2**31 - 1},
    {"terms": [1000], "k": 5, "threshold": 1000},
    {"terms": [13, 13, 40], "boosts": [1.0, 2.0, 0.5], "k": 10, "threshold": 1000},
    # query shapes of SURVEY 8f rank 3: minimumNumberShouldMatch, FILTER / MUST_NOT doc-set masks, paging on top
    {"terms": [1, 2, 5, 13, 40], "k": 20, "threshold": 1000, "msm": 3},
    {"terms": [1, 2, 5], "k": 15, "threshold": 2**31 - 1, "msm": 3},
    {"terms": [2, 13, 200], "k": 30, "threshold": 1000, "msm": 1, "filter": {"density": 0.3, "seed": 11}},
    {"terms": [1, 40, 1000], "k": 30, "threshold": 1000, "must_not": {"density": 0.1, "seed": 12}},
    {"terms": [1, 5, 13], "k": 12, "threshold": 1000, "msm": 2, "filter": {"density": 0.5, "seed": 13},
     "must_not": {"density": 0.05, "seed": 14}, "after_rank": 11},
    # the shapes whose score is not one sum (round 4): DisjunctionMaxQuery (tie breaker 0 and > 0), MUST next to SHOULD clauses
    {"terms": [1, 5, 40, 200], "k": 20, "threshold": 1000, "dismax": 0.0},
    {"terms": [2, 13, 40, 1000], "boosts": [1.0, 2.0, 0.5, 3.0], "k": 20, "threshold": 1000, "dismax": 0.3},
    {"terms": [13, 1, 40, 200], "k": 20, "threshold": 1000, "must": [True, False, False, False]},
    {"terms": [2, 40, 1, 5, 1000], "boosts": [1.0, 0.5, 2.0, 1.0, 3.0], "k": 25, "threshold": 2**31 - 1, "must": [True, True, False, False, False],
     "filter": {"density": 0.6, "seed": 15}},
]


def accept_of(corpus, c):
    """Per-segment acceptDocs of a case (None when the case has no masks): masks are PCG64(seed + segment index)."""
    if "filter" not in c and "must_not" not in c:
        return None, None, None
    fm = [synth.random_mask(s.max_doc, c["filter"]["density"], c["filter"]["seed"] + 100 * i) for i, s in enumerate(corpus.segments)] \
        if "filter" in c else None
    mn = [synth.random_mask(s.max_doc, c["must_not"]["density"], c["must_not"]["seed"] + 100 * i) for i, s in enumerate(corpus.segments)] \
        if "must_not" in c else None
    acc = [synth.accept_words(s, fm[i] if fm else None, mn[i] if mn else None) for i, s in enumerate(corpus.segments)]
    return acc, fm, mn


def run_case(corpus, c):
    acc, _, _ = accept_of(corpus, c)
    kw = dict(boosts=c.get("boosts"), total_hits_threshold=c["threshold"], min_should_match=c.get("msm", 0), accept=acc,
              dismax=c.get("dismax"), must=c.get("must"))
    after = None
    if "after_rank" in c:   # page 2: searchAfter the hit at that rank of page 1
        d, s, _, _ = oracle.search_bm25(corpus, c["terms"], c["k"], **kw)
        after = (int(d[c["after_rank"]]), float(s[c["after_rank"]]))
    return oracle.search_bm25(corpus, c["terms"], c["k"], after=after, **kw), after


def main():
    oracle.build()
    corpus = synth.build_corpus(SPEC["n_docs"], SPEC["ranks"], n_segments=SPEC["n_segments"],
                                delete_fraction=SPEC["delete_fraction"], seed=SPEC["seed"])
    out = {"spec": SPEC, "cases": []}
    for c in CASES:
        (docs, scores, total, gte), after = run_case(corpus, c)
        rec = {**c, "docs": docs.tolist(), "score_bits": scores.view(np.uint32).tolist(),
               "total_hits": int(total), "relation_gte": bool(gte)}
        if after is not None:
            rec["after"] = {"doc": after[0], "score_bits": int(np.float32(after[1]).view(np.uint32))}
        out["cases"].append(rec)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small_corpus.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
