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
]


def main():
    oracle.build()
    corpus = synth.build_corpus(SPEC["n_docs"], SPEC["ranks"], n_segments=SPEC["n_segments"],
                                delete_fraction=SPEC["delete_fraction"], seed=SPEC["seed"])
    out = {"spec": SPEC, "cases": []}
    for c in CASES:
        docs, scores, total, gte = oracle.search_bm25(corpus, c["terms"], c["k"], boosts=c.get("boosts"),
                                                      total_hits_threshold=c["threshold"])
        out["cases"].append({**c, "docs": docs.tolist(), "score_bits": scores.view(np.uint32).tolist(),
                             "total_hits": int(total), "relation_gte": bool(gte)})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small_corpus.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
