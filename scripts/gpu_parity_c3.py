#!/usr/bin/env python3
"""Full-size parity soak: N queries of the C3 set (10 M docs, 5 terms, top-1000) through the device against the
oracle's exhaustive scorer (OpenMP batch driver, one collector per slice) -- docids, score bits and relation of every query, its
totalHits exactly where the relation is EQUAL_TO and as a lower bound above the threshold where it is GREATER_THAN_OR_EQUAL_TO --
for the plain index and with 1 % deletes (folded into the postings, and under NRTGPU_FLAG_NO_LIVE_FOLD: the
masked scan variant).  Prints one JSON line per configuration."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from nrtsearch_amd import _lib, api, synth, workload  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=512)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    w = workload.C3
    n = args.queries
    qr = synth.make_queries(4096, w.n_terms, w.max_rank)[-n:]      # the tail of the bench's query set
    corpus = workload.build_shard_corpus(w, qr, 1, 0)
    rng = np.random.default_rng(17)
    for name, deletes, flags in (("plain", False, 0), ("deletes_folded", True, 0), ("deletes_masked_variant", True, _lib.NRTGPU_FLAG_NO_LIVE_FOLD)):
        for seg in corpus.segments:
            if deletes and seg.live_bits is None:
                alive = rng.random(seg.max_doc) >= 0.01
                padded = np.zeros(((seg.max_doc + 63) // 64) * 64, dtype=bool)
                padded[: seg.max_doc] = alive
                seg.live_bits = np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1)
        t0 = time.time()
        exp = oracle.PreparedBatch(corpus, [r.tolist() for r in qr], w.k, slicing=oracle.DEFAULT_SLICING).run(False, args.threads)
        t_oracle = time.time() - t0
        ctx = api.GpuContext(0, max_batch=n, flags=flags)
        leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        got = sr.search_batch(workload.boolean_queries(qr), [api.TopScoreDocCollectorManager(w.k)] * n)
        bad = 0
        for qi in range(n):
            m = int(exp[2][qi])
            # docids, ranks and score bits are the exhaustive scorer's; the relation is the reference's per-slice one; the
            # count is exact when EQUAL_TO and -- on the pruned route, as Lucene -- a lower bound above the threshold when GTE
            gte, total = bool(exp[4][qi]), int(exp[3][qi])
            ok = (got[qi].docs.tolist() == exp[0][qi][:m].tolist()
                  and got[qi].scores.view(np.uint32).tolist() == exp[1][qi][:m].view(np.uint32).tolist()
                  and got[qi].relation_gte == gte
                  and ((max(w.k, 1000) < got[qi].total_hits <= total) if gte else got[qi].total_hits == total))
            bad += not ok
        print(json.dumps({"config": name, "queries": n, "mismatches": int(bad), "oracle_s": round(t_oracle, 1),
                          "hits_checked": int(exp[2].sum())}), flush=True)
        for l in leaves:
            l.release()
        ctx.close()


if __name__ == "__main__":
    main()
