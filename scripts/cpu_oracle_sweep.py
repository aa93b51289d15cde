#!/usr/bin/env python3
"""Thread sweep of the CPU baseline (the oracle's MaxScore-pruned and exhaustive scorers, C + OpenMP) on the C3
workload: what DESIGN.md section 5 quotes next to the GPU number.  Needs no GPU.  One JSON line per thread count."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nrtsearch_amd import synth, workload  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=512)
    ap.add_argument("--threads", default="1,8,16,32,64")
    ap.add_argument("--exhaustive", action="store_true", help="also time the scorer without dynamic pruning")
    args = ap.parse_args()
    w = workload.C3
    w.n_docs = args.docs
    qr = synth.make_queries(args.queries, w.n_terms, w.max_rank)
    t0 = time.time()
    corpus = workload.build_shard_corpus(w, qr, 1, 0)
    print(json.dumps({"event": "corpus", "docs": w.n_docs, "postings": corpus.total_postings, "build_s": round(time.time() - t0, 1)}), flush=True)
    for th in [int(x) for x in args.threads.split(",")]:
        n = min(args.queries, max(16, 8 * th))
        pb = oracle.PreparedBatch(corpus, [r.tolist() for r in qr[:n]], w.k)
        pb.run(True, th)                                   # warm: page in the postings, build the block maxima
        t0 = time.perf_counter()
        res = pb.run(True, th)
        dt = time.perf_counter() - t0
        rec = {"threads": th, "queries": n, "maxscore_qps": round(n / dt, 1),
               "postings_scored_frac": round(res[5] / max(1, sum(int(corpus.doc_freq[int(t)]) for q in qr[:n] for t in q)), 4)}
        if args.exhaustive:
            t0 = time.perf_counter()
            ex = pb.run(False, th)
            rec["exhaustive_qps"] = round(n / (time.perf_counter() - t0), 1)
            rec["same_topk"] = bool((ex[0] == res[0]).all() and (ex[1] == res[1]).all())
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
