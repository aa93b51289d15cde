#!/usr/bin/env python3
"""Upper bound of cross-shard theta sharing: scan rank 0's 1/W shard of the C3 index with each query's
min_competitive_score preset to the GLOBAL k-th best score (taken from a full-index run), compare the scan
time with the unhinted shard scan, and check that the hinted result is the unhinted one cut at the bound."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from nrtsearch_amd import api, synth, workload  # noqa: E402


def run(corpus, queries, mgrs, steps):
    ctx = api.GpuContext(0, max_batch=len(queries), collect_timing=True)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    pb = api.PreparedBatch(sr, queries, mgrs)
    pb.run()
    ctx.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        pb.run()
    dt = (time.perf_counter() - t0) / steps
    st = ctx.stats()
    res = [pb.topdocs(i) for i in range(len(queries))]
    for l in leaves:
        l.release()
    ctx.close()
    return res, st["scan_ms"] / max(1, st["scan_launches"]), dt * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    w = workload.C3
    qr = synth.make_queries(a.queries, w.n_terms, w.max_rank)
    queries = workload.boolean_queries(qr)
    full = workload.build_shard_corpus(w, qr, 1, 0)
    res_full, scan_full, _ = run(full, queries, [api.TopScoreDocCollectorManager(w.k)] * a.queries, 3)
    kth = [float(r.scores[-1]) if len(r.scores) == w.k else 0.0 for r in res_full]
    del full
    shard = workload.build_shard_corpus(w, qr, a.world, 0)
    res_plain, scan_plain, step_plain = run(shard, queries, [api.TopScoreDocCollectorManager(w.k)] * a.queries, a.steps)
    res_hint, scan_hint, step_hint = run(shard, queries, [api.TopScoreDocCollectorManager(w.k, None, 1000, kth[i]) for i in range(a.queries)], a.steps)
    bad = 0
    for i in range(a.queries):
        keep = res_plain[i].scores >= np.float32(kth[i])
        ok = (res_hint[i].docs.tolist() == res_plain[i].docs[keep].tolist()
              and res_hint[i].scores.view(np.uint32).tolist() == res_plain[i].scores[keep].view(np.uint32).tolist()
              and res_hint[i].total_hits == res_plain[i].total_hits)
        bad += 0 if ok else 1
    print(json.dumps({"world": a.world, "scan_full_ms": round(scan_full, 3), "scan_shard_ms": round(scan_plain, 3),
                      "scan_shard_hinted_ms": round(scan_hint, 3), "step_shard_ms": round(step_plain, 3),
                      "step_shard_hinted_ms": round(step_hint, 3), "mismatching_queries": bad,
                      "mean_hits_returned_hinted": float(np.mean([len(r.docs) for r in res_hint]))}))


if __name__ == "__main__":
    main()
