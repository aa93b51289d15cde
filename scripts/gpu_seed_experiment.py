#!/usr/bin/env python3
"""How much of a shard's MaxScore time is threshold warm-up?  Run every batch twice: plain, then with each query's own
final k-th score handed in as min_competitive_score (a perfect seed).  Needs a one-line planner patch so that seeded
queries stay on the MaxScore route (drop `!(q.min_competitive_score > 0.0f)` from the route condition in
planner.cpp; the product keeps it: a search with a bound from outside counts its hits exactly).  Result of the round-2
run: profiles/r02_seed_experiment.log.  Also a half-strength seed (score of rank 4k of the FIRST run's shard = a weaker
bound).  Prints kernel ms per batch for world = 1 and an emulated rank of 8."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from nrtsearch_amd import api, synth, workload

w = workload.C3
B = 1024
qranks = synth.make_queries(2 * B, w.n_terms, w.max_rank)
for world in (8, 1):
    corpus = workload.build_shard_corpus(w, qranks, world, 0)
    ctx = api.GpuContext(device_id=0, max_batch=B, collect_timing=True)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    searcher = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    queries = workload.boolean_queries(qranks)
    for frac, label in ((0.0, "no seed"), (1.0, "seed = own final k-th score"), (0.9, "seed = 0.9 x"), (0.7, "seed = 0.7 x"), (0.5, "seed = 0.5 x")):
        res = []
        for b0 in (0, B):
            qs = queries[b0: b0 + B]
            plain = api.PreparedBatch(searcher, qs, [api.TopScoreDocCollectorManager(w.k)] * B)
            plain.run()
            kth = [float(plain.topdocs(i).scores[-1]) if len(plain.topdocs(i).scores) == w.k else 0.0 for i in range(B)]
            mgrs = [api.TopScoreDocCollectorManager(w.k, min_competitive_score=float(np.float32(s * frac))) for s in kth]
            seeded = api.PreparedBatch(searcher, qs, mgrs)
            seeded.run()   # warm (plan caches)
            ctx.reset_stats()
            for _ in range(3):
                seeded.run()
            st = ctx.stats()
            res.append((st["maxscore_ms"] / max(1, st["maxscore_launches"]), st["scan_ms"] / max(1, st["batches"]), st["maxscore_items"] / max(1, st["batches"])))
            if frac == 1.0:
                same = sum(plain.topdocs(i).docs.tolist() == seeded.topdocs(i).docs.tolist() for i in range(B))
                print(json.dumps({"world": world, "same_topk": same, "of": B}), flush=True)
        print(json.dumps({"world": world, "seed": label, "maxscore_ms": [round(r[0], 4) for r in res], "scan_ms": [round(r[1], 4) for r in res],
                          "maxscore_items": [r[2] for r in res]}), flush=True)
    for g in leaves:
        g.release()
    ctx.close()
