#!/usr/bin/env python3
"""SURVEY 8(d)'s metric shape: closed loop with C concurrent clients (C in {1, 8, 64, 512}), each issuing ONE C3
query at a time through nrtgpu_search_bm25_coalesced (the library merges concurrent callers into device
batches).  The clients are native threads inside the library (nrtgpu_bench_closed_loop): Python threads would
measure the GIL.  Reports queries/s and p50 / p99 latency per C."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from nrtsearch_amd import build as _build  # noqa: E402

# the load generator lives in the development build only (include/nrtgpu_dev.h), not in the product library
os.environ["NRTGPU_LIB_PATH"] = _build.DEV_OUT if os.path.exists(_build.DEV_OUT) else _build.build_dev()
from nrtsearch_amd import _lib, api, synth, workload  # noqa: E402


def main():
    w = workload.C3
    n_q = 4096
    qr = synth.make_queries(n_q, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr)
    ctx = api.GpuContext(0, max_batch=1024)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    m = sr._marshal(workload.boolean_queries(qr), [api.TopScoreDocCollectorManager(w.k)] * n_q)
    L = _lib.load()
    for linger in [int(x) for x in os.environ.get("LINGER_US", "150").split(",")]:
        _lib.check(L.nrtgpu_set_coalescing(ctx._h, linger))
        for clients in (1, 8, 64, 512, 2048):
            out = np.zeros(4, dtype=np.float64)
            ctx.reset_stats()
            _lib.check(L.nrtgpu_bench_closed_loop(ctx._h, sr._segs, sr._bases, len(sr.leaves), m.queries, n_q, clients, 2500,
                                                  out.ctypes.data))
            st = ctx.stats()
            print(json.dumps({"clients": clients, "linger_us": linger, "queries": int(out[0]), "qps": round(out[0] / out[1], 1),
                              "p50_ms": round(out[2], 3), "p99_ms": round(out[3], 3),
                              "mean_batch": round(st["queries"] / max(1, st["batches"]), 1)}), flush=True)


if __name__ == "__main__":
    main()
