#!/usr/bin/env python3
"""WHO is the tail of a MaxScore launch?  (GPU; instrumented kernel.)  One C3 batch: the five owner pieces that end last -- their
item, the windows the owner walked itself, and every helper session on that item (start, end, windows)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NRTGPU_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nrtsearch_amd", "libnrtgpu_dev.so"))   # instrumented kernels: the development library (include/nrtgpu_dev.h)

import numpy as np  # noqa: E402

from nrtsearch_amd import _lib, api, synth, workload  # noqa: E402


def main():
    w = workload.C3
    B = 1024
    qr = synth.make_queries(B * 2, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr, 1, 0)
    queries = workload.boolean_queries(qr)
    mgr = api.TopScoreDocCollectorManager(w.k)
    ctx = api.GpuContext(0, max_batch=B, collect_timing=True, flags=_lib.NRTGPU_FLAG_PROFILE)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    pbs = [api.PreparedBatch(sr, queries[i * B:(i + 1) * B], [mgr] * B) for i in range(2)]
    pbs[0].run()
    for bi, pb in enumerate(pbs):
        pb.run()
        walls, n_items = ctx.maxscore_item_walls()
        flags = (walls[:, 2] >> np.uint64(32)).astype(np.uint32)
        query = (walls[:, 6] >> np.uint64(32)).astype(np.uint32)
        walls[:, 2] &= np.uint64(0xFFFFFFFF)
        walls[:, 6] &= np.uint64(0xFFFFFFFF)
        used = walls[:, 1] > 0
        t0 = int(walls[used, 4].min())
        start = (walls[:, 0].astype(np.int64) - t0) / 100.0
        end = (walls[:, 1].astype(np.int64) - t0) / 100.0
        owners = np.arange(len(walls)) < n_items
        order = np.argsort(-np.where(used & owners, end, -1.0))[:5]
        for o in order:
            it = int(walls[o, 2])
            hs = np.where(used & ~owners & (walls[:, 2] == it))[0]
            qq = int(query[o])
            print(json.dumps({"batch": bi, "item": it, "query": qq, "mode": int(flags[o] & 3), "flags": hex(int(flags[o])), "ranks": [int(x) for x in qr[bi * B + qq]],
                              "owner": {"start_us": round(float(start[o]), 1), "end_us": round(float(end[o]), 1), "windows": int(walls[o, 3])},
                              "helpers": [{"start_us": round(float(start[h]), 1), "end_us": round(float(end[h]), 1), "windows": int(walls[h, 3])} for h in hs]}), flush=True)
    # the same batch one query per call: which queries are slow on their own, and what are their terms?
    import time
    ctx2 = api.GpuContext(0, max_batch=B, collect_timing=True)
    leaves2 = [api.GpuSegment.from_data(ctx2, s) for s in corpus.segments]
    sr2 = api.GpuIndexSearcher(ctx2, leaves2, api.IndexStatistics.from_corpus(corpus))
    times = []
    for qi in range(B):
        sr2.search(queries[qi], mgr)
        t = time.perf_counter()
        sr2.search(queries[qi], mgr)
        times.append(time.perf_counter() - t)
    for qi in np.argsort(-np.asarray(times))[:8]:
        ctx2.reset_stats()
        r = sr2.search(queries[int(qi)], mgr)
        st = ctx2.stats()
        print(json.dumps({"query": int(qi), "alone_ms": round(times[int(qi)] * 1e3, 3), "ranks": [int(x) for x in qr[int(qi)]],
                          "df": [int(corpus.doc_freq.get(int(x), 0)) for x in qr[int(qi)]], "total_hits": int(r.total_hits), "gte": bool(r.relation_gte),
                          "maxscore_items": int(st["maxscore_items"]), "scan_items": int(st["scan_items"])}), flush=True)
    for l in leaves2:
        l.release()
    ctx2.close()
    for l in leaves:
        l.release()
    ctx.close()


if __name__ == "__main__":
    main()
