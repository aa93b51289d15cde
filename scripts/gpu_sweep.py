#!/usr/bin/env python3
"""One-process GPU sweep (no torch): builds the C3 corpus once, checks parity at full size against
the oracle on a few queries, then times kernel variants / chunkings / batch sizes through the C ABI.
Prints one JSON line per variant; everything is flushed so a crash shows how far it got."""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NRTGPU_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nrtsearch_amd", "libnrtgpu_dev.so"))   # instrumented kernels: the development library (include/nrtgpu_dev.h)

import numpy as np  # noqa: E402

from nrtsearch_amd import _lib, api, synth, workload  # noqa: E402


def log(*a):
    print(*a, flush=True)


def checksum(pb, n):
    h = 0
    for qi in range(n):
        td = pb.topdocs(qi)
        h = zlib.crc32(td.docs.tobytes(), h)
        h = zlib.crc32(td.scores.tobytes(), h)
        h = zlib.crc32(np.int64(td.total_hits).tobytes(), h)
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--oracle-queries", type=int, default=6)
    ap.add_argument("--world", type=int, default=1, help="index only rank 0's docid range of a WORLD-GPU job")
    ap.add_argument("--min-rank", type=int, default=0,
                    help="replace every query term of Zipf rank < MIN_RANK by MIN_RANK + rank (sparse-only query set: "
                         "isolates the per-sub-tile cost of the walk)")
    ap.add_argument("--variants", default="0:0:1024,0:1:1024,1024:0:1024,8192:0:1024,0:0:256,0:0:64,0:0:1",
                    help="comma list of target_items:flags:batch")
    args = ap.parse_args()

    w = workload.C3
    w.n_docs = args.docs
    t0 = time.time()
    qr = synth.make_queries(args.queries, w.n_terms, w.max_rank)
    if args.min_rank > 0:
        qr = np.where(qr < args.min_rank, qr + args.min_rank, qr)
    corpus = workload.build_shard_corpus(w, qr, args.world, 0)
    ppq = workload.postings_per_query(corpus.doc_freq, qr)
    log(json.dumps({"event": "corpus", "docs": w.n_docs, "postings": corpus.total_postings, "build_s": round(time.time() - t0, 1),
                    "mean_P": float(ppq.mean())}))
    queries = workload.boolean_queries(qr)
    mgr = api.TopScoreDocCollectorManager(w.k)
    ref_sum = {}
    for vi, v in enumerate(args.variants.split(",")):
        ti, fl, B = (int(x) for x in v.split(":"))
        ctx = api.GpuContext(0, max_batch=max(B, 8), target_items=ti, collect_timing=True, flags=fl)
        leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        if vi == 0 and args.oracle_queries:
            from oracle import oracle
            n = args.oracle_queries
            got = sr.search_batch(queries[:n], [mgr] * n)
            bad = 0
            for qi in range(n):
                d, s_, tot, gte = oracle.search_bm25(corpus, qr[qi].tolist(), w.k)
                tot_ok = (w.k < got[qi].total_hits <= tot) if gte else got[qi].total_hits == tot   # pruned: a lower bound
                ok = (got[qi].docs.tolist() == d.tolist() and got[qi].scores.view(np.uint32).tolist() == s_.view(np.uint32).tolist()
                      and tot_ok and got[qi].relation_gte == gte)
                bad += (not ok)
                if not ok:
                    log(json.dumps({"event": "MISMATCH", "query": qi, "got_total": got[qi].total_hits, "exp_total": tot,
                                    "got_n": len(got[qi].docs), "exp_n": len(d),
                                    "first_docs_got": got[qi].docs[:5].tolist(), "first_docs_exp": d[:5].tolist()}))
            log(json.dumps({"event": "oracle_parity_full_size", "queries": n, "mismatches": bad}))
        nb = max(1, min(4, args.queries // B))
        pbs = [api.PreparedBatch(sr, queries[i * B:(i + 1) * B], [mgr] * B) for i in range(nb)]
        pbs[0].run()  # warm
        cs = checksum(pbs[0], B)
        key = (B, bool(fl & 16))   # total_hits of a pruned search is a lower bound: its own checksum class
        if key in ref_sum and ref_sum[key] != cs:
            log(json.dumps({"event": "CHECKSUM_MISMATCH", "variant": v}))
        ref_sum.setdefault(key, cs)
        ctx.reset_stats()
        lat = []
        t1 = time.perf_counter()
        for i in range(args.steps):
            ts = time.perf_counter()
            pbs[i % nb].run()
            lat.append(time.perf_counter() - ts)
        dt = time.perf_counter() - t1
        st = ctx.stats()
        L = max(1, st["scan_launches"], st["maxscore_launches"])
        scan_ms = (st["scan_ms"] + st["maxscore_ms"]) / L
        bytes_l = (st["scan_postings"] + st["maxscore_postings"]) / L * 8
        prof = ctx.scan_profile() if ((fl >> 8) & 15) == 7 else None
        per_item = None
        if prof:   # the counters are sums over every item of every launch since reset_stats (wave 0 of each workgroup)
            n_it = max(1.0, float(st["scan_items"]))
            per_item = {k: round(v / n_it, 2) for k, v in prof.items() if not k.endswith("finish_cycles")}
        ms_prof = ctx.maxscore_profile() if ((fl >> 8) & 15) == 7 and st["maxscore_items"] else None
        if ms_prof:
            nq_run = max(1, st["queries"])
            ms_prof = {k: round(v / nq_run, 1) for k, v in ms_prof.items()}   # per query
        log(json.dumps({"event": "variant", "maxscore_items": st["maxscore_items"] / L, "maxscore_ms": round(st["maxscore_ms"] / L, 3),
                        "maxscore_profile_per_query": ms_prof, "target_items": ti, "flags": fl, "batch": B, "qps": round(args.steps * B / dt, 1),
                        "ms_per_step": round(dt / args.steps * 1e3, 3), "p50_ms": round(float(np.median(lat)) * 1e3, 3),
                        "scan_ms": round(scan_ms, 3), "merge_ms": round(st["merge_ms"] / L, 3),
                        "plan_ms": round(st["host_plan_ms"] / L, 3), "items": st["scan_items"] / L,
                        "GBps_scan": round(bytes_l / (scan_ms * 1e-3) / 1e9, 1) if scan_ms > 0 else None,
                        "checksum": cs, "profile": prof, "profile_per_item": per_item}))
        for l in leaves:
            l.release()
        ctx.close()


if __name__ == "__main__":
    main()
