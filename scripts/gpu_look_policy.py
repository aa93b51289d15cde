#!/usr/bin/env python3
"""Which doc -> posting lookup structure for which terms, and how much HBM for it (segment.cpp: build_term_aux; plan.h: kLook*).
One process, the DEVELOPMENT library (it reads NRTGPU_LOOK_POLICY at every seal): the C3 corpus is built once, then for every
(policy, budget) a fresh context uploads it, the first queries are checked against the oracle's bits (first configuration) or
against the first configuration's answers (the others), and the MaxScore kernel is timed over 1024-query batches with HIP events
(nrtgpu_config.collect_timing).  One line per configuration: resident bytes, kernel ms per launch, queries/s of the batch call."""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NRTGPU_LIB_PATH", os.path.join(ROOT, "nrtsearch_amd", "libnrtgpu_dev.so"))

import numpy as np  # noqa: E402

from nrtsearch_amd import api, synth, workload  # noqa: E402


def log(*a):
    print(*a, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--oracle-queries", type=int, default=4)
    ap.add_argument("--packed", action="store_true")
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--profile", action="store_true", help="instrumented kernels (development library): event counts per query")
    ap.add_argument("--variant", default="iid", help="corpus variant: iid | clustered | sorted (synth.corpus_variant_arrays)")
    ap.add_argument("--configs", default="map@32,cells|150;bits:2048|100000;nib:2048|100000;map:2048|100000;cells|100000;|0",
                    help="';'-separated policy|budget_pct[|ENV=VAL,...] (budget 0 = the library's default, -1 = none)")
    args = ap.parse_args()
    w = {"C3": workload.C3, "C2": workload.C2}[args.workload]
    w.n_docs = args.docs if args.workload == "C3" else w.n_docs
    qr = synth.make_queries(args.queries, w.n_terms, w.max_rank)
    t0 = time.time()
    corpus = workload.build_shard_corpus(w, qr, 1, 0, variant=args.variant)
    log(json.dumps({"event": "corpus", "variant": args.variant, "docs": w.n_docs, "postings": corpus.total_postings, "build_s": round(time.time() - t0, 1)}))
    queries = workload.boolean_queries(qr)
    mgr = api.TopScoreDocCollectorManager(w.k)
    B = args.batch
    nb = args.queries // B
    ref = None
    for ci, cfg in enumerate(args.configs.split(";")):
        parts = cfg.split("|")
        policy, pct = parts[0], parts[1]
        os.environ["NRTGPU_LOOK_POLICY"] = policy
        extra_env = dict(kv.split("=") for kv in parts[2].split(",")) if len(parts) > 2 and parts[2] else {}
        for k_, v_ in extra_env.items():   # (development library: experiment knobs, read at every call)
            os.environ[k_] = v_
        flags = (api._lib.NRTGPU_FLAG_PACKED_POSTINGS if args.packed else 0) | (api._lib.NRTGPU_FLAG_PROFILE if args.profile else 0)
        ctx = api.GpuContext(0, max_batch=B, collect_timing=True, flags=flags, lookup_budget_pct=int(pct))
        if "SPEC" in extra_env:   # (not an environment variable: the speculation margin of this context, 0 = off)
            ctx.set_speculation(float(extra_env["SPEC"]))
        t1 = time.time()
        leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
        upload_s = time.time() - t1
        dev_bytes = sum(l.device_bytes for l in leaves)
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        pbs = [api.PreparedBatch(sr, queries[i * B:(i + 1) * B], [mgr] * B) for i in range(nb)]
        for pb in pbs:
            pb.run()   # warm: term tables resident, answers for the checks
        bad = 0
        if ci == 0 and args.oracle_queries:
            from oracle import oracle
            for qi in range(args.oracle_queries):
                d, s_, tot, gte = oracle.search_bm25(corpus, qr[qi].tolist(), w.k)
                td = pbs[0].topdocs(qi)
                bad += not (td.docs.tolist() == d.tolist() and td.scores.view(np.uint32).tolist() == s_.view(np.uint32).tolist() and td.relation_gte == gte)
        sums = []
        for pb in pbs:
            h = 0
            for qi in range(B):
                o = pb._outs[qi]
                h = zlib.crc32(pb.docs[qi, :o.n_hits].tobytes(), h)
                h = zlib.crc32(pb.scores[qi, :o.n_hits].tobytes(), h)
            sums.append(h)
        if ref is None:
            ref = sums
        same = sums == ref
        ctx.reset_stats()
        t2 = time.perf_counter()
        for i in range(args.steps):
            pbs[i % nb].run()
        dt = time.perf_counter() - t2
        st = ctx.stats()
        L = max(1, st["maxscore_launches"])
        prof = None
        if args.profile:
            prof = {k_: round(v_ / max(1, st["queries"]), 1) for k_, v_ in ctx.maxscore_profile().items()}
        for k_ in extra_env:
            os.environ.pop(k_, None)
        log(json.dumps({"event": "config", "policy": policy, "budget_pct": int(pct), "env": extra_env, "device_GB": round(dev_bytes / 1e9, 3), "upload_s": round(upload_s, 1),
                        "kernel_ms": round(st["maxscore_ms"] / L, 4), "launches": L, "step_ms": round(dt / args.steps * 1e3, 3),
                        "qps": round(args.steps * B / dt, 0), "oracle_mismatches": bad if ci == 0 else None, "answers_equal_first_config": same,
                        "spec": ctx.spec_counters(), "profile_per_query": prof}))
        for l in leaves:
            l.release()
        ctx.close()


if __name__ == "__main__":
    main()
