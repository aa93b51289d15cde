#!/usr/bin/env python3
"""WHEN do the workgroups of a MaxScore launch run?  (GPU; the instrumented kernel: NRTGPU_FLAG_PROFILE.)  One C3 batch (1024
queries, 10 M docs), a few launches; from nrtgpu_get_maxscore_item_walls: the launch's makespan against its balanced load
(sum of the workgroups' busy time / CUs), the finish-time histogram, and what the helper workgroups did (plan.h: DHelp).
    NRTGPU_MS_HELPERS=0 python scripts/gpu_makespan.py      # no helpers: the tail as round 3 had it
    python scripts/gpu_makespan.py                           # default"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NRTGPU_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nrtsearch_amd", "libnrtgpu_dev.so"))   # instrumented kernels: the development library (include/nrtgpu_dev.h)

import numpy as np  # noqa: E402

from nrtsearch_amd import _lib, api, synth, workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--cus", type=int, default=256)
    args = ap.parse_args()
    w = workload.C3
    w.n_docs = args.docs
    qr = synth.make_queries(args.batch * args.batches, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr, 1, 0)
    queries = workload.boolean_queries(qr)
    mgr = api.TopScoreDocCollectorManager(w.k)
    ctx = api.GpuContext(0, max_batch=args.batch, collect_timing=True, flags=_lib.NRTGPU_FLAG_PROFILE)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    B = args.batch
    pbs = [api.PreparedBatch(sr, queries[i * B:(i + 1) * B], [mgr] * B) for i in range(args.batches)]
    pbs[0].run()
    for bi, pb in enumerate(pbs):
        ctx.reset_stats()
        pb.run()
        st = ctx.stats()
        walls, n_items = ctx.maxscore_item_walls()
        walls[:, 2] &= np.uint64(0xFFFFFFFF)    # (the upper halves: the item's flags / its query)
        walls[:, 6] &= np.uint64(0xFFFFFFFF)
        used = walls[:, 1] > 0
        t0 = int(walls[used, 4].min())                           # the first workgroup's first instruction
        start = (walls[:, 0].astype(np.int64) - t0) / 100.0      # us: the piece's prologue begins (role chosen)
        end = (walls[:, 1].astype(np.int64) - t0) / 100.0
        entry = (walls[:, 4].astype(np.int64) - t0) / 100.0      # the workgroup's round begins
        busy = np.where(used, end - start, 0.0)
        owners = np.arange(len(walls)) < n_items
        span = float(end[used].max())
        balanced = float(busy.sum()) / args.cus
        h_used = used & ~owners
        fin = np.sort(end[used & owners])
        # per CU: the idle time between one piece's end and the next one's start on the same CU (dispatch of a fresh workgroup, or
        # a persistent workgroup's next round), and from a round's begin to its piece's start (choosing what to do)
        gaps = []
        cu = walls[:, 5]
        for c in np.unique(cu[used]):
            idx = np.where(used & (cu == c))[0]
            idx = idx[np.argsort(start[idx])]
            gaps.extend((start[idx[1:]] - end[idx[:-1]]).tolist())
        gaps = np.asarray(gaps) if gaps else np.zeros(1)
        choose = (start - entry)[used]
        rec = {
            "batch": bi, "maxscore_ms_hip_events": round(st["maxscore_ms"] / max(1, st["maxscore_launches"]), 3),
            "items": int(n_items), "helper_slots": int((~owners).sum()), "helper_sessions": int(h_used.sum()),
            "helper_windows": int(walls[h_used, 3].sum()), "owner_windows": int(walls[used & owners, 3].sum()),
            "span_us": round(span, 1), "balanced_us": round(balanced, 1), "makespan_over_balanced": round(span / balanced, 3),
            "cus_seen": int(len(np.unique(cu[used]))), "workgroups": int(len(np.unique(walls[used, 7]))), "max_round": int(walls[used, 6].max()),
            "gap_between_pieces_on_a_cu_us": {"n": int(len(gaps)), "mean": round(float(gaps.mean()), 1), "p50": round(float(np.median(gaps)), 1),
                                              "p90": round(float(np.percentile(gaps, 90)), 1), "max": round(float(gaps.max()), 1),
                                              "sum_over_cus": round(float(gaps.sum()) / args.cus, 1)},
            "choosing_us": {"mean": round(float(choose.mean()), 2), "p90": round(float(np.percentile(choose, 90)), 2), "max": round(float(choose.max()), 2)},
            "owner_busy_us": {"mean": round(float(busy[owners].mean()), 1), "p50": round(float(np.median(busy[owners])), 1),
                              "p90": round(float(np.percentile(busy[owners], 90)), 1), "max": round(float(busy[owners].max()), 1)},
            "owner_finish_us_deciles": [round(float(x), 1) for x in np.percentile(fin, [10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 99, 100])],
            "helper_busy_us": ({"mean": round(float(busy[h_used].mean()), 1), "max": round(float(busy[h_used].max()), 1),
                                "first_start_us": round(float(start[h_used].min()), 1)} if h_used.any() else None),
            # CUs busy over time: how many pieces are running at 10 points of the span
            "running_at_tenths": [int(((start <= span * f / 10) & (end > span * f / 10) & used).sum()) for f in range(1, 10)],
        }
        print(json.dumps(rec), flush=True)
    prof = ctx.maxscore_profile()
    nq = args.batch
    print(json.dumps({"last_batch_profile_per_query": {k: round(v / nq, 1) for k, v in prof.items()}}), flush=True)
    for l in leaves:
        l.release()
    ctx.close()


if __name__ == "__main__":
    main()
