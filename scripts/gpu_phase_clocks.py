#!/usr/bin/env python3
"""WHERE does a wave's walk time go inside bm25_maxscore_kernel?  (GPU; a measurement build: -DNRT_MS_PHASE_CLOCKS, instrumented
kernel.)  One C3 batch; at each mark of the walk a wave waits for everything it has requested and books the cycles since the last
mark to a phase (maxscore.hip: NRT_PH_MARK).  Printed: each phase's share of the waves' walk time.
    python -c "from nrtsearch_amd import build; build.build(force=True, extra=['-DNRT_MS_PHASE_CLOCKS=1'], out='nrtsearch_amd/libnrtgpu_phase.so')"
    NRTGPU_LIB_PATH=nrtsearch_amd/libnrtgpu_phase.so python scripts/gpu_phase_clocks.py"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NRTGPU_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nrtsearch_amd", "libnrtgpu_dev.so"))   # instrumented kernels: the development library (include/nrtgpu_dev.h)

from nrtsearch_amd import _lib, api, synth, workload  # noqa: E402

PHASES = ["window head (cells, bitset clear, clause ranges)", "posting columns requested -> arrived (incl. the group's head)",
          "values (LDS tables) + first bound", "first-to-reach test-and-set (LDS)", "later clause: bound + records requested -> arrived",
          "later clause: rank + codes requested -> arrived", "later clause: values + sums", "later clause, sparse: cells + binary search + codes",
          "hits, live / mask words, candidates (incl. compactions)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--rounds", action="store_true", help="the library is a -DNRT_MS_COUNT_ROUNDS build: print its event counts per query instead")
    args = ap.parse_args()
    w = workload.C3
    w.n_docs = args.docs
    qr = synth.make_queries(args.batch * 2, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr, 1, 0)
    queries = workload.boolean_queries(qr)
    mgr = api.TopScoreDocCollectorManager(w.k)
    ctx = api.GpuContext(0, max_batch=args.batch, collect_timing=True, flags=_lib.NRTGPU_FLAG_PROFILE)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    B = args.batch
    pbs = [api.PreparedBatch(sr, queries[i * B:(i + 1) * B], [mgr] * B) for i in range(2)]
    pbs[0].run()
    for pb in pbs:
        ctx.reset_stats()
        pb.run()
        st = ctx.stats()
        prof = list(ctx.maxscore_profile().values())
        if args.rounds:
            nq = float(B)
            print(json.dumps({"maxscore_ms_hip_events": round(st["maxscore_ms"] / max(1, st["maxscore_launches"]), 3), "per_query": {
                "windows": prof[0] / nq, "instruction_groups": prof[2] / nq, "postings_streamed": prof[3] / nq, "postings_surviving": prof[4] / nq,
                "lookups": prof[6] / nq, "candidates": prof[7] / nq, "test_and_set_rounds": prof[11] / nq, "dense_clause_rounds": prof[12] / nq,
                "sparse_clause_rounds": prof[10] / nq, "binary_search_steps": prof[13] / nq, "candidate_rounds": prof[8] / nq}}), flush=True)
            continue
        ph = prof[:9]
        walk = prof[13] + prof[10]     # the waves' walk cycles incl. their meetings
        tot = sum(ph)
        rec = {"maxscore_ms_hip_events": round(st["maxscore_ms"] / max(1, st["maxscore_launches"]), 3),
               "wave_walk_cycles": walk, "booked": round(tot / walk, 3), "meeting_share_of_walk": round(prof[10] / walk, 3),
               "part_prologue_share": round(prof[12] / walk, 3), "idle_at_item_end_share": round(prof[11] / walk, 3),
               "phases": {PHASES[i]: round(ph[i] / tot, 3) for i in range(9)}}
        print(json.dumps(rec, indent=1), flush=True)
    for l in leaves:
        l.release()
    ctx.close()


if __name__ == "__main__":
    main()
