"""Run by tests/test_planner_host.py in a subprocess with tests/mockhip preloaded (MOCKHIP_SYNC_US: what a stream synchronisation
"takes"): C threads in a closed loop through nrtgpu_search_bm25_coalesced, each sending its next query when the last one has come
back.  What is read: how many device batches the library formed for their queries (nrtgpu_stats.batches) and how long the loop
took -- the leader's rule (search.cpp): linger for company at most 150 us, leave at once when the cohort of the last batch is back.
Prints one line per C: "callers C calls N batches B seconds S"."""
import os, sys, threading, time, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nrtsearch_amd import api, synth, workload
faulthandler.dump_traceback_later(60, exit=True)
w = workload.Workload("coalescer cohort test", 150_000, 3, 50, 64, 2)
qr = synth.make_queries(64, w.n_terms, w.max_rank)
corpus = workload.build_shard_corpus(w, qr)
ctx = api.GpuContext(0, max_batch=64)
leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
queries = workload.boolean_queries(qr)
mgr = api.TopScoreDocCollectorManager(w.k)
PER = int(os.environ.get("CALLS_PER_CALLER", "30"))
for C in (1, 8, 24):
    ctx.reset_stats()
    errs = []
    def caller(t):
        try:
            for i in range(PER):
                r = sr.search_coalesced(queries[(t * PER + i) % len(queries)], mgr)
                assert len(r.scores) == 0      # the kernels did nothing
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=caller, args=(t,)) for t in range(C)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    assert not errs, errs
    st = ctx.stats()
    print(f"callers {C} calls {C * PER} batches {st['batches']} seconds {dt:.3f}", flush=True)
for g in leaves:
    g.release()
ctx.close()
print("done", flush=True)
