"""Run by tests/test_planner_host.py in a subprocess with tests/mockhip preloaded (kernels do nothing) and NRTGPU_PLAN_TRACE=1:
uploads a small C3-shaped corpus and plans / "searches" two batches; the plan trace on stderr is what the test reads."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nrtsearch_amd import api, synth, workload   # noqa: E402

w = workload.C3
w.n_docs = int(os.environ.get("DOCS", "300000"))
B = int(os.environ.get("B", "256"))
qranks = synth.make_queries(B * 2, w.n_terms, w.max_rank)
corpus = workload.build_shard_corpus(w, qranks, 1, 0)
ctx = api.GpuContext(device_id=0, max_batch=B)
leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
searcher = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
queries = workload.boolean_queries(qranks)
mgr = api.TopScoreDocCollectorManager(w.k)
for i in range(0, len(queries), B):
    res = searcher.search_batch(queries[i: i + B], [mgr] * B)
    assert len(res) == B and all(len(t.scores) == 0 for t in res)   # the kernels did nothing
    # the key the planner orders the MaxScore items by, recomputed here: postings of the query's two heaviest (rarest) clauses
    keys = []
    for q in range(i, i + B):
        df = sorted(int(corpus.doc_freq[int(r)]) for r in qranks[q])
        keys.append(df[0] + df[1])
    print("KEYS", " ".join(str(k) for k in keys), flush=True)
print("done", flush=True)
