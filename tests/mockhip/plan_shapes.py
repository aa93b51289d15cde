"""Run by tests/test_planner_host.py in a subprocess with tests/mockhip preloaded (kernels do nothing): what the PLANNER decides for
the shapes whose score is not one sum -- a DisjunctionMaxQuery with a tie breaker, MUST next to SHOULD clauses -- and for the
argument errors around them.  One line per case on stdout; the kernels' results are not looked at."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nrtsearch_amd import _lib, api, synth   # noqa: E402

INT_MAX = 2**31 - 1
ranks = [1, 2, 3, 6, 15, 50, 400, 3000]
corpus = synth.build_corpus(260_000, ranks, n_segments=3)
ctx = api.GpuContext(device_id=0, max_batch=16)
leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
tq = lambda t: api.TermQuery(0, t)   # noqa: E731


def verdict(name, query, mgr):
    """nrtgpu_query_supported through the mirror: `ok`, or the library's reason."""
    try:
        ok = sr.supported(query, mgr)
        print(name, "ok" if ok else "unsupported: " + _lib.load().nrtgpu_last_error().decode("utf-8", "replace")[:90], flush=True)
    except api.UnsupportedQuery as e:
        print(name, "mirror:", str(e)[:90], flush=True)
    except api.NrtGpuError as e:
        print(name, "error:", str(e)[:120], flush=True)


top = api.TopScoreDocCollectorManager(100)
complete = api.TopScoreDocCollectorManager(100, None, INT_MAX)
verdict("tie_small", api.DisjunctionMaxQuery((tq(15), tq(400), tq(3000)), 0.3), top)
verdict("tie_complete_small", api.DisjunctionMaxQuery((tq(15), tq(400), tq(3000)), 0.3), complete)          # exact mode on the MaxScore route
verdict("tie_complete_large", api.DisjunctionMaxQuery(tuple(tq(t) for t in (1, 2, 3, 6, 15, 50)), 0.3), complete)   # would need the exhaustive scan
verdict("tie_nine_clauses", api.DisjunctionMaxQuery(tuple(tq(t) for t in ranks + [1]), 0.3), top)            # > 8 clauses: exhaustive scan
verdict("tie_zero_nine_clauses", api.DisjunctionMaxQuery(tuple(tq(t) for t in ranks + [1]), 0.0), top)       # tie breaker 0: one accumulator, any route
verdict("tie_out_of_range", api.DisjunctionMaxQuery((tq(15), tq(400)), 1.5), top)
verdict("must_should", api.BooleanQuery((tq(1), tq(400)), must=(tq(15),)), top)
verdict("must_should_complete_large", api.BooleanQuery((tq(1), tq(2), tq(3)), must=(tq(6),)), complete)
verdict("must_should_with_msm", api.BooleanQuery((tq(1), tq(400)), 1, must=(tq(15),)), top)
verdict("all_must", api.BooleanQuery(must=(tq(1), tq(15), tq(400))), top)
verdict("must_term_nowhere", api.BooleanQuery((tq(1),), must=(tq(9999),)), top)                                 # plans to nothing: fine

# the C ABI's own argument checks (the mirror never builds these)
L = _lib.load()


def raw(name, occur=(0, 0), dismax=0, tb=0.0, msm=0):
    m = sr._marshal([api.BooleanQuery((tq(15), tq(400)))], [top])
    q = m.queries[0]
    for i, o in enumerate(occur):
        q.terms[i].occur = o
    q.disjunction_max, q.tie_breaker, q.min_should_match = dismax, tb, msm
    rc = L.nrtgpu_query_supported(ctx._h, sr._segs, len(sr.leaves), C.byref(q))
    print(name, "ok" if rc == 0 else f"rc {rc}: " + L.nrtgpu_last_error().decode("utf-8", "replace")[:90], flush=True)


raw("raw_plain")
raw("raw_occur_2", occur=(2, 0))
raw("raw_tie_without_dismax", tb=0.5)
raw("raw_tie_negative", dismax=1, tb=-0.1)
raw("raw_must_in_dismax", occur=(1, 0), dismax=1)
raw("raw_must_with_msm", occur=(1, 0), msm=1)
raw("raw_all_must", occur=(1, 1))
print("done", flush=True)
