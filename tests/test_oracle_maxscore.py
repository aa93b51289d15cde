"""The oracle's dynamically pruned scorer (MaxScore family) against its
exhaustive scorer: safe pruning means identical top-k; only totalHits may become a lower bound.
CPU only -- this checks test infrastructure (the cpu_baseline leg of bench.py), not the product."""
import numpy as np
import pytest

from nrtsearch_amd import synth
from oracle import oracle


@pytest.fixture(scope="module")
def corpus_and_queries():
    q = synth.make_queries(24, 5, max_rank=2000)
    c = synth.build_corpus(300_000, q.ravel().tolist(), n_segments=4, delete_fraction=0.01)
    return c, q


@pytest.mark.parametrize("k,thr", [(10, 1000), (100, 1000), (1000, 1000), (100, 10), (100, 2**31 - 1)])
def test_pruned_topk_equals_exhaustive(corpus_and_queries, k, thr):
    c, q = corpus_and_queries
    pruned_something = False
    for terms in q.tolist():
        stats = {}
        a = oracle.search_bm25(c, terms, k, total_hits_threshold=thr)
        b = oracle.search_bm25(c, terms, k, total_hits_threshold=thr, maxscore=True, stats=stats)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert b[2] <= a[2]
        if b[2] < a[2]:
            pruned_something = True
            assert b[3], "a pruned count must carry GREATER_THAN_OR_EQUAL_TO"
        if thr == 2**31 - 1:   # ScoreMode.COMPLETE: nothing may be skipped (…Collector.java:68-70)
            assert b[2] == a[2] and not b[3]
            assert stats["postings_scored"] == sum(c.doc_freq[t] for t in terms)
    if thr <= 1000 and k <= 100:
        assert pruned_something


def test_pruned_paging_and_single_term(corpus_and_queries):
    c, q = corpus_and_queries
    terms = q[0].tolist()
    first = oracle.search_bm25(c, terms, 50, maxscore=True)
    after = (int(first[0][-1]), float(first[1][-1]))
    a = oracle.search_bm25(c, terms, 50, after=after)
    b = oracle.search_bm25(c, terms, 50, after=after, maxscore=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    one = [int(q[1][0])]
    a = oracle.search_bm25(c, one, 20)
    b = oracle.search_bm25(c, one, 20, maxscore=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_batch_driver_matches_single_calls(corpus_and_queries):
    c, q = corpus_and_queries
    pb = oracle.PreparedBatch(c, q.tolist(), 100)
    ex = pb.run(False, 2)
    pr = pb.run(True, 3)
    assert np.array_equal(ex[0], pr[0]) and np.array_equal(ex[1], pr[1])
    assert (pr[3] <= ex[3]).all() and pr[5] > 0
    for i in (0, 7, 23):
        d, s, total, gte = oracle.search_bm25(c, q[i].tolist(), 100)
        assert np.array_equal(d, ex[0][i][: ex[2][i]]) and np.array_equal(s, ex[1][i][: ex[2][i]])
        assert total == ex[3][i] and gte == bool(ex[4][i])
