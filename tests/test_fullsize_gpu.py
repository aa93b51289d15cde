"""BASELINE.json's full C3 size (10M docs, 5-term disjunction, top-1000) on the device: oracle parity on a
few queries plus size-independent properties on a batch (ordering, idempotence, independence of how the
doc space is cut into work items, merge of per-segment searches == whole-index search).  Needs an MI355X."""
import numpy as np
import pytest

from nrtsearch_amd import _lib, api, synth, workload

pytestmark = pytest.mark.gpu
N_QUERIES = 64


@pytest.fixture(scope="module")
def c3():
    w = workload.C3
    qr = synth.make_queries(N_QUERIES, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr)
    ctx = api.GpuContext(0, max_batch=N_QUERIES)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    queries = workload.boolean_queries(qr)
    mgr = api.TopScoreDocCollectorManager(w.k)
    res = sr.search_batch(queries, [mgr] * N_QUERIES)
    yield dict(w=w, qr=qr, corpus=corpus, ctx=ctx, leaves=leaves, sr=sr, queries=queries, mgr=mgr, res=res)
    for l in leaves:
        l.release()
    ctx.close()


def _keys(td):
    return (td.scores.view(np.uint32).astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - td.docs.astype(np.uint64))


def test_full_size_oracle_parity(c3, oracle):
    for qi in (0, 17, 63):
        docs, scores, total, gte = oracle.search_bm25(c3["corpus"], c3["qr"][qi].tolist(), c3["w"].k)
        got = c3["res"][qi]
        assert got.docs.tolist() == docs.tolist()
        assert got.scores.view(np.uint32).tolist() == scores.view(np.uint32).tolist()
        assert got.relation_gte == gte
        assert (c3["w"].k < got.total_hits <= total) if gte else got.total_hits == total   # pruned: a lower bound


def test_full_size_ordering_and_counts(c3):
    ppq = workload.postings_per_query(c3["corpus"].doc_freq, c3["qr"])
    for qi, td in enumerate(c3["res"]):
        k = _keys(td)
        assert len(td.docs) == c3["w"].k                    # every query matches far more than k docs
        assert np.all(k[:-1] > k[1:])                       # (score desc, doc asc), no duplicate docs
        assert td.docs.min() >= 0 and td.docs.max() < c3["w"].n_docs
        assert c3["w"].k < td.total_hits <= ppq[qi] and td.relation_gte   # union of 5 posting lists, > threshold


def test_full_size_idempotent_and_chunking_invariant(c3):
    again = c3["sr"].search_batch(c3["queries"], [c3["mgr"]] * N_QUERIES)
    c2 = api.GpuContext(0, max_batch=N_QUERIES, target_items=4096, flags=_lib.NRTGPU_FLAG_NO_FIXED_POINT)
    leaves = [api.GpuSegment.from_data(c2, s) for s in c3["corpus"].segments]
    sr2 = api.GpuIndexSearcher(c2, leaves, api.IndexStatistics.from_corpus(c3["corpus"]))
    cut = sr2.search_batch(c3["queries"], [c3["mgr"]] * N_QUERIES)   # many items per query, fp64 accumulators
    assert c2.stats()["scan_items"] > 4 * N_QUERIES
    for a, b, c in zip(c3["res"], again, cut):
        for other in (b, c):
            assert a.docs.tolist() == other.docs.tolist()
            assert a.scores.view(np.uint32).tolist() == other.scores.view(np.uint32).tolist()
            assert a.relation_gte == other.relation_gte and (a.relation_gte or a.total_hits == other.total_hits)
    for l in leaves:
        l.release()
    c2.close()


def test_full_size_merge_of_segments_equals_whole(c3, oracle):
    # TopDocs.merge of per-leaf searches (index-global statistics) == the whole-index search
    stats = api.IndexStatistics.from_corpus(c3["corpus"])
    per_leaf = []
    for leaf in c3["leaves"]:
        sr = api.GpuIndexSearcher(c3["ctx"], [leaf], stats)
        per_leaf.append(sr.search_batch(c3["queries"][:8], [c3["mgr"]] * 8))
    for qi in range(8):
        docs, scores = oracle.topdocs_merge(c3["w"].k, [(r[qi].docs, r[qi].scores) for r in per_leaf])
        assert docs.tolist() == c3["res"][qi].docs.tolist()
        assert scores.view(np.uint32).tolist() == c3["res"][qi].scores.view(np.uint32).tolist()
        whole = c3["res"][qi]   # (pruned searches report lower bounds: the sum of the leaves' is one as well, not the same one)
        assert whole.relation_gte or sum(r[qi].total_hits for r in per_leaf) == whole.total_hits
