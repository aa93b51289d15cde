"""Config C5's shape at test size -- BM25 recall + exact-vector rescore through the device, against the
oracle's restatement of QueryRescore (src/main/java/com/yelp/nrtsearch/server/rescore/QueryRescore.java:40-57)
-- and the thread-safety the C ABI promises to the SEARCH pool (concurrent batch calls on one context)."""
import threading

import numpy as np
import pytest

from nrtsearch_amd import api, synth
from tests.test_parity_gpu import total_ok

pytestmark = pytest.mark.gpu
VEC_FIELD, DIM = 7, 64


def _hybrid_index():
    rng = np.random.default_rng(11)
    ranks = [1, 3, 8, 20, 60, 300, 2000]
    corpus = synth.build_corpus(60_000, ranks, n_segments=3)
    ctx = api.GpuContext(0, max_batch=256)
    leaves, vecs = [], []
    for seg in corpus.segments:
        g = api.GpuSegment(ctx, seg.max_doc, seg.doc_base)
        g.add_field_norms(0, seg.norms)
        g.add_terms(0, seg.term_ids, seg.offsets, seg.docids, seg.freqs)
        v = rng.standard_normal((seg.max_doc, DIM)).astype(np.float32)
        g.add_vectors(VEC_FIELD, v)
        g.seal()
        leaves.append(g)
        vecs.append(v)
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    return dict(corpus=corpus, ctx=ctx, leaves=leaves, vecs=vecs, sr=sr, rng=rng, ranks=ranks)


def _hybrid_close(h):
    for g in h["leaves"]:
        g.release()
    h["ctx"].close()


@pytest.fixture(scope="module")
def hybrid():
    h = _hybrid_index()
    yield h
    _hybrid_close(h)


def _bq(terms):
    return api.BooleanQuery(tuple(api.TermQuery(0, int(t)) for t in terms))


def test_fused_tail_equals_two_calls(hybrid, oracle):
    """nrtgpu_search_hybrid_batch == search + rescore_vectors per query, bit for bit (same arithmetic, the
    hits just never leave the device), for all similarities; then against the CPU restatement."""
    sr, rng = hybrid["sr"], hybrid["rng"]
    term_sets = [[1, 20, 300], [3, 8, 60, 2000], [2000], [1, 3, 8, 20, 60], [300, 2000]]
    for sim, recall, window, qw, rw in [("cosine", 1000, 100, 1.0, 2.5), ("l2_norm", 200, 200, 0.5, 4.0),
                                        ("max_inner_product", 64, 10, 1.0, 1.0), ("dot_product", 1000, 1000, 0.0, 1.0)]:
        qs = [_bq(t) for t in term_sets]
        mg = [api.TopScoreDocCollectorManager(recall)] * len(qs)
        qv = rng.standard_normal((len(qs), DIM)).astype(np.float32)
        if sim == "dot_product":
            qv /= np.linalg.norm(qv, axis=1, keepdims=True)
        fused = sr.search_hybrid_batch(qs, mg, VEC_FIELD, sim, qv, window, qw, rw)
        for i, q in enumerate(qs):
            first = sr.search(q, mg[i])
            two = sr.rescore_vectors(first, VEC_FIELD, sim, qv[i], window, qw, rw)
            assert fused[i].docs.tolist() == two.docs.tolist(), (sim, i)
            assert fused[i].scores.view(np.uint32).tolist() == two.scores.view(np.uint32).tolist(), (sim, i)
            assert fused[i].total_hits == first.total_hits and fused[i].relation_gte == first.relation_gte
    bases = [s.doc_base for s in hybrid["corpus"].segments]
    terms = [3, 8, 60, 2000]
    q = rng.standard_normal(DIM).astype(np.float32)
    got = sr.search_hybrid_batch([_bq(terms)], [api.TopScoreDocCollectorManager(500)], VEC_FIELD, "cosine", q[None, :], 50, 1.0, 3.0)[0]
    edocs, escores, etotal, _ = oracle.search_bm25(hybrid["corpus"], terms, 500)
    exp = []
    for doc, f in zip(edocs.tolist(), escores.tolist()):
        si = max(i for i, b in enumerate(bases) if b <= doc)
        second = float(oracle.vector_score(0, q, hybrid["vecs"][si][doc - bases[si]]))
        exp.append((float(oracle.rescore_combine(f, True, second, 1.0, 3.0)), doc))
    exp.sort(key=lambda t: (-t[0], t[1]))
    assert 1000 < got.total_hits <= etotal and got.relation_gte and len(got.docs) == 50   # (pruned first pass: a lower bound)
    assert np.allclose(got.scores, [s for s, _ in exp[:50]], rtol=1e-5, atol=1e-6)
    assert len(set(got.docs.tolist()) & set(d for _, d in exp[:50])) >= 49


def test_fused_tail_sparse_vectors_and_errors():
    """Leaves where only some docs have a vector (ord -> doc map) and one leaf without the field."""
    rng = np.random.default_rng(5)
    corpus = synth.build_corpus(30_000, [2, 9, 70], n_segments=3)
    ctx = api.GpuContext(0, max_batch=64)
    leaves = []
    try:
        for si, seg in enumerate(corpus.segments):
            g = api.GpuSegment(ctx, seg.max_doc, seg.doc_base)
            g.add_field_norms(0, seg.norms)
            g.add_terms(0, seg.term_ids, seg.offsets, seg.docids, seg.freqs)
            if si != 1:   # leaf 1 has no vectors: its hits keep queryWeight * first
                have = np.flatnonzero(rng.random(seg.max_doc) < 0.6).astype(np.int32)
                g.add_vectors(VEC_FIELD, rng.standard_normal((len(have), DIM)).astype(np.float32), have)
            g.seal()
            leaves.append(g)
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        qs = [_bq([2, 70]), _bq([9])]
        mg = [api.TopScoreDocCollectorManager(300)] * 2
        qv = rng.standard_normal((2, DIM)).astype(np.float32)
        fused = sr.search_hybrid_batch(qs, mg, VEC_FIELD, "cosine", qv, 40, 1.0, 2.0)
        for i in range(2):
            two = sr.rescore_vectors(sr.search(qs[i], mg[i]), VEC_FIELD, "cosine", qv[i], 40, 1.0, 2.0)
            assert fused[i].docs.tolist() == two.docs.tolist()
            assert fused[i].scores.view(np.uint32).tolist() == two.scores.view(np.uint32).tolist()
        with pytest.raises(Exception):   # negative weights would break the key order: refused, the caller runs two calls
            sr.search_hybrid_batch(qs, mg, VEC_FIELD, "cosine", qv, 40, -1.0, 2.0)
    finally:
        for g in leaves:
            g.release()
        ctx.close()


def test_bm25_recall_then_vector_rescore(hybrid, oracle):
    bases = [s.doc_base for s in hybrid["corpus"].segments]
    for terms, recall, window in [([1, 20, 300], 200, 50), ([3, 8, 60, 2000], 1000, 100)]:
        q = hybrid["rng"].standard_normal(DIM).astype(np.float32)
        first = hybrid["sr"].search(_bq(terms), api.TopScoreDocCollectorManager(recall))
        edocs, escores, _, _ = oracle.search_bm25(hybrid["corpus"], terms, recall)
        assert first.docs.tolist() == edocs.tolist()                       # the recall stage is bit-exact
        assert first.scores.view(np.uint32).tolist() == escores.view(np.uint32).tolist()
        got = hybrid["sr"].rescore_vectors(first, VEC_FIELD, "cosine", q, window=window, query_weight=1.0, rescore_weight=2.5)
        exp = []
        for doc, f in zip(edocs.tolist(), escores.tolist()):
            si = max(i for i, b in enumerate(bases) if b <= doc)
            second = float(oracle.vector_score(0, q, hybrid["vecs"][si][doc - bases[si]]))
            exp.append((float(oracle.rescore_combine(f, True, second, 1.0, 2.5)), doc))
        exp.sort(key=lambda t: (-t[0], t[1]))
        assert len(got.docs) == window
        assert np.allclose(got.scores, [s for s, _ in exp[:window]], rtol=1e-5, atol=1e-6)
        assert len(set(got.docs.tolist()) & set(d for _, d in exp[:window])) >= window - 1   # modulo one near-tie


def test_concurrent_batch_calls_from_many_threads(hybrid, oracle):
    term_sets = [[1, 20], [3, 60, 300], [8, 2000], [1, 3, 8, 20, 60], [300], [20, 2000, 1], [60, 8], [3, 1, 300, 2000]]
    expected = [oracle.search_bm25(hybrid["corpus"], t, 100) for t in term_sets]
    errors = []

    def worker(tix):
        try:
            for it in range(6):
                order = [(tix + it + j) % len(term_sets) for j in range(len(term_sets))]
                res = hybrid["sr"].search_batch([_bq(term_sets[i]) for i in order], [api.TopScoreDocCollectorManager(100)] * len(order))
                for i, r in zip(order, res):
                    ed, es, et, eg = expected[i]
                    if (r.docs.tolist() != ed.tolist() or r.scores.view(np.uint32).tolist() != es.view(np.uint32).tolist()
                            or not total_ok(r, et, eg, 100, 1000)):
                        errors.append((tix, it, i))
        except Exception as e:  # noqa: BLE001
            errors.append((tix, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_coalesced_single_searches_from_many_threads(hybrid, oracle):
    # what the gRPC / SEARCH pool threads do: one blocking search each; the library merges them into batches
    term_sets = [[1, 20], [3, 60, 300], [8, 2000], [1, 3, 8, 20, 60], [300], [20, 2000, 1], [60, 8], [3, 1, 300, 2000]]
    expected = [oracle.search_bm25(hybrid["corpus"], t, 100) for t in term_sets]
    errors = []
    hybrid["ctx"].reset_stats()

    def worker(tix):
        try:
            for it in range(25):
                i = (tix * 7 + it) % len(term_sets)
                r = hybrid["sr"].search_coalesced(_bq(term_sets[i]), api.TopScoreDocCollectorManager(100))
                ed, es, et, eg = expected[i]
                if (r.docs.tolist() != ed.tolist() or r.scores.view(np.uint32).tolist() != es.view(np.uint32).tolist()
                        or not total_ok(r, et, eg, 100, 1000)):
                    errors.append((tix, it, i))
        except Exception as e:  # noqa: BLE001
            errors.append((tix, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(32)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    st = hybrid["ctx"].stats()
    assert st["queries"] == 32 * 25
    assert st["batches"] <= st["queries"]         # (how many batches the free-running callers formed is the host's business)
    # an invalid request fails alone, with its own message
    with pytest.raises(Exception):
        hybrid["sr"].search_coalesced(_bq([1, 20]), api.TopScoreDocCollectorManager(0))


def test_coalesced_single_searches_form_one_batch_by_construction(dev_lib, oracle):
    """Merged into batches, BY CONSTRUCTION: 32 callers parked behind the test hook of the development library
    (include/nrtgpu_dev.h: nrtgpu_debug_hold_coalescers), released together -> the same leaves, fewer than max_batch = 256 -> ONE
    batch, whatever the host's speed."""
    import time
    term_sets = [[1, 20], [3, 60, 300], [8, 2000], [1, 3, 8, 20, 60], [300], [20, 2000, 1], [60, 8], [3, 1, 300, 2000]]
    hybrid = _hybrid_index()
    try:
        expected = [oracle.search_bm25(hybrid["corpus"], t, 100) for t in term_sets]
        errors = []
        ctx = hybrid["ctx"]
        ctx.reset_stats()
        ctx.debug_hold_coalescers(True)

        def one(tix):
            try:
                i = tix % len(term_sets)
                r = hybrid["sr"].search_coalesced(_bq(term_sets[i]), api.TopScoreDocCollectorManager(100))
                ed, es, et, eg = expected[i]
                if r.docs.tolist() != ed.tolist() or r.scores.view(np.uint32).tolist() != es.view(np.uint32).tolist() or not total_ok(r, et, eg, 100, 1000):
                    errors.append((tix, i))
            except Exception as e:  # noqa: BLE001
                errors.append((tix, repr(e)))

        threads = [threading.Thread(target=one, args=(t,)) for t in range(32)]
        try:
            for t in threads:
                t.start()
            t_end = time.monotonic() + 60.0
            while ctx.debug_coalescer_pending(0) < 32 and time.monotonic() < t_end and not errors:
                time.sleep(0.001)
            parked = ctx.debug_coalescer_pending(0)
        finally:
            ctx.debug_hold_coalescers(False)
        for t in threads:
            t.join()
        assert not errors, errors[:5]
        assert parked == 32
        st = ctx.stats()
        assert st["queries"] == 32 and st["batches"] == 1, st
    finally:
        _hybrid_close(hybrid)


def test_fused_hybrid_under_speculative_thresholds_equals_the_unspeculated_answer(dev_lib, monkeypatch):
    """The fused hybrid's first pass runs under speculative thresholds too (plan.h: kHitsSpecInvalid): the merge's tags arrive with
    the results and a query whose guess failed -- its recall set may lack docs -- is run again, first pass and tail.  On an index
    whose live docs all sit in the first third of the docid range (guesses fail there: tests/test_maxscore_gpu.py) the fused
    answer with speculation must be the fused answer without, docids and score bits, and the counters must show the re-runs."""
    monkeypatch.setenv("NRTGPU_MS_SCATTER", "0")   # (development library: windows in docid order -- where this index defeats the guesses)
    rng = np.random.default_rng(5)
    ranks = [1, 2, 5, 9, 20, 60, 150, 400]
    corpus = synth.build_corpus(3_200_000, ranks, n_segments=1)
    seg = corpus.segments[0]
    dim = 16
    ctx = api.GpuContext(0, max_batch=64)
    g = api.GpuSegment(ctx, seg.max_doc, seg.doc_base)
    g.add_field_norms(0, seg.norms)
    g.add_terms(0, seg.term_ids, seg.offsets, seg.docids, seg.freqs)
    g.add_vectors(VEC_FIELD, rng.standard_normal((seg.max_doc, dim)).astype(np.float32))
    g.seal()
    live = np.zeros((seg.max_doc + 63) // 64, dtype=np.uint64)
    live[: int(seg.max_doc * 0.3) // 64] = np.uint64(0xFFFFFFFFFFFFFFFF)
    g.set_live_docs(live)
    sr = api.GpuIndexSearcher(ctx, [g], api.IndexStatistics.from_corpus(corpus))
    try:
        qs = [_bq(t) for t in ([1, 5, 20, 150, 400], [2, 9, 60], [1, 2, 5, 9, 20, 60, 150, 400], [5, 400], [9, 20, 150])]
        mg = [api.TopScoreDocCollectorManager(1000)] * len(qs)
        qv = rng.standard_normal((len(qs), dim)).astype(np.float32)
        ctx.set_speculation(5.0)
        fused = sr.search_hybrid_batch(qs, mg, VEC_FIELD, "cosine", qv, 100, 1.0, 2.0)
        c = ctx.spec_counters()
        assert c["queries"] == len(qs) and c["reruns"] >= 2, c
        ctx.set_speculation(0.0)
        plain = sr.search_hybrid_batch(qs, mg, VEC_FIELD, "cosine", qv, 100, 1.0, 2.0)
        assert ctx.spec_counters()["queries"] == 0
        for a, b in zip(fused, plain):
            assert a.docs.tolist() == b.docs.tolist() and a.scores.view(np.uint32).tolist() == b.scores.view(np.uint32).tolist()
            assert len(a.docs) == 100 and a.relation_gte == b.relation_gte   # (the counts are lower bounds: what each run happened to evaluate)
    finally:
        g.release()
        ctx.close()
