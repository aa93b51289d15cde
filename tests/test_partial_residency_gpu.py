"""PARTIAL RESIDENCY through the C ABI ("non-resident segments are searched by Lucene and merged with TopDocs.merge").
Under NRT refresh (ShardState.java:506-527: a new small segment every second) a searcher usually holds a few leaves that are not
resident on the device yet.  What a GpuIndexSearcher does then (java/.../GpuIndexSearcher.java: search): the searcher's SLICES
whose leaves are all resident go to the device in ONE call over their leaves -- index-global statistics, the whole searcher's
slices (nrtgpu_set_thread_slices) -- every other slice runs through Lucene's own collector, and the per-slice results are reduced
as the reference reduces them: TopDocs.merge, totalHits summed, GREATER_THAN_OR_EQUAL_TO if any part's is
(LazyQueueTopScoreDocCollectorManager.java:137-144).  Here the oracle stands in for Lucene on the cold slices, and the reduced
answer must be the whole-index answer: docids, ranks, score bits, relation; totalHits exact where the relation is EQUAL_TO, a
lower bound above the threshold where the device pruned.  Needs a real MI355X."""
import numpy as np
import pytest

from nrtsearch_amd import api, synth

pytestmark = pytest.mark.gpu


def bq(terms):
    cl = tuple(api.TermQuery(0, int(t)) for t in terms)
    return cl[0] if len(cl) == 1 else api.BooleanQuery(cl)


def reduce_parts(oracle, k, parts):
    """CollectorManager.reduce of the reference over per-slice (or per-call) results (docs, scores, total, gte)."""
    docs, scores = oracle.topdocs_merge(k, [(np.asarray(p[0], dtype=np.int32), np.asarray(p[1], dtype=np.float32)) for p in parts])
    return docs, scores, int(sum(p[2] for p in parts)), bool(any(p[3] for p in parts))


@pytest.mark.parametrize("slicing", [(250_000, 5), (40_000, 2)])
def test_resident_slices_on_the_device_cold_slices_on_the_cpu_equal_the_whole_index(slicing, oracle):
    ranks = [2, 7, 30, 100, 300, 1000, 3000]
    corpus = synth.build_corpus(900_000, ranks, n_segments=9, delete_fraction=0.01)
    groups = oracle.corpus_slices(corpus, slicing)       # the WHOLE searcher's slices (MyIndexSearcher.slices)
    assert len(groups) >= 3
    ctx = api.GpuContext(device_id=0, max_batch=64)
    ctx.set_slicing(*slicing)
    slice_of = {li: s for s, g in enumerate(groups) for li in g}
    # the youngest (last, smallest) leaf is cold, and with it its whole slice; one more slice in the middle as well
    cold_slices = {slice_of[len(corpus.segments) - 1], sorted(set(range(len(groups))) - {slice_of[len(corpus.segments) - 1]})[len(groups) // 2 - 1]}
    hot = [li for li in range(len(corpus.segments)) if slice_of[li] not in cold_slices]
    assert hot and len(hot) < len(corpus.segments)
    leaves = [api.GpuSegment.from_data(ctx, corpus.segments[li]) for li in hot]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))     # index-global statistics
    try:
        seen = set()
        for terms in ([100], [3000], [300, 1000], [2, 7], [7, 100, 3000], [30, 300], [2, 30, 100, 1000, 3000]):
            for k, thr in ((10, 1000), (100, 200), (1000, 1000), (10, 2**31 - 1), (50, 20)):
                api.GpuContext.set_thread_slices([slice_of[li] for li in hot])
                try:
                    dev = sr.search(bq(terms), api.TopScoreDocCollectorManager(k, total_hits_threshold=thr))
                finally:
                    api.GpuContext.set_thread_slices(None)
                parts = [(dev.docs, dev.scores, dev.total_hits, dev.relation_gte)]
                for s in sorted(cold_slices):                 # "Lucene": one collector per cold slice, index-global statistics
                    parts.append(oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr, segments=groups[s]))
                docs, scores, total, gte = reduce_parts(oracle, k, parts)
                edocs, escores, etotal, egte = oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr, slicing=slicing)
                name = f"{terms}_k{k}_thr{thr}"
                assert docs.tolist() == edocs.tolist(), f"{name}: docids / ranks"
                assert scores.view(np.uint32).tolist() == escores.view(np.uint32).tolist(), f"{name}: score bits"
                assert gte == egte, f"{name}: relation"
                if egte:
                    assert max(thr, k) < total <= etotal, f"{name}: {total} not in ({max(thr, k)}, {etotal}]"
                else:
                    assert total == etotal, f"{name}: totalHits {total} != {etotal}"
                seen.add(egte)
        assert seen == {True, False}
    finally:
        for g in leaves:
            g.release()
        ctx.close()


def test_virtual_shards_need_the_callers_slices(oracle):
    """With virtual shards the leaves are dealt to shards over ALL of them first (MyIndexSearcher.slicesForShards): a call over a
    subset cannot reproduce that by itself.  With the caller's slices stated the per-slice counts -- and so the relation -- follow
    the whole searcher's; without them the library slices the subset on its own (and may report another relation: that is the
    point of the override, not a requirement)."""
    ranks = [3, 30, 100, 300, 1000]
    corpus = synth.build_corpus(400_000, ranks, n_segments=8, delete_fraction=0.01)
    ctx = api.GpuContext(device_id=0, max_batch=64)
    ctx.set_slicing(30_000, 2, 3)
    max_docs = np.asarray([s.max_doc for s in corpus.segments], dtype=np.int32)
    num_docs = np.asarray([s.max_doc if s.live_bits is None else int(np.unpackbits(s.live_bits.view(np.uint8)).sum()) for s in corpus.segments],
                          dtype=np.int32)
    bases = np.asarray([s.doc_base for s in corpus.segments], dtype=np.int32)
    from nrtsearch_amd import _lib
    slice_of = np.zeros(len(max_docs), dtype=np.int32)
    shard_of = np.zeros(len(max_docs), dtype=np.int32)
    n_slices = _lib.load().nrtgpu_slices(len(max_docs), max_docs.ctypes.data, num_docs.ctypes.data, bases.ctypes.data, 3, 30_000, 2,
                                         slice_of.ctypes.data, shard_of.ctypes.data)
    assert n_slices >= 3
    cold = int(slice_of[len(max_docs) - 1])
    hot = [li for li in range(len(max_docs)) if int(slice_of[li]) != cold]
    leaves = [api.GpuSegment.from_data(ctx, corpus.segments[li]) for li in hot]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    try:
        for terms in ([100], [300, 1000], [3, 30]):
            for k, thr in ((10, 1000), (100, 200), (10, 50)):
                api.GpuContext.set_thread_slices([int(slice_of[li]) for li in hot])
                try:
                    dev = sr.search(bq(terms), api.TopScoreDocCollectorManager(k, total_hits_threshold=thr))
                finally:
                    api.GpuContext.set_thread_slices(None)
                # the same leaves, one oracle collector per slice of the WHOLE searcher
                parts = [oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr, segments=[li for li in hot if int(slice_of[li]) == s])
                         for s in sorted(set(int(slice_of[li]) for li in hot))]
                docs, scores, total, gte = reduce_parts(oracle, k, parts)
                assert dev.docs.tolist() == docs.tolist() and dev.scores.view(np.uint32).tolist() == scores.view(np.uint32).tolist()
                assert dev.relation_gte == gte, (terms, k, thr)
                if gte:
                    assert max(thr, k) < dev.total_hits <= total
                else:
                    assert dev.total_hits == total
    finally:
        for g in leaves:
            g.release()
        ctx.close()


def test_coalesced_callers_with_different_slices_do_not_share_a_batch(dev_lib, oracle):
    """nrtgpu_search_bm25_coalesced plans a batch on its LEADER's thread, under the leader's nrtgpu_set_thread_slices.  Two searcher
    versions over the same resident leaves may slice them differently (virtual shards dealt over different leaf lists, a cold leaf
    in one of them): callers whose slices differ must not travel in one batch, or the followers' per-slice hit counts -- the
    totalHits relation -- would be the leader's.  Sixteen callers parked behind the development library's hook,
    eight with "every leaf its own slice", eight with "all leaves one slice", the same leaves: TWO batches, and every caller gets
    the relation of ITS slices -- one collector per leaf never passes the threshold (exact count), one collector over all leaves
    does (lower bound)."""
    import threading
    import time

    ranks = [200, 400, 2000]   # rank 200: 1592 postings, 796 of them in the largest leaf: above the threshold together, below it leaf by leaf
    corpus = synth.build_corpus(320_000, ranks, n_segments=8)
    n_leaves = len(corpus.segments)
    ctx = api.GpuContext(device_id=0, max_batch=64)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    k, thr = 10, 1000
    slicings = {"per_leaf": list(range(n_leaves)), "one": [0] * n_leaves}
    try:
        expected = {}
        for name, sl in slicings.items():
            for terms in ([200], [400, 2000]):
                groups = [[li for li in range(n_leaves) if sl[li] == s] for s in sorted(set(sl))]
                parts = [oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr, segments=g) for g in groups]
                expected[(name, tuple(terms))] = reduce_parts(oracle, k, parts)
        # (the point of the test: the two slicings disagree about the relation of the frequent term)
        assert expected[("per_leaf", (200,))][3] is False and expected[("one", (200,))][3] is True
        errors, results = [], {}
        ctx.reset_stats()
        ctx.debug_hold_coalescers(True)

        def one(tix):
            name = "per_leaf" if tix % 2 == 0 else "one"
            terms = [200] if (tix // 2) % 2 == 0 else [400, 2000]
            try:
                api.GpuContext.set_thread_slices(slicings[name])
                try:
                    r = sr.search_coalesced(bq(terms), api.TopScoreDocCollectorManager(k, total_hits_threshold=thr))
                finally:
                    api.GpuContext.set_thread_slices(None)
                results[tix] = (name, tuple(terms), r)
            except Exception as e:  # noqa: BLE001
                errors.append((tix, repr(e)))

        threads = [threading.Thread(target=one, args=(t,)) for t in range(16)]
        try:
            for t in threads:
                t.start()
            t_end = time.monotonic() + 60.0
            while ctx.debug_coalescer_pending(0) < 16 and time.monotonic() < t_end and not errors:
                time.sleep(0.001)
            parked = ctx.debug_coalescer_pending(0)
        finally:
            ctx.debug_hold_coalescers(False)
        for t in threads:
            t.join()
        assert not errors, errors[:5]
        assert parked == 16
        st = ctx.stats()
        assert st["queries"] == 16 and st["batches"] == 2, st     # one batch per slicing, not one for all
        for tix, (name, terms, r) in results.items():
            docs, scores, total, gte = expected[(name, terms)]
            assert r.docs.tolist() == docs.tolist() and r.scores.view(np.uint32).tolist() == scores.view(np.uint32).tolist(), (tix, name, terms)
            assert r.relation_gte == gte, (tix, name, terms, r.relation_gte, gte)
            assert (max(thr, k) < r.total_hits <= total) if gte else r.total_hits == total, (tix, name, terms, r.total_hits, total)
    finally:
        for g in leaves:
            g.release()
        ctx.close()
