"""Compressed postings (NRTGPU_FLAG_PACKED_POSTINGS): one 32-bit word per posting in HBM instead of a
docid and a code column.  Same results bit for bit as the two-column layout and as the oracle, on both routes (MaxScore
and the exhaustive scan), with high freqs / long docs (the exception list), deletes (a mask here: not folded), coarse
cells (rare terms in segments larger than a super-window's 2^20 docs) and the query shapes.
The whole GPU suite also runs on this layout with NRTGPU_PACKED_POSTINGS=1 in the environment."""
import numpy as np
import pytest

from nrtsearch_amd import _lib, api, synth
from oracle import oracle

from tests.test_parity_gpu import Index, assert_same, bq

pytestmark = pytest.mark.gpu


def spiced_corpus(n_docs, ranks, n_segments, deletes, seed=11, long_docs=True):
    """A synthetic corpus with many postings the score tables cannot serve: freqs up to 300 and (long_docs) norm bytes
    >= 128.  (Norm bytes up to 255 stretch a term's score range past the fixed-point accumulators: such a corpus runs
    the exhaustive scan in fp64; without them the MaxScore route is taken.)"""
    corpus = synth.build_corpus(n_docs, ranks, n_segments=n_segments, delete_fraction=deletes)
    rng = np.random.Generator(np.random.PCG64(seed))
    for seg in corpus.segments:
        hot = rng.random(len(seg.freqs)) < 0.03
        seg.freqs[hot] = rng.integers(13, 300, size=int(hot.sum()), dtype=np.int32)
        if long_docs:
            far = rng.random(seg.max_doc) < 0.01
            seg.norms[far] = rng.integers(128, 256, size=int(far.sum())).astype(np.uint8)
    return corpus


@pytest.mark.parametrize("deletes,long_docs", [(0.0, False), (0.04, False), (0.04, True)])
def test_packed_equals_two_columns_and_oracle(deletes, long_docs, monkeypatch):
    ranks = [1, 2, 3, 7, 20, 90, 400, 2500, 9000]
    corpus = spiced_corpus(1_300_000, ranks, 2, deletes, long_docs=long_docs)
    packed = api.GpuContext(0, 256, flags=_lib.NRTGPU_FLAG_PACKED_POSTINGS, collect_timing=True)
    plain = api.GpuContext(0, 256, flags=0)
    if not (plain.flags & _lib.NRTGPU_FLAG_PACKED_POSTINGS):   # (not under NRTGPU_PACKED_POSTINGS=1, which packs every context)
        # half the posting bytes, and -- since round 5 -- less for the lookup structures as well: their budget is a share of the
        # RESIDENT posting bytes (nrtgpu_config.lookup_budget_pct), so a context that packs its postings also keeps fewer records.
        # Everything counted: columns, lookup structures, cell tables, norms, liveDocs, per-term records.
        ip, iu = Index(packed, corpus), Index(plain, corpus)
        try:
            assert sum(l.device_bytes for l in ip.leaves) < 0.66 * sum(l.device_bytes for l in iu.leaves)
        finally:
            ip.close()
            iu.close()
    ip, iu = Index(packed, corpus), Index(plain, corpus)
    try:
        cases = [([1, 3, 20, 400, 9000], None), ([2, 7], None), ([9000], None), ([1, 2, 3, 7, 20, 90, 400, 2500, 9000], None),
                 ([3, 90, 2500], [2.0, 0.5, 3.0]), ([1], None)]
        for terms, boosts in cases:
            for k, thr in ((10, 1000), (1000, 1000), (100, 2**31 - 1), (1000, 10)):
                mgr = api.TopScoreDocCollectorManager(k, None, thr)
                gp = ip.searcher.search(bq(terms, boosts), mgr)
                gu = iu.searcher.search(bq(terms, boosts), mgr)
                exp = oracle.search_bm25(corpus, terms, k, boosts=boosts, total_hits_threshold=thr)
                assert_same(f"packed_{terms[0]}_{len(terms)}_{k}_{thr}_{deletes}", gp, exp, k, thr)
                assert gp.docs.tolist() == gu.docs.tolist() and gp.scores.view(np.uint32).tolist() == gu.scores.view(np.uint32).tolist()
        st = packed.stats()
        assert st["scan_launches"] > 0
        if not long_docs:
            assert st["maxscore_launches"] > 0     # both routes ran on the packed layout (deletes: the kernel tests the mask)
        # paging, clause counting, DisjunctionMaxQuery, a mask, a batch
        terms = [1, 3, 20, 400, 9000]
        first = ip.searcher.search(bq(terms), api.TopScoreDocCollectorManager(50))
        after = api.ScoreDoc(int(first.docs[-1]), float(first.scores[-1]))
        second = ip.searcher.search(bq(terms), api.TopScoreDocCollectorManager(50, after))
        assert_same("packed_page2", second, oracle.search_bm25(corpus, terms, 50, after=(after.doc, after.score)), 50, 1000)
        should = tuple(api.TermQuery(0, t) for t in terms)
        if not long_docs:   # (clause counts / best-clause scores need the fixed-point accumulators)
            got = ip.searcher.search(api.BooleanQuery(should, 3), api.TopScoreDocCollectorManager(200))
            assert_same("packed_msm", got, oracle.search_bm25(corpus, terms, 200, min_should_match=3), 200, 1000)
            got = ip.searcher.search(api.DisjunctionMaxQuery(should), api.TopScoreDocCollectorManager(200))
            assert_same("packed_dismax", got, oracle.search_bm25(corpus, terms, 200, dismax=0.0), 200, 1000)
        masks = [synth.random_mask(s.max_doc, 0.3, 40 + i) for i, s in enumerate(corpus.segments)]
        for leaf, m in zip(ip.leaves, masks):
            leaf.set_mask(2, m)
        acc = [synth.accept_words(s, masks[i], None) for i, s in enumerate(corpus.segments)]
        got = ip.searcher.search(api.BooleanQuery(should, 1, (api.MaskFilter(2),)), api.TopScoreDocCollectorManager(300))
        assert_same("packed_mask", got, oracle.search_bm25(corpus, terms, 300, accept=acc), 300, 1000)
        qs = [bq([1, 3, 20, 400, 9000]), bq([2, 7]), bq([9000]), bq([3, 90, 2500], [2.0, 0.5, 3.0])] * 8
        res = ip.searcher.search_batch(qs, [api.TopScoreDocCollectorManager(1000)] * len(qs))
        exp4 = [oracle.search_bm25(corpus, t, 1000, boosts=b) for t, b in (([1, 3, 20, 400, 9000], None), ([2, 7], None), ([9000], None),
                                                                          ([3, 90, 2500], [2.0, 0.5, 3.0]))]
        for i, r in enumerate(res):
            assert_same(f"packed_batch{i}", r, exp4[i % 4], 1000, 1000)
    finally:
        ip.close()
        iu.close()
        packed.close()
        plain.close()


def test_packed_segment_larger_than_a_super_window():
    """One segment of 2.6M docs (> 2 x 2^20): rare terms get coarse cells capped at a super-window, windows and sub-tiles
    on both sides of the 2^20-doc boundaries decode their own offsets."""
    ranks = [1, 5, 60, 900, 9000, 9999]
    corpus = spiced_corpus(2_600_000, ranks, 1, 0.0, seed=13, long_docs=False)
    ctx = api.GpuContext(0, 64, flags=_lib.NRTGPU_FLAG_PACKED_POSTINGS)
    ix = Index(ctx, corpus)
    try:
        for terms in ([1, 5, 60, 900, 9000], [9000, 9999], [5, 9999], [900]):
            for k, thr in ((1000, 1000), (64, 2**31 - 1)):
                got = ix.searcher.search(bq(terms), api.TopScoreDocCollectorManager(k, None, thr))
                assert_same(f"packed_big_{terms[0]}_{k}", got, oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr), k, thr)
    finally:
        ix.close()
        ctx.close()


def test_packed_many_distinct_exceptions():
    """Thousands of distinct (freq, norm) pairs outside the score tables, whole runs of consecutive exception postings
    (blocks of 2048 postings full of them): every one is found through the exception list's directory."""
    corpus = synth.build_corpus(120_000, [1, 2, 30], n_segments=1)
    seg = corpus.segments[0]
    n = min(len(seg.freqs), 9000)
    seg.freqs[:n] = 13 + np.arange(n, dtype=np.int32)      # 9000 consecutive postings, each with its own freq > 12
    seg.freqs[-500:] = 1000 + np.arange(500, dtype=np.int32)
    ctx = api.GpuContext(0, 64, flags=_lib.NRTGPU_FLAG_PACKED_POSTINGS)
    ix = Index(ctx, corpus)
    try:
        for terms in ([1], [1, 2, 30], [30], [2, 30]):
            for k, thr in ((1000, 1000), (20, 2**31 - 1)):
                got = ix.searcher.search(bq(terms), api.TopScoreDocCollectorManager(k, None, thr))
                assert_same(f"packed_exc_{terms[0]}_{k}", got, oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr), k, thr)
    finally:
        ix.close()
        ctx.close()
