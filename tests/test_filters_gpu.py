"""FILTER / MUST_NOT clauses as resident doc-set masks through the C ABI, against
the oracle run with acceptDocs = liveDocs & filter & ~must_not.  Bit-exact, like every BM25 test."""
import numpy as np
import pytest

from nrtsearch_amd import _lib, api, synth
from oracle import oracle

from tests.test_parity_gpu import Index, assert_same

pytestmark = pytest.mark.gpu


random_mask = synth.random_mask


def accept_of(seg, f, mn):
    return synth.accept_words(seg, f, mn)


@pytest.fixture(scope="module")
def ctx():
    c = api.GpuContext(device_id=0, max_batch=256)
    yield c
    c.close()


@pytest.mark.parametrize("deletes", [0.0, 0.03])
def test_filter_and_must_not_masks(ctx, deletes):
    ranks = [1, 2, 4, 9, 30, 120, 700, 4000]
    corpus = synth.build_corpus(200_000, ranks, n_segments=3, delete_fraction=deletes)
    ix = Index(ctx, corpus)
    try:
        masks = {}
        for si, (seg, leaf) in enumerate(zip(corpus.segments, ix.leaves)):
            masks[(si, 7)] = random_mask(seg.max_doc, 0.30, 100 + si)   # a selective filter
            masks[(si, 9)] = random_mask(seg.max_doc, 0.05, 200 + si)   # an exclusion list
            masks[(si, 11)] = random_mask(seg.max_doc, 0.0005, 300 + si)  # nearly empty filter
            for mid in (7, 9, 11):
                leaf.set_mask(mid, masks[(si, mid)])
        should = tuple(api.TermQuery(0, r) for r in (2, 30, 700))
        terms = [2, 30, 700]
        cases = [
            ("filter", (7,), ()), ("must_not", (), (9,)), ("both", (7,), (9,)), ("tiny_filter", (11,), ()),
            ("same_mask_both_ways", (7,), (7,)),
        ]
        ctx.reset_stats()
        for name, f, mn in cases:
            for k, thr in ((10, 1000), (200, 1000), (50, 2**31 - 1)):
                q = api.BooleanQuery(should, 1, tuple(api.MaskFilter(i) for i in f), tuple(api.MaskFilter(i) for i in mn))
                got = ix.searcher.search(q, api.TopScoreDocCollectorManager(k, None, thr))
                acc = [accept_of(seg, masks[(si, f[0])] if f else None, masks[(si, mn[0])] if mn else None)
                       for si, seg in enumerate(corpus.segments)]
                exp = oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr, accept=acc)
                assert_same(f"mask_{name}_{k}_{deletes}", got, exp, k, thr)
                if name == "same_mask_both_ways":
                    assert got.total_hits == 0 and len(got.docs) == 0
        # masked queries take the MaxScore route (a mask probe per evaluated doc; exact counting until a slice passes the
        # threshold); only ScoreMode.COMPLETE on a big query is scanned exhaustively
        st = ctx.stats()
        assert st["maxscore_items"] >= 2 * len(cases) and st["scan_items"] <= len(cases), st
        # a batch mixing masked and unmasked queries over the same leaves
        qs = [api.BooleanQuery(should, 1, (api.MaskFilter(7),)), api.BooleanQuery(should), api.BooleanQuery(should, 0, (), (api.MaskFilter(9),))]
        mg = [api.TopScoreDocCollectorManager(100)] * 3
        res = ix.searcher.search_batch(qs, mg)
        accs = [[accept_of(s, masks[(si, 7)], None) for si, s in enumerate(corpus.segments)], None,
                [accept_of(s, None, masks[(si, 9)]) for si, s in enumerate(corpus.segments)]]
        for i in range(3):
            assert_same(f"mask_batch_{i}", res[i], oracle.search_bm25(corpus, terms, 100, accept=accs[i]), 100, 1000)
        # new liveDocs invalidate the combined sets
        if deletes:
            for leaf in ix.leaves:
                leaf.set_live_docs(None)
            got = ix.searcher.search(qs[0], mg[0])
            acc = [masks[(si, 7)] for si in range(len(corpus.segments))]
            assert_same("mask_after_live_reset", got, oracle.search_bm25(corpus, terms, 100, accept=acc), 100, 1000)
    finally:
        ix.close()


def test_several_filter_and_must_not_clauses(ctx):
    """Any number of FILTER / MUST_NOT clauses (QueryNodeMapper.java:257-283 builds what the request holds): every clause is a
    resident mask of its own, the library ANDs / AND-NOTs them at plan time (nrtgpu_bm25_query.more_filters / more_must_not).
    Against the oracle with the combined accept set; on the pruned route; the combination is cached per leaf; in a batch next to
    single-mask and unmasked queries; the coalesced single-query entry takes them too."""
    ranks = [1, 2, 4, 9, 30, 120, 700, 4000]
    corpus = synth.build_corpus(200_000, ranks, n_segments=3, delete_fraction=0.02)
    ix = Index(ctx, corpus)
    try:
        masks = {}
        spec = {3: 0.6, 4: 0.5, 5: 0.7, 6: 0.04, 7: 0.02, 8: 0.10}
        for si, (seg, leaf) in enumerate(zip(corpus.segments, ix.leaves)):
            for mid, frac in spec.items():
                masks[(si, mid)] = random_mask(seg.max_doc, frac, 1000 * mid + si)
                leaf.set_mask(mid, masks[(si, mid)])
        should = tuple(api.TermQuery(0, r) for r in (2, 30, 700))
        terms = [2, 30, 700]

        def acc_of(f, mn):
            out = []
            for si, seg in enumerate(corpus.segments):
                n = (seg.max_doc + 63) // 64
                a = np.full(n, ~np.uint64(0), dtype=np.uint64) if seg.live_bits is None else seg.live_bits[:n].copy()
                for i in f:
                    a &= masks[(si, i)][:n]
                for i in mn:
                    a &= ~masks[(si, i)][:n]
                out.append(a)
            return out

        cases = [((3, 4), ()), ((3, 4, 5), (6,)), ((), (6, 7, 8)), ((4,), (6, 7)), ((3, 4, 5), (6, 7, 8)), ((5, 3), (8, 6)), ((3, 3), (6, 6))]
        ctx.reset_stats()
        for f, mn in cases:
            for k, thr in ((10, 1000), (300, 1000), (50, 2**31 - 1)):
                q = api.BooleanQuery(should, 1, tuple(api.MaskFilter(i) for i in f), tuple(api.MaskFilter(i) for i in mn))
                got = ix.searcher.search(q, api.TopScoreDocCollectorManager(k, None, thr))
                exp = oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr, accept=acc_of(f, mn))
                assert_same(f"masks_{f}_{mn}_{k}", got, exp, k, thr)
        st = ctx.stats()
        assert st["maxscore_items"] >= 2 * len(cases), st
        # (5, 3) / (8, 6) is the same combination as (3, 5) / (6, 8): one resident set per leaf, not two
        b0 = [l.device_bytes for l in ix.leaves]
        ix.searcher.search(api.BooleanQuery(should, 1, (api.MaskFilter(3), api.MaskFilter(5)), (api.MaskFilter(6), api.MaskFilter(8))),
                           api.TopScoreDocCollectorManager(10))
        assert [l.device_bytes for l in ix.leaves] == b0
        # a batch: several masks, one mask, none
        qs = [api.BooleanQuery(should, 1, (api.MaskFilter(3), api.MaskFilter(4)), (api.MaskFilter(6), api.MaskFilter(7))),
              api.BooleanQuery(should, 1, (api.MaskFilter(5),)), api.BooleanQuery(should)]
        mg = [api.TopScoreDocCollectorManager(100)] * 3
        res = ix.searcher.search_batch(qs, mg)
        accs = [acc_of((3, 4), (6, 7)), acc_of((5,), ()), acc_of((), ())]
        for i in range(3):
            assert_same(f"masks_batch_{i}", res[i], oracle.search_bm25(corpus, terms, 100, accept=accs[i]), 100, 1000)
        got = ix.searcher.search_coalesced(qs[0], mg[0])
        assert_same("masks_coalesced", got, oracle.search_bm25(corpus, terms, 100, accept=accs[0]), 100, 1000)
        # a mask of the list that is not resident: this request's own error
        with pytest.raises(_lib.NrtGpuError) as e:
            ix.searcher.search(api.BooleanQuery(should, 1, (api.MaskFilter(3), api.MaskFilter(99))), api.TopScoreDocCollectorManager(10))
        assert e.value.code == _lib.NRTGPU_ERR_UNSUPPORTED
    finally:
        ix.close()


def test_more_combinations_than_the_accept_set_cache_holds(ctx):
    """The resident combined doc sets of a leaf are a bounded cache (64 per leaf, like the reference's LRUQueryCache bounds its
    entries): the least recently used combination makes room (an evicted set may still be read by a search in flight, so it is
    freed by the last search to leave the handle -- segment.cpp: accept_set_of_ids), instead of every new combination being
    refused once the cache is full.  150 distinct combinations of 8 masks, then the first ones again: every
    answer is the oracle's, and the leaves hold at most 64 sets afterwards."""
    import itertools

    ranks = [2, 30, 700]
    corpus = synth.build_corpus(60_000, ranks, n_segments=2, delete_fraction=0.01)
    ix = Index(ctx, corpus)
    try:
        masks = {}
        ids = list(range(11, 19))
        for si, (seg, leaf) in enumerate(zip(corpus.segments, ix.leaves)):
            for mid in ids:
                masks[(si, mid)] = random_mask(seg.max_doc, 0.5 + 0.05 * (mid - 11), 77 * mid + si)
                leaf.set_mask(mid, masks[(si, mid)])
        b_masks = [l.device_bytes for l in ix.leaves]
        should = tuple(api.TermQuery(0, r) for r in ranks)

        def acc_of(f, mn):
            out = []
            for si, seg in enumerate(corpus.segments):
                n = (seg.max_doc + 63) // 64
                a = np.full(n, ~np.uint64(0), dtype=np.uint64) if seg.live_bits is None else seg.live_bits[:n].copy()
                for i in f:
                    a &= masks[(si, i)][:n]
                for i in mn:
                    a &= ~masks[(si, i)][:n]
                out.append(a)
            return out

        combos = ([((a,), (b,)) for a, b in itertools.permutations(ids, 2)] +
                  [((a, b), (c,)) for a, b in itertools.combinations(ids, 2) for c in ids if c not in (a, b)])[:150]
        assert len(set(combos)) == 150
        for f, mn in combos + combos[:10]:
            q = api.BooleanQuery(should, 1, tuple(api.MaskFilter(i) for i in f), tuple(api.MaskFilter(i) for i in mn))
            got = ix.searcher.search(q, api.TopScoreDocCollectorManager(20))
            assert_same(f"lru_{f}_{mn}", got, oracle.search_bm25(corpus, ranks, 20, accept=acc_of(f, mn)), 20, 1000)
        for l, b0, seg in zip(ix.leaves, b_masks, corpus.segments):
            set_bytes = (seg.max_doc + 63) // 64 * 8
            assert l.device_bytes - b0 <= 64 * set_bytes, (l.device_bytes - b0, set_bytes)   # (the evicted ones were freed: nothing was in flight)
        # the same while a search is IN FLIGHT over the leaves (begun, not waited for): its combination is evicted by the 70 that
        # follow -- retired, not freed -- its answer is still the oracle's, and the retired sets are gone once it has left
        import torch
        f0, mn0 = combos[120]
        q0 = api.BooleanQuery(should, 1, tuple(api.MaskFilter(i) for i in f0), tuple(api.MaskFilter(i) for i in mn0))
        pb = api.PreparedBatch(ix.searcher, [q0], [api.TopScoreDocCollectorManager(20)])
        keys = torch.zeros((1, 32), dtype=torch.int64, device="cuda")
        cnt = torch.zeros((1,), dtype=torch.int32, device="cuda")
        hits = torch.zeros((1,), dtype=torch.int64, device="cuda")
        h = pb.begin_device(32, keys.data_ptr(), cnt.data_ptr(), hits.data_ptr())
        for f, mn in combos[:70]:
            q = api.BooleanQuery(should, 1, tuple(api.MaskFilter(i) for i in f), tuple(api.MaskFilter(i) for i in mn))
            got = ix.searcher.search(q, api.TopScoreDocCollectorManager(20))
            assert_same(f"lru_in_flight_{f}_{mn}", got, oracle.search_bm25(corpus, ranks, 20, accept=acc_of(f, mn)), 20, 1000)
        api.PreparedBatch.wait_device(h)
        torch.cuda.synchronize()
        kk = keys.cpu().numpy().view(np.uint64)[0, : int(cnt.cpu()[0])]
        edocs, escores, _, _ = oracle.search_bm25(corpus, ranks, 20, accept=acc_of(f0, mn0))
        assert (0xFFFFFFFF - (kk & np.uint64(0xFFFFFFFF))).astype(np.int64).tolist() == edocs.tolist()
        assert (kk >> np.uint64(32)).astype(np.uint32).tolist() == escores.view(np.uint32).tolist()
        for l, b0, seg in zip(ix.leaves, b_masks, corpus.segments):
            assert l.device_bytes - b0 <= 64 * ((seg.max_doc + 63) // 64 * 8)
    finally:
        ix.close()


def test_mask_errors(ctx):
    corpus = synth.build_corpus(5_000, [3, 50], n_segments=1)
    ix = Index(ctx, corpus)
    try:
        should = (api.TermQuery(0, 3), api.TermQuery(0, 50))
        with pytest.raises(_lib.NrtGpuError) as e:   # mask not resident: the caller runs the CPU path
            ix.searcher.search(api.BooleanQuery(should, 1, (api.MaskFilter(5),)), api.TopScoreDocCollectorManager(10))
        assert e.value.code == _lib.NRTGPU_ERR_UNSUPPORTED
        with pytest.raises(api.UnsupportedQuery):    # Lucene would add score-0 hits for filter-only docs
            ix.searcher.search(api.BooleanQuery(should, 0, (api.MaskFilter(5),)), api.TopScoreDocCollectorManager(10))
        with pytest.raises(_lib.NrtGpuError) as e:
            ix.leaves[0].set_mask(0, np.zeros(100, np.uint64))
        assert e.value.code == _lib.NRTGPU_ERR_INVALID_ARG
        with pytest.raises(_lib.NrtGpuError):
            ix.leaves[0].set_mask(5, np.zeros(3, np.uint64))   # too few words
        ix.leaves[0].set_mask(5, random_mask(5_000, 0.5, 1))
        ix.leaves[0].set_mask(5, None)                          # dropped again
        with pytest.raises(_lib.NrtGpuError):
            ix.searcher.search(api.BooleanQuery(should, 1, (api.MaskFilter(5),)), api.TopScoreDocCollectorManager(10))
    finally:
        ix.close()


# ---- minimumNumberShouldMatch > 1 (clause count carried in the fixed-point accumulators) ----------------
@pytest.mark.parametrize("deletes", [0.0, 0.02])
def test_minimum_should_match(ctx, deletes):
    ranks = [1, 2, 3, 6, 15, 50, 400]
    corpus = synth.build_corpus(250_000, ranks, n_segments=3, delete_fraction=deletes)
    ix = Index(ctx, corpus)
    try:
        terms = [1, 3, 6, 15, 400]
        should = tuple(api.TermQuery(0, t) for t in terms)
        for msm in (2, 3, 5, 6):
            for k, thr in ((10, 1000), (1000, 1000), (100, 2**31 - 1)):
                got = ix.searcher.search(api.BooleanQuery(should, msm), api.TopScoreDocCollectorManager(k, None, thr))
                exp = oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr, min_should_match=msm)
                assert_same(f"msm{msm}_{k}_{deletes}", got, exp, k, thr)
                if msm == 6:
                    assert got.total_hits == 0
        # a repeated clause counts twice (every clause is its own scorer)
        dup = [2, 2, 50]
        got = ix.searcher.search(api.BooleanQuery(tuple(api.TermQuery(0, t) for t in dup), 2), api.TopScoreDocCollectorManager(50))
        assert_same("msm_dup", got, oracle.search_bm25(corpus, dup, 50, min_should_match=2), 50, 1000)
        # boosted clauses, paging
        boosts = [1.0, 2.5, 0.5, 3.0, 1.0]
        bq_ = api.BooleanQuery(tuple(api.BoostQuery(api.TermQuery(0, t), b) for t, b in zip(terms, boosts)), 2)
        first = ix.searcher.search(bq_, api.TopScoreDocCollectorManager(40))
        assert_same("msm_boost_p1", first, oracle.search_bm25(corpus, terms, 40, boosts=boosts, min_should_match=2), 40, 1000)
        after = api.ScoreDoc(int(first.docs[-1]), float(first.scores[-1]))
        second = ix.searcher.search(bq_, api.TopScoreDocCollectorManager(40, after))
        assert_same("msm_boost_p2", second,
                    oracle.search_bm25(corpus, terms, 40, boosts=boosts, min_should_match=2, after=(after.doc, after.score)), 40, 1000)
        # a batch mixing counted and plain queries, and a mask next to the count
        masks = [random_mask(s.max_doc, 0.4, 900 + i) for i, s in enumerate(corpus.segments)]
        for leaf, m in zip(ix.leaves, masks):
            leaf.set_mask(3, m)
        qs = [api.BooleanQuery(should, 3), api.BooleanQuery(should), api.BooleanQuery(should, 2, (api.MaskFilter(3),)),
              api.TermQuery(0, 50)]
        res = ix.searcher.search_batch(qs, [api.TopScoreDocCollectorManager(200)] * 4)
        acc = [accept_of(s, masks[i], None) for i, s in enumerate(corpus.segments)]
        assert_same("msm_batch0", res[0], oracle.search_bm25(corpus, terms, 200, min_should_match=3), 200, 1000)
        assert_same("msm_batch1", res[1], oracle.search_bm25(corpus, terms, 200), 200, 1000)
        assert_same("msm_batch2", res[2], oracle.search_bm25(corpus, terms, 200, min_should_match=2, accept=acc), 200, 1000)
        assert_same("msm_batch3", res[3], oracle.search_bm25(corpus, [50], 200), 200, 1000)
        assert ctx.stats()["maxscore_items"] > 0     # clause counting on the MaxScore route
        # weights too far apart for the fixed-point accumulators: refused, the caller runs Lucene
        wide = api.BooleanQuery((api.BoostQuery(api.TermQuery(0, 1), 1e-6), api.BoostQuery(api.TermQuery(0, 400), 1e6)), 2)
        with pytest.raises(_lib.NrtGpuError) as e:
            ix.searcher.search(wide, api.TopScoreDocCollectorManager(10))
        assert e.value.code == _lib.NRTGPU_ERR_UNSUPPORTED
        with pytest.raises(_lib.NrtGpuError) as e:   # ... also as a coalesced request (only the offender of a batch sees the error)
            ix.searcher.search_coalesced(wide, api.TopScoreDocCollectorManager(10))
        assert e.value.code == _lib.NRTGPU_ERR_UNSUPPORTED
        got = ix.searcher.search_coalesced(api.BooleanQuery(should, 2), api.TopScoreDocCollectorManager(10))
        assert_same("msm_coalesced", got, oracle.search_bm25(corpus, terms, 10, min_should_match=2), 10, 1000)
    finally:
        ix.close()


@pytest.mark.parametrize("deletes", [0.0, 0.03])
def test_disjunction_max_query(ctx, deletes):
    """DisjunctionMaxQuery over (boosted) term queries with tie breaker 0 (QueryNodeMapper.java:350-358; the reference's
    test: QueryTest.java:541-583): the doc's best clause instead of the sum -- docids, ranks, score bits == oracle."""
    ranks = [1, 2, 3, 6, 15, 50, 400, 3000]
    corpus = synth.build_corpus(260_000, ranks, n_segments=3, delete_fraction=deletes)
    ix = Index(ctx, corpus)
    try:
        terms = [1, 3, 15, 400, 3000]
        dq = api.DisjunctionMaxQuery(tuple(api.TermQuery(0, t) for t in terms))
        for k, thr in ((10, 1000), (1000, 1000), (100, 2**31 - 1), (1000, 10)):
            got = ix.searcher.search(dq, api.TopScoreDocCollectorManager(k, None, thr))
            assert_same(f"dismax_{k}_{thr}_{deletes}", got, oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr, dismax=0.0), k, thr)
        # it is not the sum: the plain disjunction ranks differently
        plain = ix.searcher.search(api.BooleanQuery(dq.disjuncts), api.TopScoreDocCollectorManager(100))
        dm = ix.searcher.search(dq, api.TopScoreDocCollectorManager(100))
        assert dm.scores[0] < plain.scores[0]
        # boosted disjuncts (the usual multi-field shape: one clause dominates), a repeated term, paging
        boosts = [0.5, 3.0, 1.0, 2.0, 4.0]
        bdq = api.DisjunctionMaxQuery(tuple(api.BoostQuery(api.TermQuery(0, t), b) for t, b in zip(terms, boosts)))
        first = ix.searcher.search(bdq, api.TopScoreDocCollectorManager(60))
        assert_same("dismax_boost_p1", first, oracle.search_bm25(corpus, terms, 60, boosts=boosts, dismax=0.0), 60, 1000)
        after = api.ScoreDoc(int(first.docs[-1]), float(first.scores[-1]))
        second = ix.searcher.search(bdq, api.TopScoreDocCollectorManager(60, after))
        assert_same("dismax_boost_p2", second, oracle.search_bm25(corpus, terms, 60, boosts=boosts, dismax=0.0, after=(after.doc, after.score)), 60, 1000)
        dup = [2, 2, 50]
        got = ix.searcher.search(api.DisjunctionMaxQuery(tuple(api.TermQuery(0, t) for t in dup)), api.TopScoreDocCollectorManager(50))
        assert_same("dismax_dup", got, oracle.search_bm25(corpus, dup, 50, dismax=0.0), 50, 1000)
        # one batch: dismax, plain sum, clause counting, dismax behind a FILTER mask, a single term
        masks = [random_mask(s.max_doc, 0.35, 500 + i) for i, s in enumerate(corpus.segments)]
        for leaf, m in zip(ix.leaves, masks):
            leaf.set_mask(5, m)
        should = dq.disjuncts
        qs = [dq, api.BooleanQuery(should), api.BooleanQuery(should, 3), api.BooleanQuery(must=(dq,), filter=(api.MaskFilter(5),)),
              api.DisjunctionMaxQuery((api.TermQuery(0, 50),))]
        res = ix.searcher.search_batch(qs, [api.TopScoreDocCollectorManager(300)] * 5)
        acc = [accept_of(s, masks[i], None) for i, s in enumerate(corpus.segments)]
        assert_same("dismax_batch0", res[0], oracle.search_bm25(corpus, terms, 300, dismax=0.0), 300, 1000)
        assert_same("dismax_batch1", res[1], oracle.search_bm25(corpus, terms, 300), 300, 1000)
        assert_same("dismax_batch2", res[2], oracle.search_bm25(corpus, terms, 300, min_should_match=3), 300, 1000)
        assert_same("dismax_batch3", res[3], oracle.search_bm25(corpus, terms, 300, dismax=0.0, accept=acc), 300, 1000)
        assert_same("dismax_batch4", res[4], oracle.search_bm25(corpus, [50], 300), 300, 1000)
        assert ix.searcher.supported(dq, api.TopScoreDocCollectorManager(10))
        ctx.reset_stats()
        got = ix.searcher.search_coalesced(dq, api.TopScoreDocCollectorManager(10))
        assert_same("dismax_coalesced", got, oracle.search_bm25(corpus, terms, 10, dismax=0.0), 10, 1000)
        assert ctx.stats()["maxscore_items"] > 0 and ctx.stats()["scan_items"] == 0   # `max` instead of `+` on the MaxScore route
    finally:
        ix.close()


def test_minimum_should_match_full_tiles(ctx):
    """Dense terms over many sub-tiles and several items per query (k-th best shared between items)."""
    ranks = [1, 2, 3, 4, 5]
    corpus = synth.build_corpus(1_500_000, ranks, n_segments=2)
    c2 = api.GpuContext(device_id=0, max_batch=64, target_items=64)
    ix = Index(c2, corpus)
    try:
        for msm, k in ((2, 1000), (4, 100), (5, 10)):
            got = ix.searcher.search(api.BooleanQuery(tuple(api.TermQuery(0, t) for t in ranks), msm),
                                     api.TopScoreDocCollectorManager(k))
            assert_same(f"msm_dense_{msm}", got, oracle.search_bm25(corpus, ranks, k, min_should_match=msm), k, 1000)
    finally:
        ix.close()
        c2.close()


# ---- deletes: folded into the posting columns by default; the mask paths stay reachable by flag ---------
@pytest.mark.parametrize("flags", [0, _lib.NRTGPU_FLAG_NO_LIVE_FOLD, _lib.NRTGPU_FLAG_NO_LIVE_FOLD | _lib.NRTGPU_FLAG_NO_MASK_VARIANT,
                                   _lib.NRTGPU_FLAG_NO_FIXED_POINT, _lib.NRTGPU_FLAG_NO_FIXED_POINT | _lib.NRTGPU_FLAG_NO_LIVE_FOLD])
def test_deletes_all_routes(flags):
    """liveDocs three ways -- re-coded postings (default), the masked scan variant, the per-doc check -- and in
    both accumulator modes: same bits.  8 clauses: some have no score table (division path), long docs and high
    freqs take the escape code."""
    ranks = [1, 2, 3, 5, 9, 17, 60, 250, 1200]
    corpus = synth.build_corpus(220_000, ranks, n_segments=3, delete_fraction=0.05)
    c = api.GpuContext(device_id=0, max_batch=64, flags=flags)
    ix = Index(c, corpus)
    try:
        for terms in ([1, 2, 3, 5, 9, 17, 60, 250], [2, 60, 1200], [1], [1200]):
            for k, thr in ((10, 1000), (1000, 1000), (100, 2**31 - 1)):
                got = ix.searcher.search(api.BooleanQuery(tuple(api.TermQuery(0, t) for t in terms)) if len(terms) > 1
                                         else api.TermQuery(0, terms[0]), api.TopScoreDocCollectorManager(k, None, thr))
                assert_same(f"del_{flags}_{len(terms)}_{k}", got, oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr), k, thr)
        # a later reader version deletes more; then one without deletes (not possible in Lucene, allowed here)
        rng = np.random.Generator(np.random.PCG64(3))
        for seg, leaf in zip(corpus.segments, ix.leaves):
            alive = np.unpackbits(seg.live_bits.view(np.uint8), bitorder="little")[: seg.max_doc].astype(bool)
            alive &= rng.random(seg.max_doc) >= 0.10
            padded = np.zeros(((seg.max_doc + 63) // 64) * 64, dtype=bool)
            padded[: seg.max_doc] = alive
            seg.live_bits = np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1)
            leaf.set_live_docs(seg.live_bits)
        terms = [1, 3, 9, 60, 250, 1200]
        q = api.BooleanQuery(tuple(api.TermQuery(0, t) for t in terms))
        assert_same(f"del2_{flags}", ix.searcher.search(q, api.TopScoreDocCollectorManager(300)), oracle.search_bm25(corpus, terms, 300), 300, 1000)
        for seg, leaf in zip(corpus.segments, ix.leaves):
            seg.live_bits = None
            leaf.set_live_docs(None)
        assert_same(f"del3_{flags}", ix.searcher.search(q, api.TopScoreDocCollectorManager(300)), oracle.search_bm25(corpus, terms, 300), 300, 1000)
    finally:
        ix.close()
        c.close()


@pytest.mark.parametrize("deletes", [0.0, 0.03])
def test_disjunction_max_query_with_a_tie_breaker(ctx, deletes):
    """DisjunctionMaxQuery with tieBreakerMultiplier > 0 (QueryNodeMapper.java:350-358): (float)(best + tieBreaker x the others),
    computed in double from the doc's best clause AND the sum of its clauses -- the MaxScore kernel's second accumulator.  Docids,
    ranks, score bits == the oracle's DisjunctionMaxScorer restatement."""
    ranks = [1, 2, 3, 6, 15, 50, 400, 3000]
    corpus = synth.build_corpus(260_000, ranks, n_segments=3, delete_fraction=deletes)
    ix = Index(ctx, corpus)
    try:
        terms = [1, 3, 15, 400, 3000]
        should = tuple(api.TermQuery(0, t) for t in terms)
        ctx.reset_stats()
        for tb in (0.1, 0.5, 1.0, float(np.float32(0.3))):
            dq = api.DisjunctionMaxQuery(should, tb)
            for k, thr in ((10, 1000), (1000, 1000), (100, 10)):
                got = ix.searcher.search(dq, api.TopScoreDocCollectorManager(k, None, thr))
                assert_same(f"tie_{tb}_{k}_{thr}_{deletes}", got, oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr, dismax=tb), k, thr)
        st = ctx.stats()
        assert st["maxscore_items"] > 0 and st["scan_items"] == 0
        # tie breaker 1 is NOT the plain sum's float: best + others in double, one cast -- the same value here, and the oracle agrees
        # boosted disjuncts, a repeated term, paging, a mask, mixed into one batch with the other shapes
        boosts = [0.5, 3.0, 1.0, 2.0, 4.0]
        bdq = api.DisjunctionMaxQuery(tuple(api.BoostQuery(api.TermQuery(0, t), b) for t, b in zip(terms, boosts)), 0.25)
        first = ix.searcher.search(bdq, api.TopScoreDocCollectorManager(60))
        assert_same("tie_boost_p1", first, oracle.search_bm25(corpus, terms, 60, boosts=boosts, dismax=0.25), 60, 1000)
        after = api.ScoreDoc(int(first.docs[-1]), float(first.scores[-1]))
        second = ix.searcher.search(bdq, api.TopScoreDocCollectorManager(60, after))
        assert_same("tie_boost_p2", second, oracle.search_bm25(corpus, terms, 60, boosts=boosts, dismax=0.25, after=(after.doc, after.score)), 60, 1000)
        dup = [2, 2, 50]
        got = ix.searcher.search(api.DisjunctionMaxQuery(tuple(api.TermQuery(0, t) for t in dup), 0.7), api.TopScoreDocCollectorManager(50))
        assert_same("tie_dup", got, oracle.search_bm25(corpus, dup, 50, dismax=0.7), 50, 1000)
        masks = [random_mask(s.max_doc, 0.35, 900 + i) for i, s in enumerate(corpus.segments)]
        for leaf, m in zip(ix.leaves, masks):
            leaf.set_mask(6, m)
        acc = [accept_of(s, masks[i], None) for i, s in enumerate(corpus.segments)]
        dq = api.DisjunctionMaxQuery(should, 0.4)
        qs = [dq, api.BooleanQuery(should), api.BooleanQuery(should, 3), api.BooleanQuery(must=(dq,), filter=(api.MaskFilter(6),)),
              api.DisjunctionMaxQuery(should), api.DisjunctionMaxQuery((api.TermQuery(0, 50),), 0.9)]
        res = ix.searcher.search_batch(qs, [api.TopScoreDocCollectorManager(300)] * len(qs))
        assert_same("tie_batch0", res[0], oracle.search_bm25(corpus, terms, 300, dismax=0.4), 300, 1000)
        assert_same("tie_batch1", res[1], oracle.search_bm25(corpus, terms, 300), 300, 1000)
        assert_same("tie_batch2", res[2], oracle.search_bm25(corpus, terms, 300, min_should_match=3), 300, 1000)
        assert_same("tie_batch3", res[3], oracle.search_bm25(corpus, terms, 300, dismax=0.4, accept=acc), 300, 1000)
        assert_same("tie_batch4", res[4], oracle.search_bm25(corpus, terms, 300, dismax=0.0), 300, 1000)
        assert_same("tie_batch5", res[5], oracle.search_bm25(corpus, [50], 300, dismax=0.9), 300, 1000)
        got = ix.searcher.search_coalesced(dq, api.TopScoreDocCollectorManager(10))
        assert_same("tie_coalesced", got, oracle.search_bm25(corpus, terms, 10, dismax=0.4), 10, 1000)
        # ScoreMode.COMPLETE over a large query would need the exhaustive scan, which carries one accumulator: the caller's path
        big = api.DisjunctionMaxQuery(tuple(api.TermQuery(0, t) for t in (1, 2, 3, 6, 15, 50)), 0.4)   # (> 2^18 postings)
        with pytest.raises(api.NrtGpuError):
            ix.searcher.search(big, api.TopScoreDocCollectorManager(10, None, 2**31 - 1))
        got = ix.searcher.search(api.DisjunctionMaxQuery(should[2:], 0.4), api.TopScoreDocCollectorManager(10, None, 2**31 - 1))   # a small one: exact mode
        assert_same("tie_complete_small", got, oracle.search_bm25(corpus, terms[2:], 10, total_hits_threshold=2**31 - 1, dismax=0.4), 10, 2**31 - 1)
    finally:
        ix.close()


@pytest.mark.parametrize("deletes", [0.0, 0.03])
def test_must_next_to_should_clauses(ctx, deletes):
    """BooleanQuery with MUST and SHOULD term clauses (QueryNodeMapper.java:257-283; minimumNumberShouldMatch 0): a hit matches
    every MUST clause, its score is (float) MUST sum + (float) SHOULD sum added in float (ReqOptSumScorer) -- the kernel's second
    accumulator.  Docids, ranks, score bits and hit counts == the oracle's restatement."""
    ranks = [1, 2, 3, 6, 15, 50, 400, 3000]
    corpus = synth.build_corpus(260_000, ranks, n_segments=3, delete_fraction=deletes)
    ix = Index(ctx, corpus)
    try:
        tq = lambda t: api.TermQuery(0, t)   # noqa: E731
        cases = [([15], [1, 3, 400]),            # a mid-frequency MUST term, frequent and rare SHOULD terms
                 ([1], [3000, 400]),             # the MUST term is the most frequent one: the rare SHOULD clauses are streamed first
                 ([3000], [1, 2, 3]),            # a rare MUST term
                 ([2, 50], [6, 400, 3000]),      # two MUST clauses (ConjunctionScorer's double sum) + three SHOULD
                 ([6, 15, 400], [1]),            # three MUST + one SHOULD
                 ([50], [50, 3]),                # the same term on both sides
                 ([3, 9999], [1])]               # a MUST term the index does not hold: no hits
        ctx.reset_stats()
        for ci, (must, should) in enumerate(cases):
            q = api.BooleanQuery(tuple(tq(t) for t in should), must=tuple(tq(t) for t in must))
            terms, flags = must + should, [True] * len(must) + [False] * len(should)
            for k, thr in ((10, 1000), (1000, 1000), (100, 10), (7, 1000)):
                got = ix.searcher.search(q, api.TopScoreDocCollectorManager(k, None, thr))
                assert_same(f"reqopt_{ci}_{k}_{thr}_{deletes}", got, oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr, must=flags), k, thr)
        st = ctx.stats()
        assert st["maxscore_items"] > 0 and st["scan_items"] == 0
        # ScoreMode.COMPLETE: a small query is counted exactly on the same route; a large one would need the exhaustive scan
        small = api.BooleanQuery((tq(3000), tq(50)), must=(tq(400),))
        got = ix.searcher.search(small, api.TopScoreDocCollectorManager(50, None, 2**31 - 1))
        assert_same("reqopt_complete_small", got, oracle.search_bm25(corpus, [400, 3000, 50], 50, total_hits_threshold=2**31 - 1, must=[True, False, False]), 50, 2**31 - 1)
        with pytest.raises(api.NrtGpuError):
            ix.searcher.search(api.BooleanQuery((tq(1), tq(2), tq(3)), must=(tq(6),)), api.TopScoreDocCollectorManager(50, None, 2**31 - 1))
        # it is not the one-sum score: somewhere in the top hits float(a) + float(b) != float(a + b)
        must, should = [2, 50], [6, 400, 3000]
        q = api.BooleanQuery(tuple(tq(t) for t in should), must=tuple(tq(t) for t in must))
        got = ix.searcher.search(q, api.TopScoreDocCollectorManager(1000))
        one_sum = ix.searcher.search(api.BooleanQuery(tuple(tq(t) for t in must + should)), api.TopScoreDocCollectorManager(1000, None, 2**31 - 1))
        as_one = dict(zip(one_sum.docs.tolist(), one_sum.scores.view(np.uint32).tolist()))
        both = [(d, b) for d, b in zip(got.docs.tolist(), got.scores.view(np.uint32).tolist()) if d in as_one]
        assert both and any(as_one[d] != b for d, b in both), "every MUST + SHOULD score equals the one-sum score: the case proves nothing"
        # boosts, paging, a FILTER and a MUST_NOT mask, one batch with the other shapes
        boosts = [2.0, 0.5, 1.0, 3.0, 0.25]
        boosted = api.BooleanQuery(tuple(api.BoostQuery(tq(t), b) for t, b in zip(should, boosts[2:])), must=tuple(api.BoostQuery(tq(t), b) for t, b in zip(must, boosts[:2])))
        first = ix.searcher.search(boosted, api.TopScoreDocCollectorManager(40))
        okw = dict(boosts=boosts, must=[True, True, False, False, False])
        assert_same("reqopt_boost_p1", first, oracle.search_bm25(corpus, must + should, 40, **okw), 40, 1000)
        after = api.ScoreDoc(int(first.docs[-1]), float(first.scores[-1]))
        second = ix.searcher.search(boosted, api.TopScoreDocCollectorManager(40, after))
        assert_same("reqopt_boost_p2", second, oracle.search_bm25(corpus, must + should, 40, after=(after.doc, after.score), **okw), 40, 1000)
        masks = [random_mask(s.max_doc, 0.4, 700 + i) for i, s in enumerate(corpus.segments)]
        masks2 = [random_mask(s.max_doc, 0.1, 800 + i) for i, s in enumerate(corpus.segments)]
        for leaf, m, m2 in zip(ix.leaves, masks, masks2):
            leaf.set_mask(7, m)
            leaf.set_mask(8, m2)
        acc = [accept_of(s, masks[i], masks2[i]) for i, s in enumerate(corpus.segments)]
        qf = api.BooleanQuery(q.should, 0, (api.MaskFilter(7),), (api.MaskFilter(8),), q.must)
        sh = tuple(tq(t) for t in must + should)
        qs = [q, qf, api.BooleanQuery(sh), api.BooleanQuery(must=sh[:3]), api.DisjunctionMaxQuery(sh, 0.5), api.BooleanQuery(sh, 2)]
        res = ix.searcher.search_batch(qs, [api.TopScoreDocCollectorManager(200)] * len(qs))
        fl = [True, True, False, False, False]
        assert_same("reqopt_batch0", res[0], oracle.search_bm25(corpus, must + should, 200, must=fl), 200, 1000)
        assert_same("reqopt_batch1", res[1], oracle.search_bm25(corpus, must + should, 200, must=fl, accept=acc), 200, 1000)
        assert_same("reqopt_batch2", res[2], oracle.search_bm25(corpus, must + should, 200), 200, 1000)
        assert_same("reqopt_batch3", res[3], oracle.search_bm25(corpus, (must + should)[:3], 200, min_should_match=3), 200, 1000)
        assert_same("reqopt_batch4", res[4], oracle.search_bm25(corpus, must + should, 200, dismax=0.5), 200, 1000)
        assert_same("reqopt_batch5", res[5], oracle.search_bm25(corpus, must + should, 200, min_should_match=2), 200, 1000)
        got = ix.searcher.search_coalesced(q, api.TopScoreDocCollectorManager(10))
        assert_same("reqopt_coalesced", got, oracle.search_bm25(corpus, must + should, 10, must=fl), 10, 1000)
    finally:
        ix.close()


def test_must_conjunction_of_terms(ctx):
    """MatchQuery with operator MUST: every term required, scores of all of them summed."""
    corpus = synth.build_corpus(150_000, [1, 2, 4, 30], n_segments=2, delete_fraction=0.01)
    ix = Index(ctx, corpus)
    try:
        for terms in ([1, 2], [1, 2, 4], [2, 4, 30]):
            q = api.BooleanQuery(must=tuple(api.TermQuery(0, t) for t in terms))
            got = ix.searcher.search(q, api.TopScoreDocCollectorManager(100))
            assert_same(f"must_{len(terms)}", got, oracle.search_bm25(corpus, terms, 100, min_should_match=len(terms)), 100, 1000)
    finally:
        ix.close()


def test_mask_with_search_after_and_min_competitive(ctx):
    """A masked query that also pages (searchAfter takes the per-doc path inside the masked variant) or carries
    a min competitive score from another shard."""
    corpus = synth.build_corpus(180_000, [1, 3, 12, 90], n_segments=3, delete_fraction=0.02)
    ix = Index(ctx, corpus)
    try:
        masks = [random_mask(s.max_doc, 0.5, 40 + i) for i, s in enumerate(corpus.segments)]
        for leaf, m in zip(ix.leaves, masks):
            leaf.set_mask(6, m)
        acc = [accept_of(s, masks[i], None) for i, s in enumerate(corpus.segments)]
        terms = [1, 12, 90]
        q = api.BooleanQuery(tuple(api.TermQuery(0, t) for t in terms), 1, (api.MaskFilter(6),))
        p1 = ix.searcher.search(q, api.TopScoreDocCollectorManager(60))
        assert_same("mask_page1", p1, oracle.search_bm25(corpus, terms, 60, accept=acc), 60, 1000)
        after = api.ScoreDoc(int(p1.docs[-1]), float(p1.scores[-1]))
        p2 = ix.searcher.search(q, api.TopScoreDocCollectorManager(60, after))
        assert_same("mask_page2", p2, oracle.search_bm25(corpus, terms, 60, accept=acc, after=(after.doc, after.score)), 60, 1000)
        assert not set(p1.docs.tolist()) & set(p2.docs.tolist())
        # a bound from elsewhere: everything strictly below it is counted but not collected
        bound = float(p1.scores[30])
        got = ix.searcher.search(q, api.TopScoreDocCollectorManager(60, None, 1000, bound))
        exp = oracle.search_bm25(corpus, terms, 60, accept=acc)
        keep = exp[1] >= np.float32(bound)
        assert got.docs[: keep.sum()].tolist() == exp[0][keep].tolist()
        assert got.total_hits == exp[2]
    finally:
        ix.close()


def test_live_docs_change_while_searching(ctx):
    """A reader-version change (new liveDocs) while searches run on other threads: every answer is the oracle's
    for one of the two versions -- never a mixture (the fold rewrites posting columns in place, so searches and
    set_live_docs exclude each other per segment inside the library)."""
    import threading

    corpus = synth.build_corpus(120_000, [1, 3, 9, 40], n_segments=2, delete_fraction=0.02)
    ix = Index(ctx, corpus)
    try:
        rng = np.random.Generator(np.random.PCG64(5))
        v_a = [s.live_bits.copy() for s in corpus.segments]
        v_b = []
        for s in corpus.segments:
            alive = np.unpackbits(s.live_bits.view(np.uint8), bitorder="little")[: s.max_doc].astype(bool) & (rng.random(s.max_doc) >= 0.2)
            padded = np.zeros(((s.max_doc + 63) // 64) * 64, dtype=bool)
            padded[: s.max_doc] = alive
            v_b.append(np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1))
        terms = [1, 9, 40]
        q = api.BooleanQuery(tuple(api.TermQuery(0, t) for t in terms))
        # the leaves switch one after the other, so a search may legally see any combination of versions per leaf
        exp = []
        for c0 in (v_a, v_b):
            for c1 in (v_a, v_b):
                d, s_, tot, gte = oracle.search_bm25(corpus, terms, 100, accept=[c0[0], c1[1]])
                exp.append((d.tolist(), s_.view(np.uint32).tolist(), tot, gte))
        stop = threading.Event()
        errors = []

        def flip():
            i = 0
            while not stop.is_set():
                ver = v_b if i % 2 == 0 else v_a
                for leaf, bits in zip(ix.leaves, ver):
                    leaf.set_live_docs(bits)
                i += 1

        def search():
            for _ in range(150):
                got = ix.searcher.search(q, api.TopScoreDocCollectorManager(100))
                rec = (got.docs.tolist(), got.scores.view(np.uint32).tolist(), got.total_hits, got.relation_gte)
                # (a pruned search reports a lower bound above the threshold where the relation is GTE)
                if not any(rec[:2] == e[:2] and rec[3] == e[3] and (rec[2] == e[2] or (e[3] and 1000 < rec[2] <= e[2])) for e in exp):
                    errors.append(rec[2])

        t_flip = threading.Thread(target=flip)
        t_search = [threading.Thread(target=search) for _ in range(3)]
        t_flip.start()
        for t in t_search:
            t.start()
        for t in t_search:
            t.join()
        stop.set()
        t_flip.join()
        assert not errors, f"{len(errors)} searches saw a torn segment"
        for leaf, bits in zip(ix.leaves, v_a):
            leaf.set_live_docs(bits)
        got = ix.searcher.search(q, api.TopScoreDocCollectorManager(100))
        assert got.docs.tolist() == exp[0][0] and got.relation_gte == exp[0][3]   # exp[0] = both leaves at version A
        assert (1000 < got.total_hits <= exp[0][2]) if exp[0][3] else got.total_hits == exp[0][2]
    finally:
        ix.close()


def test_forked_reader_versions_are_point_in_time(ctx):
    """nrtgpu_segment_fork: a refresh that only changed liveDocs gets new handles that SHARE the postings; searches over
    the previous handles keep the previous liveDocs (Lucene's point-in-time IndexSearcher) while the new ones see the new
    deletes -- both bit-exact against the oracle, on the MaxScore route and on the exhaustive scan."""
    import copy

    ranks = [1, 2, 3, 9, 40, 300, 2500]
    v1 = synth.build_corpus(400_000, ranks, n_segments=3, delete_fraction=0.01)     # reader version 1: 1 % deleted
    v2 = copy.deepcopy(v1)                                                           # version 2: more deletes on top
    rng = np.random.Generator(np.random.PCG64(77))
    for seg in v2.segments:
        more = rng.random(seg.max_doc) < 0.05
        words = seg.live_bits.copy()
        idx = np.nonzero(more)[0]
        np.bitwise_and.at(words, idx // 64, ~(np.uint64(1) << (idx % 64).astype(np.uint64)))
        seg.live_bits = words
    ix1 = Index(ctx, v1)
    forks = [leaf.fork(seg.live_bits) for leaf, seg in zip(ix1.leaves, v2.segments)]
    sr2 = api.GpuIndexSearcher(ctx, forks, api.IndexStatistics.from_corpus(v2))
    try:
        assert sum(f.device_bytes for f in forks) < 0.05 * sum(l.device_bytes for l in ix1.leaves)   # only the bit sets are new
        for terms in ([1, 3, 40, 300, 2500], [2, 9], [2500]):
            for k, thr in ((1000, 1000), (50, 2**31 - 1)):
                mgr = api.TopScoreDocCollectorManager(k, None, thr)
                assert_same(f"fork_v1_{terms[0]}_{k}", ix1.searcher.search(bq_(terms), mgr), oracle.search_bm25(v1, terms, k, total_hits_threshold=thr), k, thr)
                assert_same(f"fork_v2_{terms[0]}_{k}", sr2.search(bq_(terms), mgr), oracle.search_bm25(v2, terms, k, total_hits_threshold=thr), k, thr)
        # a version may not resurrect a doc the shared postings already carry as deleted
        all_live = np.full(len(v1.segments[0].live_bits), np.uint64(0xFFFFFFFFFFFFFFFF))
        if ctx.flags & _lib.NRTGPU_FLAG_PACKED_POSTINGS:     # (packed postings never carry deletes: any liveDocs may be forked)
            ix1.leaves[0].fork(all_live).release()
        else:
            with pytest.raises(_lib.NrtGpuError) as e:
                ix1.leaves[0].fork(all_live)
            assert e.value.code == _lib.NRTGPU_ERR_UNSUPPORTED
        # the old reader closes: the forks keep the data alive
        ix1.close()
        got = sr2.search(bq_([1, 3, 40, 300, 2500]), api.TopScoreDocCollectorManager(100))
        assert_same("fork_after_release", got, oracle.search_bm25(v2, [1, 3, 40, 300, 2500], 100), 100, 1000)
    finally:
        for f in forks:
            f.release()
        ix1.close()


def bq_(terms):
    from tests.test_parity_gpu import bq

    return bq(terms)


def test_segments_are_uploaded_and_sealed_while_searches_run(ctx):
    """NRT churn (index/ShardState.java:506-527: a refresh builds the next searcher while the current one serves): a new
    segment is uploaded and sealed on one thread -- norms, postings, the seal-time passes -- while other threads keep searching
    the leaves that are already resident.  Those searches keep returning the oracle's answer for THEIR leaf set; the searcher
    built afterwards over all leaves returns the oracle's answer for the whole index.  (Uploads run on the legacy stream, the
    searches on non-blocking streams of their own: neither waits for the other on the device.)"""
    import threading

    corpus = synth.build_corpus(240_000, [1, 2, 7, 30, 200], n_segments=4, delete_fraction=0.01)
    old_segs, new_segs = corpus.segments[:2], corpus.segments[2:]
    old_leaves = [api.GpuSegment.from_data(ctx, s) for s in old_segs]
    stats = api.IndexStatistics.from_corpus(corpus)       # index-global statistics, as the reference computes them
    old_searcher = api.GpuIndexSearcher(ctx, old_leaves, stats)
    terms = [2, 7, 200]
    q = api.BooleanQuery(tuple(api.TermQuery(0, t) for t in terms))
    first = old_searcher.search(q, api.TopScoreDocCollectorManager(50))
    errors, new_leaves, stop = [], [], threading.Event()

    def search():
        while not stop.is_set():
            got = old_searcher.search(q, api.TopScoreDocCollectorManager(50))
            if got.docs.tolist() != first.docs.tolist() or got.scores.view(np.uint32).tolist() != first.scores.view(np.uint32).tolist():
                errors.append("a search over the resident leaves changed its answer during an upload")
                return

    def upload():
        try:
            for rep in range(3):            # the same two segments three times over: as many seal-time passes under the searches
                built = [api.GpuSegment.from_data(ctx, s) for s in new_segs]
                if rep < 2:
                    for g in built:
                        g.release()
                else:
                    new_leaves.extend(built)
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    searchers = [threading.Thread(target=search) for _ in range(3)]
    up = threading.Thread(target=upload)
    for t in searchers:
        t.start()
    up.start()
    up.join()
    stop.set()
    for t in searchers:
        t.join()
    try:
        assert not errors, errors
        assert len(new_leaves) == len(new_segs)
        whole = api.GpuIndexSearcher(ctx, old_leaves + new_leaves, stats)
        got = whole.search(q, api.TopScoreDocCollectorManager(50))
        d, s_, tot, gte = oracle.search_bm25(corpus, terms, 50)
        assert got.docs.tolist() == d.tolist() and got.scores.view(np.uint32).tolist() == s_.view(np.uint32).tolist()
        assert got.relation_gte == gte and ((1000 < got.total_hits <= tot) if gte else got.total_hits == tot)
        # and the leaves searched during the upload gave the whole index's answer restricted to them (same statistics): the
        # whole-index top-50's docs that lie in those leaves are the first so many of their own top-50
        old_limit = old_segs[-1].doc_base + old_segs[-1].max_doc
        restricted = [(doc, bits) for doc, bits in zip(d.tolist(), s_.view(np.uint32).tolist()) if doc < old_limit]
        assert restricted == list(zip(first.docs.tolist(), first.scores.view(np.uint32).tolist()))[: len(restricted)]
    finally:
        for g in old_leaves + new_leaves:
            g.release()
