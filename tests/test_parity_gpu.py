"""Parity of the HIP path (through the C ABI) against the CPU oracle: docids/ranks, scores and hit
counts must be bit-exact (integer / fp32-bit comparisons, no tolerance).  Needs a real MI355X."""
import json
import os

import numpy as np
import pytest

from nrtsearch_amd import _lib, api, synth

pytestmark = pytest.mark.gpu
f32 = np.float32
INT_MAX = 2**31 - 1
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _dump(name, payload):
    if os.environ.get("NRTGPU_TEST_NO_DUMPS"):   # (tests/test_planner_host.py: under the stand-in HIP runtime every comparison fails by design)
        return
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, f"parity_fail_{name}.json"), "w") as f:
            json.dump(payload, f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    except Exception:
        pass


def total_ok(got: api.TopDocs, etotal, egte, k, threshold) -> bool:
    """totalHits against the oracle's exhaustive count: exact where the relation is EQUAL_TO; where it is
    GREATER_THAN_OR_EQUAL_TO the search may have run with dynamic pruning (MaxScore route) and then reports a lower
    bound above the threshold -- as Lucene does: parity is defined on hits + relation (hard part 3)."""
    if got.relation_gte != egte:
        return False
    if egte:
        return max(threshold, k) < got.total_hits <= etotal
    return got.total_hits == etotal


def assert_same(name, got: api.TopDocs, exp, k, threshold):
    edocs, escores, etotal, egte = exp
    ok = (got.docs.tolist() == edocs.tolist()
          and got.scores.view(np.uint32).tolist() == escores.view(np.uint32).tolist()
          and total_ok(got, etotal, egte, k, threshold))
    if not ok:
        nd = min(len(got.docs), len(edocs))
        first = next((i for i in range(nd) if got.docs[i] != edocs[i] or
                      got.scores[i].view(np.uint32) != escores[i].view(np.uint32)), nd)
        _dump(name, dict(k=k, threshold=threshold, first_mismatch=first, n_got=len(got.docs), n_exp=len(edocs),
                         got_docs=got.docs[max(0, first - 3): first + 5], exp_docs=edocs[max(0, first - 3): first + 5],
                         got_scores=got.scores[max(0, first - 3): first + 5], exp_scores=escores[max(0, first - 3): first + 5],
                         got_total=got.total_hits, exp_total=etotal, got_gte=got.relation_gte, exp_gte=egte))
    assert got.relation_gte == egte, f"{name}: relation"
    assert total_ok(got, etotal, egte, k, threshold), f"{name}: totalHits {got.total_hits} vs exhaustive {etotal} (gte={egte})"
    assert got.docs.tolist() == edocs.tolist(), f"{name}: docids/ranks differ"
    assert got.scores.view(np.uint32).tolist() == escores.view(np.uint32).tolist(), f"{name}: score bits differ"


@pytest.fixture(scope="module")
def ctx():
    c = api.GpuContext(device_id=0, max_batch=4096)
    yield c
    c.close()


class Index:
    """A synthetic corpus uploaded to the GPU + its searcher."""

    def __init__(self, ctx, corpus, **kw):
        self.corpus = corpus
        self.leaves = [api.GpuSegment.from_data(ctx, s, **kw) for s in corpus.segments]
        self.searcher = api.GpuIndexSearcher(ctx, self.leaves, api.IndexStatistics.from_corpus(corpus))

    def close(self):
        for l in self.leaves:
            l.release()


def bq(terms, boosts=None):
    cl = []
    for i, t in enumerate(terms):
        q = api.TermQuery(0, int(t))
        if boosts is not None:
            q = api.BoostQuery(q, float(boosts[i]))
        cl.append(q)
    return cl[0] if len(cl) == 1 else api.BooleanQuery(tuple(cl))


@pytest.fixture(scope="module")
def mid(ctx):
    ranks = [1, 2, 3, 5, 8, 13, 40, 100, 333, 1000, 5000, 9999]
    corpus = synth.build_corpus(300_000, ranks, n_segments=4, delete_fraction=0.02)
    ix = Index(ctx, corpus)
    yield ix
    ix.close()


# ---- the reference's own golden values, now through the device -----------------------------------
def _tiny_index(ctx, docs_terms, lengths):
    """docs_terms: {term_id: [(doc, freq), ...]}."""
    term_ids = sorted(docs_terms)
    offs, d, f = [0], [], []
    for t in term_ids:
        for doc, fr in docs_terms[t]:
            d.append(doc)
            f.append(fr)
        offs.append(len(d))
    seg = synth.SegmentData(max_doc=len(lengths), doc_base=0, norms=synth.int_to_byte4(np.array(lengths)),
                            term_ids=np.array(term_ids, np.int64), offsets=np.array(offs, np.int64),
                            docids=np.array(d, np.int32), freqs=np.array(f, np.int32))
    corpus = synth.Corpus(n_docs=len(lengths), doc_count=len(lengths), sum_total_term_freq=int(sum(lengths)),
                          segments=[seg], doc_freq={t: len(docs_terms[t]) for t in term_ids})
    return Index(ctx, corpus)


def test_docker_compose_known_answer(ctx):
    # docker-compose-config/docs.csv + search.json: "first vendor", "second vendor";
    # query vendor_name:first OR vendor_name:vendor -> doc 0 (0.3979403), doc 1 (0.0828734)
    FIRST, SECOND, VENDOR = 11, 12, 13
    ix = _tiny_index(ctx, {FIRST: [(0, 1)], SECOND: [(1, 1)], VENDOR: [(0, 1), (1, 1)]}, [2, 2])
    td = ix.searcher.search(bq([FIRST, VENDOR]), api.TopScoreDocCollectorManager(100))
    assert td.docs.tolist() == [0, 1] and td.total_hits == 2 and not td.relation_gte
    assert abs(float(td.scores[0]) - 0.3979403) < 1e-7 and abs(float(td.scores[1]) - 0.0828734) < 1e-7
    ix.close()


def test_search_state_golden(ctx):
    # SearchStateTest.java:43-63,117: lastDocId 1, lastScore 0.0766057
    VENDOR = 7
    ix = _tiny_index(ctx, {VENDOR: [(0, 1), (1, 1)], 8: [(0, 1)], 9: [(1, 1)], 10: [(1, 1)]}, [2, 3])
    td = ix.searcher.search(bq([VENDOR]), api.TopScoreDocCollectorManager(10))
    assert td.docs.tolist() == [0, 1]
    assert td.scores[1] == f32(0.0766057)
    ix.close()


def test_similarity_test_bm25_component(ctx):
    # SimilarityTest.java:115-120: "first" freq 2 in doc 0, "vendor" in both, dl 4 -> 0.433217 + 0.0828734
    FIRST, VENDOR = 1, 2
    ix = _tiny_index(ctx, {FIRST: [(0, 2)], VENDOR: [(0, 1), (1, 1)], 3: [(0, 1)], 4: [(1, 2)], 5: [(1, 1)]}, [4, 4])
    td = ix.searcher.search(bq([FIRST, VENDOR]), api.TopScoreDocCollectorManager(10))
    assert td.docs.tolist() == [0, 1]
    assert abs(float(td.scores[0]) - (12.12609 - 11.11 - 0.5)) < 1e-4
    assert abs(float(td.scores[1]) - 0.0828734) < 1e-7
    ix.close()


# ---- randomized parity vs the oracle ---------------------------------------------------------------
@pytest.mark.parametrize("k", [1, 10, 100, 1000, 1024])
def test_disjunction_parity(mid, oracle, k):
    rng = np.random.default_rng(k)
    ranks = sorted(mid.corpus.doc_freq)
    queries, mgrs, exps = [], [], []
    for i in range(24):
        n = int(rng.integers(1, 6))
        terms = rng.choice(ranks, size=n, replace=False).tolist()
        thr = [1000, INT_MAX, 0, 10][i % 4]
        queries.append(bq(terms))
        mgrs.append(api.TopScoreDocCollectorManager(k, None, thr))
        exps.append((terms, thr, oracle.search_bm25(mid.corpus, terms, k, total_hits_threshold=thr)))
    got = mid.searcher.search_batch(queries, mgrs)
    for i, (terms, thr, exp) in enumerate(exps):
        assert_same(f"disj_k{k}_{i}", got[i], exp, k, thr)


def test_single_query_call_equals_batch(mid, oracle):
    terms = [1, 40, 1000]
    exp = oracle.search_bm25(mid.corpus, terms, 50)
    got = mid.searcher.search(bq(terms), api.TopScoreDocCollectorManager(50))
    assert_same("single", got, exp, 50, 1000)


def test_boosts_and_duplicate_clauses(mid, oracle):
    terms, boosts = [2, 2, 13, 333], [1.0, 2.5, 0.25, 7.0]
    exp = oracle.search_bm25(mid.corpus, terms, 200, boosts=boosts, total_hits_threshold=INT_MAX)
    got = mid.searcher.search(bq(terms, boosts), api.TopScoreDocCollectorManager(200, None, INT_MAX))
    assert_same("boosts", got, exp, 200, INT_MAX)


def test_missing_term_and_no_hits(mid, oracle):
    got = mid.searcher.search(bq([777777]), api.TopScoreDocCollectorManager(10))
    assert len(got.docs) == 0 and got.total_hits == 0 and not got.relation_gte
    exp = oracle.search_bm25(mid.corpus, [777777, 9999], 10)
    got = mid.searcher.search(bq([777777, 9999]), api.TopScoreDocCollectorManager(10))
    assert_same("missing", got, exp, 10, 1000)


def test_k_larger_than_matches(mid, oracle):
    exp = oracle.search_bm25(mid.corpus, [9999], 1000)
    got = mid.searcher.search(bq([9999]), api.TopScoreDocCollectorManager(1000))
    assert len(exp[0]) < 1000
    assert_same("k_gt_matches", got, exp, 1000, 1000)


def test_search_after_paging(mid, oracle):
    # RelevanceCollectorITest.java:117-190: pages concatenate to the full ranking, no dup, no gap
    terms = [3, 100]
    full = oracle.search_bm25(mid.corpus, terms, 1000, total_hits_threshold=INT_MAX)
    after, seen_docs, seen_scores = None, [], []
    for page in range(6):
        mgr = api.TopScoreDocCollectorManager(37, after, INT_MAX)
        got = mid.searcher.search(bq(terms), mgr)
        exp = oracle.search_bm25(mid.corpus, terms, 37, after=(after.doc, after.score) if after else None,
                                 total_hits_threshold=INT_MAX)
        assert_same(f"after_{page}", got, exp, 37, INT_MAX)
        seen_docs += got.docs.tolist()
        seen_scores += got.scores.tolist()
        after = api.ScoreDoc(int(got.docs[-1]), float(got.scores[-1]))
    assert seen_docs == full[0][: len(seen_docs)].tolist()
    assert seen_scores == full[1][: len(seen_scores)].tolist()


def test_ties_rank_by_docid(ctx, oracle):
    # TotalHitsThresholdTest.java:43-51,72-99: equal scores => ascending docid.  ATOM-like field:
    # freqs and norms omitted => every matching doc has the same score.
    corpus = synth.build_corpus(100_000, [2, 50], n_segments=2)
    ix = Index(ctx, corpus, omit_norms=True, omit_freqs=True)
    for terms, k in [([2], 100), ([2, 50], 1000), ([50], 7)]:
        exp = oracle.search_bm25(corpus, terms, k, omit_norms=True, omit_freqs=True, total_hits_threshold=INT_MAX)
        got = ix.searcher.search(bq(terms), api.TopScoreDocCollectorManager(k, None, INT_MAX))
        assert_same(f"ties_{terms}", got, exp, k, INT_MAX)
        if len(terms) == 1:
            assert len(set(got.scores.tolist())) == 1 and got.docs.tolist() == sorted(got.docs.tolist())
    ix.close()


def test_ragged_shapes(ctx, oracle):
    # max_doc around multiples of the 768-doc sub-tile and of a 16-wave round (12288), one-doc segment
    for n_docs, nseg in [(1, 1), (767, 1), (768, 1), (769, 1), (1023, 1), (1024, 1), (1025, 1), (12_289, 1), (16_385, 1), (20_000, 3), (70_001, 5)]:
        corpus = synth.build_corpus(n_docs, [1, 3, 9], n_segments=nseg, delete_fraction=0.1 if n_docs > 1 else 0.0)
        ix = Index(ctx, corpus)
        for terms in ([1], [1, 3, 9]):
            exp = oracle.search_bm25(corpus, terms, 64, total_hits_threshold=INT_MAX)
            got = ix.searcher.search(bq(terms), api.TopScoreDocCollectorManager(64, None, INT_MAX))
            assert_same(f"ragged_{n_docs}_{len(terms)}", got, exp, 64, INT_MAX)
        ix.close()


def test_chunking_and_prefetch_invariance(mid, oracle):
    # the result must not depend on how the doc space is cut into items nor on the kernel variant
    terms = [1, 5, 100, 5000]
    exp = oracle.search_bm25(mid.corpus, terms, 1000)
    for target_items, flags in [(1, 0), (64, 0), (4096, 0), (0, _lib.NRTGPU_FLAG_NO_PREFETCH), (4096, _lib.NRTGPU_FLAG_NO_PREFETCH)]:
        c2 = api.GpuContext(0, 64, target_items=target_items, flags=flags)
        leaves = [api.GpuSegment.from_data(c2, s) for s in mid.corpus.segments]
        sr = api.GpuIndexSearcher(c2, leaves, api.IndexStatistics.from_corpus(mid.corpus))
        got = sr.search(bq(terms), api.TopScoreDocCollectorManager(1000))
        assert_same(f"chunk_{target_items}_{flags}", got, exp, 1000, 1000)
        for l in leaves:
            l.release()
        c2.close()


def test_frozen_oracle_fixture_through_the_device(ctx):
    # tests/golden/oracle_small_corpus.json: file-based expectation (no live oracle involved)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_small_corpus.json")) as f:
        g = json.load(f)
    sp = g["spec"]
    corpus = synth.build_corpus(sp["n_docs"], sp["ranks"], n_segments=sp["n_segments"],
                                delete_fraction=sp["delete_fraction"], seed=sp["seed"])
    ix = Index(ctx, corpus)
    try:
        for ci, c in enumerate(g["cases"]):
            q = bq(c["terms"], c.get("boosts"))
            if "msm" in c or "filter" in c or "must_not" in c or "must" in c:   # masks: PCG64(seed + 100 * segment), as the generator
                ids = {}
                for name in ("filter", "must_not"):
                    if name in c:
                        ids[name] = 50 + 2 * ci + (name == "must_not")
                        for i, (seg, leaf) in enumerate(zip(corpus.segments, ix.leaves)):
                            leaf.set_mask(ids[name], synth.random_mask(seg.max_doc, c[name]["density"], c[name]["seed"] + 100 * i))
                cl = q.should if isinstance(q, api.BooleanQuery) else (q,)
                q = api.BooleanQuery(cl, c.get("msm", 0), (api.MaskFilter(ids["filter"]),) if "filter" in ids else (),
                                     (api.MaskFilter(ids["must_not"]),) if "must_not" in ids else ())
            if "dismax" in c:    # the clauses as the disjuncts of a DisjunctionMaxQuery
                q = api.DisjunctionMaxQuery(q.should if isinstance(q, api.BooleanQuery) else (q,), c["dismax"])
            if "must" in c:      # MUST next to SHOULD clauses (minimumNumberShouldMatch 0), the masks kept
                cl = q.should
                q = api.BooleanQuery(tuple(x for x, m in zip(cl, c["must"]) if not m), 0, q.filter, q.must_not, tuple(x for x, m in zip(cl, c["must"]) if m))
            after = None
            if "after" in c:
                after = api.ScoreDoc(c["after"]["doc"], float(np.uint32(c["after"]["score_bits"]).view(np.float32)))
            got = ix.searcher.search(q, api.TopScoreDocCollectorManager(c["k"], after, c["threshold"]))
            assert got.docs.tolist() == c["docs"]
            assert got.scores.view(np.uint32).tolist() == c["score_bits"]
            assert total_ok(got, c["total_hits"], c["relation_gte"], c["k"], c["threshold"])
    finally:
        ix.close()


def test_min_competitive_score_from_other_shards(mid, oracle):
    # Scorable.setMinCompetitiveScore fed from outside (the k-th best score other shards already hold): docs
    # scoring strictly below the bound are counted but not collected; everything at or above it is unchanged
    terms = [1, 5, 100, 5000]
    plain = mid.searcher.search(bq(terms), api.TopScoreDocCollectorManager(1000))
    exact_total = oracle.search_bm25(mid.corpus, terms, 1000)[2]   # (a search with a bound from outside always counts exactly)
    assert_same("mcs_plain", plain, oracle.search_bm25(mid.corpus, terms, 1000), 1000, 1000)
    for rank in (10, 400, 999):
        bound = float(plain.scores[rank])
        got = mid.searcher.search(bq(terms), api.TopScoreDocCollectorManager(1000, None, 1000, bound))
        keep = plain.scores >= np.float32(bound)
        assert got.docs.tolist() == plain.docs[keep].tolist()
        assert got.scores.view(np.uint32).tolist() == plain.scores[keep].view(np.uint32).tolist()
        assert got.total_hits == exact_total
    none = mid.searcher.search(bq(terms), api.TopScoreDocCollectorManager(1000, None, 1000, float(plain.scores[0]) * 2))
    assert len(none.docs) == 0 and none.total_hits == exact_total
    with pytest.raises(_lib.NrtGpuError):
        mid.searcher.search(bq(terms), api.TopScoreDocCollectorManager(10, None, 1000, -1.0))


def test_fixed_point_and_fp64_accumulators_agree(mid, oracle):
    # the scan accumulates in exact fixed point when the host's range analysis allows it, else in fp64
    # (like the reference's double sum); both must give the oracle's bits
    cases = [([1, 5, 100, 5000], None), ([2, 3, 13, 40, 333], [1.0, 0.5, 3.0, 2.0, 1.5]), ([9999], None)]
    for flags, want_fixed in [(0, True), (_lib.NRTGPU_FLAG_NO_FIXED_POINT, False)]:
        c2 = api.GpuContext(0, 64, flags=flags | _lib.NRTGPU_FLAG_NO_PRUNE)   # (this test is about the exhaustive scan's accumulators)
        leaves = [api.GpuSegment.from_data(c2, s) for s in mid.corpus.segments]
        sr = api.GpuIndexSearcher(c2, leaves, api.IndexStatistics.from_corpus(mid.corpus))
        for terms, boosts in cases:
            exp = oracle.search_bm25(mid.corpus, terms, 1000, boosts=boosts)
            got = sr.search(bq(terms, boosts), api.TopScoreDocCollectorManager(1000))
            assert_same(f"acc_{flags}_{terms[0]}", got, exp, 1000, 1000)
        st = c2.stats()
        assert (st["fixed_point_launches"] == st["scan_launches"]) == want_fixed and st["scan_launches"] == len(cases)
        for l in leaves:
            l.release()
        c2.close()


def test_wide_score_range_falls_back_to_fp64(mid, oracle):
    # boosts 2^20 apart: the term scores do not fit one fixed-point scale -> fp64 accumulators, same answer
    terms, boosts = [1, 100, 5000], [1.0, 1048576.0, 3.0]
    c2 = api.GpuContext(0, 64)
    leaves = [api.GpuSegment.from_data(c2, s) for s in mid.corpus.segments]
    sr = api.GpuIndexSearcher(c2, leaves, api.IndexStatistics.from_corpus(mid.corpus))
    exp = oracle.search_bm25(mid.corpus, terms, 100, boosts=boosts, total_hits_threshold=INT_MAX)
    got = sr.search(bq(terms, boosts), api.TopScoreDocCollectorManager(100, None, INT_MAX))
    assert_same("wide_range", got, exp, 100, INT_MAX)
    st = c2.stats()
    assert st["scan_launches"] == 1 and st["fixed_point_launches"] == 0
    for l in leaves:
        l.release()
    c2.close()


def test_four_fields_parity_five_fields_fall_back(ctx, oracle):
    # up to four scored fields (multi-field match, DisjunctionMax over fields): their normInverse tables live in LDS; a
    # fifth field is beyond the device fast path (NRTGPU_ERR_UNSUPPORTED -> the caller runs Lucene)
    import ctypes as C

    from oracle import oracle as o

    corpus = synth.build_corpus(50_000, [1, 4, 20], n_segments=1)
    seg = corpus.segments[0]
    g = api.GpuSegment(ctx, seg.max_doc, 0)
    stats = api.IndexStatistics()
    for field in range(5):
        g.add_field_norms(field, seg.norms)
        g.add_terms(field, seg.term_ids, seg.offsets, seg.docids, seg.freqs)
        stats.fields[field] = api.CollectionStatistics(corpus.doc_count, corpus.sum_total_term_freq * (field + 1))
        for t, df in corpus.doc_freq.items():
            stats.doc_freq[(field, t)] = df
    g.seal()
    sr = api.GpuIndexSearcher(ctx, [g], stats)

    def reference(clauses, dismax):
        col = o.Collector(100, None, INT_MAX)
        arr = (o._Term * len(clauses))()
        keep = []
        for i, (field, t) in enumerate(clauses):
            d, f = seg.postings(t)
            d = np.ascontiguousarray(d)
            f = np.ascontiguousarray(f)
            cs = stats.fields[field]
            cache = o.bm25_norm_cache(float(o.bm25_avgdl(cs.sum_total_term_freq, cs.doc_count)))
            keep += [d, f, cache]
            arr[i].docids, arr[i].freqs, arr[i].n = d.ctypes.data, f.ctypes.data, len(d)
            arr[i].weight = float(o.bm25_idf(cs.doc_count, corpus.doc_freq[t]))
            arr[i].norms, arr[i].cache = seg.norms.ctypes.data, cache.ctypes.data
        if dismax:
            o.lib().nrt_oracle_search_segment_dismax(seg.max_doc, 0, None, len(clauses), C.byref(arr), C.c_float(0.0), col._h)
        else:
            o.lib().nrt_oracle_search_segment(seg.max_doc, 0, None, len(clauses), C.byref(arr), col._h)
        return col.topdocs()

    for clauses in ([(0, 1), (1, 4), (1, 20), (0, 20)], [(0, 4), (1, 4), (2, 4), (3, 4)], [(3, 1), (2, 20), (0, 4), (1, 1), (3, 20)]):
        q = api.BooleanQuery(tuple(api.TermQuery(f, t) for f, t in clauses))
        assert_same(f"fields_sum_{len(clauses)}", sr.search(q, api.TopScoreDocCollectorManager(100, None, INT_MAX)), reference(clauses, False), 100, INT_MAX)
        dq = api.DisjunctionMaxQuery(tuple(api.TermQuery(f, t) for f, t in clauses))      # the multi-field "best field" shape
        assert_same(f"fields_max_{len(clauses)}", sr.search(dq, api.TopScoreDocCollectorManager(100, None, INT_MAX)), reference(clauses, True), 100, INT_MAX)
    q5 = api.BooleanQuery(tuple(api.TermQuery(f, 4) for f in range(5)))
    with pytest.raises(api.NrtGpuError) as e:
        sr.search(q5, api.TopScoreDocCollectorManager(100))
    assert e.value.code == _lib.NRTGPU_ERR_UNSUPPORTED
    g.release()


def test_argument_errors(mid):
    # LazyQueueTopScoreDocCollectorManager.java:90-98 -> IllegalArgumentException
    with pytest.raises(api.NrtGpuError) as e:
        mid.searcher.search(bq([1]), api.TopScoreDocCollectorManager(0))
    assert e.value.code == _lib.NRTGPU_ERR_INVALID_ARG
    with pytest.raises(api.NrtGpuError) as e:
        mid.searcher.search(bq([1]), api.TopScoreDocCollectorManager(10, None, -1))
    assert e.value.code == _lib.NRTGPU_ERR_INVALID_ARG
    with pytest.raises(api.NrtGpuError) as e:   # beyond the device fast path: caller falls back to Lucene
        mid.searcher.search(bq([1]), api.TopScoreDocCollectorManager(5000))
    assert e.value.code == _lib.NRTGPU_ERR_UNSUPPORTED
    with pytest.raises(api.UnsupportedQuery):   # a nested clause is not an eligible shape
        mid.searcher.search(api.BooleanQuery((api.BooleanQuery((api.TermQuery(0, 1),)), api.TermQuery(0, 2))),
                            api.TopScoreDocCollectorManager(5))


def test_unsealed_segment_is_a_state_error(ctx):
    g = api.GpuSegment(ctx, 10, 0)
    st = api.IndexStatistics()
    st.fields[0] = api.CollectionStatistics(10, 40)
    st.doc_freq[(0, 1)] = 1
    sr = api.GpuIndexSearcher(ctx, [g], st)
    with pytest.raises(api.NrtGpuError) as e:
        sr.search(bq([1]), api.TopScoreDocCollectorManager(5))
    assert e.value.code == _lib.NRTGPU_ERR_STATE
    g.release()


def test_live_docs_update_after_seal(ctx, oracle):
    corpus = synth.build_corpus(40_000, [1, 6], n_segments=1)
    ix = Index(ctx, corpus)
    exp = oracle.search_bm25(corpus, [1, 6], 20, total_hits_threshold=INT_MAX)
    assert_same("live0", ix.searcher.search(bq([1, 6]), api.TopScoreDocCollectorManager(20, None, INT_MAX)), exp, 20, INT_MAX)
    # delete the current top hit: only liveDocs change between reader versions of one segment
    top = int(exp[0][0])
    bits = np.full((40_000 + 63) // 64, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    bits[top >> 6] &= ~np.uint64(1 << (top & 63))
    corpus.segments[0].live_bits = bits
    ix.leaves[0].set_live_docs(bits)
    exp2 = oracle.search_bm25(corpus, [1, 6], 20, total_hits_threshold=INT_MAX)
    got2 = ix.searcher.search(bq([1, 6]), api.TopScoreDocCollectorManager(20, None, INT_MAX))
    assert top not in got2.docs.tolist() and exp2[2] == exp[2] - 1
    assert_same("live1", got2, exp2, 20, INT_MAX)
    ix.close()


def test_query_supported_is_the_planners_predicate(mid):
    ok = api.TopScoreDocCollectorManager(100)
    assert mid.searcher.supported(bq([1, 100, 5000]), ok)
    assert mid.searcher.supported(bq([424242]), ok)                                   # a term no leaf holds: runs, matches nothing
    assert not mid.searcher.supported(bq([1, 5]), api.TopScoreDocCollectorManager(5000))            # numHits > NRTGPU_MAX_K
    assert not mid.searcher.supported(bq(list(range(1, 40))), ok)                     # more clauses than NRTGPU_MAX_TERMS
    q = api.BooleanQuery(tuple(api.TermQuery(0, t) for t in (1, 100)), 1, (api.MaskFilter(9999),), ())
    assert not mid.searcher.supported(q, ok)                                          # FILTER mask not resident
    with pytest.raises(_lib.NrtGpuError):
        mid.searcher.supported(bq([1]), api.TopScoreDocCollectorManager(0))           # invalid argument, not "unsupported"


def test_deadline_and_diagnostics(mid, oracle):
    """A request thread states its deadline (gRPC Context / timeoutSec in the reference: SearchHandler.java:158-277,
    SearchCutoffWrapper.java:149-159); past it, calls return NRTGPU_ERR_TIMEOUT without launching work.  Every completed call
    leaves its cost for SearchResponse.Diagnostics."""
    terms = [1, 5, 100, 5000]
    mgr = api.TopScoreDocCollectorManager(50)
    try:
        api.GpuContext.set_thread_deadline(-0.001)      # already expired
        for call in (lambda: mid.searcher.search(bq(terms), mgr), lambda: mid.searcher.search_coalesced(bq(terms), mgr),
                     lambda: mid.searcher.search_batch([bq(terms)] * 3, [mgr] * 3)):
            with pytest.raises(_lib.NrtGpuError) as e:
                call()
            assert e.value.code == _lib.NRTGPU_ERR_TIMEOUT
        api.GpuContext.set_thread_deadline(30.0)        # far away: the answers are the oracle's
        got = mid.searcher.search(bq(terms), mgr)
        assert_same("deadline_far", got, oracle.search_bm25(mid.corpus, terms, 50), 50, 1000)
        d = api.GpuContext.last_diagnostics()
        assert d["queries"] == 1 and d["postings"] > 0 and d["items_maxscore"] + d["items_scan"] >= 1
        assert d["total_ms"] >= d["plan_ms"] > 0.0 and d["queue_ms"] >= 0.0
        got = mid.searcher.search_coalesced(bq(terms), mgr)
        assert_same("deadline_far_coalesced", got, oracle.search_bm25(mid.corpus, terms, 50), 50, 1000)
        assert api.GpuContext.last_diagnostics()["queries"] >= 1
    finally:
        api.GpuContext.set_thread_deadline(None)
    got = mid.searcher.search(bq(terms), mgr)           # no deadline again
    assert_same("deadline_none", got, oracle.search_bm25(mid.corpus, terms, 50), 50, 1000)
