"""Pins the CPU oracle against every golden value the reference's own tests hold for the hot path
(SURVEY.md section 8c).  All paths below are relative to /root/reference."""
import numpy as np
import pytest

f32 = np.float32


# ---- SmallFloat (Lucene-recall values re-derived in SURVEY Appendix A.1) -----------------------
def test_smallfloat_known_values(oracle):
    assert oracle.int_to_byte4(23) == 23 and oracle.byte4_to_int(23) == 23      # NUM_FREE_VALUES = 24
    assert oracle.int_to_byte4(24) == 24 and oracle.byte4_to_int(24) == 24
    for length, byte, decoded in [(100, 57, 96), (1000, 87, 984), (100000, 140, 98328)]:
        assert oracle.int_to_byte4(length) == byte
        assert oracle.byte4_to_int(byte) == decoded
    assert oracle.byte4_to_int(255) == 2013265944
    assert oracle.int_to_byte4(2147483647) == 255


def test_smallfloat_roundtrip_monotone(oracle):
    prev = -1
    for b in range(256):
        v = oracle.byte4_to_int(b)
        assert v > prev                      # strictly increasing decode table (LENGTH_TABLE)
        assert oracle.int_to_byte4(v) == b   # decode then encode is the identity
        prev = v
    # encoding rounds down: every length maps to the largest representable value <= length
    for length in list(range(0, 3000)) + [4000, 65535, 10**6, 10**9]:
        b = oracle.int_to_byte4(length)
        assert oracle.byte4_to_int(b) <= length
        if b < 255:
            assert oracle.byte4_to_int(b + 1) > length


def test_vectorised_int_to_byte4_matches_scalar(oracle):
    from nrtsearch_amd.synth import int_to_byte4

    lens = np.concatenate([np.arange(0, 5000), [2**k + d for k in range(3, 31) for d in (-1, 0, 1)]])
    got = int_to_byte4(lens)
    exp = np.array([oracle.int_to_byte4(int(v)) for v in lens], dtype=np.uint8)
    assert np.array_equal(got, exp)


# ---- src/test/java/com/yelp/nrtsearch/server/grpc/SearchStateTest.java:43-63,117 ---------------
def test_search_state_last_score(oracle):
    # docs: "first vendor" (dl 2), "second vendor review" (dl 3); TermQuery vendor_name:vendor
    N, n = 2, 2
    idf = oracle.bm25_idf(N, n)
    avgdl = oracle.bm25_avgdl(2 + 3, N)
    assert avgdl == f32(2.5)
    cache = oracle.bm25_norm_cache(avgdl)
    s0 = oracle.bm25_score(idf, 1.0, cache[oracle.int_to_byte4(2)])
    s1 = oracle.bm25_score(idf, 1.0, cache[oracle.int_to_byte4(3)])
    assert s0 > s1                                  # lastDocId == 1: the longer doc ranks last
    assert abs(float(s1) - 0.0766057) <= 1e-7       # assertEquals(0.0766057, lastScore, 1e-7)
    assert s1 == f32(0.0766057)                     # and it is that float exactly


# ---- src/test/java/com/yelp/nrtsearch/server/grpc/QueryTest.java:1003-1019 (explain tree) ------
def test_query_explain_tree(oracle):
    idf1 = oracle.bm25_idf(2, 1)
    idf2 = oracle.bm25_idf(2, 2)
    assert idf1 == f32(0.6931472)
    assert idf2 == f32(0.18232156)
    weight = f32(idf1 + idf2)                       # phrase weight = sum of idfs (float)
    assert weight == f32(0.87546873)
    avgdl = oracle.bm25_avgdl(8, 2)                 # dl = avgdl = 4
    cache = oracle.bm25_norm_cache(avgdl)
    ninv = cache[oracle.int_to_byte4(4)]
    # explain's tf = freq / (freq + k1 * (1 - b + b * dl / avgdl)) in float
    k1, b = f32(1.2), f32(0.75)
    tf = f32(1.0) / (f32(1.0) + k1 * ((f32(1) - b) + b * f32(4.0) / f32(4.0)))
    assert f32(tf) == f32(0.45454544)
    score = oracle.bm25_score(weight, 1.0, ninv)
    assert score == f32(0.3979403)


# ---- src/test/java/com/yelp/nrtsearch/server/similarity/SimilarityTest.java:115-135 ------------
def test_similarity_test_scores(oracle):
    # vendor_name docs: ["first vendor","first again"] and ["second vendor","second again"], dl = 4
    avgdl = oracle.bm25_avgdl(8, 2)
    cache = oracle.bm25_norm_cache(avgdl)
    ninv = cache[oracle.int_to_byte4(4)]
    vendor = oracle.bm25_score(oracle.bm25_idf(2, 2), 1.0, ninv)    # both docs
    first = oracle.bm25_score(oracle.bm25_idf(2, 1), 2.0, ninv)     # doc1 only, freq 2
    assert abs(float(vendor) - 0.0828734) < 1e-7
    # doc2 = custom 11.11 + classic 0.5 + BM25(vendor); doc1 adds BM25(first); sums in double
    doc2 = f32(float(f32(11.11)) + 0.5 + float(vendor))
    doc1 = f32(float(f32(11.11)) + float(f32(0.5)) + float(vendor) + float(first))
    assert abs(float(doc2) - 11.692873) <= 1e-4
    # testDefaultSimilarity hit 0: the classic "first" clause is absent for field vendor_name;
    # 12.12609 = 11.11 + 0.5 + 0.0828734 + 0.433217
    assert abs(float(doc1) - 12.12609) <= 1e-4


# ---- docker-compose fixture: config C1 known answer -------------------------------
def test_docker_compose_known_answer(oracle):
    # vendor_name: "first vendor", "second vendor"; N=2, dl=2, avgdl=2
    avgdl = oracle.bm25_avgdl(4, 2)
    cache = oracle.bm25_norm_cache(avgdl)
    ninv = cache[oracle.int_to_byte4(2)]
    assert ninv == f32(f32(1.0) / f32(1.2))
    s_first = oracle.bm25_score(oracle.bm25_idf(2, 1), 1.0, ninv)
    s_vendor = oracle.bm25_score(oracle.bm25_idf(2, 2), 1.0, ninv)
    assert abs(float(s_first) - 0.3150669) < 1e-7
    assert abs(float(s_vendor) - 0.0828734) < 1e-7
    doc0 = f32(float(s_first) + float(s_vendor))
    assert abs(float(doc0) - 0.3979403) < 1e-7


# ---- src/test/java/com/yelp/nrtsearch/server/script/ScoreScriptTest.java:456-460 ---------------
def test_score_script_test_values(oracle):
    avgdl = oracle.bm25_avgdl(8, 2)
    cache = oracle.bm25_norm_cache(avgdl)
    ninv = cache[oracle.int_to_byte4(4)]
    both = float(oracle.bm25_score(oracle.bm25_idf(2, 2), 1.0, ninv)) + float(oracle.bm25_score(oracle.bm25_idf(2, 1), 2.0, ninv))
    assert abs(both - 0.516) <= 1e-3
    assert abs(float(oracle.bm25_score(oracle.bm25_idf(2, 2), 1.0, ninv)) - 0.0828) <= 1e-3


# ---- src/test/java/com/yelp/nrtsearch/server/grpc/QueryTest.java:398-441 (QueryRescorer) -------
def test_query_rescore_combine(oracle):
    assert oracle.rescore_combine(5.0, True, 10.0, 1.0, 4.0) == f32(45.0)
    assert oracle.rescore_combine(5.0, False, 10.0, 1.0, 4.0) == f32(5.0)


# ---- collector semantics: src/main/java/org/apache/lucene/search/LazyQueueTopScoreDocCollector.java
def test_collector_ties_prefer_lower_doc(oracle):
    # TotalHitsThresholdTest.java:43-51,72-99: equal scores => ascending docid, EQUAL_TO with 3 hits
    c = oracle.Collector(2, None, 1000)
    for d in range(3):
        c.collect(d, 1.0)
    docs, scores, total, gte = c.topdocs()
    assert docs.tolist() == [0, 1] and total == 3 and not gte


def test_collector_threshold_relation(oracle):
    c = oracle.Collector(2, None, 2)            # threshold = max(2, numHits)
    for d in range(5):
        c.collect(d, float(d))
    docs, scores, total, gte = c.topdocs()
    assert docs.tolist() == [4, 3] and total == 5 and gte
    c = oracle.Collector(2, None, 2**31 - 1)    # COMPLETE mode: exact
    for d in range(5):
        c.collect(d, float(d))
    assert c.topdocs()[2:] == (5, False)


def test_collector_search_after(oracle):
    # RelevanceCollectorITest.java:117-190: paging by `after` yields no duplicate and no gap
    rng = np.random.default_rng(5)
    scores = rng.integers(0, 6, size=200).astype(np.float32)  # many ties
    def page(after):
        c = oracle.Collector(7, after, 1000)
        for d, s in enumerate(scores):
            c.collect(d, float(s))
        return c.topdocs()
    seen = []
    after = None
    while True:
        docs, sc, total, _ = page(after)
        assert total == 200
        if len(docs) == 0:
            break
        seen += list(zip(sc.tolist(), docs.tolist()))
        after = (int(docs[-1]), float(sc[-1]))
    expect = sorted([(float(s), d) for d, s in enumerate(scores)], key=lambda t: (-t[0], t[1]))
    assert seen == expect


def test_collector_argument_checks(oracle):
    # LazyQueueTopScoreDocCollectorManager.java:90-98
    with pytest.raises(ValueError):
        oracle.Collector(0, None, 10)
    with pytest.raises(ValueError):
        oracle.Collector(5, None, -1)


def test_topdocs_merge_order(oracle):
    a = (np.array([10, 3], np.int32), np.array([2.0, 1.0], np.float32))
    b = (np.array([7, 1], np.int32), np.array([2.0, 1.0], np.float32))
    d, s = oracle.topdocs_merge(3, [a, b])
    assert d.tolist() == [7, 10, 1] and s.tolist() == [2.0, 2.0, 1.0]


# ---- vector similarity maps: docs/field_types/vector.rst:26-35; VectorFieldDef.java:664-673 -----
def test_vector_score_maps(oracle):
    q = np.array([1.0, 0.0, 0.0], np.float32)
    v = np.array([0.0, 1.0, 0.0], np.float32)
    assert oracle.vector_score(0, q, v) == f32(0.5)        # cosine 0 -> (1+0)/2
    assert oracle.vector_score(0, q, q) == f32(1.0)
    assert oracle.vector_score(0, q, -q) == f32(0.0)
    assert oracle.vector_score(1, q, q) == f32(1.0)        # dot of unit vectors
    assert oracle.vector_score(2, q, v) == f32(1.0 / 3.0)  # 1/(1+2)
    assert oracle.vector_score(3, q, -2 * q) == f32(1.0 / 3.0)   # mip: dot=-2 -> 1/(1+2)
    assert oracle.vector_score(3, q, 2 * q) == f32(3.0)          # mip: dot=2 -> 3


def test_knn_exact_is_score_every_row_then_sort(oracle):
    """nrt_oracle_knn_exact (the C4 CPU baseline / checker) == vector_score of every live row, sorted by
    (score desc, docid asc) -- ExactVectorQuery.java:137-173 behind a top-k collector; ties lose to the earlier doc."""
    rng = np.random.default_rng(5)
    n, dim, k = 150_000, 16, 37
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    vecs[1000] = vecs[10]            # an exact tie: doc 10 must rank before doc 1000
    vecs[140_000] = vecs[10]
    q = np.vstack([vecs[10] * 2.0, rng.standard_normal(dim)]).astype(np.float32)
    live = np.ones(((n + 63) // 64) * 64, dtype=bool)
    live[[3, 70_000]] = False
    words = np.packbits(live.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1)
    for sim in range(4):
        docs, scores, cnt = oracle.knn_exact(sim, q, vecs, k, live_words=words, doc_base=7, boost=1.5, n_threads=3)
        for qi in range(2):
            # (a python loop over 150k rows is too slow: score the rows the oracle returned plus a random sample and
            # check the order properties)
            assert cnt[qi] == k
            got = [(float(scores[qi, i]), int(docs[qi, i])) for i in range(k)]
            assert got == sorted(got, key=lambda t: (-t[0], t[1]))
            for sc, d in got[:5]:
                assert live[d - 7]
                assert np.float32(oracle.vector_score(sim, q[qi], vecs[d - 7]) * np.float32(1.5)) == np.float32(sc)
            sample = rng.choice(n, size=300, replace=False)
            worst = got[-1]
            for r in sample.tolist():
                if not live[r] or (r + 7) in set(docs[qi].tolist()):
                    continue
                sc = float(np.float32(oracle.vector_score(sim, q[qi], vecs[r]) * np.float32(1.5)))
                assert (sc, -(r + 7)) < (worst[0], -worst[1]) or sc < worst[0]
        if sim in (0, 1, 3):   # the planted ties of query 0 (cosine of identical directions / equal dots)
            top = docs[0].tolist()
            assert top.index(17) < top.index(1007) < top.index(140_007)


def test_segment_search_matches_bruteforce(oracle):
    from nrtsearch_amd import synth

    corpus = synth.build_corpus(20000, [1, 2, 7, 40, 300], n_segments=3, delete_fraction=0.05)
    terms = [1, 7, 300]
    docs, scores, total, gte = oracle.search_bm25(corpus, terms, 50, total_hits_threshold=2**31 - 1)
    # brute force in numpy with float32 ops + float64 sums
    w, cache = oracle.bm25_query_stats(corpus, terms)
    allhits = []
    for seg in corpus.segments:
        acc = np.zeros(seg.max_doc, np.float64)
        m = np.zeros(seg.max_doc, bool)
        for wi, t in zip(w, terms):
            d, f = seg.postings(t)
            ninv = cache[seg.norms[d]]
            s = wi - wi / (f32(1.0) + f.astype(np.float32) * ninv)
            assert s.dtype == np.float32
            acc[d] += s.astype(np.float64)
            m[d] = True
        live = np.unpackbits(seg.live_bits.view(np.uint8), bitorder="little")[: seg.max_doc].astype(bool)
        m &= live
        idx = np.nonzero(m)[0]
        allhits += [(float(f32(acc[i])), int(i) + seg.doc_base) for i in idx]
    allhits.sort(key=lambda t: (-t[0], t[1]))
    assert total == len(allhits) and not gte
    assert docs.tolist() == [d for _, d in allhits[:50]]
    assert scores.tolist() == [s for s, _ in allhits[:50]]


def test_dismax_matches_bruteforce(oracle):
    """DisjunctionMaxQuery over term clauses (QueryNodeMapper.java:350-358): the oracle against a numpy restatement of
    DisjunctionMaxScorer.score() -- best clause (float) + tie breaker x the double sum of the others, one cast."""
    from nrtsearch_amd import synth

    corpus = synth.build_corpus(25000, [1, 3, 9, 50, 350], n_segments=3, delete_fraction=0.04)
    terms, boosts = [1, 9, 350, 9], [1.0, 2.5, 0.5, 1.0]      # (a repeated term is two disjuncts)
    w, cache = oracle.bm25_query_stats(corpus, terms, boosts)
    for tie in (0.0, 0.25):
        hits = []
        for seg in corpus.segments:
            best = np.zeros(seg.max_doc, np.float32)
            other = np.zeros(seg.max_doc, np.float64)
            m = np.zeros(seg.max_doc, bool)
            for wi, t in zip(w, terms):
                d, f = seg.postings(t)
                sc = (wi - wi / (f32(1.0) + f.astype(np.float32) * cache[seg.norms[d]])).astype(np.float32)
                ge = sc >= best[d]
                other[d] += np.where(ge, best[d], sc).astype(np.float64)
                best[d] = np.where(ge, sc, best[d])
                m[d] = True
            live = np.unpackbits(seg.live_bits.view(np.uint8), bitorder="little")[: seg.max_doc].astype(bool)
            for i in np.nonzero(m & live)[0]:
                hits.append((float(f32(np.float64(best[i]) + other[i] * np.float64(f32(tie)))), int(i) + seg.doc_base))
        hits.sort(key=lambda t: (-t[0], t[1]))
        docs, scores, total, gte = oracle.search_bm25(corpus, terms, 40, boosts=boosts, total_hits_threshold=2**31 - 1, dismax=tie)
        assert total == len(hits) and not gte
        assert docs.tolist() == [d for _, d in hits[:40]]
        assert scores.tolist() == [s for s, _ in hits[:40]]
    # tie breaker 0: the score IS the best clause's score, never above a single clause's maximum
    docs, scores, _, _ = oracle.search_bm25(corpus, terms, 5, boosts=boosts, dismax=0.0)
    plain = oracle.search_bm25(corpus, terms, 5, boosts=boosts)
    assert scores[0] <= plain[1][0]


def test_query_shapes_match_bruteforce(oracle):
    """minimumNumberShouldMatch, FILTER / MUST_NOT doc sets, searchAfter and repeated clauses of the oracle
    against an independent numpy restatement (float32 term scores, float64 sums, one cast, HitQueue order)."""
    from nrtsearch_amd import synth

    corpus = synth.build_corpus(30000, [1, 2, 5, 11, 60, 400], n_segments=3, delete_fraction=0.03)
    fm = [synth.random_mask(s.max_doc, 0.4, 70 + i) for i, s in enumerate(corpus.segments)]
    mn = [synth.random_mask(s.max_doc, 0.1, 80 + i) for i, s in enumerate(corpus.segments)]

    def brute(terms, msm, use_f, use_mn):
        w, cache = oracle.bm25_query_stats(corpus, terms)
        hits = []
        for si, seg in enumerate(corpus.segments):
            acc = np.zeros(seg.max_doc, np.float64)
            cnt = np.zeros(seg.max_doc, np.int32)
            for wi, t in zip(w, terms):          # a repeated term is two clauses: counted and scored twice
                d, f = seg.postings(t)
                acc[d] += (wi - wi / (f32(1.0) + f.astype(np.float32) * cache[seg.norms[d]])).astype(np.float64)
                cnt[d] += 1
            ok = cnt >= max(msm, 1)
            for words, keep in ((seg.live_bits, True), (fm[si] if use_f else None, True), (mn[si] if use_mn else None, False)):
                if words is not None:
                    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[: seg.max_doc].astype(bool)
                    ok &= bits if keep else ~bits
            hits += [(float(f32(acc[i])), int(i) + seg.doc_base) for i in np.nonzero(ok)[0]]
        hits.sort(key=lambda t: (-t[0], t[1]))
        return hits

    for terms, msm, use_f, use_mn in (([1, 5, 60], 2, False, False), ([1, 2, 5, 11, 400], 4, True, False),
                                      ([2, 2, 60], 2, False, True), ([1, 11, 60, 400], 1, True, True)):
        acc = [synth.accept_words(s, fm[i] if use_f else None, mn[i] if use_mn else None) for i, s in enumerate(corpus.segments)] \
            if (use_f or use_mn) else None
        exp = brute(terms, msm, use_f, use_mn)
        docs, scores, total, gte = oracle.search_bm25(corpus, terms, 40, total_hits_threshold=2**31 - 1, min_should_match=msm, accept=acc)
        assert total == len(exp) and not gte
        assert docs.tolist() == [d for _, d in exp[:40]] and scores.tolist() == [s for s, _ in exp[:40]]
        # page 2: searchAfter the 25th hit
        if len(exp) > 30:
            after = (exp[24][1], exp[24][0])
            d2, s2, t2, _ = oracle.search_bm25(corpus, terms, 10, total_hits_threshold=2**31 - 1, min_should_match=msm, accept=acc, after=after)
            assert d2.tolist() == [d for _, d in exp[25:35]] and s2.tolist() == [s for s, _ in exp[25:35]] and t2 == len(exp)


def test_must_next_to_should_matches_bruteforce(oracle):
    """MUST next to SHOULD term clauses (QueryNodeMapper.java:257-283, minimumNumberShouldMatch 0): the oracle against a numpy
    restatement of ReqOptSumScorer.score() [Lucene-recall]: float32 term scores, a float64 sum per side, each cast to float32,
    the two added in float32; a hit matches every MUST clause."""
    from nrtsearch_amd import synth

    corpus = synth.build_corpus(30000, [1, 2, 5, 11, 60, 400], n_segments=3, delete_fraction=0.03)
    for terms, must, boosts in (([5, 1, 60], [True, False, False], None), ([2, 11, 1, 400, 60], [True, True, False, False, False], [1.0, 2.0, 0.5, 3.0, 1.0]),
                                ([60, 60, 1], [True, False, False], None), ([1, 2, 5], [False, True, False], None), ([400, 7777, 1], [True, True, False], None)):
        w, cache = oracle.bm25_query_stats(corpus, terms, boosts)
        hits, differs = [], 0
        for seg in corpus.segments:
            req = np.zeros(seg.max_doc, np.float64)
            opt = np.zeros(seg.max_doc, np.float64)
            n_req = np.zeros(seg.max_doc, np.int32)
            n_opt = np.zeros(seg.max_doc, np.int32)
            for wi, t, m in zip(w, terms, must):
                d, f = seg.postings(t)
                sc = (wi - wi / (f32(1.0) + f.astype(np.float32) * cache[seg.norms[d]])).astype(np.float64)
                if m:
                    req[d] += sc
                    n_req[d] += 1
                else:
                    opt[d] += sc
                    n_opt[d] += 1
            live = np.unpackbits(seg.live_bits.view(np.uint8), bitorder="little")[: seg.max_doc].astype(bool)
            for i in np.nonzero((n_req == sum(must)) & live)[0]:
                score = f32(req[i]) + (f32(opt[i]) if n_opt[i] else f32(0.0))   # float32 + float32
                differs += int(score != f32(req[i] + opt[i]))
                hits.append((float(score), int(i) + seg.doc_base))
        hits.sort(key=lambda t: (-t[0], t[1]))
        docs, scores, total, gte = oracle.search_bm25(corpus, terms, 40, boosts=boosts, total_hits_threshold=2**31 - 1, must=must)
        assert total == len(hits) and not gte
        assert docs.tolist() == [d for _, d in hits[:40]] and scores.tolist() == [s for s, _ in hits[:40]]
        if 7777 in terms:
            assert total == 0     # a MUST term the index does not hold
        elif len(terms) == 5:
            assert differs > 0    # (the two-float addition is NOT the one-sum score: the shape needs its own restatement)
    # every clause MUST: the conjunction == minimumNumberShouldMatch = n
    a = oracle.search_bm25(corpus, [1, 5, 60], 30, must=[True, True, True])
    b = oracle.search_bm25(corpus, [1, 5, 60], 30, min_should_match=3)
    assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist() and a[2] == b[2]


# ---- tests/golden: the catalogue of the reference's golden values and the frozen oracle fixture ----------
def _golden(name):
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)) as f:
        return json.load(f)


def test_reference_vector_catalogue(oracle):
    g = _golden("reference_vectors.json")
    v = g["bm25_scalar"][0]        # SearchStateTest.java:117
    idf = oracle.bm25_idf(v["doc_count"], v["doc_freq"])
    cache = oracle.bm25_norm_cache(oracle.bm25_avgdl(v["sum_total_term_freq"], v["doc_count"]))
    s = oracle.bm25_score(idf, float(v["freq"]), cache[oracle.int_to_byte4(v["field_length"])])
    assert abs(float(s) - v["expected_score"]) <= v["tolerance"] and s == f32(v["expected_score"])
    v = g["bm25_scalar"][1]        # QueryTest.java:1003-1019
    assert oracle.bm25_idf(2, 1) == f32(v["idf_n1"]) and oracle.bm25_idf(2, 2) == f32(v["idf_n2"])
    cache = oracle.bm25_norm_cache(f32(v["avgdl"]))
    w = f32(f32(v["idf_n1"]) + f32(v["idf_n2"]))
    assert oracle.bm25_score(w, 1.0, cache[oracle.int_to_byte4(v["field_length"])]) == f32(v["expected_score"])
    v = g["rescore"][0]            # QueryTest.java:398-441
    assert oracle.rescore_combine(v["first"], True, v["second"], v["query_weight"], v["rescore_weight"]) == f32(v["expected"])
    for length, byte, decoded in g["smallfloat"][0]["pairs_length_byte_decoded"]:
        assert oracle.int_to_byte4(length) == byte and oracle.byte4_to_int(byte) == decoded
    assert oracle.byte4_to_int(255) == g["smallfloat"][0]["byte4_to_int_255"]


def test_oracle_matches_its_frozen_fixture(oracle):
    from nrtsearch_amd import synth

    g = _golden("oracle_small_corpus.json")
    sp = g["spec"]
    corpus = synth.build_corpus(sp["n_docs"], sp["ranks"], n_segments=sp["n_segments"],
                                delete_fraction=sp["delete_fraction"], seed=sp["seed"])
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_fixtures", os.path.join(here, "make_oracle_fixtures.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert len(g["cases"]) >= 14
    for c in g["cases"]:
        (docs, scores, total, gte), _ = gen.run_case(corpus, c)
        assert docs.tolist() == c["docs"] and scores.view(np.uint32).tolist() == c["score_bits"]
        assert int(total) == c["total_hits"] and bool(gte) == c["relation_gte"]
