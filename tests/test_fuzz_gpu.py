"""Seeded differential fuzz of the query shapes against the oracle: random corpora (segment counts, deletes,
ragged sizes), random clause sets (1..10 terms, boosts, repeats), numHits, thresholds, paging, masks,
minimumNumberShouldMatch, DisjunctionMaxQuery, min competitive scores -- mixed inside batches so that every kernel variant and the
batch-level switches between them are exercised together.  Bit-exact like every BM25 test.
NRT_FUZZ_ROUNDS scales the number of corpora (default 8)."""
import os

import numpy as np
import pytest

from nrtsearch_amd import _lib, api, synth
from oracle import oracle

from tests.test_parity_gpu import Index, assert_same
from tests.test_filters_gpu import accept_of, random_mask

pytestmark = pytest.mark.gpu
ROUNDS = int(os.environ.get("NRT_FUZZ_ROUNDS", "8"))


@pytest.mark.parametrize("round_", range(ROUNDS))
def test_fuzz_query_shapes(round_, monkeypatch):
    rng = np.random.Generator(np.random.PCG64(20260925 + round_))
    # which terms get membership records (segment.cpp: build_term_aux) decides between the two lookup paths of the MaxScore walk:
    # a posting per 16 docs (nearly every lookup a binary search), the default, or every term (no binary search at all)
    monkeypatch.setenv("NRTGPU_RECORD_DOCS_PER_POSTING", ["16", "128", "4096", str(1 << 30)][round_ % 4])
    if round_ % 8 >= 4:
        monkeypatch.setenv("NRTGPU_RECORD_MAX_TERMS", "3")
    n_docs = int(rng.choice([900, 1024, 5_000, 33_000, 70_001, 200_000]))
    n_seg = int(rng.integers(1, 5))
    ranks = sorted(set(int(r) for r in np.floor(np.exp(rng.uniform(0, np.log(3000), size=14))).clip(1, 3000)))
    deletes = float(rng.choice([0.0, 0.0, 0.03, 0.3]))
    corpus = synth.build_corpus(n_docs, ranks, n_segments=n_seg, delete_fraction=deletes)
    flags = int(rng.choice([0, 0, _lib.NRTGPU_FLAG_NO_LIVE_FOLD, _lib.NRTGPU_FLAG_NO_FIXED_POINT]))
    ctx = api.GpuContext(device_id=0, max_batch=64, flags=flags, target_items=int(rng.choice([0, 0, 64])))
    ix = Index(ctx, corpus)
    try:
        masks = {}
        for mid in (1, 2):
            density = float(rng.choice([0.02, 0.4, 0.97]))
            masks[mid] = [random_mask(s.max_doc, density, 1000 * round_ + 10 * mid + i) for i, s in enumerate(corpus.segments)]
            for leaf, m in zip(ix.leaves, masks[mid]):
                leaf.set_mask(mid, m)
        for batch_no in range(4):
            qs, mgrs, expect = [], [], []
            allow_msm = flags != _lib.NRTGPU_FLAG_NO_FIXED_POINT and batch_no % 2 == 0
            for _ in range(int(rng.integers(1, 9))):
                nt = int(rng.integers(1, 11))
                terms = [int(t) for t in rng.choice(ranks, size=nt, replace=True)]
                boosts = [float(np.float32(rng.choice([1.0, 1.0, 0.5, 2.0, 3.25]))) for _ in terms]
                k = int(rng.choice([1, 7, 64, 300, 1000]))
                thr = int(rng.choice([1000, 1000, 10, 2**31 - 1]))
                msm = int(rng.integers(2, nt + 2)) if (allow_msm and nt > 1 and rng.random() < 0.5) else 0
                f = int(rng.choice([0, 0, 1, 2]))
                mn = int(rng.choice([0, 0, 0, 1, 2]))
                forced = bool(f) and msm == 0
                if forced:
                    msm = 1
                clauses = tuple(api.BoostQuery(api.TermQuery(0, t), b) if b != 1.0 else api.TermQuery(0, t) for t, b in zip(terms, boosts))
                dismax = allow_msm and msm <= 1 and rng.random() < 0.25   # (same kernel variant as the clause counts: fixed point only)
                tie = 0.0
                must_flags = None
                # the second-accumulator shapes (a tie breaker > 0, MUST next to SHOULD clauses): MaxScore route only -- at most 8
                # clauses, and no ScoreMode.COMPLETE (a large query would need the exhaustive scan)
                two_ok = allow_msm and nt <= 8 and thr != 2**31 - 1
                if dismax and two_ok and rng.random() < 0.6:
                    tie = float(np.float32(rng.choice([0.05, 0.3, 0.5, 1.0])))
                if not dismax and two_ok and nt >= 2 and (msm == 0 or forced) and rng.random() < 0.3:
                    must_flags = [bool(x) for x in rng.integers(0, 2, size=nt)]
                    if all(must_flags) or not any(must_flags):
                        must_flags[0], must_flags[1] = True, False
                    msm = 0   # (MUST clauses make the hit: a FILTER next to them needs no minimumNumberShouldMatch)
                if dismax:
                    dq = api.DisjunctionMaxQuery(clauses, tie)
                    q = dq if not (f or mn) else api.BooleanQuery(must=(dq,), filter=(api.MaskFilter(f),) if f else (),
                                                                  must_not=(api.MaskFilter(mn),) if mn else ())
                    msm = 0
                elif must_flags is not None:
                    q = api.BooleanQuery(tuple(c for c, m_ in zip(clauses, must_flags) if not m_), 0, (api.MaskFilter(f),) if f else (),
                                         (api.MaskFilter(mn),) if mn else (), tuple(c for c, m_ in zip(clauses, must_flags) if m_))
                    # (the clauses as the mirror hands them over: MUST first)
                    order = [i for i, m_ in enumerate(must_flags) if m_] + [i for i, m_ in enumerate(must_flags) if not m_]
                    terms, boosts, must_flags = [terms[i] for i in order], [boosts[i] for i in order], [must_flags[i] for i in order]
                elif nt == 1 and not f and not mn and msm == 0:
                    q = clauses[0]
                else:
                    q = api.BooleanQuery(clauses, msm, (api.MaskFilter(f),) if f else (), (api.MaskFilter(mn),) if mn else ())
                acc = None
                if f or mn:
                    acc = [accept_of(s, masks[f][i] if f else None, masks[mn][i] if mn else None) for i, s in enumerate(corpus.segments)]
                after = None
                okw = dict(boosts=boosts, total_hits_threshold=thr, accept=acc, min_should_match=msm, dismax=tie if dismax else None, must=must_flags)
                if rng.random() < 0.25:
                    first = oracle.search_bm25(corpus, terms, k, **okw)
                    if len(first[0]):
                        j = int(rng.integers(0, len(first[0])))
                        after = (int(first[0][j]), float(first[1][j]))
                qs.append(q)
                mgrs.append(api.TopScoreDocCollectorManager(k, api.ScoreDoc(*after) if after else None, thr))
                expect.append((terms, k, thr, dict(okw, after=after)))
            got = ix.searcher.search_batch(qs, mgrs)
            for i, (terms, k, thr, okw) in enumerate(expect):
                assert_same(f"fuzz_{round_}_{batch_no}_{i}", got[i], oracle.search_bm25(corpus, terms, k, **okw), k, thr)
    finally:
        ix.close()
        ctx.close()


KNN_ROUNDS = int(os.environ.get("NRT_KNN_FUZZ_ROUNDS", "16"))


@pytest.mark.parametrize("round_", range(KNN_ROUNDS))
def test_fuzz_exact_vector_search(round_):
    """Seeded differential fuzz of the exact vector search against the oracle, bit for bit: ragged leaves (1 row, 15 / 16 / 17
    rows, thousands), dimensions that are no multiple of 32, k from 1 to 1024 (more than there are rows; at the nomination
    list's capacity), all four similarities, deletes, sparse ordinals, copies of rows (ties by docid), zero vectors, rows and
    queries scaled by 1e4 / 1e-4 or with one huge element next to tiny ones (what the fp16 sketch flushes), 1 - 150 queries per
    call (several panels), the sketch on and off, and the knn request path with a pre-filter and a score threshold."""
    rng = np.random.Generator(np.random.PCG64(777000 + round_))
    dim = int(rng.choice([16, 48, 64, 96, 128, 400, 768]))
    sizes = [int(rng.choice([1, 15, 16, 17, 100, 999, 4097, 20_000])) for _ in range(int(rng.integers(1, 4)))]
    style = str(rng.choice(["normal", "big", "tiny", "spiky", "clustered"]))
    flags = int(rng.choice([0, 0, 0, _lib.NRTGPU_FLAG_NO_VECTOR_SKETCH]))
    ctx = api.GpuContext(device_id=0, max_batch=64, flags=flags)
    leaves, mats, lives, base = [], [], [], 0
    try:
        for si, n in enumerate(sizes):
            v = rng.standard_normal((n, dim)).astype(np.float32)
            if style == "big":
                v *= np.float32(1e4)
            elif style == "tiny":
                v *= np.float32(1e-4)
            elif style == "spiky":          # one element dominates each row: the others fall under fp16's range after scaling
                v *= np.float32(1e-5)
                v[np.arange(n), rng.integers(0, dim, size=n)] = rng.standard_normal(n).astype(np.float32) * np.float32(300.0)
            elif style == "clustered":      # rows close to each other and far from the origin
                v = (v * np.float32(0.05) + np.float32(3.0)).astype(np.float32)
            if n >= 100:
                dup = rng.choice(n, size=n // 10, replace=False)          # copies: equal scores, the docid decides
                v[dup] = v[int(dup[0])]
                v[rng.choice(n, size=3, replace=False)] = 0.0             # zero vectors
            max_doc, o2d = n, None
            if rng.random() < 0.3 and n > 1:
                max_doc = 2 * n
                o2d = np.sort(rng.choice(max_doc, size=n, replace=False)).astype(np.int32)
            live = None
            if rng.random() < 0.4:
                live = rng.random(max_doc) > 0.15
            g = api.GpuSegment(ctx, max_doc, base)
            g.add_vectors(5, v, o2d)
            g.seal()
            if live is not None:
                padded = np.zeros(((max_doc + 63) // 64) * 64, dtype=bool)
                padded[:max_doc] = live
                g.set_live_docs(np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1))
            mask = rng.random(max_doc) < 0.5
            padded = np.zeros(((max_doc + 63) // 64) * 64, dtype=bool)
            padded[:max_doc] = mask
            g.set_mask(9, np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1))
            leaves.append(g)
            mats.append((base, v, o2d, live, mask))
            base += max_doc
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics())
        n_q = int(rng.choice([1, 3, 17, 70, 150]))
        queries = rng.standard_normal((n_q, dim)).astype(np.float32)
        if style in ("big", "tiny", "clustered"):
            queries = (queries * np.float32({"big": 1e4, "tiny": 1e-4, "clustered": 0.05}[style]) + np.float32(3.0 if style == "clustered" else 0.0)).astype(np.float32)
        queries[0] = mats[0][1][0]          # one query IS a row (distance 0, cosine 1)

        def reference(sim, q, k, use_mask=False, min_score=0.0):
            hits = []
            for b, v, o2d, live, mask in mats:
                for r in range(len(v)):
                    doc = int(o2d[r]) if o2d is not None else r
                    if (live is not None and not live[doc]) or (use_mask and not mask[doc]):
                        continue
                    s = float(oracle.vector_score(sim, q, v[r]))
                    if s >= min_score:
                        hits.append((-s, b + doc, s))
            hits.sort()
            return [(d, s) for _, d, s in hits[:k]], len(hits)

        if dim % 16 != 0:
            with pytest.raises(api.NrtGpuError) as e:
                sr.knn_exact(5, "cosine", queries, 5)
            assert e.value.code == -4      # the caller's path
            return
        total_rows = sum(len(m[1]) for m in mats)
        checked = [int(i) for i in rng.choice(n_q, size=min(n_q, 3 if total_rows > 5000 else 6), replace=False)] + [0]
        for sim_name, sim in (("cosine", 0), ("dot_product", 1), ("l2_norm", 2), ("max_inner_product", 3)):
            k = int(rng.choice([1, 7, 100, 700, 1024]))
            got = sr.knn_exact(5, sim_name, queries, k)
            for qi in checked:
                exp, total = reference(sim, queries[qi], k)
                assert got[qi].docs.tolist() == [d for d, _ in exp], (round_, sim_name, qi, style, dim, sizes, k)
                assert got[qi].scores.view(np.uint32).tolist() == np.array([s for _, s in exp], np.float32).view(np.uint32).tolist(), \
                    (round_, sim_name, qi, style)
                assert got[qi].total_hits == total
        # the knn request path: pre-filter + threshold on the unboosted score, boost applied afterwards (a power of two: exact)
        for sim_name, sim in (("cosine", 0), ("l2_norm", 2)):
            qi = checked[0]
            full, _ = reference(sim, queries[qi], 10**9, use_mask=True)
            if len(full) < 3:
                continue
            thr = full[len(full) // 2][1]
            exp = [(d, float(np.float32(s) * np.float32(2.0))) for d, s in full if s >= thr][:50]
            got1 = sr.knn_search(5, sim_name, queries[qi], 50, boost=2.0, filter=api.MaskFilter(9), min_score=thr)[0]
            assert got1.docs.tolist() == [d for d, _ in exp], (round_, sim_name, "knn_search", style)
            assert got1.scores.view(np.uint32).tolist() == np.array([s for _, s in exp], np.float32).view(np.uint32).tolist()
    finally:
        for g in leaves:
            g.release()
        ctx.close()
