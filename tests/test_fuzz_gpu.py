"""Seeded differential fuzz of the query shapes against the oracle: random corpora (segment counts, deletes,
ragged sizes), random clause sets (1..10 terms, boosts, repeats), numHits, thresholds, paging, masks,
minimumNumberShouldMatch, DisjunctionMaxQuery, min competitive scores -- mixed inside batches so that every kernel variant and the
batch-level switches between them are exercised together.  Bit-exact like every BM25 test.
NRT_FUZZ_ROUNDS scales the number of corpora (default 6)."""
import os

import numpy as np
import pytest

from nrtsearch_amd import _lib, api, synth
from oracle import oracle

from tests.test_parity_gpu import Index, assert_same
from tests.test_filters_gpu import accept_of, random_mask

pytestmark = pytest.mark.gpu
ROUNDS = int(os.environ.get("NRT_FUZZ_ROUNDS", "6"))


@pytest.mark.parametrize("round_", range(ROUNDS))
def test_fuzz_query_shapes(round_):
    rng = np.random.Generator(np.random.PCG64(20260925 + round_))
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
                if f and msm == 0:
                    msm = 1
                clauses = tuple(api.BoostQuery(api.TermQuery(0, t), b) if b != 1.0 else api.TermQuery(0, t) for t, b in zip(terms, boosts))
                dismax = allow_msm and msm <= 1 and rng.random() < 0.25   # (same kernel variant as the clause counts: fixed point only)
                if dismax:
                    dq = api.DisjunctionMaxQuery(clauses)
                    q = dq if not (f or mn) else api.BooleanQuery(must=(dq,), filter=(api.MaskFilter(f),) if f else (),
                                                                  must_not=(api.MaskFilter(mn),) if mn else ())
                    msm = 0
                elif nt == 1 and not f and not mn and msm == 0:
                    q = clauses[0]
                else:
                    q = api.BooleanQuery(clauses, msm, (api.MaskFilter(f),) if f else (), (api.MaskFilter(mn),) if mn else ())
                acc = None
                if f or mn:
                    acc = [accept_of(s, masks[f][i] if f else None, masks[mn][i] if mn else None) for i, s in enumerate(corpus.segments)]
                after = None
                okw = dict(boosts=boosts, total_hits_threshold=thr, accept=acc, min_should_match=msm, dismax=0.0 if dismax else None)
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
