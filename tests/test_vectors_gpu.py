"""Exact vector search / vector rescore on the GPU vs the oracle (scalar left-to-right fp32, one member of Lucene's
tolerance class -- SURVEY A.7).  The exact SEARCH is bit-exact: the matrix cores only nominate rows, every returned score is
recomputed in the oracle's order and the answer is certified against the rows left out, so docs, ranks and
score bits are the oracle's for all four similarities.  The vector RESCORE (one wave per hit, lanes striding the dimensions)
keeps a tolerance: 1e-5 relative (+1e-6 absolute), ranks identical except among hits closer than that.
Reference for the tolerance: src/test/java/com/yelp/nrtsearch/server/field/VectorFieldDefTest.java:1917 (1e-4)."""
import numpy as np
import pytest

from nrtsearch_amd import api, synth

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 1e-6   # the rescore path only


@pytest.fixture(scope="module")
def ctx():
    c = api.GpuContext(device_id=0, max_batch=64)
    yield c
    c.close()


def brute_force(oracle, sim, q, segs, k, boost=1.0):
    """The in-test oracle of VectorFieldDefTest.java:2283-2330: score every vector, sort."""
    hits = []
    for base, vecs, ord_to_doc, live in segs:
        for r in range(len(vecs)):
            doc = int(ord_to_doc[r]) if ord_to_doc is not None else r
            if live is not None and not live[doc]:
                continue
            hits.append((float(np.float32(oracle.vector_score(sim, q, vecs[r]) * np.float32(boost))), base + doc))
    hits.sort(key=lambda t: (-t[0], t[1]))
    return hits[:k], len(hits), {d: s for s, d in hits}


def check_hits(got: api.TopDocs, exp):
    """Bit for bit: the oracle's docs in the oracle's order (score desc, docid asc among equal scores) with the oracle's scores."""
    assert got.docs.tolist() == [d for _, d in exp]
    assert got.scores.view(np.uint32).tolist() == np.array([s for s, _ in exp], dtype=np.float32).view(np.uint32).tolist()


def make_segments(rng, n_list, dim, sparse_ords=False, deletes=False):
    segs, base = [], 0
    for si, n in enumerate(n_list):
        vecs = rng.standard_normal((n, dim)).astype(np.float32)
        max_doc = n if not (sparse_ords and si == 1) else 2 * n
        ord_to_doc = None
        if sparse_ords and si == 1:
            ord_to_doc = np.sort(rng.choice(max_doc, size=n, replace=False)).astype(np.int32)
        live = None
        if deletes and si == 0:
            live = rng.random(max_doc) > 0.1
        segs.append((base, vecs, ord_to_doc, live, max_doc))
        base += max_doc
    return segs


def upload(ctx, segs, field=3, normalize=False):
    leaves = []
    for base, vecs, ord_to_doc, live, max_doc in segs:
        g = api.GpuSegment(ctx, max_doc, base)
        v = vecs
        if normalize:
            v = (vecs / np.linalg.norm(vecs, axis=1, keepdims=True)).astype(np.float32)
        g.add_vectors(field, v, ord_to_doc)
        g.seal()
        if live is not None:
            padded = np.zeros(((max_doc + 63) // 64) * 64, dtype=bool)
            padded[:max_doc] = live
            g.set_live_docs(np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1))
        leaves.append(g)
    return leaves


@pytest.mark.parametrize("sim_name,sim", [("cosine", 0), ("dot_product", 1), ("l2_norm", 2), ("max_inner_product", 3)])
def test_knn_exact_matches_bruteforce(ctx, oracle, sim_name, sim):
    rng = np.random.default_rng(12345678)          # VectorFieldDefTest.java:122 uses Random(12345678L)
    dim = 64
    segs = make_segments(rng, [3000, 1500, 700], dim, sparse_ords=True, deletes=True)
    normalize = sim == 1
    leaves = upload(ctx, segs, normalize=normalize)
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics())
    queries = rng.standard_normal((5, dim)).astype(np.float32)
    if normalize:
        queries = (queries / np.linalg.norm(queries, axis=1, keepdims=True)).astype(np.float32)
    osegs = [(b, (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32) if normalize else v, o, l)
             for b, v, o, l, _ in segs]
    for k in (1, 10, 100):
        got = sr.knn_exact(3, sim_name, queries, k, boost=1.5)
        for qi in range(len(queries)):
            exp, total, ref = brute_force(oracle, sim, queries[qi], osegs, k, boost=1.5)
            check_hits(got[qi], exp)
            assert got[qi].total_hits == total and not got[qi].relation_gte   # live docs with a vector (segment 0 has deletes)
    for g in leaves:
        g.release()


def test_knn_768d_many_queries_and_rounds(ctx, oracle):
    # d = 768 (config C4 shape), 40 queries (two MFMA panels), enough rows for several rounds
    rng = np.random.default_rng(777)
    dim, n = 768, 90_000
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    g = api.GpuSegment(ctx, n, 0)
    g.add_vectors(0, vecs)
    g.seal()
    sr = api.GpuIndexSearcher(ctx, [g], api.IndexStatistics())
    queries = rng.standard_normal((40, dim)).astype(np.float32)
    got = sr.knn_exact(0, "cosine", queries, 100)
    st = ctx.stats()
    assert st["knn_second_passes"] == 0       # random rows: the nominations certify every answer
    odocs, oscores, ocnt = oracle.knn_exact(0, queries[[0, 17, 39]], vecs, 100, n_threads=8)
    for j, qi in enumerate((0, 17, 39)):
        assert got[qi].docs.tolist() == odocs[j].tolist()
        assert got[qi].scores.view(np.uint32).tolist() == oscores[j].view(np.uint32).tolist()
        assert got[qi].total_hits == n
    g.release()


def test_knn_sketch_and_fp32_nominations_give_the_same_bits(oracle):
    """The pass over all rows nominates from the fp16 sketch (half the bytes) unless the context keeps none
    (NRTGPU_FLAG_NO_VECTOR_SKETCH); the answer is the rescored, certified one either way: identical bits, and the oracle's.
    Dimensions that are no multiple of 32 / 128 (zero-padded steps), 1 - 4 query panels, deletes, sparse ordinals."""
    from nrtsearch_amd import _lib
    rng = np.random.default_rng(2024)
    for dim, n_q, k in ((48, 3, 7), (96, 17, 30), (384, 33, 64), (768, 64, 100)):
        segs = make_segments(rng, [5000, 2600], dim, sparse_ords=True, deletes=True)
        queries = rng.standard_normal((n_q, dim)).astype(np.float32)
        answers = []
        for flags in (0, _lib.NRTGPU_FLAG_NO_VECTOR_SKETCH):
            c = api.GpuContext(device_id=0, max_batch=64, flags=flags)
            leaves = upload(c, segs)
            sr = api.GpuIndexSearcher(c, leaves, api.IndexStatistics())
            per_sim = {}
            for sim_name in ("cosine", "l2_norm", "max_inner_product"):
                c.reset_stats()
                per_sim[sim_name] = sr.knn_exact(3, sim_name, queries, k)
                st = c.stats()
                assert (st["knn_sketch_launches"] > 0) == (flags == 0), (dim, sim_name, st)
                assert st["knn_second_passes"] == 0
            answers.append(per_sim)
            for g in leaves:
                g.release()
            c.close()
        for sim_name, sim in (("cosine", 0), ("l2_norm", 2), ("max_inner_product", 3)):
            for qi in range(n_q):
                a, b = answers[0][sim_name][qi], answers[1][sim_name][qi]
                assert a.docs.tolist() == b.docs.tolist() and a.scores.view(np.uint32).tolist() == b.scores.view(np.uint32).tolist()
            for qi in (0, n_q - 1):
                exp, total, _ = brute_force(oracle, sim, queries[qi], [(b_, v, o, l) for b_, v, o, l, _ in segs], k)
                check_hits(answers[0][sim_name][qi], exp)


def test_knn_euclidean_near_duplicates_and_large_norms(ctx, oracle):
    """|q|^2 + |v|^2 - 2 q.v cancels when rows sit close to the query and far from the origin: norms ~ 1e4, distances ~ 1e-2,
    so the matrix-core estimate of |q - v|^2 is all rounding.  The answer must still be the oracle's, bit for bit: the
    estimate only nominates, and what it cannot separate within its error bound goes through the second pass."""
    rng = np.random.default_rng(404)
    dim, n, k = 64, 20_000, 25
    centre = (rng.standard_normal(dim) * 12.0 + 100.0).astype(np.float32)            # |centre|^2 ~ 6.5e5
    vecs = (centre + rng.standard_normal((n, dim)).astype(np.float32) * np.float32(0.02)).astype(np.float32)
    vecs[n // 2:] = rng.standard_normal((n - n // 2, dim)).astype(np.float32)          # half of the rows are ordinary
    queries = np.stack([centre, (centre + np.float32(0.01)).astype(np.float32), rng.standard_normal(dim).astype(np.float32)])
    g = api.GpuSegment(ctx, n, 0)
    g.add_vectors(0, vecs)
    g.seal()
    sr = api.GpuIndexSearcher(ctx, [g], api.IndexStatistics())
    ctx.reset_stats()
    got = sr.knn_exact(0, "l2_norm", queries, k)
    odocs, oscores, _ = oracle.knn_exact(2, queries, vecs, k, n_threads=4)
    for qi in range(len(queries)):
        assert got[qi].docs.tolist() == odocs[qi].tolist()
        assert got[qi].scores.view(np.uint32).tolist() == oscores[qi].view(np.uint32).tolist()
    assert ctx.stats()["knn_second_passes"] >= 1     # the cluster's 10,000 rows all lie within the bound of each other
    # the estimate alone cannot rank these rows: one ulp of |q|^2 is larger than the winner's whole squared distance
    q64, v64 = queries[0].astype(np.float64), vecs[odocs[0][0]].astype(np.float64)
    assert np.spacing(np.float32(q64 @ q64)) > ((q64 - v64) ** 2).sum()
    g.release()


def test_knn_more_equal_rows_than_nominations(oracle):
    """3000 copies of one row, all of them the best match: the nominations (k + max(32, k / 2) rows) cannot certify the answer,
    the second pass sees every copy, and ties go to the lowest docids as in the oracle.  All four similarities, two leaves.
    The second pass runs from the sketch as well (the fp32 rows are only read when its wider bound overflows a list)."""
    ctx = api.GpuContext(device_id=0, max_batch=64, collect_timing=True)    # (timing on: the launches are counted by kind)
    rng = np.random.default_rng(7)
    dim, k = 32, 40
    base_rows = rng.standard_normal((5000, dim)).astype(np.float32)
    star = rng.standard_normal(dim).astype(np.float32)
    where = np.sort(rng.choice(5000, size=3000, replace=False))
    base_rows[where] = star
    for sim_name, sim in (("cosine", 0), ("dot_product", 1), ("l2_norm", 2), ("max_inner_product", 3)):
        rows = base_rows
        q = star.copy()
        if sim == 1:
            rows = (rows / np.linalg.norm(rows, axis=1, keepdims=True)).astype(np.float32)
            q = (q / np.linalg.norm(q)).astype(np.float32)
        leaves = []
        for lo, hi in ((0, 2048), (2048, 5000)):
            g = api.GpuSegment(ctx, hi - lo, lo)
            g.add_vectors(0, rows[lo:hi])
            g.seal()
            leaves.append(g)
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics())
        ctx.reset_stats()
        got = sr.knn_exact(0, sim_name, q[None, :], k, boost=2.0)[0]
        odocs, oscores, _ = oracle.knn_exact(sim, q[None, :], rows, k, boost=2.0)
        assert got.docs.tolist() == odocs[0].tolist() == where[:k].tolist()
        assert got.scores.view(np.uint32).tolist() == oscores[0].view(np.uint32).tolist()
        st = ctx.stats()
        assert st["knn_second_passes"] == 1
        assert st["knn_score_launches"] == st["knn_sketch_launches"] == 2   # one nomination launch, one second pass: both from the sketch
        for g in leaves:
            g.release()
    ctx.close()


def test_vector_rescore_matches_queryrescore(ctx, oracle):
    # QueryTest.java:398-441 shape: final = queryWeight * first + rescoreWeight * second, window trims
    rng = np.random.default_rng(5)
    dim = 32
    segs = make_segments(rng, [400, 300], dim, sparse_ords=True)
    leaves = upload(ctx, segs)
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics())
    q = rng.standard_normal(dim).astype(np.float32)
    total_docs = segs[-1][0] + segs[-1][4]
    docs = np.sort(rng.choice(total_docs, size=120, replace=False)).astype(np.int32)
    first = rng.random(120).astype(np.float32) * 5
    hits = api.TopDocs(docs, first, 120, False)
    got = sr.rescore_vectors(hits, 3, "cosine", q, window=50, query_weight=1.0, rescore_weight=4.0)
    exp = []
    for doc, f in zip(docs.tolist(), first.tolist()):
        second, matched = 0.0, False
        for base, vecs, ord_to_doc, live, max_doc in segs:
            if base <= doc < base + max_doc:
                local = doc - base
                row = local if ord_to_doc is None else (int(np.searchsorted(ord_to_doc, local)) if local in set(ord_to_doc.tolist()) else -1)
                if 0 <= row < len(vecs):
                    second, matched = float(oracle.vector_score(0, q, vecs[row])), True
        exp.append((float(oracle.rescore_combine(f, matched, second, 1.0, 4.0)), doc))
    exp.sort(key=lambda t: (-t[0], t[1]))
    assert len(got.docs) == 50
    assert np.allclose(got.scores, [s for s, _ in exp[:50]], rtol=1e-5, atol=1e-6)
    assert len(set(got.docs.tolist()) & set(d for _, d in exp[:50])) >= 49
    for g in leaves:
        g.release()


def test_knn_segment_with_more_than_24_rounds(ctx):
    # 6.6M rows in ONE segment = 26 rounds of <= 256k rows (the round size once overflowed after 24 and hung)
    rng = np.random.default_rng(9)
    n, dim = 6_600_000, 16
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    g = api.GpuSegment(ctx, n, 0)
    g.add_vectors(0, vecs)
    g.seal()
    sr = api.GpuIndexSearcher(ctx, [g], api.IndexStatistics())
    qs = rng.standard_normal((3, dim)).astype(np.float32)
    got = sr.knn_exact(0, "max_inner_product", qs, 10)
    for qi in range(3):
        dots = vecs @ qs[qi]
        scores = np.where(dots < 0, 1.0 / (1.0 - dots), dots + 1.0).astype(np.float32)
        order = np.lexsort((np.arange(n), -scores))[:10]
        assert np.allclose(got[qi].scores, scores[order], rtol=1e-5, atol=1e-6)
        assert len(set(got[qi].docs.tolist()) & set(order.tolist())) >= 9
        assert got[qi].total_hits == n
    g.release()


def test_knn_rows_ordered_by_rising_similarity(ctx):
    """Long rounds assume rows in no particular order; here EVERY row beats the running theta (dot product grows with the
    row), so a long round overflows the candidate list: the library must notice and redo the panel in bounded rounds."""
    n, dim, k = 700_000, 16, 100
    t = (np.arange(n, dtype=np.float64) + 1.0) / (n + 1.0)
    vecs = np.zeros((n, dim), dtype=np.float32)
    vecs[:, 0] = t
    vecs[:, 1] = np.sqrt(1.0 - t * t)
    g = api.GpuSegment(ctx, n, 0)
    g.add_vectors(0, vecs)
    g.seal()
    sr = api.GpuIndexSearcher(ctx, [g], api.IndexStatistics())
    q = np.zeros((2, dim), dtype=np.float32)
    q[0, 0] = 1.0            # dot = t: rising
    q[1, 1] = 1.0            # dot = sqrt(1 - t^2): falling (nothing after the first round is competitive)
    got = sr.knn_exact(0, "dot_product", q, k)
    for qi in range(2):
        dots = vecs.astype(np.float64) @ q[qi].astype(np.float64)
        scores = np.maximum((1.0 + dots) / 2.0, 0.0)
        order = np.lexsort((np.arange(n), -scores))[:k]
        assert np.allclose(got[qi].scores, scores[order].astype(np.float32), rtol=1e-5, atol=1e-6)
        # fp32 scores of neighbouring rows tie; the docs must come from the reference's near-tie set
        lo = scores[order[-1]] * (1 - 1e-5) - 1e-6
        assert all(scores[d] >= lo for d in got[qi].docs.tolist())
        assert len(set(got[qi].docs.tolist())) == k
        assert got[qi].total_hits == n
    g.release()


def test_knn_theta_unknown_after_the_first_round(ctx):
    """Fewer than k live rows among the first 65536: theta is still unknown when the long rounds start, every row keeps its
    own slot and a round longer than the list must be caught (not silently truncated)."""
    rng = np.random.default_rng(21)
    n, dim, k = 500_000, 16, 64
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    live = np.ones(n, dtype=bool)
    live[:70_000] = False
    live[5:45] = True          # 40 live rows < k in the first round
    g = api.GpuSegment(ctx, n, 0)
    g.add_vectors(0, vecs)
    g.seal()
    padded = np.zeros(((n + 63) // 64) * 64, dtype=bool)
    padded[:n] = live
    g.set_live_docs(np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1))
    sr = api.GpuIndexSearcher(ctx, [g], api.IndexStatistics())
    q = rng.standard_normal((1, dim)).astype(np.float32)
    got = sr.knn_exact(0, "max_inner_product", q, k)[0]
    dots = vecs.astype(np.float64) @ q[0].astype(np.float64)
    scores = np.where(dots < 0, 1.0 / (1.0 - dots), dots + 1.0)
    scores[~live] = -1.0
    order = np.lexsort((np.arange(n), -scores))[:k]
    assert np.allclose(got.scores, scores[order].astype(np.float32), rtol=1e-5, atol=1e-6)
    assert len(set(got.docs.tolist()) & set(order.tolist())) >= k - 1
    assert got.total_hits == int(live.sum())
    g.release()


def test_knn_large_dimensions_one_panel(ctx, oracle):
    """d = 1536 and 2048 (above 1280 the query panel is one 16-query MFMA panel per pass: LDS): 20 queries = two passes,
    against fp64 numpy; d = 2064 is refused."""
    rng = np.random.default_rng(31)
    for dim in (1536, 2048):
        n = 30_000
        vecs = rng.standard_normal((n, dim)).astype(np.float32)
        g = api.GpuSegment(ctx, n, 0)
        g.add_vectors(0, vecs)
        g.seal()
        sr = api.GpuIndexSearcher(ctx, [g], api.IndexStatistics())
        qs = rng.standard_normal((20, dim)).astype(np.float32)
        got = sr.knn_exact(0, "cosine", qs, 50)
        vn = np.linalg.norm(vecs.astype(np.float64), axis=1)
        for qi in (0, 15, 16, 19):
            q = qs[qi].astype(np.float64)
            sc = np.maximum((1.0 + (vecs.astype(np.float64) @ q) / (vn * np.linalg.norm(q))) / 2.0, 0.0)
            order = np.lexsort((np.arange(n), -sc))[:50]
            assert np.allclose(got[qi].scores, sc[order], rtol=2e-5, atol=2e-6)
            assert len(set(got[qi].docs.tolist()) & set(order.tolist())) >= 49
            assert got[qi].total_hits == n
        odocs, oscores, _ = oracle.knn_exact(0, qs[[3, 18]], vecs, 50, n_threads=8)   # and the oracle's bits
        for j, qi in enumerate((3, 18)):
            assert got[qi].docs.tolist() == odocs[j].tolist()
            assert got[qi].scores.view(np.uint32).tolist() == oscores[j].view(np.uint32).tolist()
        g.release()
    g = api.GpuSegment(ctx, 8, 0)
    with pytest.raises(api.NrtGpuError) as e:   # more than 2048 dimensions: the field stays on the caller's path (refused at upload)
        g.add_vectors(0, np.ones((8, 2064), np.float32))
    assert e.value.code == -4
    g.release()


def test_knn_dimensions_that_are_no_multiple_of_16(ctx, oracle):
    """Round 3 refused them (NRTGPU_ERR_UNSUPPORTED -> Lucene); the reference's own vector tests run at d = 3
    (VectorFieldDefTest.java:1885-1919: 10 000 docs, query (0.25, 0.5, 0.75), top 5, EUCLIDEAN; :1925-1965 normalized cosine).
    Rows are resident zero-padded to a multiple of 16 elements and queries are padded alike: zeros add nothing to any of the four
    similarities' sums, so the answers are the field's own -- the oracle's docids and score BITS at the field's dimension."""
    rng = np.random.default_rng(1885)
    for dim, n in ((3, 10_000), (5, 3_000), (17, 3_000), (100, 4_000), (770, 2_000)):
        vecs = rng.standard_normal((n, dim)).astype(np.float32)
        if dim == 3:
            vecs[:64] = np.float32(0.5) * vecs[64:128]        # ties among the cosine scores, near-duplicates
        unit = (vecs / np.linalg.norm(vecs, axis=1, keepdims=True)).astype(np.float32)
        live = np.ones(((n + 63) // 64) * 64, dtype=bool)
        live[n:] = False
        live[rng.choice(n, n // 50, replace=False)] = False
        live_words = np.packbits(live.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1)
        g = api.GpuSegment(ctx, n, 7)
        g.add_vectors(0, vecs)
        g.add_vectors(1, unit)
        g.seal()
        g.set_live_docs(live_words)
        sr = api.GpuIndexSearcher(ctx, [g], api.IndexStatistics())
        queries = np.concatenate([np.array([[0.25, 0.5, 0.75]], np.float32) if dim == 3 else rng.standard_normal((1, dim)).astype(np.float32),
                                  rng.standard_normal((6, dim)).astype(np.float32)])
        for sim_name, sim, field, rows in (("cosine", 0, 0, vecs), ("dot_product", 1, 1, unit), ("l2_norm", 2, 0, vecs), ("max_inner_product", 3, 0, vecs)):
            qs = (queries / np.linalg.norm(queries, axis=1, keepdims=True)).astype(np.float32) if sim == 1 else queries
            for k in (5, 100):
                got = sr.knn_exact(field, sim_name, qs, k, boost=1.5)
                od, os_, oc = oracle.knn_exact(sim, qs, rows, k, live_words=live_words, doc_base=7, boost=1.5, n_threads=4)
                for qi in range(len(qs)):
                    m = int(oc[qi])
                    assert got[qi].docs.tolist() == od[qi][:m].tolist(), (dim, sim_name, k, qi)
                    assert got[qi].scores.view(np.uint32).tolist() == os_[qi][:m].view(np.uint32).tolist(), (dim, sim_name, k, qi)
                    assert got[qi].total_hits == int(live[:n].sum())
        one = sr.knn_exact_coalesced(0, "l2_norm", queries[0], 5)
        od, os_, _ = oracle.knn_exact(2, queries[:1], vecs, 5, live_words=live_words, doc_base=7, n_threads=2)
        assert one.docs.tolist() == od[0].tolist() and one.scores.view(np.uint32).tolist() == os_[0].view(np.uint32).tolist()
        # the vector rescorer at the same dimension (QueryRescore.combine in double, 1e-5: the one path with a tolerance)
        hits = sr.knn_exact(0, "cosine", queries[:1], 20)[0]
        first = np.linspace(2.0, 1.0, len(hits.docs)).astype(np.float32)
        res = sr.rescore_vectors(api.TopDocs(hits.docs, first, hits.total_hits, False), 0, "cosine", queries[0], window=10, query_weight=1.0, rescore_weight=4.0)
        exp = sorted(((float(oracle.rescore_combine(float(f), True, float(oracle.vector_score(0, queries[0], vecs[d - 7])), 1.0, 4.0)), int(d))
                      for d, f in zip(hits.docs.tolist(), first.tolist())), key=lambda t: (-t[0], t[1]))[:10]
        assert res.docs.tolist() == [d for _, d in exp]
        assert np.allclose(res.scores, [s_ for s_, _ in exp], rtol=1e-5, atol=1e-6)
        with pytest.raises(api.NrtGpuError) as e:      # a query of another dimension: this request's own error
            sr.knn_exact(0, "cosine", np.ones((1, dim + 1), np.float32), 5)
        assert e.value.code == -1
        g.release()


def test_exact_vector_query_relation_is_the_reference_collectors():
    """TotalHits.relation of an ExactVectorQuery: the scorer ignores min competitive scores, so every live doc
    with a vector is collected and the COUNT is exact -- but the reference's collector still flips its relation to
    GREATER_THAN_OR_EQUAL_TO once a slice has collected more than max(totalHitsThreshold, numHits) hits
    (LazyQueueTopScoreDocCollector.java:176-199; one collector per slice: MyIndexSearcher.java:163-208).
    nrtgpu_knn_exact_relation applies that rule to the live vectors of each slice (host only)."""
    rng = np.random.default_rng(176)
    c = api.GpuContext(device_id=0, max_batch=16)
    leaves = []
    try:
        for si, n in enumerate((600, 600, 600)):
            g = api.GpuSegment(c, n, si * 600)
            g.add_vectors(3, rng.standard_normal((n, 16)).astype(np.float32))
            g.seal()
            leaves.append(g)
        sr = api.GpuIndexSearcher(c, leaves, api.IndexStatistics())
        # default slicing (250 000 docs / 5 segments per slice): the three leaves are ONE slice of 1800 vectors
        assert sr.knn_exact_relation(3, 10, 1000) is True            # 1800 > max(1000, 10)
        assert sr.knn_exact_relation(3, 10, 1800) is False           # not MORE than the threshold
        assert sr.knn_exact_relation(3, 1024, 1000) is True          # max(threshold, numHits) = 1024 < 1800
        assert sr.knn_exact_relation(3, 10, 2**31 - 1) is False      # ScoreMode.COMPLETE
        assert sr.knn_exact_relation(7, 10, 1000) is False           # a field without vectors: nothing collected
        c.set_slicing(500, 5, 1)                                       # every leaf exceeds sliceMaxDocs: three slices of 600
        assert sr.knn_exact_relation(3, 10, 1000) is False           # 1800 hits in all, no slice above 1000: EQUAL_TO 1800
        assert sr.knn_exact_relation(3, 10, 500) is True
        live = np.ones(640, dtype=bool)
        live[rng.choice(600, 150, replace=False)] = False              # 450 live vectors in leaf 0
        live[600:] = False
        leaves[0].set_live_docs(np.packbits(live.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1))
        assert sr.knn_exact_relation(3, 10, 500) is True             # leaves 1 and 2 still hold 600
        for leaf in leaves[1:]:
            leaf.set_live_docs(np.packbits(live.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1))
        assert sr.knn_exact_relation(3, 10, 500) is False            # 450 live vectors per slice
        assert sr.knn_exact(3, "cosine", rng.standard_normal((1, 16)).astype(np.float32), 10)[0].total_hits == 3 * 450
    finally:
        for g in leaves:
            g.release()
        c.close()


def test_knn_search_prefilter_and_threshold(ctx, oracle):
    """The `knn` request path answered exactly: pre-filter mask (KnnFloatVectorQuery's filter), score threshold
    on the unboosted score (MinThresholdQuery, MinThresholdQuery.java:201), boost afterwards."""
    rng = np.random.default_rng(99)
    dim = 64
    segs = make_segments(rng, [4000, 2500, 900], dim, sparse_ords=True, deletes=True)
    leaves = upload(ctx, segs)
    masks = []
    for (base, vecs, o2d, live, max_doc), leaf in zip(segs, leaves):
        m = rng.random(max_doc) < 0.25
        masks.append(m)
        padded = np.zeros(((max_doc + 63) // 64) * 64, dtype=bool)
        padded[:max_doc] = m
        leaf.set_mask(4, np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1))
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics())
    queries = rng.standard_normal((3, dim)).astype(np.float32)
    for sim_name, sim in (("cosine", 0), ("l2_norm", 2)):
        for use_filter in (False, True):
            osegs = []
            for (b, v, o, l, max_doc), m in zip(segs, masks):
                acc = np.ones(max_doc, bool) if l is None else l.copy()
                if use_filter:
                    acc &= m
                osegs.append((b, v, o, acc))
            for qi in range(len(queries)):
                full, _, ref = brute_force(oracle, sim, queries[qi], osegs, 10**9)
                thr = full[60][0]          # a threshold that cuts the list inside the top 100
                for k, min_score, boost in ((20, 0.0, 1.0), (100, thr, 1.0), (100, thr, 2.0), (100, full[0][0] * 2 + 1, 1.0)):
                    got = sr.knn_search(3, sim_name, queries[qi], k, boost=boost, filter=api.MaskFilter(4) if use_filter else None,
                                        min_score=min_score)[0]
                    # the threshold is compared with the RESULT score (the oracle's bits), so the cut is the oracle's too
                    exp = [(float(np.float32(s) * np.float32(boost)), d) for s, d in full if s >= min_score][:k]
                    check_hits(got, exp)
                    assert got.total_hits == len(got.docs)
                    if use_filter:
                        for d in got.docs.tolist():
                            si = max(i for i, sg in enumerate(segs) if sg[0] <= d)
                            assert masks[si][d - segs[si][0]]
    with pytest.raises(api.NrtGpuError):
        sr.knn_search(3, "cosine", queries[0], 10, filter=api.MaskFilter(77))   # mask not resident
    for g in leaves:
        g.release()


def test_knn_concurrent_callers_take_turns(ctx, oracle):
    """Several threads in nrtgpu_knn_exact at once (each call has its own workspace and stream; their kernels take turns on the
    device, the staging and the result handling of one overlap the kernels of another): every call returns what it returns alone."""
    import threading
    rng = np.random.default_rng(4242)
    dim, n = 128, 60_000
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    g = api.GpuSegment(ctx, n, 0)
    g.add_vectors(0, vecs)
    g.seal()
    sr = api.GpuIndexSearcher(ctx, [g], api.IndexStatistics())
    panels = [rng.standard_normal((nq, dim)).astype(np.float32) for nq in (1, 20, 64, 7)]
    sims = ["cosine", "l2_norm", "max_inner_product", "dot_product"]
    alone = [sr.knn_exact(0, sims[i], panels[i], 25) for i in range(4)]
    results, errors = {}, []

    def caller(i):
        try:
            for rep in range(6):
                results[(i, rep)] = sr.knn_exact(0, sims[i], panels[i], 25)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=caller, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for (i, rep), got in results.items():
        for a, b in zip(alone[i], got):
            assert a.docs.tolist() == b.docs.tolist() and a.scores.view(np.uint32).tolist() == b.scores.view(np.uint32).tolist(), (i, rep)
    odocs, oscores, _ = oracle.knn_exact(2, panels[1][:2], vecs, 25, n_threads=4)
    for qi in range(2):
        assert alone[1][qi].docs.tolist() == odocs[qi].tolist()
    g.release()


def test_sketch_is_built_by_the_first_exact_search_only(oracle):
    """The fp16 sketch (+50 % of the fp32 matrix) is paid by fields that are searched exactly, on their first search: uploading
    and RESCORING leave the segment's footprint alone; a context that declines the sketch never builds it."""
    from nrtsearch_amd import _lib
    rng = np.random.default_rng(88)
    n, dim = 10_000, 96            # 3 steps of 32 dimensions, padded to 4: 16 rows x 4 KiB per tile
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal(dim).astype(np.float32)
    for flags in (0, _lib.NRTGPU_FLAG_NO_VECTOR_SKETCH):
        c = api.GpuContext(device_id=0, max_batch=8, flags=flags)
        g = api.GpuSegment(c, n, 0)
        g.add_vectors(0, vecs)
        g.seal()
        sr = api.GpuIndexSearcher(c, [g], api.IndexStatistics())
        b0 = g.device_bytes
        hits = api.TopDocs(np.arange(50, dtype=np.int32), np.ones(50, dtype=np.float32), 50, False)
        sr.rescore_vectors(hits, 0, "cosine", q, window=10, query_weight=1.0, rescore_weight=1.0)
        assert g.device_bytes == b0
        got = sr.knn_exact(0, "cosine", q[None, :], 5)[0]
        grown = g.device_bytes - b0
        assert grown == (0 if flags else ((n + 15) // 16) * 4 * 1024 + 256), (flags, grown)
        sr.knn_exact(0, "l2_norm", q[None, :], 5)
        assert g.device_bytes - b0 == grown           # once
        odocs, oscores, _ = oracle.knn_exact(0, q[None, :], vecs, 5)
        assert got.docs.tolist() == odocs[0].tolist() and got.scores.view(np.uint32).tolist() == oscores[0].view(np.uint32).tolist()
        g.release()
        c.close()


def test_knn_coalesced_callers_share_passes(dev_lib, oracle):
    """nrtgpu_knn_exact_coalesced: what a request thread calls with ONE query.  48 concurrent callers with their own queries and
    their own k (two similarities: requests that cannot share a panel) get exactly what nrtgpu_knn_exact gives each of them alone,
    from fewer passes over the rows than there were calls; a lone caller runs at once.

    The panel count is asserted BY CONSTRUCTION, not by the host's speed (round 3's version failed on a slow-host box: the
    coalescer has no linger, a leader that finds the device free leaves alone, so how many callers share a panel depended on
    how fast Python started its threads): the callers of a repetition are parked behind nrtgpu_debug_hold_coalescers until
    nrtgpu_debug_coalescer_pending reports all 48, then released -- 32 cosine callers form one panel, 16 l2 callers another,
    whichever leads."""
    import threading
    import time
    rng = np.random.default_rng(515)
    dim, n = 64, 40_000
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    c = api.GpuContext(device_id=0, max_batch=64)
    g = api.GpuSegment(c, n, 0)
    g.add_vectors(0, vecs)
    g.seal()
    sr = api.GpuIndexSearcher(c, [g], api.IndexStatistics())
    n_callers, reps = 48, 5
    queries = rng.standard_normal((n_callers, dim)).astype(np.float32)
    ks = [int(x) for x in rng.choice([1, 10, 37, 100], size=n_callers)]
    sims = ["cosine" if i % 3 else "l2_norm" for i in range(n_callers)]
    alone = [sr.knn_exact(0, sims[i], queries[i][None, :], ks[i])[0] for i in range(n_callers)]
    lone = sr.knn_exact_coalesced(0, "cosine", queries[1], ks[1])
    assert lone.docs.tolist() == alone[1].docs.tolist() and lone.scores.view(np.uint32).tolist() == alone[1].scores.view(np.uint32).tolist()
    results, errors = {}, []

    def caller(i, rep):
        try:
            results[(i, rep)] = sr.knn_exact_coalesced(0, sims[i], queries[i], ks[i])
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    for rep in range(reps):
        c.reset_stats()
        c.debug_hold_coalescers(True)
        threads = [threading.Thread(target=caller, args=(i, rep)) for i in range(n_callers)]
        try:
            for t in threads:
                t.start()
            t_end = time.monotonic() + 60.0
            while c.debug_coalescer_pending(1) < n_callers and time.monotonic() < t_end and not errors:
                time.sleep(0.001)
            parked = c.debug_coalescer_pending(1)
        finally:
            c.debug_hold_coalescers(False)
        for t in threads:
            t.join()
        assert not errors, errors
        assert parked == n_callers, parked
        # two groups of compatible requests, each at most one panel of 64: exactly two panels, whatever the host's speed
        assert c.stats()["knn_panels"] == 2, (rep, c.stats()["knn_panels"])
    for (i, rep), got in results.items():
        assert got.docs.tolist() == alone[i].docs.tolist(), (i, rep)
        assert got.scores.view(np.uint32).tolist() == alone[i].scores.view(np.uint32).tolist(), (i, rep)
        assert got.total_hits == n
    assert len(results) == n_callers * reps
    with pytest.raises(api.NrtGpuError) as e:
        sr.knn_exact_coalesced(0, "cosine", np.ones(24, np.float32), 5)      # dimension the device does not take: this request alone
    assert e.value.code in (-1, -4)
    g.release()
    c.close()
