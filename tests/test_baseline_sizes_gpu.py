"""Parity at BASELINE.json's configuration sizes, run by the driver (`-m gpu`): C3 (10M docs, 5-term disjunction,
top-1000; 256 queries, plain and with 1 % deletes), C2 (1M docs, 2-term, top-100; 256 queries), C4 at its full size
(10M x 768 fp32 cosine, top-100: fp64 over every row + the oracle's score bits of the hits) and at 2M rows (the oracle's own
top-k bit for bit), and the C5-shaped hybrid
(5M docs, BM25 recall-1000 -> 768-d cosine rescore -> top-100; 32 queries).  BM25: docids, ranks and score bits
bit-exact against the CPU oracle's EXHAUSTIVE scorer (one C call, OpenMP over queries).  Needs an MI355X."""
import os

import numpy as np
import pytest

from nrtsearch_amd import _lib, api, synth, workload

pytestmark = pytest.mark.gpu
N_Q = 256


def _cpus():
    return max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))


def _check_batch(got, exp, k, thr=1000):
    docs, scores, n, total, gte, _ = exp
    bad = []
    for qi, g in enumerate(got):
        m = int(n[qi])
        ok = (g.docs.tolist() == docs[qi, :m].tolist() and g.scores.view(np.uint32).tolist() == scores[qi, :m].view(np.uint32).tolist()
              and g.relation_gte == bool(gte[qi])
              and ((max(thr, k) < g.total_hits <= int(total[qi])) if gte[qi] else g.total_hits == int(total[qi])))
        if not ok:
            bad.append(qi)
    assert not bad, f"{len(bad)} of {len(got)} queries differ from the oracle: {bad[:8]}"


def _vectors(rng, n, dim):
    """n x dim fp32, uniform in [-1, 1) (the generator's fastest path: these tests move gigabytes)."""
    v = rng.random((n, dim), dtype=np.float32)
    v *= 2.0
    v -= 1.0
    return v


def _deletes(corpus, fraction, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    for s in corpus.segments:
        alive = rng.random(s.max_doc) >= fraction
        padded = np.zeros(((s.max_doc + 63) // 64) * 64, dtype=bool)
        padded[: s.max_doc] = alive
        s.live_bits = np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1)


@pytest.mark.parametrize("wname", ["C3", "C2"])
def test_bm25_full_size_against_the_exhaustive_oracle(oracle, wname):
    w = getattr(workload, wname)
    qr = synth.make_queries(N_Q, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr)
    ctx = api.GpuContext(0, max_batch=N_Q)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    try:
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        queries = workload.boolean_queries(qr)
        mgr = api.TopScoreDocCollectorManager(w.k)
        sample = [q.tolist() for q in qr]
        # plain index
        ctx.reset_stats()
        got = sr.search_batch(queries, [mgr] * N_Q)
        assert ctx.stats()["maxscore_items"] >= N_Q - 8       # the default route for this workload: dynamic pruning
        exp = oracle.PreparedBatch(corpus, sample, w.k, slicing=oracle.DEFAULT_SLICING).run(False, _cpus())
        _check_batch(got, exp, w.k)
        # 1 % deletes (liveDocs folded into the posting columns)
        _deletes(corpus, 0.01, 99)
        for leaf, seg in zip(leaves, corpus.segments):
            leaf.set_live_docs(seg.live_bits)
        got = sr.search_batch(queries, [mgr] * N_Q)
        exp = oracle.PreparedBatch(corpus, sample, w.k, slicing=oracle.DEFAULT_SLICING).run(False, _cpus())
        _check_batch(got, exp, w.k)
        # ScoreMode.COMPLETE: exhaustive route, exact counts
        mgr_c = api.TopScoreDocCollectorManager(w.k, total_hits_threshold=2**31 - 1)
        got = sr.search_batch(queries[:64], [mgr_c] * 64)
        exp = oracle.PreparedBatch(corpus, sample[:64], w.k, total_hits_threshold=2**31 - 1).run(False, _cpus())
        _check_batch(got, exp, w.k, thr=2**31 - 1)
    finally:
        for l in leaves:
            l.release()
        ctx.close()


def test_bm25_c3_numbered_by_doc_length_is_cured_by_the_scattered_window_order(oracle):
    """C3's size, docs numbered by length (synth.corpus_variant_arrays: "sorted" -- an index sorted by a field the BM25 score
    follows: scores fall along the docid axis).  In docid order practically every speculative threshold fails there (the first
    windows hold the best docs); the library then walks this leaf set's windows in the scattered order -- any prefix of the
    windows taken is spread over the docs -- and from there the guesses hold: few queries are run again, speculation stays on.
    Every answer on the way is the oracle's, whatever the guesses do."""
    w = workload.C3
    n_q = 1024
    qr = synth.make_queries(5 * n_q, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr, variant="sorted")
    ctx = api.GpuContext(0, max_batch=n_q)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    try:
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        mgr = api.TopScoreDocCollectorManager(w.k)
        ctx.set_speculation(5.0)
        history = []
        for b in range(5):
            rows = qr[b * n_q:(b + 1) * n_q]
            got = sr.search_batch(workload.boolean_queries(rows), [mgr] * n_q)
            history.append(ctx.spec_counters())
            if b in (0, 4):    # the oracle's exhaustive answer for a slice of the batch: docids, ranks, score bits
                exp = oracle.PreparedBatch(corpus, [q.tolist() for q in rows[:64]], w.k, slicing=oracle.DEFAULT_SLICING).run(False, _cpus())
                _check_batch(got[:64], exp, w.k)
        assert history[0]["reruns"] > 0.5 * n_q, history       # docid order: the guesses fail wholesale
        assert history[-1]["scattered"] and not history[-1]["switched_off"], history
        late = (history[-1]["reruns"] - history[2]["reruns"], history[-1]["queries"] - history[2]["queries"])   # batches 3 and 4: scattered order
        assert late[1] == 2 * n_q and late[0] * 50 <= late[1], history
    finally:
        for l in leaves:
            l.release()
        ctx.close()


def test_knn_c4_shape_against_fp64(oracle):
    n, dim, k, nq = 2_000_000, 768, 100, 40
    rng = np.random.default_rng(777)
    n_seg = 4
    per = n // n_seg
    ctx = api.GpuContext(0, max_batch=64)
    leaves, host = [], []
    try:
        for si in range(n_seg):
            v = _vectors(rng, per, dim)
            host.append(v)
            leaf = api.GpuSegment(ctx, per, si * per)
            leaf.add_vectors(0, v)
            leaf.seal()
            leaves.append(leaf)
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics())
        queries = np.random.default_rng(778).standard_normal((nq, dim)).astype(np.float32)
        got = sr.knn_exact(0, "cosine", queries, k)
        for qi in (0, 19, 39):
            q = queries[qi].astype(np.float64)
            sc = np.empty(n, dtype=np.float64)
            for si, v in enumerate(host):   # fp64 reference, a chunk at a time
                for a in range(0, per, 250_000):
                    blk = v[a: a + 250_000].astype(np.float64)
                    cos = (blk @ q) / (np.linalg.norm(blk, axis=1) * np.linalg.norm(q))
                    sc[si * per + a: si * per + a + len(blk)] = np.maximum((1.0 + cos) / 2.0, 0.0)
            order = np.lexsort((np.arange(n), -sc))[:k]
            gd, gs = got[qi].docs, got[qi].scores
            assert got[qi].total_hits == n and len(gd) == k and len(set(gd.tolist())) == k
            assert np.allclose(gs, sc[order], rtol=2e-5, atol=2e-6)
            for r in range(k):   # a docid that differs at its rank must be a near-tie of the reference's doc there
                if gd[r] != order[r]:
                    assert abs(sc[gd[r]] - sc[order[r]]) <= 2e-5 * sc[order[r]] + 2e-6, f"query {qi} rank {r}"
            assert len(set(gd.tolist()) & set(order.tolist())) >= k - 2
        # and the oracle itself (scalar fp32, its order of summation), leaf by leaf, on two queries: bit for bit
        for qi in (0, 39):
            hits = []
            for si, v in enumerate(host):
                d_, s_, c_ = oracle.knn_exact(0, queries[[qi]], v, k, doc_base=si * per, n_threads=_cpus())
                hits += list(zip((-s_[0][: c_[0]].astype(np.float64)).tolist(), d_[0][: c_[0]].tolist(), s_[0][: c_[0]].tolist()))
            hits.sort()
            assert got[qi].docs.tolist() == [d for _, d, _ in hits[:k]]
            assert got[qi].scores.view(np.uint32).tolist() == np.array([s for _, _, s in hits[:k]], np.float32).view(np.uint32).tolist()
        assert ctx.stats()["knn_second_passes"] == 0
    finally:
        for l in leaves:
            l.release()
        ctx.close()


def _normal_chunk(seed, chunk, rows, dim):
    """Chunk `chunk` of the C4 row matrix: Normal(0, 1) fp32 (SURVEY 8d's row distribution; the stream is PCG64 seeded per
    chunk from `seed`, so any chunk can be generated again on its own -- the 30.7 GB matrix is never held on the host)."""
    return np.random.Generator(np.random.PCG64([seed, chunk])).standard_normal((rows, dim), dtype=np.float32)


def test_knn_c4_full_size_against_fp64(oracle):
    """BASELINE config 4 at its FULL size in the driver-run suite: 10 M x 768 fp32 rows ~ Normal(0, 1)
    in 4 segments, 40 queries (PCG64(778)), cosine top-100 through nrtgpu_knn_exact --
      * against the fp64 cosine score of EVERY row (host, 16 threads, chunk by chunk as the rows are generated): scores within
        2e-5 relative, a docid off its fp64 rank only among near-ties, the sets equal but for <= 2 docs per query;
      * the score BITS of every returned hit against the oracle's pinned summation order (its rows generated again from their
        chunk seeds), and the order (score desc, doc asc) -- what tests at 2 M rows check against the oracle's own top-k.
    bench.py --workload C4 verifies the same on its device-generated rows; this one needs no torch."""
    from concurrent.futures import ThreadPoolExecutor
    n, dim, k, nq, n_seg, chunk_rows = 10_000_000, 768, 100, 40, 4, 250_000
    per = n // n_seg
    queries = np.random.Generator(np.random.PCG64(778)).standard_normal((nq, dim), dtype=np.float32)
    check_q = (0, 19, 39)
    q64 = queries[list(check_q)].astype(np.float64)
    qn = np.linalg.norm(q64, axis=1)
    ctx = api.GpuContext(0, max_batch=64)
    leaves = []
    cand = [[] for _ in check_q]     # per checked query: (fp64 score, docid) candidates of every chunk

    def make(args):
        seg_buf, c, a0, rows = args
        blk = _normal_chunk(777, c, rows, dim)
        seg_buf[a0: a0 + rows] = blk
        b64 = blk.astype(np.float64)
        cos = (b64 @ q64.T) / (np.linalg.norm(b64, axis=1)[:, None] * qn[None, :])
        sc = np.maximum((1.0 + cos) / 2.0, 0.0)
        out = []
        for j in range(len(check_q)):
            top = np.argpartition(-sc[:, j], 2 * k)[: 2 * k]
            out.append((sc[top, j], top))
        return c, out

    try:
        with ThreadPoolExecutor(max_workers=_cpus()) as ex:
            for si in range(n_seg):
                seg_buf = np.empty((per, dim), dtype=np.float32)
                jobs = [(seg_buf, si * (per // chunk_rows) + ci, ci * chunk_rows, chunk_rows) for ci in range(per // chunk_rows)]
                for c, out in ex.map(make, jobs):
                    for j, (sc, idx) in enumerate(out):
                        cand[j].append((sc, idx.astype(np.int64) + c * chunk_rows))
                leaf = api.GpuSegment(ctx, per, si * per)
                leaf.add_vectors(0, seg_buf)
                leaf.seal()
                leaves.append(leaf)
                del seg_buf
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics())
        got = sr.knn_exact(0, "cosine", queries, k)
        for j, qi in enumerate(check_q):
            cs = np.concatenate([c[0] for c in cand[j]])
            cd = np.concatenate([c[1] for c in cand[j]])
            order = np.lexsort((cd, -cs))[:k]
            ref_s, ref_d = cs[order], cd[order]
            score_of = dict(zip(cd.tolist(), cs.tolist()))
            gd, gs = got[qi].docs, got[qi].scores
            assert got[qi].total_hits == n and len(gd) == k and len(set(gd.tolist())) == k
            assert np.allclose(gs, ref_s, rtol=2e-5, atol=2e-6)
            for r in range(k):   # a docid that differs at its rank must be a near-tie of the reference's doc there
                if gd[r] != ref_d[r]:
                    assert int(gd[r]) in score_of and abs(score_of[int(gd[r])] - ref_s[r]) <= 2e-5 * ref_s[r] + 2e-6, f"query {qi} rank {r}"
            assert len(set(gd.tolist()) & set(ref_d.tolist())) >= k - 2
        # the returned hits' score bits in the oracle's pinned order; their order (score desc, doc asc)
        need = {}
        for qi in check_q:
            for d in got[qi].docs.tolist():
                need.setdefault(d // chunk_rows, set()).add(d)
        with ThreadPoolExecutor(max_workers=_cpus()) as ex:
            chunks = dict(zip(need, ex.map(lambda c: _normal_chunk(777, c, chunk_rows, dim), list(need))))
        for qi in check_q:
            gd, gs = got[qi].docs.tolist(), got[qi].scores
            exp = np.array([oracle.vector_score(0, queries[qi], chunks[d // chunk_rows][d % chunk_rows]) for d in gd], dtype=np.float32)
            assert gs.view(np.uint32).tolist() == exp.view(np.uint32).tolist(), f"query {qi}: score bits differ from the oracle's order"
            keys = [(-float(s_), d) for s_, d in zip(gs, gd)]
            assert keys == sorted(keys)
        assert ctx.stats()["knn_second_passes"] == 0
    finally:
        for l in leaves:
            l.release()
        ctx.close()


def test_hybrid_c5_shape(oracle):
    n_docs, dim, nq, recall, window = 5_000_000, 768, 32, 1000, 100
    w = workload.Workload("C5-shaped: 5M docs, recall-1000 -> rescore top-100", n_docs, 5, recall, nq, 4)
    qr = synth.make_queries(nq, 5, 10000)
    corpus = workload.build_shard_corpus(w, qr)
    ctx = api.GpuContext(0, max_batch=64)
    leaves, host = [], []
    rng = np.random.default_rng(7)
    try:
        for seg in corpus.segments:
            v = _vectors(rng, seg.max_doc, dim)
            host.append(v)
            leaf = api.GpuSegment(ctx, seg.max_doc, seg.doc_base)
            leaf.add_field_norms(0, seg.norms)
            leaf.add_terms(0, seg.term_ids, seg.offsets, seg.docids, seg.freqs)
            leaf.add_vectors(7, v)
            leaf.seal()
            leaves.append(leaf)
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        queries = workload.boolean_queries(qr)
        mgr = api.TopScoreDocCollectorManager(recall)
        qv = np.random.default_rng(8).standard_normal((nq, dim)).astype(np.float32)
        fused = sr.search_hybrid_batch(queries, [mgr] * nq, 7, "cosine", qv, window, 1.0, 2.0)
        first = oracle.PreparedBatch(corpus, [q.tolist() for q in qr], recall, slicing=oracle.DEFAULT_SLICING).run(False, _cpus())
        bases = [s.doc_base for s in corpus.segments]
        for qi in range(nq):
            docs, scores = first[0][qi, : first[2][qi]], first[1][qi, : first[2][qi]]
            exp = []
            for doc, f in zip(docs.tolist(), scores.tolist()):   # QueryRescore.combine over the oracle's first pass
                si = max(i for i, b in enumerate(bases) if b <= doc)
                second = float(oracle.vector_score(0, qv[qi], host[si][doc - bases[si]]))
                exp.append((float(oracle.rescore_combine(f, True, second, 1.0, 2.0)), doc))
            exp.sort(key=lambda t: (-t[0], t[1]))
            got = fused[qi]
            assert len(got.docs) == window and got.relation_gte
            assert np.allclose(got.scores, [s for s, _ in exp[:window]], rtol=1e-5, atol=1e-6)
            ref = dict((d, s) for s, d in exp)
            for r, d in enumerate(got.docs.tolist()):   # every returned doc was recalled; off-rank only among near-ties
                assert d in ref
                assert abs(ref[d] - exp[r][0]) <= 1e-5 * abs(exp[r][0]) + 1e-6
    finally:
        for l in leaves:
            l.release()
        ctx.close()
