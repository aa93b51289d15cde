"""Cross-GPU bound exchange (nrtgpu_exchange_open): two processes, each scanning its docid shard of one index
on the same MI355X with the shared-memory exchange open, must merge to exactly the whole-index answer --
shards may return fewer low-ranked hits, never lose one of the merged top-k."""
import os
import subprocess
import sys
import tempfile
import uuid

import numpy as np
import pytest

from nrtsearch_amd import synth, workload

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
N_DOCS, N_QUERIES, K, WORLD = 1_500_000, 48, 1000, 2


def _run_ranks(shm_name, epochs):
    with tempfile.TemporaryDirectory() as d:
        procs, outs = [], []
        for r in range(WORLD):
            out = os.path.join(d, f"rank{r}.npz")
            outs.append(out)
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_exchange_worker.py"), str(r), str(WORLD),
                                           shm_name, d, out, str(N_DOCS), str(N_QUERIES), str(K), str(epochs)],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
        for p, log in zip(procs, logs):
            assert p.returncode == 0, log[-2000:]
        return [dict(np.load(o)) for o in outs]


def _lists(res, qi):
    out = []
    for r in res:
        n = int(r["cnt"][qi])
        keys = r["keys"][qi, :n]
        docs = (np.uint64(0xFFFFFFFF) - (keys & np.uint64(0xFFFFFFFF))).astype(np.int32)
        scores = (keys >> np.uint64(32)).astype(np.uint32).view(np.float32)
        out.append((docs, scores))
    return out


@pytest.mark.parametrize("exchange", [False, True])
def test_two_shards_merge_to_the_whole_index_answer(oracle, exchange):
    w = workload.Workload("exchange test", N_DOCS, 5, K, N_QUERIES, 4)
    qr = synth.make_queries(N_QUERIES, w.n_terms, w.max_rank)
    full = workload.build_shard_corpus(w, qr, 1, 0)
    name = f"/nrtgpu_test_{uuid.uuid4().hex[:12]}" if exchange else "-"
    try:
        res = _run_ranks(name, epochs=3)
    finally:
        if exchange and os.path.exists("/dev/shm" + name):
            os.unlink("/dev/shm" + name)
    returned = []
    for qi in range(N_QUERIES):
        edocs, escores, etotal, _ = oracle.search_bm25(full, qr[qi].tolist(), K)
        docs, scores = oracle.topdocs_merge(K, _lists(res, qi))
        assert docs.tolist() == edocs.tolist(), f"query {qi}: merged docids differ"
        assert scores.view(np.uint32).tolist() == escores.view(np.uint32).tolist()
        tot = int(sum(int(r["hits"][qi]) for r in res))   # per shard: exact count, or (1 << 48) + a lower bound where it pruned
        low = tot & ((1 << 48) - 1)
        assert (K < low <= etotal) if (tot >> 48) else low == etotal
        returned.append(sum(int(r["cnt"][qi]) for r in res))
    print("exchange" if exchange else "plain", "mean hits returned per query by the two shards:", float(np.mean(returned)))
    if not exchange:
        assert min(returned) >= K


def test_exchange_table_written_and_honoured():
    """One process plays rank 0 of 2; the test writes rank 1's rows by hand.  Checks the table protocol (slot,
    tag, own-row publication) and that a peer's published bound prunes rank 0's result without changing hits."""
    import torch  # noqa: F401  (HIP runtime first)

    from nrtsearch_amd import api

    n_q, world, max_batch, slots = 32, 2, 64, 8
    w = workload.Workload("exchange table test", 1_200_000, 5, K, n_q, 4)
    qr = synth.make_queries(n_q, w.n_terms, w.max_rank)
    shard0 = workload.build_shard_corpus(w, qr, world, 0)
    ctx = api.GpuContext(0, max_batch=max_batch)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in shard0.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(shard0))
    queries = workload.boolean_queries(qr)
    plain = sr.search_batch(queries, [api.TopScoreDocCollectorManager(K)] * n_q)
    name = f"/nrtgpu_test_{uuid.uuid4().hex[:12]}"
    try:
        ctx.exchange_open(name, world, 0)
        table = np.memmap("/dev/shm" + name, dtype=np.uint64, mode="r+", shape=(slots, world, max_batch))
        pb = api.PreparedBatch(sr, queries, [api.TopScoreDocCollectorManager(K)] * n_q)
        k_stride = (K + 15) // 16 * 16
        import torch

        keys = torch.zeros((n_q, k_stride), dtype=torch.int64, device="cuda")
        cnt = torch.zeros((n_q,), dtype=torch.int32, device="cuda")
        hits = torch.zeros((n_q,), dtype=torch.int64, device="cuda")

        def run(epoch):
            pb.run_device(k_stride, keys.data_ptr(), cnt.data_ptr(), hits.data_ptr(), epoch=epoch)
            torch.cuda.synchronize()
            return keys.cpu().numpy().view(np.uint64), cnt.cpu().numpy(), hits.cpu().numpy()

        # epoch 3, peer silent: nothing may be pruned; rank 0 publishes into slot 3, row 0, tag 4
        kk, cc, hh = run(3)
        assert cc.tolist() == [len(p.docs) for p in plain]
        mine = np.array(table[3, 0, :n_q])
        pub = (mine >> np.uint64(32)) == np.uint64(4)
        assert pub.any()                                   # queries that met a rendezvous have published
        for qi in np.nonzero(pub)[0]:
            bound = np.uint32(mine[qi] & np.uint64(0xFFFFFFFF)).view(np.float32)
            assert (plain[qi].scores >= bound).sum() >= -(-K // (world - 1))   # >= ceil(k / (world - 1)) of my docs reach it
        # epoch 4, the peer "publishes" a high bound under the right tag -> rank 0 bounds itself by it
        # epoch 5: the same values under a stale tag -> ignored
        for epoch, tag_ok in ((4, True), (5, False)):
            peer_bound = np.array([p.scores[K // 4] for p in plain], dtype=np.float32)   # plausible: my own rank-250 score
            tag = np.uint64(epoch + 1 if tag_ok else epoch)
            table[epoch % slots, 1, :n_q] = (tag << np.uint64(32)) | peer_bound.view(np.uint32).astype(np.uint64)
            table.flush()
            kk, cc, hh = run(epoch)
            differs = 0
            for qi in range(n_q):
                n = int(cc[qi])
                docs = (np.uint64(0xFFFFFFFF) - (kk[qi, :n] & np.uint64(0xFFFFFFFF))).astype(np.int32)
                scores = (kk[qi, :n] >> np.uint64(32)).astype(np.uint32).view(np.float32)
                # (the device-resident entry point counts exactly; the plain search may have pruned: a lower bound)
                assert hh[qi] >= plain[qi].total_hits if plain[qi].relation_gte else hh[qi] == plain[qi].total_hits
                if not tag_ok:
                    assert docs.tolist() == plain[qi].docs.tolist()
                    continue
                # everything at or above the other rank's published bound is returned, in the same order (with
                # or without a publication of my own); below it the shard may return any docs it collected
                # before the bound arrived
                floor = peer_bound[qi]
                keep_p = plain[qi].scores >= floor
                keep_g = scores >= floor
                assert docs[keep_g].tolist() == plain[qi].docs[keep_p].tolist()
                assert keep_g[: int(keep_g.sum())].all()             # the kept part is the head of the list
                differs += int(docs.tolist() != plain[qi].docs.tolist())
            if tag_ok:
                assert differs > 0   # the bound took effect: below it the tail is what was collected before it arrived
        del table
    finally:
        ctx.exchange_close()
        if os.path.exists("/dev/shm" + name):
            os.unlink("/dev/shm" + name)
        for l in leaves:
            l.release()
        ctx.close()


def test_collective_inside_the_library_world_of_one(oracle):
    """nrtgpu_dist_*: local search -> RCCL all-gather -> merge, all behind the C ABI (no torch in the data path).  A
    one-rank communicator exercises every step on the one GPU this pool has; the N-rank logic is the same call."""
    from nrtsearch_amd import api

    w = workload.Workload("dist test", 300_000, 4, 100, 24, 3)
    qr = synth.make_queries(24, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr)
    ctx = api.GpuContext(0, max_batch=64)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    try:
        ctx.dist_init(1, 0, api.GpuContext.dist_unique_id())
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        queries = workload.boolean_queries(qr)
        mgr = api.TopScoreDocCollectorManager(w.k)
        got = sr.dist_search_batch(queries, [mgr] * len(queries))
        for qi in range(len(queries)):
            edocs, escores, etotal, egte = oracle.search_bm25(corpus, qr[qi].tolist(), w.k)
            assert got[qi].docs.tolist() == edocs.tolist() and got[qi].scores.view(np.uint32).tolist() == escores.view(np.uint32).tolist()
            assert got[qi].relation_gte == egte
            assert (1000 < got[qi].total_hits <= etotal) if egte else got[qi].total_hits == etotal
    finally:
        ctx.dist_close()
        for l in leaves:
            l.release()
        ctx.close()


def test_exchange_modes_and_vector_search_world_of_one(oracle):
    """The all-to-all form of the exchange (rank r merges and delivers its slice of the batch) and the row-partitioned exact
    vector search (nrtgpu_dist_knn_exact), through a one-rank communicator: the slice is the whole batch, the other ranks'
    lists are none -- every step of the N-rank call runs, on the one GPU this pool has."""
    from nrtsearch_amd import api

    w = workload.Workload("dist test", 200_000, 3, 50, 16, 2)
    qr = synth.make_queries(16, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr)
    ctx = api.GpuContext(0, max_batch=64)
    leaves = []
    rng = np.random.Generator(np.random.PCG64(11))
    dim = 64
    mats = []
    for s in corpus.segments:
        g = api.GpuSegment(ctx, s.max_doc, s.doc_base)
        g.add_field_norms(0, s.norms)
        g.add_terms(0, s.term_ids, s.offsets, s.docids, s.freqs)
        m = rng.standard_normal((s.max_doc, dim), dtype=np.float32)
        g.add_vectors(5, m)
        g.seal()
        mats.append(m)
        leaves.append(g)
    try:
        ctx.dist_init(1, 0, api.GpuContext.dist_unique_id())
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        queries = workload.boolean_queries(qr)
        mgr = api.TopScoreDocCollectorManager(w.k)
        for mode in (api.EXCHANGE_ALLGATHER, api.EXCHANGE_ALLTOALL):
            got = sr.dist_search_batch(queries, [mgr] * len(queries), mode=mode)
            assert all(g is not None for g in got)          # a world of one owns every query
            for qi in range(len(queries)):
                edocs, escores, etotal, egte = oracle.search_bm25(corpus, qr[qi].tolist(), w.k)
                assert got[qi].docs.tolist() == edocs.tolist() and got[qi].scores.view(np.uint32).tolist() == escores.view(np.uint32).tolist()
                assert got[qi].relation_gte == egte
            qv = rng.standard_normal((5, dim), dtype=np.float32)
            local = sr.knn_exact(5, "cosine", qv, 20)
            dist_ = sr.dist_knn_exact(5, "cosine", qv, 20, mode=mode)
            for a, b in zip(local, dist_):
                assert a.docs.tolist() == b.docs.tolist() and a.scores.view(np.uint32).tolist() == b.scores.view(np.uint32).tolist()
                assert a.total_hits == b.total_hits == sum(len(m) for m in mats) and not b.relation_gte
            # the hybrid over shards (config 5): first pass -> all-gather + merge -> this rank's docs rescored -> windows exchanged
            hq = rng.standard_normal((len(queries), dim), dtype=np.float32)
            one = sr.search_hybrid_batch(queries, [mgr] * len(queries), 5, "cosine", hq, 20, 1.0, 2.0)
            many = sr.dist_search_hybrid_batch(queries, [mgr] * len(queries), 5, "cosine", hq, 20, 1.0, 2.0, mode=mode)
            for a, b in zip(one, many):
                assert a.docs.tolist() == b.docs.tolist() and a.scores.view(np.uint32).tolist() == b.scores.view(np.uint32).tolist()
                assert a.total_hits == b.total_hits and a.relation_gte == b.relation_gte
    finally:
        ctx.dist_close()
        for l in leaves:
            l.release()
        ctx.close()


def test_device_resident_search_in_two_halves(oracle):
    """nrtgpu_search_bm25_batch_device_begin + nrtgpu_pending_wait == the synchronous call: same keys / counts / hit totals in
    HBM, several batches in flight from one submitting thread."""
    import torch

    from nrtsearch_amd import api

    w = workload.Workload("begin-wait test", 250_000, 4, 100, 48, 3)
    qr = synth.make_queries(48, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr)
    ctx = api.GpuContext(0, max_batch=16)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    try:
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        queries = workload.boolean_queries(qr)
        mgr = api.TopScoreDocCollectorManager(w.k)
        k_stride = 112
        pbs = [api.PreparedBatch(sr, queries[i: i + 16], [mgr] * 16) for i in range(0, 48, 16)]
        bufs = [(torch.zeros((16, k_stride), dtype=torch.int64, device="cuda"), torch.zeros((16,), dtype=torch.int32, device="cuda"),
                 torch.zeros((16,), dtype=torch.int64, device="cuda")) for _ in range(6)]
        for b, pb in enumerate(pbs):   # the synchronous call
            pb.run_device(k_stride, *(t.data_ptr() for t in bufs[b]))
        handles = [pb.begin_device(k_stride, *(t.data_ptr() for t in bufs[3 + b])) for b, pb in enumerate(pbs)]   # three in flight
        for h in handles:
            api.PreparedBatch.wait_device(h)
        torch.cuda.synchronize()
        for b in range(3):
            for x, y in zip(bufs[b], bufs[3 + b]):
                assert torch.equal(x, y)
            keys = bufs[3 + b][0].cpu().numpy().view(np.uint64)
            cnt = bufs[3 + b][1].cpu().numpy()
            for j in range(16):
                qi = b * 16 + j
                edocs, escores, _, _ = oracle.search_bm25(corpus, qr[qi].tolist(), w.k)
                docs = (0xFFFFFFFF - (keys[j, : cnt[j]] & np.uint64(0xFFFFFFFF))).astype(np.int64)
                assert docs.tolist() == edocs.tolist()
    finally:
        for l in leaves:
            l.release()
        ctx.close()


def _small_begin_wait_index(max_batch=16):
    from nrtsearch_amd import api

    w = workload.Workload("begin-wait lifetime test", 250_000, 4, 100, 48, 3)
    qr = synth.make_queries(48, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr)
    ctx = api.GpuContext(0, max_batch=max_batch)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    return w, qr, corpus, ctx, leaves


def test_segment_release_while_searches_are_in_flight(dev_lib):
    """nrtgpu_segment_release under running searches (the reference closes readers while SEARCH-pool
    threads run: ShardState.java:506-527).  Three batches are begun over forked reader versions (planned, enqueued, NOT waited
    for), every handle they use -- the forks AND the base segments -- is released, then the batches are waited for: same keys,
    counts and hit totals as the synchronous run before; nothing crashes; the last search to let go of a handle frees it (the
    device's free memory comes back)."""
    import torch

    from nrtsearch_amd import api

    w, qr, corpus, ctx, leaves = _small_begin_wait_index()
    try:
        forks = [l.fork(None) for l in leaves]
        sr = api.GpuIndexSearcher(ctx, forks, api.IndexStatistics.from_corpus(corpus))
        queries = workload.boolean_queries(qr)
        mgr = api.TopScoreDocCollectorManager(w.k)
        k_stride = 112
        pbs = [api.PreparedBatch(sr, queries[i: i + 16], [mgr] * 16) for i in range(0, 48, 16)]
        bufs = [(torch.zeros((16, k_stride), dtype=torch.int64, device="cuda"), torch.zeros((16,), dtype=torch.int32, device="cuda"),
                 torch.zeros((16,), dtype=torch.int64, device="cuda")) for _ in range(6)]
        for b, pb in enumerate(pbs):
            pb.run_device(k_stride, *(t.data_ptr() for t in bufs[b]))
        torch.cuda.synchronize()
        assert ctx.debug_live_segments() == 2 * len(leaves)
        handles = [pb.begin_device(k_stride, *(t.data_ptr() for t in bufs[3 + b])) for b, pb in enumerate(pbs)]   # three in flight
        for g in leaves:              # the segments are merged away: nobody searches the base handles -> freed at once
            g.release()
        assert ctx.debug_live_segments() == len(forks)
        for g in forks:               # the reader versions close WHILE three searches over them are in flight: deferred
            g.release()
        assert ctx.debug_live_segments() == len(forks)
        for i, h in enumerate(handles):
            api.PreparedBatch.wait_device(h)
            assert ctx.debug_live_segments() == (len(forks) if i < len(handles) - 1 else 0)   # the LAST search to let go frees them
        torch.cuda.synchronize()
        for b in range(3):
            for x, y in zip(bufs[b], bufs[3 + b]):
                assert torch.equal(x, y)
        assert int(bufs[3][1].sum().item()) > 0
    finally:
        ctx.close()


def test_begin_wait_pipeline_against_a_writer_of_the_same_segments():
    """a thread that begins search i + 1 before it waits for search i must not deadlock against a writer
    (nrtgpu_segment_set_mask / set_live_docs) that waits for search i -- the old std::shared_mutex parked the second begin
    behind the writer, and the wait that would have released the first never came.  And the wait may come from another thread
    than the begin.  A writer thread rewrites a mask of every leaf in a loop while a pipeline of depth 3 runs 40 rounds, its waits
    on a third thread; everything must finish, with the synchronous run's results."""
    import queue
    import threading

    import torch

    from nrtsearch_amd import api

    w, qr, corpus, ctx, leaves = _small_begin_wait_index()
    try:
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        queries = workload.boolean_queries(qr)
        mgr = api.TopScoreDocCollectorManager(w.k)
        k_stride = 112
        pbs = [api.PreparedBatch(sr, queries[i: i + 16], [mgr] * 16) for i in range(0, 48, 16)]
        bufs = [(torch.zeros((16, k_stride), dtype=torch.int64, device="cuda"), torch.zeros((16,), dtype=torch.int32, device="cuda"),
                 torch.zeros((16,), dtype=torch.int64, device="cuda")) for _ in range(6)]
        for b, pb in enumerate(pbs):
            pb.run_device(k_stride, *(t.data_ptr() for t in bufs[b]))
        torch.cuda.synchronize()
        stop, errors, done = threading.Event(), [], threading.Event()
        pending = queue.Queue()
        state = {"writer": "-", "waiter": "-", "pipeline": "-", "writes": 0, "begun": 0, "waited": 0}   # (what a hang looked like)
        masks9 = [[synth.random_mask(seg.max_doc, 0.5, i) for i in range(4)] for seg in corpus.segments]

        def writer():
            i = 0
            while not stop.is_set():
                for li, leaf in enumerate(leaves):
                    state["writer"] = f"set_mask leaf {li}"
                    leaf.set_mask(9, masks9[li][i % 4])   # (no query names mask 9: the results do not change)
                    state["writes"] += 1
                state["writer"] = "between"
                i += 1

        def waiter():
            try:
                while True:
                    state["waiter"] = "queue"
                    h = pending.get()
                    if h is None:
                        break
                    state["waiter"] = "wait_device"
                    api.PreparedBatch.wait_device(h)     # another thread than the one that began it
                    state["waited"] += 1
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))
            done.set()

        def pipeline():
            try:
                for rnd in range(40):
                    hs = []
                    for b, pb in enumerate(pbs):   # i + 1, i + 2 begun before i is waited for
                        state["pipeline"] = f"begin round {rnd} batch {b}"
                        hs.append(pb.begin_device(k_stride, *(t.data_ptr() for t in bufs[3 + b])))
                        state["begun"] += 1
                    state["pipeline"] = "put"
                    for h in hs:
                        pending.put(h)
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))
            pending.put(None)

        threads = [threading.Thread(target=writer, daemon=True), threading.Thread(target=waiter, daemon=True), threading.Thread(target=pipeline, daemon=True)]
        for t in threads:
            t.start()
        finished = done.wait(timeout=60.0)
        stop.set()
        assert finished, f"the begin / wait pipeline did not finish next to a mask writer (deadlock): {state}"
        threads[0].join(timeout=30.0)
        assert not errors, errors
        torch.cuda.synchronize()
        for b in range(3):
            for x, y in zip(bufs[b], bufs[3 + b]):
                assert torch.equal(x, y)
    finally:
        for l in leaves:
            l.release()
        ctx.close()


def test_device_resident_results_under_speculative_thresholds(dev_lib, monkeypatch, oracle):
    """The device-resident path (what a rank of a multi-GPU search runs) speculates too since round 5: the merge's tags are read in
    nrtgpu_pending_wait and a batch with a failed guess is run again without speculation into the same buffers.  On an index that
    defeats the guesses -- every live doc in the first third of the docid range, windows walked in docid order (development library,
    NRTGPU_MS_SCATTER=0) -- the keys left in HBM are the oracle's top-k all the same, no tag is left in the hit totals, and the
    counters show the re-runs."""
    import torch

    from nrtsearch_amd import api

    monkeypatch.setenv("NRTGPU_MS_SCATTER", "0")
    ranks = [1, 2, 5, 9, 20, 60, 150, 400]
    corpus = synth.build_corpus(3_200_000, ranks, n_segments=1)
    seg = corpus.segments[0]
    live = np.zeros((seg.max_doc + 63) // 64, dtype=np.uint64)
    live[: int(seg.max_doc * 0.3) // 64] = np.uint64(0xFFFFFFFFFFFFFFFF)
    seg.live_bits = live
    ctx = api.GpuContext(0, max_batch=16)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    try:
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        qs = [[1, 5, 20, 150, 400], [2, 9, 60], [1, 2, 5, 9, 20, 60, 150, 400], [5, 400], [9, 20, 150]]
        k, k_stride = 1000, 1008
        queries = [api.BooleanQuery(tuple(api.TermQuery(0, t) for t in q)) for q in qs]
        pb = api.PreparedBatch(sr, queries, [api.TopScoreDocCollectorManager(k)] * len(qs))
        keys = torch.zeros((len(qs), k_stride), dtype=torch.int64, device="cuda")
        cnt = torch.zeros((len(qs),), dtype=torch.int32, device="cuda")
        hits = torch.zeros((len(qs),), dtype=torch.int64, device="cuda")
        ctx.set_speculation(5.0)
        pb.run_device(k_stride, keys.data_ptr(), cnt.data_ptr(), hits.data_ptr())
        torch.cuda.synchronize()
        c = ctx.spec_counters()
        assert c["queries"] == len(qs) and c["reruns"] >= 2, c
        hk, hc, hh = keys.cpu().numpy().view(np.uint64), cnt.cpu().numpy(), hits.cpu().numpy().view(np.uint64)
        for j, q in enumerate(qs):
            edocs, escores, _, _ = oracle.search_bm25(corpus, q, k)
            docs = (0xFFFFFFFF - (hk[j, : hc[j]] & np.uint64(0xFFFFFFFF))).astype(np.int64)
            bits = (hk[j, : hc[j]] >> np.uint64(32)).astype(np.uint32)
            assert docs.tolist() == edocs.tolist() and bits.tolist() == escores.view(np.uint32).tolist(), j
            assert (int(hh[j]) >> 47) & 1 == 0      # plan.h: kHitsSpecInvalid -- never handed to the caller
    finally:
        for l in leaves:
            l.release()
        ctx.close()
