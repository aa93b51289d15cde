"""N > 1 path on CPU: 2 ranks (gloo), each owning one docid range of the same index.  Every rank
produces its local top-k (here with the CPU oracle standing in for the GPU kernels -- this test is
about the partition and the exchange), all-gathers the packed keys exactly as bench.py does over
RCCL, merges, and must reproduce the single-process result bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank: int, world: int, port: int, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from nrtsearch_amd import dist as nd
    from nrtsearch_amd import synth, workload
    from oracle import oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = workload.Workload("dist-test", 60_000, 3, 50, 6, 2, max_rank=300)
        qr = synth.make_queries(w.n_queries, w.n_terms, w.max_rank)
        shard = workload.build_shard_corpus(w, qr, world, rank)      # my docid range, global statistics
        k, k_stride = w.k, 64
        keys = np.zeros((w.n_queries, k_stride), dtype=np.int64)
        cnt = np.zeros(w.n_queries, dtype=np.int32)
        hits = np.zeros(w.n_queries, dtype=np.int64)
        for qi in range(w.n_queries):
            d, s, total, _ = oracle.search_bm25(shard, qr[qi].tolist(), k, total_hits_threshold=2**31 - 1)
            keys[qi] = nd.pack_keys(d, s, k_stride)
            cnt[qi], hits[qi] = len(d), total
        g_keys, g_cnt, g_hits = nd.all_gather_topk(torch.from_numpy(keys), torch.from_numpy(cnt), torch.from_numpy(hits))
        ok = True
        if rank == 0:
            full = workload.build_shard_corpus(w, qr, 1, 0)
            for qi in range(w.n_queries):
                lists = [nd.unpack_keys(g_keys[r, qi].numpy(), int(g_cnt[r, qi])) for r in range(world)]
                md, ms = oracle.topdocs_merge(k, lists)
                ed, es, etotal, _ = oracle.search_bm25(full, qr[qi].tolist(), k, total_hits_threshold=2**31 - 1)
                ok &= md.tolist() == ed.tolist() and ms.view(np.uint32).tolist() == es.view(np.uint32).tolist()
                ok &= int(g_hits[:, qi].sum()) == etotal
        # the all-to-all form (what bench.py uses when the batch divides by the world size): every rank merges its
        # own slice of the queries and must reproduce the single-process answers for that slice
        a_keys, a_cnt, a_hits = nd.all_to_all_topk(torch.from_numpy(keys), torch.from_numpy(cnt), torch.from_numpy(hits))
        per = w.n_queries // world
        full = workload.build_shard_corpus(w, qr, 1, 0)
        for j in range(per):
            qi = rank * per + j
            lists = [nd.unpack_keys(a_keys[r, j].numpy(), int(a_cnt[r, j])) for r in range(world)]
            md, ms = oracle.topdocs_merge(k, lists)
            ed, es, etotal, _ = oracle.search_bm25(full, qr[qi].tolist(), k, total_hits_threshold=2**31 - 1)
            ok &= md.tolist() == ed.tolist() and ms.view(np.uint32).tolist() == es.view(np.uint32).tolist()
            ok &= int(a_hits[:, j].sum()) == etotal
        # Row-partitioned exact vector search (BASELINE config 4; nrtgpu_dist_knn_exact): every rank scores its rows for every
        # query, the per-rank top-k lists are all-gathered and merged -- the whole matrix's answer
        rng = np.random.Generator(np.random.PCG64(5))
        n_rows, dim, kk = 4000, 32, 20
        mat = rng.standard_normal((n_rows, dim), dtype=np.float32)
        qv = rng.standard_normal((4, dim), dtype=np.float32)
        lo, hi = n_rows * rank // world, n_rows * (rank + 1) // world
        d_, s_, c_ = oracle.knn_exact(0, qv, mat[lo:hi], kk, doc_base=lo)
        vkeys = np.zeros((len(qv), 32), dtype=np.int64)
        for qi in range(len(qv)):
            vkeys[qi] = nd.pack_keys(d_[qi, : c_[qi]], s_[qi, : c_[qi]], 32)
        gk, gc, gh = nd.all_gather_topk(torch.from_numpy(vkeys), torch.from_numpy(c_.astype(np.int32)),
                                        torch.from_numpy(np.full(len(qv), hi - lo, dtype=np.int64)))
        wd, ws, wc = oracle.knn_exact(0, qv, mat, kk)
        for qi in range(len(qv)):
            lists = [nd.unpack_keys(gk[r, qi].numpy(), int(gc[r, qi])) for r in range(world)]
            md, ms = oracle.topdocs_merge(kk, lists)
            ok &= md.tolist() == wd[qi, : wc[qi]].tolist() and ms.view(np.uint32).tolist() == ws[qi, : wc[qi]].view(np.uint32).tolist()
            ok &= int(gh[:, qi].sum()) == n_rows
        # The hybrid over shards (BASELINE config 5; nrtgpu_dist_search_hybrid_batch): shard first passes -> all-gather + merge
        # (the GLOBAL first pass) -> every rank rescores ITS docs of the merged list -> the windows are gathered and merged.
        # Must equal: the global first pass rescored by one process.
        vec = np.random.Generator(np.random.PCG64(9)).standard_normal((w.n_docs, 16), dtype=np.float32)
        hq = np.random.Generator(np.random.PCG64(10)).standard_normal((w.n_queries, 16), dtype=np.float32)
        lo_d, hi_d = workload.shard_range(w.n_docs, world, rank)
        window, qw, rw = 10, 1.0, 3.0

        def rescored(docs_, scores_, qi, only_mine):
            rows = [(float(oracle.rescore_combine(float(s0), True, float(oracle.vector_score(0, hq[qi], vec[d0])), qw, rw)), int(d0))
                    for d0, s0 in zip(docs_, scores_) if (not only_mine or lo_d <= d0 < hi_d)]
            rows.sort(key=lambda r: (-r[0], r[1]))
            rows = rows[:window]
            return np.array([r[1] for r in rows], dtype=np.int32), np.array([r[0] for r in rows], dtype=np.float32)

        wkeys = np.zeros((w.n_queries, 16), dtype=np.int64)
        wcnt = np.zeros(w.n_queries, dtype=np.int32)
        merged_first = []
        for qi in range(w.n_queries):
            lists = [nd.unpack_keys(g_keys[r, qi].numpy(), int(g_cnt[r, qi])) for r in range(world)]
            md, ms = oracle.topdocs_merge(k, lists)                      # every rank holds the merged first pass
            merged_first.append((md, ms))
            wd, ws = rescored(md, ms, qi, True)
            wkeys[qi] = nd.pack_keys(wd, ws, 16)
            wcnt[qi] = len(wd)
        hk, hc, _ = nd.all_gather_topk(torch.from_numpy(wkeys), torch.from_numpy(wcnt), torch.from_numpy(np.zeros(w.n_queries, dtype=np.int64)))
        for qi in range(w.n_queries):
            lists = [nd.unpack_keys(hk[r, qi].numpy(), int(hc[r, qi])) for r in range(world)]
            fd, fs = oracle.topdocs_merge(window, lists)
            ed, es = rescored(merged_first[qi][0], merged_first[qi][1], qi, False)
            ok &= fd.tolist() == ed.tolist() and fs.view(np.uint32).tolist() == es.view(np.uint32).tolist()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_partition_and_allgather():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = 29500 + (os.getpid() % 2000)
        procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        assert ret.get(0) is True and ret.get(1) is True


def _topology_worker(rank: int, world: int, doc_shards: int, port: int, ret):
    """N = D doc-shards x R query-groups (bench.py --doc-shards): the D ranks of a group share the index and all-gather their lists
    inside the group's process group; group g runs batches g, g + R, ... of the query set.  Every group's merged answers for ITS
    batches are the oracle's whole-index answers; together the groups cover every batch exactly once."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from nrtsearch_amd import dist as nd
    from nrtsearch_amd import synth, workload
    from oracle import oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D, R, group, grank = nd.topology(world, rank, doc_shards)
        pg = nd.doc_shard_groups(world, doc_shards)
        w = workload.Workload("topology-test", 40_000, 3, 20, 8, 2, max_rank=300)
        qr = synth.make_queries(w.n_queries, w.n_terms, w.max_rank)
        B = 2                                                        # queries per batch: 4 batches
        n_batches = w.n_queries // B
        shard = workload.build_shard_corpus(w, qr, D, grank)         # my docid range of the index, global statistics
        full = workload.build_shard_corpus(w, qr, 1, 0)
        k, k_stride = w.k, 32
        ok, ran = True, []
        for step in range(n_batches // R):
            bi = (step * R + group) % n_batches                       # bench.py: batch_index
            ran.append(bi)
            keys = np.zeros((B, k_stride), dtype=np.int64)
            cnt = np.zeros(B, dtype=np.int32)
            hits = np.zeros(B, dtype=np.int64)
            for j in range(B):
                d, s_, total, _ = oracle.search_bm25(shard, qr[bi * B + j].tolist(), k, total_hits_threshold=2**31 - 1)
                keys[j] = nd.pack_keys(d, s_, k_stride)
                cnt[j], hits[j] = len(d), total
            if D > 1:
                g_keys, g_cnt, g_hits = nd.all_gather_topk(torch.from_numpy(keys), torch.from_numpy(cnt), torch.from_numpy(hits), group=pg)
            else:                                                     # replicas: nothing is exchanged
                g_keys, g_cnt, g_hits = torch.from_numpy(keys)[None], torch.from_numpy(cnt)[None], torch.from_numpy(hits)[None]
            ok &= g_keys.shape[0] == D
            for j in range(B):
                lists = [nd.unpack_keys(g_keys[r, j].numpy(), int(g_cnt[r, j])) for r in range(D)]
                md, ms = oracle.topdocs_merge(k, lists)
                ed, es, etotal, _ = oracle.search_bm25(full, qr[bi * B + j].tolist(), k, total_hits_threshold=2**31 - 1)
                ok &= md.tolist() == ed.tolist() and ms.view(np.uint32).tolist() == es.view(np.uint32).tolist()
                ok &= int(g_hits[:, j].sum()) == etotal
        ret[rank] = (bool(ok), group, grank, ran)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,doc_shards", [(4, 2), (2, 1), (4, 4)])
def test_doc_shards_times_query_groups(world, doc_shards):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = 31500 + (os.getpid() % 2000) + world * 7 + doc_shards
        procs = [ctx.Process(target=_topology_worker, args=(r, world, doc_shards, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
            assert p.exitcode == 0
        res = [ret.get(r) for r in range(world)]
        assert all(x is not None and x[0] is True for x in res), res
        R = world // doc_shards
        assert [x[1] for x in res] == [r // doc_shards for r in range(world)] and [x[2] for x in res] == [r % doc_shards for r in range(world)]
        # the ranks of a group ran the same batches; the groups together every batch once
        by_group = {}
        for x in res:
            by_group.setdefault(x[1], []).append(tuple(x[3]))
        assert all(len(set(v)) == 1 for v in by_group.values()), by_group
        assert sorted(b for g in range(R) for b in by_group[g][0]) == list(range(4)), by_group


def test_topology_arithmetic():
    sys.path.insert(0, ROOT)
    from nrtsearch_amd import dist as nd

    assert nd.topology(8, 5, 0) == (8, 1, 0, 5)
    assert nd.topology(8, 5, 2) == (2, 4, 2, 1)
    assert nd.topology(8, 5, 1) == (1, 8, 5, 0)
    assert nd.topology(1, 0, 0) == (1, 1, 0, 0)
    for bad in (3, 16):
        with pytest.raises(ValueError):
            nd.topology(8, 0, bad)


def test_shard_ranges_cover_the_index():
    sys.path.insert(0, ROOT)
    from nrtsearch_amd import workload

    for n_docs in (1, 1023, 1024, 10_000_000, 50_000_001):
        for world in (1, 2, 4, 8):
            ranges = [workload.shard_range(n_docs, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n_docs
            for (a, b), (c, d) in zip(ranges, ranges[1:]):
                assert b == c and a <= b
            assert all(a % 1024 == 0 for a, b in ranges if b > a)   # non-empty ranges start on a sub-tile boundary


def test_shard_leaves_tile_the_index():
    """layout "index": a rank owns the pieces of the INDEX's segments inside its docid range; the ranks' leaves tile the index and
    a boundary next to a segment boundary snaps onto it (no slivers)."""
    sys.path.insert(0, ROOT)
    import dataclasses

    from nrtsearch_amd import synth, workload

    for n_docs in (10_000_000, 3_000_000, 1_234_567):
        w = dataclasses.replace(workload.C3, n_docs=n_docs)
        whole = synth.tiered_segment_sizes(n_docs, w.segments_per_shard)
        for world in (1, 2, 4, 8):
            pos = 0
            pieces = []
            for r in range(world):
                lo, hi, sizes = workload.shard_leaves(w, world, r)
                assert lo == pos and sum(sizes) == hi - lo
                pos = hi
                pieces += sizes
            assert pos == n_docs
            if world == 1:
                assert pieces == whole
            # cutting never merges segments: the pieces refine the index's segments
            edges = set(np.cumsum(whole).tolist())
            assert edges <= set(np.cumsum(pieces).tolist())
            assert min(pieces) > n_docs // world // 100 or min(pieces) in whole   # no slivers besides the index's own small segments


def test_balanced_shard_layout_tiles_the_index_and_spreads_the_small_segments():
    """layout "balanced": the ranks' leaves partition the index's docid space, refine its segments (a cut falls on a 1024-doc
    boundary inside a segment), every rank holds about the same number of docs and at most a handful of leaves -- no rank is left
    with all the small segments; the corpus builder follows it (a rank's leaves need not be one contiguous docid range)."""
    sys.path.insert(0, ROOT)
    import dataclasses

    from nrtsearch_amd import synth, workload

    for n_docs in (10_000_000, 3_000_000, 1_234_567):
        w = dataclasses.replace(workload.C3, n_docs=n_docs)
        whole = synth.tiered_segment_sizes(n_docs, w.segments_per_shard)
        seg_edges = np.concatenate([[0], np.cumsum(whole)])
        for world in (2, 3, 4, 8):
            per_rank = [workload.shard_pieces(w, world, r, "balanced") for r in range(world)]
            pos = 0
            for b, g in sorted(sum(per_rank, [])):
                assert b == pos and g > 0
                seg = int(np.searchsorted(seg_edges, b, side="right")) - 1
                assert b + g <= seg_edges[seg + 1]                       # a piece lies inside one segment of the index
                assert (b - seg_edges[seg]) % 1024 == 0                   # and starts on a sub-tile boundary of it
                pos += g
            assert pos == n_docs
            loads = [sum(g for _, g in p) for p in per_rank]
            assert max(loads) - min(loads) <= max(whole[2:]) if world <= 3 else max(loads) - min(loads) <= n_docs // world // 20
            assert max(len(p) for p in per_rank) <= -(-len(whole) // world) + 3
    w = dataclasses.replace(workload.SMOKE, n_docs=60_000)
    qr = synth.make_queries(4, 3, 500)
    docs = 0
    for r in range(4):
        c = workload.build_shard_corpus(w, qr, 4, r, layout="balanced")
        assert [s.doc_base for s in c.segments] == sorted(s.doc_base for s in c.segments)
        assert [(s.doc_base, s.max_doc) for s in c.segments] == workload.shard_pieces(w, 4, r, "balanced")
        docs += sum(s.max_doc for s in c.segments)
    assert docs == w.n_docs


def test_key_packing_roundtrip_and_order():
    sys.path.insert(0, ROOT)
    from nrtsearch_amd import dist as nd

    docs = np.array([5, 9, 2, 7], dtype=np.int32)
    scores = np.array([3.5, 3.5, 1.25, 0.0], dtype=np.float32)
    keys = nd.pack_keys(docs, scores, 8)
    d, s = nd.unpack_keys(keys, 4)
    assert d.tolist() == docs.tolist() and s.tolist() == scores.tolist()
    u = keys.view(np.uint64)[:4]
    assert all(u[i] > u[i + 1] for i in range(3))   # (score desc, doc asc) == descending keys
