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
