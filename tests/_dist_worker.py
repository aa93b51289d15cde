"""One rank of the two-rank test of the library's own multi-GPU path (tests/test_dist_two_ranks_gpu.py): a process with its own
context on the one GPU of the box, holding ITS docid shard of the index, that calls nrtgpu_dist_search_bm25_batch_mode /
nrtgpu_dist_knn_exact / nrtgpu_dist_search_hybrid_batch with the collective carried by tests/mockrccl (a stand-in librccl.so.1 on
LD_LIBRARY_PATH: messages are /dev/shm files).  No torch here: its bundled RCCL must not be the one dist.cpp binds to."""
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, sync_dir, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    n_docs, n_queries, k = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    variant = sys.argv[8] if len(sys.argv) > 8 else "iid"
    host_only = os.environ.get("NRTGPU_TEST_HOST_ONLY") == "1"   # tests/test_dist_two_ranks_host.py: against tests/mockhip, BM25 only
    import numpy as np

    from nrtsearch_amd import api, synth, workload

    w = workload.Workload("two-rank dist test", n_docs, 4, k, n_queries, 4)
    qr = synth.make_queries(n_queries, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr, world, rank, variant=variant)   # this rank's leaves, index-global statistics
    ctx = api.GpuContext(0, max_batch=max(64, n_queries))
    leaves = []
    rng = np.random.default_rng(4242)                                  # the same rows on every rank: each keeps its docid range's
    dim = 32
    all_vecs = rng.standard_normal((n_docs, dim)).astype(np.float32)
    for seg in corpus.segments:
        g = api.GpuSegment(ctx, seg.max_doc, seg.doc_base)
        g.add_field_norms(0, seg.norms)
        g.add_terms(0, seg.term_ids, seg.offsets, seg.docids, seg.freqs)
        if not host_only:
            g.add_vectors(1, all_vecs[seg.doc_base: seg.doc_base + seg.max_doc])
        g.seal()
        leaves.append(g)
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    # the unique id travels through the file system (a deployment would use its own means)
    id_path = os.path.join(sync_dir, "unique_id")
    if rank == 0:
        uid = api.GpuContext.dist_unique_id()
        with open(id_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(id_path + ".tmp", id_path)
    t0 = time.time()
    while not os.path.exists(id_path):
        if time.time() - t0 > 120:
            sys.exit("rank 0 never published the unique id")
        time.sleep(0.005)
    uid = open(id_path, "rb").read()
    ctx.dist_init(world, rank, uid)
    queries = workload.boolean_queries(qr)
    mgr = api.TopScoreDocCollectorManager(k)
    out = {}
    for name, mode in (("allgather", api.EXCHANGE_ALLGATHER), ("alltoall", api.EXCHANGE_ALLTOALL)):
        c0 = ctx.spec_counters()
        got = sr.dist_search_batch(queries, [mgr] * n_queries, mode=mode)
        c1 = ctx.spec_counters()
        out["bm25_" + name] = [None if g is None else (g.docs, g.scores, g.total_hits, g.relation_gte) for g in got]
        out["spec_" + name] = {k_: c1[k_] - c0[k_] for k_ in ("queries", "reruns")}   # the shard-level guesses of this call
        got = sr.dist_search_batch(queries, [mgr] * n_queries, mode=mode | api.EXCHANGE_NO_SPECULATION)
        out["bm25_nospec_" + name] = [None if g is None else (g.docs, g.scores, g.total_hits, g.relation_gte) for g in got]
        if variant != "iid" or host_only:
            continue
        qv = all_vecs[:8] + np.float32(0.25)
        kn = sr.dist_knn_exact(1, "cosine", qv, 10, mode=mode)
        out["knn_" + name] = [None if g is None else (g.docs, g.scores, g.total_hits) for g in kn]
        hy = sr.dist_search_hybrid_batch(queries[:8], [mgr] * 8, 1, "cosine", qv, 20, 1.0, 2.0, mode=mode)
        out["hybrid_" + name] = [None if g is None else (g.docs, g.scores, g.total_hits, g.relation_gte) for g in hy]
    # the same search as a caller that pipelines would run it (bench.py at N > 1): nrtgpu_search_bm25_shard_device_begin leaves this
    # shard's lists and guesses in HBM, nrtgpu_dist_exchange_merge_checked exchanges, merges and checks them; what fails is run
    # again by every rank without speculation
    import ctypes as C
    hip = C.CDLL(os.environ.get("NRTGPU_TEST_HIP_LIB", "libamdhip64.so"))

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        assert hip.hipMemset(p, 0, C.c_size_t(nbytes)) == 0
        return p.value

    ks = (k + 15) // 16 * 16
    d_keys, d_cnt, d_hits, d_guess = dmalloc(n_queries * ks * 8), dmalloc(n_queries * 4), dmalloc(n_queries * 8), dmalloc(n_queries * 8)
    pb = api.PreparedBatch(sr, queries, [mgr] * n_queries)
    for name, mode in (("allgather", api.EXCHANGE_ALLGATHER), ("alltoall", api.EXCHANGE_ALLTOALL)):
        h = pb.begin_shard_device(ks, d_keys, d_cnt, d_hits, world, d_guess)
        api.PreparedBatch.wait_device(h)
        if host_only and os.environ.get("NRTGPU_TEST_POKE_GUESSES") == "1":
            # (against the stand-in HIP runtime "device" memory is host memory and no kernel ran: plant guesses -- rank 0 for queries
            #  1 and 5, rank 1 for 5 and 9 -- that no merged list can reach: every rank must come to know all three, whoever owns them)
            planted = (C.c_uint64 * n_queries)()
            for q_ in ((1, 5) if rank == 0 else (5, 9)):
                planted[q_] = 1 << 62
            C.memmove(d_guess, planted, n_queries * 8)
        pm = api.PreparedMerge(ctx, world, n_queries, ks, [k] * n_queries, [api.TOTAL_HITS_THRESHOLD] * n_queries)
        bad = pm.run_dist_checked(d_keys, d_cnt, d_hits, d_guess, mode)
        pb.note_shard_speculation(n_queries, len(bad))
        got = [pm.topdocs(qi) if pm.owned(qi) else None for qi in range(n_queries)]
        if len(bad):
            again = sr.dist_search_batch([queries[int(j)] for j in bad], [mgr] * len(bad), mode=api.EXCHANGE_ALLGATHER | api.EXCHANGE_NO_SPECULATION)
            for i, j in enumerate(bad):
                if got[int(j)] is not None:
                    got[int(j)] = again[i]
        out["bm25_pipelined_" + name] = [None if g is None else (g.docs, g.scores, g.total_hits, g.relation_gte) for g in got]
        out["pipelined_failed_" + name] = [int(j) for j in bad]
    for p_ in (d_keys, d_cnt, d_hits, d_guess):
        hip.hipFree(C.c_void_p(p_))
    # ONE rank's part of a one-call search fails (rank 1: its thread's deadline has passed before the shard search is planned):
    # it must still enter the exchange -- with empty lists and its status -- so that the other rank is not left waiting in the
    # collective; EVERY rank gets an error (the failed one its own, the others one that names it), in both exchange forms, and the
    # communicator is usable afterwards ("a failing rank strands the others")
    for name, mode in (("allgather", api.EXCHANGE_ALLGATHER), ("alltoall", api.EXCHANGE_ALLTOALL)):
        if rank == 1:
            api.GpuContext.set_thread_deadline(-1.0)
        try:
            sr.dist_search_batch(queries, [mgr] * n_queries, mode=mode)
            out["one_rank_fails_" + name] = "no error"
        except Exception as e:   # noqa: BLE001
            out["one_rank_fails_" + name] = str(e)
        finally:
            api.GpuContext.set_thread_deadline(None)
    got = sr.dist_search_batch(queries, [mgr] * n_queries, mode=api.EXCHANGE_ALLGATHER)
    out["after_failure_allgather"] = [None if g is None else (g.docs, g.scores, g.total_hits, g.relation_gte) for g in got]
    out["stats"] = ctx.stats()
    with open(out_path, "wb") as f:
        pickle.dump(out, f)
    ctx.dist_close()
    for l in leaves:
        l.release()
    ctx.close()


if __name__ == "__main__":
    main()
