"""The library's OWN multi-GPU path at world = 2, end to end, on the one GPU the pool has ("a byte has never moved
between two ranks through nrtgpu_dist_*"; tests/test_dist_gloo.py drives torch.distributed and the oracle, not dist.cpp).  Two
processes, each a rank with its own context and ITS docid shard of the index, call nrtgpu_dist_search_bm25_batch_mode,
nrtgpu_dist_knn_exact and nrtgpu_dist_search_hybrid_batch -- real kernels, real lists; the collective is carried by
tests/mockrccl (a stand-in librccl.so.1 that moves a message through /dev/shm and checks that both ranks agree about its size:
TEST INFRASTRUCTURE, see its header).  The merged answers must be the whole-index answers of one context over all leaves: docids,
ranks, score bits -- for the all-gather form (every rank holds every answer) and the all-to-all form (a rank holds its slice of
the batch).  What this does NOT show: RCCL itself, xGMI, or any rate."""
import os
import pickle
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from nrtsearch_amd import synth, workload

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def mockrccl(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is not here")
    d = tmp_path_factory.mktemp("mockrccl")
    out = str(d / "librccl.so.1")
    subprocess.run([HIPCC, "-O1", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", "-Wl,-soname,librccl.so.1",
                    os.path.join(ROOT, "tests", "mockrccl", "mockrccl.cpp"), "-o", out], check=True)
    return str(d)


@pytest.mark.parametrize("variant", ["iid", "sorted"])
def test_two_ranks_through_the_librarys_collective_equal_the_whole_index(mockrccl, variant, oracle):
    """variant "iid": every posting list an independent draw -- a shard IS a sample of the index, the shards' speculative
    thresholds (guesses at the WHOLE search's k-th score, search.cpp: spec_world) stand.  "sorted": docs numbered by length, so
    rank 0's docid range holds the short docs and with them most of every top-k -- its guesses are too high, the check against
    the merged lists catches them and every rank runs those queries again: the answers are the whole-index answers either way."""
    from nrtsearch_amd import api

    world, n_docs, n_q, k = 2, 600_000, 32, 100
    sync_dir = tempfile.mkdtemp(prefix="nrtgpu_dist2_")
    outs = [os.path.join(sync_dir, f"rank{r}.pkl") for r in range(world)]
    env = dict(os.environ, LD_LIBRARY_PATH=mockrccl + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    env.pop("NRTGPU_LIB_PATH", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(r), str(world), sync_dir, outs[r],
                               str(n_docs), str(n_q), str(k), variant], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=600)
            logs.append(o)
        assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
        ranks = [pickle.load(open(o, "rb")) for o in outs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(sync_dir, ignore_errors=True)
    # the whole index in ONE context: what the merged answers must equal
    w = workload.Workload("two-rank dist test", n_docs, 4, k, n_q, 4)
    qr = synth.make_queries(n_q, w.n_terms, w.max_rank)
    pieces = [workload.build_shard_corpus(w, qr, world, r, variant=variant) for r in range(world)]
    ctx = api.GpuContext(0, max_batch=64)
    rng = np.random.default_rng(4242)
    dim = 32
    all_vecs = rng.standard_normal((n_docs, dim)).astype(np.float32)
    leaves = []
    for c in pieces:
        for seg in c.segments:
            g = api.GpuSegment(ctx, seg.max_doc, seg.doc_base)
            g.add_field_norms(0, seg.norms)
            g.add_terms(0, seg.term_ids, seg.offsets, seg.docids, seg.freqs)
            g.add_vectors(1, all_vecs[seg.doc_base: seg.doc_base + seg.max_doc])
            g.seal()
            leaves.append(g)
    try:
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(pieces[0]))
        queries = workload.boolean_queries(qr)
        mgr = api.TopScoreDocCollectorManager(k)
        whole = sr.search_batch(queries, [mgr] * n_q)
        qv = all_vecs[:8] + np.float32(0.25)
        whole_knn = sr.knn_exact(1, "cosine", qv, 10)
        whole_hy = sr.search_hybrid_batch(queries[:8], [mgr] * 8, 1, "cosine", qv, 20, 1.0, 2.0)

        def same(got, exp, with_relation=True):
            docs, scores = got[0], got[1]
            assert docs.tolist() == exp.docs.tolist() and scores.view(np.uint32).tolist() == exp.scores.view(np.uint32).tolist()
            if with_relation:
                assert got[3] == exp.relation_gte
                assert got[2] > 0 if exp.relation_gte else got[2] == exp.total_hits   # (GTE: each shard's lower bound, summed)

        # the shards speculated (every rank the same verdicts: the re-runs are collective calls); a shard that is no sample of the
        # index had its guesses caught
        for r in range(world):
            for form in ("allgather", "alltoall"):
                sp = ranks[r]["spec_" + form]
                assert sp == ranks[0]["spec_" + form] or sp["queries"] == 0 or ranks[0]["spec_" + form]["queries"] == 0, (r, form, sp)
        print("shard-level speculation", variant, [(ranks[r]["spec_allgather"], ranks[r]["spec_alltoall"]) for r in range(world)])
        assert any(ranks[r]["spec_allgather"]["queries"] == n_q for r in range(world)), [ranks[r]["spec_allgather"] for r in range(world)]
        reruns = max(ranks[r]["spec_allgather"]["reruns"] for r in range(world))
        assert (reruns <= 2) if variant == "iid" else (reruns >= 1), reruns
        # (the pipelined form: both ranks must have been told the same queries to run again)
        for form in ("allgather", "alltoall"):
            assert ranks[0]["pipelined_failed_" + form] == ranks[1]["pipelined_failed_" + form]
        if variant == "sorted":
            assert len(ranks[0]["pipelined_failed_allgather"]) >= 1
        # ... and against the ORACLE directly (the lists nrtgpu_dist_* merged were only ever compared with another HIP
        # run): docids, ranks and score bits of every form on every rank that holds the answer equal the CPU restatement's exhaustive
        # search over the whole index (one corpus object over all leaves, index-global statistics); the relation is the oracle's too
        whole_corpus = synth.Corpus(n_docs=pieces[0].n_docs, doc_count=pieces[0].doc_count, sum_total_term_freq=pieces[0].sum_total_term_freq,
                                    segments=[seg for c in pieces for seg in c.segments], doc_freq=pieces[0].doc_freq)   # the leaves the ranks hold
        for qi in range(n_q):
            edocs, escores, etotal, egte = oracle.search_bm25(whole_corpus, [int(t) for t in qr[qi]], k)
            for what in ("bm25", "bm25_nospec", "bm25_pipelined"):
                for form in ("allgather", "alltoall"):
                    for r in range(world):
                        g = ranks[r][f"{what}_{form}"][qi]
                        if g is None:
                            continue
                        assert g[0].tolist() == edocs.tolist(), f"{what} {form} rank {r} query {qi}: docids / ranks differ from the oracle's"
                        assert g[1].view(np.uint32).tolist() == escores.view(np.uint32).tolist(), f"{what} {form} rank {r} query {qi}: score bits"
                        assert (max(k, 1000) < g[2] <= etotal) if g[3] else (g[2] == etotal and not egte), (what, form, r, qi, g[2], g[3], etotal, egte)
        # a rank whose part of a call failed (tests/_dist_worker.py: rank 1's deadline) entered the exchange with its status: nobody
        # hung, both ranks got an error (rank 0's names rank 1), and the next call went through with the whole-index answers
        for r in range(world):
            for form in ("allgather", "alltoall"):
                msg = ranks[r]["one_rank_fails_" + form]
                assert ("deadline" in msg) if r == 1 else ("rank 1 of 2 failed" in msg), (r, form, msg)
            for qi in range(n_q):
                same(ranks[r]["after_failure_allgather"][qi], whole[qi])
        cases = [("bm25", whole, True), ("bm25_nospec", whole, True), ("bm25_pipelined", whole, True)]
        if variant == "iid":
            cases += [("knn", whole_knn, False), ("hybrid", whole_hy, True)]
        for what, exp_list, rel in cases:
            n = len(exp_list)
            # all-gather: every rank holds every answer
            for r in range(world):
                got = ranks[r][what + "_allgather"]
                assert all(g is not None for g in got)
                for qi in range(n):
                    same(got[qi], exp_list[qi], rel)
            # all-to-all: every answer on exactly one rank (its slice of the batch); a batch the ranks cannot share evenly is gathered whole
            owners = [[r for r in range(world) if ranks[r][what + "_alltoall"][qi] is not None] for qi in range(n)]
            if n % world == 0:
                assert all(len(o) == 1 for o in owners), owners
                assert [o[0] for o in owners] == [qi * world // n for qi in range(n)]
            for qi in range(n):
                for r in owners[qi]:
                    same(ranks[r][what + "_alltoall"][qi], exp_list[qi], rel)
    finally:
        for g in leaves:
            g.release()
        ctx.close()
