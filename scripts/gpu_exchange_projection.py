#!/usr/bin/env python3
"""Projection of the cross-GPU bound exchange on ONE GPU: rank 0's shard of a W-GPU C3 job is scanned with the
exchange table open while the test process plays the other W-1 ranks, writing into their rows what they would
publish: the score that ceil(k / (W - 1)) of THEIR docs reach (taken from a run of each shard alone).  Compared with
rank 0's shard scanned without exchange and with the (unattainable) global k-th score preset."""
import argparse
import json
import os
import sys
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (HIP runtime first)

from nrtsearch_amd import api, synth, workload  # noqa: E402


def searcher_for(corpus, n_q):
    ctx = api.GpuContext(0, max_batch=n_q, collect_timing=True)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    return ctx, leaves, api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=6)
    a = ap.parse_args()
    w, W, n_q = workload.C3, a.world, a.queries
    qr = synth.make_queries(n_q, w.n_terms, w.max_rank)
    queries = workload.boolean_queries(qr)
    mgrs = [api.TopScoreDocCollectorManager(w.k)] * n_q
    k2 = -(-w.k // (W - 1))
    peer_bounds = np.zeros((W, n_q), dtype=np.float32)
    for r in range(1, W):   # what rank r would publish once it has seen its shard: its k2-th best score
        corpus = workload.build_shard_corpus(w, qr, W, r)
        ctx, leaves, sr = searcher_for(corpus, n_q)
        res = sr.search_batch(queries, mgrs)
        peer_bounds[r] = [float(t.scores[k2 - 1]) if len(t.scores) >= k2 else 0.0 for t in res]
        for l in leaves:
            l.release()
        ctx.close()
    corpus = workload.build_shard_corpus(w, qr, W, 0)
    ctx, leaves, sr = searcher_for(corpus, n_q)
    pb = api.PreparedBatch(sr, queries, mgrs)
    k_stride = (w.k + 15) // 16 * 16
    keys = torch.zeros((n_q, k_stride), dtype=torch.int64, device="cuda")
    cnt = torch.zeros((n_q,), dtype=torch.int32, device="cuda")
    hits = torch.zeros((n_q,), dtype=torch.int64, device="cuda")

    def timed(epoch0):
        pb.run_device(k_stride, keys.data_ptr(), cnt.data_ptr(), hits.data_ptr(), epoch=epoch0)
        ctx.reset_stats()
        for i in range(a.steps):
            pb.run_device(k_stride, keys.data_ptr(), cnt.data_ptr(), hits.data_ptr(), epoch=(epoch0 + 1 + i) if epoch0 >= 0 else -1)
        st = ctx.stats()   # (both scorers take part in the exchange: the batch's device time per step)
        return (st["scan_ms"] + st["maxscore_ms"]) / max(1, st["batches"])

    plain = timed(-1)
    name = f"/nrtgpu_proj_{uuid.uuid4().hex[:12]}"
    ctx.exchange_open(name, W, 0)
    table = np.memmap("/dev/shm" + name, dtype=np.uint64, mode="r+", shape=(8, W, n_q))
    for e in range(0, a.steps + 2):   # the peers' rows for every epoch used below (tag = epoch + 1)
        for r in range(1, W):
            table[e % 8, r, :] = (np.uint64(e + 1) << np.uint64(32)) | peer_bounds[r].view(np.uint32).astype(np.uint64)
        if e % 8 == 7 or e == a.steps + 1:
            table.flush()
    assert a.steps + 2 <= 8, "one slot per epoch in this experiment"
    exch = timed(0)
    del table
    ctx.exchange_close()
    os.unlink("/dev/shm" + name)
    print(json.dumps({"world": W, "scorer_ms_rank0_shard": round(plain, 3), "scorer_ms_with_peers_published": round(exch, 3)}))


if __name__ == "__main__":
    main()
