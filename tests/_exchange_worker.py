"""One rank of the two-process cross-GPU exchange test (tests/test_exchange_gpu.py): scans its docid shard of a
small index with the exchange open and saves the device-resident results of the last epoch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    shm_name, sync_dir, out_path = sys.argv[3], sys.argv[4], sys.argv[5]
    n_docs, n_queries, k, epochs = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9])
    import torch  # first: its bundled HIP runtime must be the one libnrtgpu.so binds to
    import numpy as np

    from nrtsearch_amd import api, synth, workload

    torch.cuda.set_device(0)
    w = workload.Workload("exchange test", n_docs, 5, k, n_queries, 4)
    qr = synth.make_queries(n_queries, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr, world, rank)
    ctx = api.GpuContext(0, max_batch=n_queries)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    pb = api.PreparedBatch(sr, workload.boolean_queries(qr), [api.TopScoreDocCollectorManager(k)] * n_queries)
    if shm_name != "-":
        ctx.exchange_open(shm_name, world, rank)
    # ranks synchronise once between exchange_open and the first search
    open(os.path.join(sync_dir, f"ready_{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(sync_dir, f"ready_{r}")) for r in range(world)):
        if time.time() - t0 > 120:
            sys.exit("peer never became ready")
        time.sleep(0.005)
    k_stride = (k + 15) // 16 * 16
    keys = torch.zeros((n_queries, k_stride), dtype=torch.int64, device="cuda")
    cnt = torch.zeros((n_queries,), dtype=torch.int32, device="cuda")
    hits = torch.zeros((n_queries,), dtype=torch.int64, device="cuda")
    for e in range(epochs):
        pb.run_device(k_stride, keys.data_ptr(), cnt.data_ptr(), hits.data_ptr(), epoch=e if shm_name != "-" else -1)
    torch.cuda.synchronize()
    np.savez(out_path, keys=keys.cpu().numpy().view(np.uint64), cnt=cnt.cpu().numpy(), hits=hits.cpu().numpy())
    ctx.exchange_close()
    for l in leaves:
        l.release()
    ctx.close()


if __name__ == "__main__":
    main()
