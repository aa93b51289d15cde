"""One rank of a two-process stress of the library's multi-GPU loop without a GPU (tests/mockhip preloaded, the collective carried by
tests/mockrccl): bench.py's shape at N > 1 -- two submitting threads begin shard searches (nrtgpu_search_bm25_shard_device_begin)
over three result buffers, the main thread waits in step order and runs nrtgpu_dist_exchange_merge_checked; every PLANT-th step
guesses are planted that no merged list reaches, so the ranks must agree on the failed queries and re-run them together
(nrtgpu_dist_search_bm25_batch_mode with NRTGPU_EXCHANGE_NO_SPECULATION).  argv: rank world sync_dir mode steps."""
import ctypes as C
import os, sys, threading, time, faulthandler
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nrtsearch_amd import api, synth, workload
rank, world, sync_dir, mode_name, STEPS = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
PLANT = int(os.environ.get("PLANT", "7"))
faulthandler.dump_traceback_later(float(os.environ.get("WATCHDOG", "150")), exit=True)
mode = api.EXCHANGE_ALLTOALL if mode_name == "alltoall" else api.EXCHANGE_ALLGATHER
w = workload.Workload("dist pipeline stress", 400_000, 2, 100, 2048, 4)
B = 512
qr = synth.make_queries(2048, w.n_terms, w.max_rank)
corpus = workload.build_shard_corpus(w, qr, world, rank)
ctx = api.GpuContext(0, max_batch=B, host_threads=4)
leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
id_path = os.path.join(sync_dir, "unique_id")
if rank == 0:
    with open(id_path + ".tmp", "wb") as f:
        f.write(api.GpuContext.dist_unique_id())
    os.rename(id_path + ".tmp", id_path)
t0 = time.time()
while not os.path.exists(id_path):
    assert time.time() - t0 < 120, "rank 0 never published the unique id"
    time.sleep(0.005)
ctx.dist_init(world, rank, open(id_path, "rb").read())
queries = workload.boolean_queries(qr)
mgr = api.TopScoreDocCollectorManager(w.k)
batches = [api.PreparedBatch(sr, queries[i: i + B], [mgr] * B) for i in range(0, len(queries), B)]
ks, NB = 112, 3
bufs = [(np.zeros((B, ks), np.int64), np.zeros(B, np.int32), np.zeros(B, np.int64), np.zeros(B, np.uint64)) for _ in range(NB)]
merger = api.PreparedMerge(ctx, world, B, ks, [w.k] * B, [api.TOTAL_HITS_THRESHOLD] * B)
free = [threading.Semaphore(1) for _ in range(NB)]
pending = [None] * STEPS

def produce(i):
    b = i % NB
    free[b].acquire()
    k, c, h, g = bufs[b]
    pending[i] = batches[i % len(batches)].begin_shard_device(ks, k.ctypes.data, c.ctypes.data, h.ctypes.data, world, g.ctypes.data)
    return b

reruns = 0
with ThreadPoolExecutor(max_workers=2) as ex:
    futs = [ex.submit(produce, i) for i in range(STEPS)]
    for i in range(STEPS):
        b = futs[i].result()
        api.PreparedBatch.wait_device(pending[i])
        k, c, h, g = bufs[b]
        want = []
        if PLANT and i % PLANT == PLANT - 1:      # (no kernel ran: "device" memory is host memory -- guesses nobody's list reaches)
            mine = [(3 * i + 11 * rank + j * 37) % B for j in range(3)]
            g[mine] = np.uint64(1) << np.uint64(62)
            want = sorted({(3 * i + 11 * r + j * 37) % B for r in range(world) for j in range(3)})
        bad = merger.run_dist_checked(k.ctypes.data, c.ctypes.data, h.ctypes.data, g.ctypes.data, mode)
        assert [int(x) for x in bad] == want, (i, [int(x) for x in bad], want)
        g[:] = 0
        free[b].release()
        if len(bad):
            q0 = (i % len(batches)) * B
            again = sr.dist_search_batch([queries[q0 + int(j)] for j in bad], [mgr] * len(bad), mode=api.EXCHANGE_ALLGATHER | api.EXCHANGE_NO_SPECULATION)
            assert len(again) == len(bad) and all(a is not None for a in again)
            reruns += 1
print(f"rank {rank}: {STEPS} steps, {reruns} re-run calls", flush=True)
ctx.dist_close()
for g_ in leaves:
    g_.release()
ctx.close()
print("done", flush=True)
