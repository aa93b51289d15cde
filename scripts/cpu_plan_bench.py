"""The HOST cost of one step of one rank of an 8-GPU C3 job, without a GPU: tests/mockhip preloaded (kernels do nothing), this
rank's shard uploaded, 1024-query batches begun and waited for in a loop -- what is timed is nrtgpu_search_bm25_shard_device_begin
(plan + marshal + enqueue).  Usage:
    LD_PRELOAD=<libmockhip.so> [NRTGPU_LIB_PATH=.../libnrtgpu_dev.so NRTGPU_PLAN_TRACE=1] python scripts/cpu_plan_bench.py [world] [threads]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nrtsearch_amd import api, synth, workload   # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w = workload.C3
w.n_docs = int(os.environ.get("DOCS", str(w.n_docs)))
B = 1024
qranks = synth.make_queries(B * 4, w.n_terms, w.max_rank)
t0 = time.time()
corpus = workload.build_shard_corpus(w, qranks, world, 0)
print(f"corpus {time.time() - t0:.1f} s, {len(corpus.segments)} leaves", flush=True)
ctx = api.GpuContext(device_id=0, max_batch=B, host_threads=threads)
leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
queries = workload.boolean_queries(qranks)
mgr = api.TopScoreDocCollectorManager(w.k)
pbs = [api.PreparedBatch(sr, queries[i: i + B], [mgr] * B) for i in range(0, len(queries), B)]
ks = (w.k + 15) // 16 * 16
keys = np.zeros((B, ks), dtype=np.int64)
cnt = np.zeros(B, dtype=np.int32)
hits = np.zeros(B, dtype=np.int64)
guess = np.zeros(B, dtype=np.int64)
best = 1e9
for rep in range(3):
    n = 60
    t0 = time.perf_counter()
    prev = None
    for i in range(n):   # one submitting thread, two searches in flight: the next one is begun before the last one is waited for
        tb = time.perf_counter()
        h = pbs[i % len(pbs)].begin_shard_device(ks, keys.ctypes.data, cnt.ctypes.data, hits.ctypes.data, world if world > 1 else 0, guess.ctypes.data)
        best = min(best, time.perf_counter() - tb)
        if prev is not None:
            api.PreparedBatch.wait_device(prev)
        prev = h
    api.PreparedBatch.wait_device(prev)
    dt = (time.perf_counter() - t0) / n * 1e3
    st = ctx.stats()
    print(f"begin + wait: {dt:.3f} ms per 1024-query batch (mean); plan {st['host_plan_ms'] / max(1, st['batches']):.3f} ms (mean); fastest begin {best * 1e3:.3f} ms", flush=True)
