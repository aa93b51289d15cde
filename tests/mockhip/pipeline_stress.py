"""Run by hand or by tests/test_planner_host.py (a short run) in a subprocess with tests/mockhip preloaded: the shape of bench.py's
multi-GPU loop without a GPU -- two submitting threads begin shard searches over three result buffers (1024 queries over 3 leaves:
the planner's helpers take part), the main thread waits for them in step order, merges and hands the buffer back -- for STEPS
steps.  Any error of a call, a wrong count or a hang (the watchdog dumps every thread's stack) ends it with a non-zero code."""
import os, sys, threading, time, faulthandler
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nrtsearch_amd import api, synth, workload
STEPS = int(os.environ.get("STEPS", "200"))
faulthandler.dump_traceback_later(float(os.environ.get("WATCHDOG", "120")), exit=True)
w = workload.Workload("pipeline stress", 300_000, 2, 100, 4096, 3)
B = 1024
qr = synth.make_queries(4096, w.n_terms, w.max_rank)
corpus = workload.build_shard_corpus(w, qr)
ctx = api.GpuContext(0, max_batch=B, host_threads=4)
leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
queries = workload.boolean_queries(qr)
mgr = api.TopScoreDocCollectorManager(w.k)
batches = [api.PreparedBatch(sr, queries[i: i + B], [mgr] * B) for i in range(0, len(queries), B)]
ks = 112
NB = 3
bufs = [(np.zeros((B, ks), np.int64), np.zeros(B, np.int32), np.zeros(B, np.int64), np.zeros(B, np.int64)) for _ in range(NB)]
merger = api.PreparedMerge(ctx, 1, B, ks, [w.k] * B, [api.TOTAL_HITS_THRESHOLD] * B)
free = [threading.Semaphore(1) for _ in range(NB)]
pending = [None] * STEPS

def produce(i):
    b = i % NB
    free[b].acquire()
    k, c, h, g = bufs[b]
    pending[i] = batches[i % len(batches)].begin_shard_device(ks, k.ctypes.data, c.ctypes.data, h.ctypes.data, 2, g.ctypes.data)
    return b

t0 = time.perf_counter()
with ThreadPoolExecutor(max_workers=2) as ex:
    futs = [ex.submit(produce, i) for i in range(STEPS)]
    for i in range(STEPS):
        b = futs[i].result()
        api.PreparedBatch.wait_device(pending[i])
        k, c, h, g = bufs[b]
        merger.run(k.ctypes.data, c.ctypes.data, h.ctypes.data)
        batches[i % len(batches)].note_shard_speculation(B, 0)
        free[b].release()
st = ctx.stats()
assert st["batches"] >= STEPS, st["batches"]
print(f"{STEPS} steps in {time.perf_counter() - t0:.2f} s, batches {st['batches']}", flush=True)
for g_ in leaves:
    g_.release()
ctx.close()
print("done", flush=True)
