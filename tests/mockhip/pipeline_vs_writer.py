"""Run by tests/test_planner_host.py in a subprocess with tests/mockhip preloaded (MOCKHIP_SYNC_US > 0: a stream synchronisation
"takes" that long, so searches stay in flight for a while): a begin / wait pipeline of depth 3 whose handles are only handed to
the waiting thread after a whole round has been begun, next to a thread that rewrites a mask of every leaf in a loop.  With a
content lock that parks a pipelined search behind a WAITING writer this deadlocks (the writer waits for a search whose handle the
parked thread still holds): found on the GPU in round 4, reproduced here without one.  Prints "finished True ..." or dumps the
threads' stacks and exits 1."""
import os, sys, threading, queue, time, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nrtsearch_amd import api, synth, workload
faulthandler.dump_traceback_later(40, exit=True)
w = workload.Workload("begin-wait lifetime test", 250_000, 4, 100, 48, 3)
qr = synth.make_queries(48, w.n_terms, w.max_rank)
corpus = workload.build_shard_corpus(w, qr)
ctx = api.GpuContext(0, max_batch=16)
leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
queries = workload.boolean_queries(qr)
mgr = api.TopScoreDocCollectorManager(w.k)
k_stride = 112
pbs = [api.PreparedBatch(sr, queries[i: i + 16], [mgr] * 16) for i in range(0, 48, 16)]
bufs = [(np.zeros((16, k_stride), np.int64), np.zeros(16, np.int32), np.zeros(16, np.int64)) for _ in range(6)]
stop, errors, done = threading.Event(), [], threading.Event()
pending = queue.Queue()
cnt = {"w": 0, "b": 0, "wait": 0}
def writer():
    i = 0
    while not stop.is_set():
        for leaf, seg in zip(leaves, corpus.segments):
            leaf.set_mask(9, synth.random_mask(seg.max_doc, 0.5, i))
            cnt["w"] += 1
        i += 1
def waiter():
    while True:
        h = pending.get()
        if h is None: break
        api.PreparedBatch.wait_device(h)
        cnt["wait"] += 1
    done.set()
def pipeline():
    for rnd in range(40):
        hs = [pb.begin_device(k_stride, *(t.ctypes.data for t in bufs[3 + b])) for b, pb in enumerate(pbs)]
        cnt["b"] += 3
        for h in hs: pending.put(h)
    pending.put(None)
ts = [threading.Thread(target=writer, daemon=True), threading.Thread(target=waiter, daemon=True), threading.Thread(target=pipeline, daemon=True)]
for t in ts: t.start()
ok = done.wait(30)
print("finished", ok, cnt, flush=True)
if not ok: faulthandler.dump_traceback(all_threads=True)
stop.set()
os._exit(0 if ok else 1)
