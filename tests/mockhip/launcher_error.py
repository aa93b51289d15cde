"""Run by tests/test_planner_host.py in a subprocess with tests/mockhip preloaded: a search begun with
nrtgpu_search_bm25_batch_device_begin is enqueued by the context's launcher thread (runtime_internal.h: Launcher) -- so a launch
that fails is found out on THAT thread, after _begin has returned.  The stand-in makes the next launch fail: _begin still says OK,
nrtgpu_pending_wait reports the failure with its message, the workspace is free again and the following searches run, in both forms."""
import ctypes as C
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nrtsearch_amd import _lib, api, synth, workload
faulthandler.dump_traceback_later(60, exit=True)
mock = C.CDLL(os.environ["LD_PRELOAD"])
w = workload.Workload("launcher error test", 120_000, 3, 50, 32, 2)
qr = synth.make_queries(32, w.n_terms, w.max_rank)
corpus = workload.build_shard_corpus(w, qr)
ctx = api.GpuContext(0, max_batch=16)
leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
queries = workload.boolean_queries(qr)
mgr = api.TopScoreDocCollectorManager(w.k)
pbs = [api.PreparedBatch(sr, queries[i: i + 16], [mgr] * 16) for i in (0, 16)]
ks = 64
bufs = [(np.zeros((16, ks), np.int64), np.zeros(16, np.int32), np.zeros(16, np.int64)) for _ in range(2)]

def begin(i):
    k, c, h = bufs[i % 2]
    return pbs[i % 2].begin_device(ks, k.ctypes.data, c.ctypes.data, h.ctypes.data)

api.PreparedBatch.wait_device(begin(0))                      # a search that works (term tables resident from here on)
mock.mockhip_fail_next_launches(1)
h = begin(1)                                                 # _begin only plans: it cannot know
try:
    api.PreparedBatch.wait_device(h)
    print("the failed launch went unnoticed")
    sys.exit(1)
except _lib.NrtGpuError as e:
    assert e.code == _lib.NRTGPU_ERR_HIP, e.code
    print("wait reported:", str(e)[:120])
for i in range(6):                                           # nothing is left behind: four workspaces, six more searches, both forms
    api.PreparedBatch.wait_device(begin(i))
    pbs[i % 2].run()
# the synchronous form, on the caller's own thread: its failed launch is its own error, and nothing sticks for the next call
mock.mockhip_fail_next_launches(1)
try:
    pbs[0].run()
    print("the synchronous search's failed launch went unnoticed")
    sys.exit(1)
except _lib.NrtGpuError:
    pass
pbs[0].run()
for g in leaves:
    g.release()
ctx.close()
print("done", flush=True)
