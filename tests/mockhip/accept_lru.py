"""Run by tests/test_planner_host.py in a subprocess with tests/mockhip preloaded: the per-leaf cache of combined doc sets
(liveDocs & FILTER masks & ~MUST_NOT masks; segment.cpp: accept_set_of_ids) holds 64 sets and evicts the least recently used one;
a set evicted while a search is in flight is retired and freed by the last search to leave the handle.  What is read is the leaves'
device bytes: 150 combinations one after another leave at most 64 sets resident; 50 more while a begun search is not yet waited
for leave 64 + 50 (the retired ones wait: at most 64 of them, then the cache refuses instead), and after the wait the 64 again."""
import itertools, os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nrtsearch_amd import api, synth
faulthandler.dump_traceback_later(60, exit=True)
ranks = [2, 30, 700]
corpus = synth.build_corpus(60_000, ranks, n_segments=2, delete_fraction=0.01)
ctx = api.GpuContext(0, max_batch=16)
leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
ids = list(range(11, 19))
for si, (seg, leaf) in enumerate(zip(corpus.segments, leaves)):
    for mid in ids:
        leaf.set_mask(mid, synth.random_mask(seg.max_doc, 0.5, 77 * mid + si))
base = [l.device_bytes for l in leaves]
set_bytes = [(s.max_doc + 63) // 64 * 8 for s in corpus.segments]
should = tuple(api.TermQuery(0, r) for r in ranks)
combos = ([((a,), (b,)) for a, b in itertools.permutations(ids, 2)] + [((a, b), (c,)) for a, b in itertools.combinations(ids, 2) for c in ids if c not in (a, b)])[:150]
mgr = api.TopScoreDocCollectorManager(20)

def q_of(f, mn):
    return api.BooleanQuery(should, 1, tuple(api.MaskFilter(i) for i in f), tuple(api.MaskFilter(i) for i in mn))

def resident_sets():
    return [(l.device_bytes - b) // sb for l, b, sb in zip(leaves, base, set_bytes)]

for f, mn in combos:
    sr.search(q_of(f, mn), mgr)
print("after 150 combinations:", resident_sets())
assert all(n <= 64 for n in resident_sets()), resident_sets()
pb = api.PreparedBatch(sr, [q_of(*combos[120])], [mgr])
keys, cnt, hits = np.zeros((1, 32), np.int64), np.zeros(1, np.int32), np.zeros(1, np.int64)
h = pb.begin_device(32, keys.ctypes.data, cnt.ctypes.data, hits.ctypes.data)     # in flight from here ...
for f, mn in combos[:50]:
    sr.search(q_of(f, mn), mgr)
mid = resident_sets()
print("50 more with a search in flight:", mid)
assert all(n == 64 + 50 for n in mid), mid                                      # ... so what was evicted waits
api.PreparedBatch.wait_device(h)
print("after the wait:", resident_sets())
assert all(n <= 64 for n in resident_sets()), resident_sets()
for g in leaves:
    g.release()
ctx.close()
print("done", flush=True)
