"""Run by tests/test_planner_host.py in a subprocess with tests/mockhip preloaded: the leaf set's verdict on speculative thresholds
(search.cpp: note_speculation_of) driven through nrtgpu_note_shard_speculation -- more than 2 % of >= 2048 queries run again moves
the leaf set to the scattered window order with a fresh count, the same again switches speculation off for it;
nrtgpu_set_speculation starts the verdicts over; another leaf set has a verdict of its own."""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nrtsearch_amd import api, synth, workload
faulthandler.dump_traceback_later(60, exit=True)
w = workload.Workload("verdict test", 120_000, 3, 50, 16, 3)
qr = synth.make_queries(16, w.n_terms, w.max_rank)
corpus = workload.build_shard_corpus(w, qr)
ctx = api.GpuContext(0, max_batch=16)
leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
sr_all = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
sr_one = api.GpuIndexSearcher(ctx, leaves[:1], api.IndexStatistics.from_corpus(corpus))
queries = workload.boolean_queries(qr)
mgr = api.TopScoreDocCollectorManager(w.k)
pb_all = api.PreparedBatch(sr_all, queries, [mgr] * 16)
pb_one = api.PreparedBatch(sr_one, queries, [mgr] * 16)

def state():
    c = ctx.spec_counters()
    return c["queries"], c["reruns"], bool(c["scattered"]), bool(c["switched_off"])

assert state() == (0, 0, False, False), state()
pb_all.note_shard_speculation(2000, 100)           # 5 % failed, but fewer than 2048 queries seen: no verdict yet
assert state() == (2000, 100, False, False), state()
pb_all.note_shard_speculation(100, 0)              # 2100 seen, 100 failed = 4.8 %: the scattered window order, a fresh count
assert state()[2:] == (True, False), state()
pb_all.note_shard_speculation(3000, 30)            # 1 %: stays
assert state()[2:] == (True, False), state()
pb_all.note_shard_speculation(1000, 200)           # 4000 seen, 230 failed = 5.75 %: off for this leaf set
assert state()[2:] == (True, True), state()
# (the context's flags say "some leaf set"; the other leaf set's own verdict is untouched: it still speculates)
pb_one.note_shard_speculation(4096, 0)
ctx.set_speculation(5.0)                            # the verdicts start over
assert state() == (0, 0, False, False), state()
# the verdict by CALLS (round 6): a second pass costs per call.  1024-query calls of which every one re-runs 3 queries: 0.3 % of
# the queries -- the query rule never fires -- but every call pays a second pass: after 32 calls the scattered order, after 32
# more speculation is off.  (pb_one's leaf set: 200 calls of 8 queries with one re-run in ten of them keep their speculation.)
for i in range(31):
    pb_all.note_shard_speculation(1024, 3)
assert state()[2:] == (False, False), state()
pb_all.note_shard_speculation(1024, 3)
assert state()[2:] == (True, False), state()
for i in range(31):
    pb_all.note_shard_speculation(1024, 3)
assert state()[2:] == (True, False), state()
pb_all.note_shard_speculation(1024, 3)
assert state()[2:] == (True, True), state()
ctx.set_speculation(5.0)
for i in range(200):
    pb_one.note_shard_speculation(8, 1 if i % 10 == 0 else 0)
assert state()[2:] == (False, False), state()
pb_all.note_shard_speculation(1024, 0)
print("after set_speculation", state())
for g in leaves:
    g.release()
ctx.close()
print("done", flush=True)
