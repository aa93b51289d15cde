"""The HOST side of the library on a box without a GPU: segment upload bookkeeping, the planner and the result unpacking run
against tests/mockhip (a stand-in HIP runtime whose kernels do nothing -- test infrastructure, see its header), in a
subprocess with the stand-in preloaded.  What is checked is the PLAN (NRTGPU_PLAN_TRACE): results under the stand-in are empty.

The launch order of the MaxScore route's items (planner.cpp): longest-first by the postings of the query's
two heaviest clauses.  The order must not change WHAT is launched -- the same items, as a set, under either key -- and the
heaviest queries by that key must lead."""
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "nrtsearch_amd", "libnrtgpu.so")


@pytest.fixture(scope="module")
def mockhip(tmp_path_factory):
    if not (shutil.which("gcc") and os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h") and os.path.exists(LIB)):
        pytest.skip("gcc, the HIP headers or the built library are not here")
    out = str(tmp_path_factory.mktemp("mockhip") / "libmockhip.so")
    subprocess.run(["gcc", "-O1", "-w", "-fPIC", "-shared", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "mockhip", "mockhip.c"), "-o", out],
                   check=True)
    return out


DEV_LIB = os.path.join(ROOT, "nrtsearch_amd", "libnrtgpu_dev.so")   # the plan trace and the A/B keys are knobs of the development build


def plan_of(mockhip, **env):
    from nrtsearch_amd import build
    build.build_dev()
    e = dict(os.environ, LD_PRELOAD=mockhip, NRTGPU_PLAN_TRACE="1", NRTGPU_LIB_PATH=DEV_LIB, **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mockhip", "plan_batch.py")], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "done" in r.stdout, r.stderr[-2000:]
    plans = [(m.group(1), m.group(2), [int(x) for x in m.group(3).split()])
             for m in re.finditer(r"items as a set ([0-9a-f]+), in launch order ([0-9a-f]+); first items' queries:([ 0-9]*)", r.stderr)]
    counts = [int(m.group(1)) for m in re.finditer(r"(\d+) parts (\d+) items", r.stderr)]
    keys = [np.array([int(x) for x in line.split()[1:]]) for line in r.stdout.split("\n") if line.startswith("KEYS")]
    return plans, keys, counts


def test_host_runtime_plans_a_batch_without_a_gpu_and_the_launch_order_is_only_an_order(mockhip):
    new, keys, counts = plan_of(mockhip, NRTGPU_MS_LPT="1")    # A/B key: longest-first by the two heaviest clauses' postings
    old, _, _ = plan_of(mockhip)                               # the default (measured faster, round 4): by all postings
    assert len(new) == len(old) == 2 and len(keys) == 2
    for (set_new, order_new, first_new), (set_old, order_old, first_old), k in zip(new, old, keys):
        assert set_new == set_old, "the launch order changed WHAT is launched"
        assert order_new != order_old and first_new != first_old
        rank = (-k).argsort(kind="stable").argsort()           # 0 = the query with the most postings in its two heaviest clauses
        assert all(rank[q] < len(k) // 10 for q in first_new), f"leading items' queries {first_new} rank {[int(rank[q]) for q in first_new]} by the key"
        assert not all(rank[q] < len(k) // 10 for q in first_old)


def test_host_paths_of_the_gpu_suites_run_through_when_the_kernels_do_nothing(mockhip):
    """The BM25 suites of `-m gpu` against the stand-in: every search comes back empty, so their comparisons with the oracle fail
    -- what is asked here is only that the host side (uploads, seals, masks, slicing, every planner route and query shape,
    unpacking, the coalescer's threads) neither crashes nor hangs on a device that answers with zeros."""
    files = [os.path.join(ROOT, "tests", f) for f in ("test_parity_gpu.py", "test_maxscore_gpu.py", "test_filters_gpu.py", "test_packed_gpu.py")]
    e = dict(os.environ, LD_PRELOAD=mockhip, NRTGPU_TEST_NO_DUMPS="1")   # (no gpurun_out/parity_fail_*.json from failures that are the point)
    e.pop("NRTGPU_LIB_PATH", None)
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-q", "-p", "no:cacheprovider", "--tb=no"], env=e, capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    tail = r.stdout.strip().split("\n")[-1]
    assert r.returncode in (0, 1), f"pytest under the stand-in ended with {r.returncode}: {r.stdout[-1500:]} {r.stderr[-1500:]}"
    m = re.search(r"(\d+) failed", tail)
    assert m and int(m.group(1)) >= 30 and "error" not in tail, tail      # (they ran, and failed on their assertions, not on set-up)
    assert "Fatal Python error" not in r.stderr and "Segmentation" not in r.stderr


def test_a_begin_wait_pipeline_does_not_deadlock_against_a_writer(mockhip):
    """ADVICE round 3 (search.cpp: nrtgpu_pending keeps its segments' content locks from begin to wait): a thread that begins
    search i + 1 before anybody waits for search i must not be parked behind a writer (nrtgpu_segment_set_mask) that waits for
    search i.  The GPU test of the same name found the round's first fix (a bounded number of passes) wanting; this is its CPU
    twin, against the stand-in HIP runtime with searches that stay in flight for 0.5 ms / 3 ms each."""
    for us in ("500", "3000"):
        e = dict(os.environ, LD_PRELOAD=mockhip, MOCKHIP_SYNC_US=us)
        e.pop("NRTGPU_LIB_PATH", None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mockhip", "pipeline_vs_writer.py")], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "finished True" in r.stdout, (us, r.stdout[-500:], r.stderr[-1500:])
        assert "'b': 120, 'wait': 120" in r.stdout, r.stdout[-300:]


def test_the_planner_on_the_second_accumulator_shapes(mockhip):
    """What the planner decides for a DisjunctionMaxQuery with a tie breaker and for MUST next to SHOULD clauses (the MaxScore
    kernel's second accumulator: that route or the caller's path) and the C ABI's argument checks around them -- host logic,
    against the stand-in HIP runtime (tests/mockhip/plan_shapes.py)."""
    e = dict(os.environ, LD_PRELOAD=mockhip)
    e.pop("NRTGPU_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mockhip", "plan_shapes.py")], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "done" in r.stdout, r.stderr[-2000:]
    got = {}
    for line in r.stdout.split("\n"):
        if " " in line:
            name, rest = line.split(" ", 1)
            got[name] = rest
    ok = ("tie_small", "tie_complete_small", "tie_zero_nine_clauses", "must_should", "all_must", "must_term_nowhere", "raw_plain", "raw_all_must")
    assert all(got[n] == "ok" for n in ok), {n: got[n] for n in ok}
    for n in ("tie_complete_large", "tie_nine_clauses", "must_should_complete_large"):     # would need the exhaustive scan: one accumulator there
        assert got[n].startswith("unsupported:") and "MaxScore route only" in got[n], got[n]
    assert got["tie_out_of_range"].startswith("mirror:") and got["must_should_with_msm"].startswith("mirror:")
    for n, what in (("raw_occur_2", "occur must be 0"), ("raw_tie_without_dismax", "tie_breaker without disjunction_max"), ("raw_tie_negative", "[0, 1]"),
                    ("raw_must_in_dismax", "have no occur")):
        assert got[n].startswith("rc -1:") and what in got[n], got[n]      # NRTGPU_ERR_INVALID_ARG
    assert got["raw_must_with_msm"].startswith("rc -4:")                     # NRTGPU_ERR_UNSUPPORTED: Lucene's third sum structure


def test_coalesced_callers_in_a_closed_loop_form_cohorts(mockhip):
    """nrtgpu_search_bm25_coalesced against the stand-in HIP runtime (a stream synchronisation "takes" 300 us): a lone caller's
    queries run alone, one batch each -- it never lingers; C callers in a closed loop are served as cohorts (a few batches per
    round, not one per query), and the loop ends (no caller is left behind by the leader's leave-when-the-cohort-is-back rule,
    search.cpp)."""
    e = dict(os.environ, LD_PRELOAD=mockhip, MOCKHIP_SYNC_US="300")
    e.pop("NRTGPU_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mockhip", "coalesce_cohort.py")], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "done" in r.stdout, r.stderr[-2000:]
    rows = {int(m.group(1)): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"callers (\d+) calls (\d+) batches (\d+)", r.stdout)}
    assert rows[1][1] == rows[1][0], rows                    # alone: one batch per query
    # cohorts (how large depends on how the interpreter's threads happen to be scheduled on a busy box: on an idle one 7 and 16 of
    # a batch, under the rest of this suite 3: what is asked is company at all, and that the loop ends)
    assert rows[8][1] <= rows[8][0] * 6 // 10, rows
    assert rows[24][1] <= rows[24][0] * 6 // 10, rows


def test_a_launch_that_fails_behind_begin_is_reported_by_wait(mockhip):
    """nrtgpu_search_bm25_batch_device_begin hands the enqueue to the context's launcher thread: a launch that fails there is what
    nrtgpu_pending_wait returns (code and message), the workspace is free again, later searches run (tests/mockhip/launcher_error.py:
    the stand-in HIP runtime makes one launch fail)."""
    e = dict(os.environ, LD_PRELOAD=mockhip)
    e.pop("NRTGPU_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mockhip", "launcher_error.py")], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "done" in r.stdout and "wait reported: nrtgpu error" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])


def test_the_leaf_sets_verdict_on_speculation(mockhip):
    """The two-step verdict (search.cpp: note_speculation_of) without a GPU, driven through nrtgpu_note_shard_speculation: > 2 % of
    >= 2048 queries run again -- or, round 6, more than a quarter of >= 32 CALLS needing a second pass -- -> scattered window order
    and a fresh count; again -> off for that leaf set; nrtgpu_set_speculation starts over; small calls under the same failure
    rate keep their speculation (tests/mockhip/spec_verdict.py asserts every step)."""
    e = dict(os.environ, LD_PRELOAD=mockhip)
    e.pop("NRTGPU_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mockhip", "spec_verdict.py")], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "done" in r.stdout and "after set_speculation (2624, 20, False, False)" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])


def test_the_accept_set_cache_evicts_and_retires(mockhip):
    """The per-leaf cache of combined doc sets (segment.cpp: accept_set_of_ids) without a GPU: 150 combinations leave 64 sets
    resident; sets evicted while a begun search is in flight wait for it (64 + 50 resident) and are freed when it has been waited
    for (tests/mockhip/accept_lru.py reads the leaves' device bytes at every step)."""
    e = dict(os.environ, LD_PRELOAD=mockhip)
    e.pop("NRTGPU_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mockhip", "accept_lru.py")], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "done" in r.stdout and "after the wait: [64, 64]" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])


def test_two_submitting_threads_a_launcher_and_the_planners_helpers(mockhip):
    """The shape of bench.py's multi-GPU loop without a GPU (tests/mockhip/pipeline_stress.py): two threads begin shard searches
    over three result buffers -- 1024 queries over 3 leaves, so the planner's helper threads take part, and every enqueue goes
    through the context's launcher thread -- while the main thread waits in step order, merges and hands the buffers back; 400
    steps with searches that stay in flight for 100 us.  45 000 steps of it ran clean by hand in round 5."""
    e = dict(os.environ, LD_PRELOAD=mockhip, MOCKHIP_SYNC_US="100", STEPS="400", WATCHDOG="120")
    e.pop("NRTGPU_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mockhip", "pipeline_stress.py")], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "done" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
