"""The HOST side of the library's own multi-GPU path at world = 2 on a box without a GPU: two processes, each a rank with its own
context and its docid shard, call nrtgpu_dist_search_bm25_batch_mode (both exchange forms, with and without shard-level
speculation) and the pipelined pair nrtgpu_search_bm25_shard_device_begin + nrtgpu_dist_exchange_merge_checked -- against
tests/mockhip (a HIP runtime whose kernels do nothing) with the collective carried by tests/mockrccl (messages are /dev/shm files; a
size the two ranks disagree about is an error).  Results under the stand-in are empty; what is asked is that dist.cpp's control
flow -- buffers, the grouped exchange with the guesses riding along, the verdicts' all-gather, who owns which answer -- runs to
the end on both ranks, without a protocol mismatch and without a hang; and, with guesses PLANTED in the ranks' buffers that no merged
list can reach, that every rank comes to know the same set of failed queries and runs them again together.  The answers themselves:
tests/test_dist_two_ranks_gpu.py."""
import os
import pickle
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "nrtsearch_amd", "libnrtgpu.so")


@pytest.fixture(scope="module")
def stand_ins(tmp_path_factory):
    if not (shutil.which("gcc") and shutil.which("g++") and os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h") and os.path.exists(LIB)):
        pytest.skip("gcc, the HIP headers or the built library are not here")
    d = tmp_path_factory.mktemp("standins")
    mockhip = str(d / "libmockhip.so")
    subprocess.run(["gcc", "-O1", "-w", "-fPIC", "-shared", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "mockhip", "mockhip.c"), "-o", mockhip], check=True)
    rccl_dir = d / "rccl"
    rccl_dir.mkdir()
    subprocess.run(["g++", "-O1", "-w", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-Wl,-soname,librccl.so.1",
                    os.path.join(ROOT, "tests", "mockrccl", "mockrccl.cpp"), "-o", str(rccl_dir / "librccl.so.1")], check=True)
    return mockhip, str(rccl_dir)


def test_two_ranks_run_the_librarys_exchange_to_the_end_without_a_gpu(stand_ins):
    mockhip, rccl_dir = stand_ins
    world, n_docs, n_q, k = 2, 120_000, 16, 50
    sync_dir = tempfile.mkdtemp(prefix="nrtgpu_dist2h_")
    outs = [os.path.join(sync_dir, f"rank{r}.pkl") for r in range(world)]
    env = dict(os.environ, LD_PRELOAD=mockhip, LD_LIBRARY_PATH=rccl_dir + ":" + os.environ.get("LD_LIBRARY_PATH", ""), NRTGPU_TEST_HOST_ONLY="1",
               NRTGPU_TEST_HIP_LIB=mockhip, NRTGPU_TEST_POKE_GUESSES="1")
    env.pop("NRTGPU_LIB_PATH", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(r), str(world), sync_dir, outs[r], str(n_docs), str(n_q),
                               str(k), "iid"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=300)
            logs.append(o)
        assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
        ranks = [pickle.load(open(o, "rb")) for o in outs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(sync_dir, ignore_errors=True)
    for r in range(world):
        for what in ("bm25", "bm25_nospec", "bm25_pipelined"):
            got = ranks[r][what + "_allgather"]
            assert len(got) == n_q and all(g is not None and len(g[0]) == 0 for g in got)           # every answer on every rank (empty: the kernels did nothing)
            owned = [g is not None for g in ranks[r][what + "_alltoall"]]
            assert owned == [qi * world // n_q == r for qi in range(n_q)], (what, r, owned)          # its slice of the batch, nothing else
        # the planted guesses (tests/_dist_worker.py: rank 0's for queries 1 and 5, rank 1's for 5 and 9; no merged list reaches them):
        # every rank is told all three, in both forms -- all-gather: every rank holds every guess; all-to-all: the owners' verdicts are
        # all-gathered -- and the re-run of those three (a collective call of its own) went through
        assert ranks[r]["pipelined_failed_allgather"] == [1, 5, 9] and ranks[r]["pipelined_failed_alltoall"] == [1, 5, 9], ranks[r]
        # rank 1's shard search timed out before it was planned: it entered the exchange all the same (empty lists + its status), so
        # nobody hung; rank 1 reports its own error, rank 0 one that names rank 1; the next call on the communicator goes through
        for name in ("allgather", "alltoall"):
            msg = ranks[r]["one_rank_fails_" + name]
            assert ("deadline" in msg) if r == 1 else ("rank 1 of 2 failed" in msg), (r, name, msg)
        assert len(ranks[r]["after_failure_allgather"]) == n_q and all(g is not None for g in ranks[r]["after_failure_allgather"])


@pytest.mark.parametrize("mode", ["alltoall", "allgather"])
def test_two_ranks_pipeline_with_planted_failures(stand_ins, mode):
    """bench.py's loop at N = 2 without a GPU (tests/mockhip/dist_pipeline_stress.py): per rank two submitting threads, the launcher,
    the planner's helpers, and the main thread running nrtgpu_dist_exchange_merge_checked per step; every fifth step guesses are
    planted that no merged list reaches -- both ranks must name the same failed queries (the worker asserts the exact set) and
    re-run them together.  200 steps here; 12 000 steps with 2 300 re-run calls ran clean by hand in round 5."""
    mockhip, rccl_dir = stand_ins
    sync_dir = tempfile.mkdtemp(prefix="nrtgpu_dist2s_")
    env = dict(os.environ, LD_PRELOAD=mockhip, LD_LIBRARY_PATH=rccl_dir + ":" + os.environ.get("LD_LIBRARY_PATH", ""), MOCKHIP_SYNC_US="50", PLANT="5",
               WATCHDOG="150")
    env.pop("NRTGPU_LIB_PATH", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mockhip", "dist_pipeline_stress.py"), str(r), "2", sync_dir, mode, "200"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=300)
            logs.append(o)
        assert all(p.returncode == 0 for p in procs) and all("done" in l for l in logs), "\n".join(l[-3000:] for l in logs)
        assert all("200 steps, 40 re-run calls" in l for l in logs), [l[-200:] for l in logs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(sync_dir, ignore_errors=True)
