"""bench.py at N = 2 on the one GPU of the pool, through the path the driver's N > 1 runs take: one process per rank
(torch.distributed.run), each holding its docid shard, scan threads that begin shard searches under SHARD-LEVEL speculative
thresholds (nrtgpu_search_bm25_shard_device_begin), one thread that issues nrtgpu_dist_exchange_merge_checked in batch order
and has every rank run failed queries again -- the library's own collective, carried here by tests/mockrccl (the two ranks share
the GPU: --debug-same-gpu; the development library binds the stand-in by path, NRTGPU_RCCL_LIB, because the process holds
torch's RCCL too).  What is asked: rank 0 prints ONE well-formed line that says the library's collective and the shard-level
speculation were what ran, with every owned query's guess checked.  No rate is read from it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode", ["alltoall", "allgather"])
def test_bench_two_ranks_through_the_librarys_collective(tmp_path, dev_lib, mode):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is not here")
    mock = str(tmp_path / "librccl_mock.so")
    subprocess.run([HIPCC, "-O1", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "mockrccl", "mockrccl.cpp"), "-o", mock],
                   check=True)
    env = dict(os.environ, NRTGPU_LIB_PATH=os.path.join(ROOT, "nrtsearch_amd", "libnrtgpu_dev.so"), NRTGPU_RCCL_LIB=mock,
               NRTGPU_BENCH_DEBUG_LIB_COLLECTIVE="1", NRTGPU_BENCH_COLLECTIVE_TIMEOUT="60", MASTER_ADDR="127.0.0.1", NRTGPU_BENCH_WATCHDOG="120")
    # ONE attempt (round 6).  Round 5 tried twice after one unexplained failure in ~30 runs; reproduced in round 6 (3 of 32 runs of
    # scripts/gpu_two_rank_loop.sh, all-gather form only) and traced to the stand-in collective: tests/mockrccl copied the rank's own
    # block with a device-to-device hipMemcpy -- not host-synchronous, ordered against the NULL stream only -- so the merge (another
    # stream) could read the previous exchange's block, the ranks reached different verdicts on the shards' guesses and one entered
    # the re-run's collective alone.  The copy is stream-ordered now (mockrccl.cpp: ncclAllGather); the loop ran clean after it
    # (profiles/r06_two_rank_loop.log).
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-same-gpu", "--workload", "C2", "--steps", "12", "--warmup", "3",
           "--no-cpu-baseline", "--closed-loop", "", "--exhaustive-steps", "0", "--c4-steps", "0", "--exchange-mode", mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    err = "\n".join(l for l in r.stderr.split("\n") if not l.startswith(("RCCL", "HIP", "ROCm", "Host", "Libr")) and "amdgpu.ids" not in l)
    if r.returncode != 0:
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"bench_two_ranks_failure_{mode}.log"), "w") as f:
                f.write(err)
        except OSError:
            pass
    assert r.returncode == 0, err[:3000] + "\n[...]\n" + err[-3000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["value"] > 0
    c = d["config"]
    assert "collective inside the library" in c["sharding"] and "shard-level speculative thresholds" in c["sharding"], c["sharding"]
    sp = c["shard_speculation"]
    assert sp and sp["queries"] == (12 + 3) * c["batch_queries"], sp        # every step's batch went through the checked exchange
    assert sp["failed"] <= sp["queries"] // 50, sp                          # (i.i.d. shards: guesses stand)


@pytest.mark.parametrize("n_gpus,doc_shards", [(2, 1), (4, 2)])
def test_bench_topology_doc_shards_times_query_groups(tmp_path, dev_lib, n_gpus, doc_shards):
    """bench.py --gpus N --doc-shards D (round 6): N = D doc-shards x R query-groups on the one GPU of the pool.  (2, 1): two
    replicas, nothing exchanged, each runs ITS batches; (4, 2): two groups of two doc shards, ONE library communicator per group
    (carried by tests/mockrccl: two directories under /dev/shm), the groups take different batches.  The line counts every
    group's queries and says what ran."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc is not here")
    mock = str(tmp_path / "librccl_mock.so")
    subprocess.run([HIPCC, "-O1", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "mockrccl", "mockrccl.cpp"), "-o", mock],
                   check=True)
    env = dict(os.environ, NRTGPU_LIB_PATH=os.path.join(ROOT, "nrtsearch_amd", "libnrtgpu_dev.so"), NRTGPU_RCCL_LIB=mock,
               NRTGPU_BENCH_DEBUG_LIB_COLLECTIVE="1", NRTGPU_BENCH_COLLECTIVE_TIMEOUT="90", MASTER_ADDR="127.0.0.1", NRTGPU_BENCH_WATCHDOG="200")
    steps, warmup = 8, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1", "--master-port",
           str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n_gpus), "--doc-shards", str(doc_shards), "--debug-same-gpu", "--workload", "C2",
           "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--closed-loop", "", "--exhaustive-steps", "0", "--c4-steps", "0",
           "--exchange-mode", "allgather"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400, cwd=ROOT)
    err = "\n".join(l for l in r.stderr.split("\n") if not l.startswith(("RCCL", "HIP", "ROCm", "Host", "Libr")) and "amdgpu.ids" not in l)
    assert r.returncode == 0, err[:3000] + "\n[...]\n" + err[-3000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    R = n_gpus // doc_shards
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and c["topology"]["doc_shards"] == doc_shards and c["topology"]["query_groups"] == R
    # value counts every group's queries: steps x batch x R over the slowest rank's time
    assert abs(d["value"] - steps * c["batch_queries"] * R / (d["ms_per_step"] * steps * 1e-3)) / d["value"] < 0.01
    if doc_shards == 1:
        assert "replicas, no exchange" in c["sharding"] and c["shard_speculation"] is None
    else:
        assert "collective inside the library" in c["sharding"] and f"{R} query groups of {doc_shards} doc shards" in c["sharding"], c["sharding"]
        assert c["shard_speculation"]["queries"] == (steps + warmup) * c["batch_queries"]   # rank 0's group: every step's batch checked
