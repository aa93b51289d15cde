#!/usr/bin/env python3
"""bench.py -- queries/sec + p50 latency of the BM25 top-k hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver launches one rank per GPU with torch.distributed.run.  A *step* is one pass of the hot path
over one batch of `--batch` synthetic queries (postings already resident in HBM): plan upload,
postings-scan kernel, top-k merge kernel, results back on the host (N = 1) or RCCL all-gather of
the per-GPU top-k + merge on every rank (N > 1; --all-to-all: each rank merges a slice of the batch).  Rank 0 prints ONE JSON line.

Workload at N = 1 = BASELINE.json config C3 (10M docs, 5-term BM25 disjunction, top-1000); for
N > 1 the same index is sharded by contiguous docid range over the ranks (strong scaling).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Algorithmic bytes per posting, SURVEY.md 8(d): 4 B docid + 4 B freq + 1 B norm gather = 9.  The HBM layout of
# this library folds the norm into the freq word at seal time, so the kernel physically streams 8 B per
# posting; the 8-byte figure is reported next to the contract's 9-byte one.
BYTES_PER_POSTING = 9
BYTES_PER_POSTING_FUSED = 8


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="queries per step")
    ap.add_argument("--host-threads", type=int, default=2,
                    help="N=1 only: host threads submitting steps (plan building of step i+1 overlaps the kernels of step i)")
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C4", "SMOKE"],
                    help="C3 (default, the headline), C2, or C4 = exact kNN over --docs (default 10M) x 768 fp32 rows, "
                         "cosine, top-100, --knn-queries per step (1 GPU)")
    ap.add_argument("--knn-queries", type=int, default=32, help="C4: queries per step (one panel = up to 32 queries)")
    ap.add_argument("--docs", type=int, default=0, help="override the number of docs (debug)")
    ap.add_argument("--corpus-variant", default="iid", choices=["iid", "clustered", "sorted"],
                    help="the synthetic corpus: iid = SURVEY 8d (every posting list an independent uniform draw: the headline); clustered = "
                         "terms in docid bursts; sorted = docs numbered by length (nrtsearch_amd/synth.py: corpus_variant_arrays)")
    ap.add_argument("--target-items", type=int, default=0)
    ap.add_argument("--no-prefetch", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=8192, help="queries timed on the CPU oracle (0 = skip): ~10 s of CPU work on 16 cores")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exchange", action="store_true",
                    help="N>1: do not share score bounds between the GPUs' shards (nrtgpu_exchange_open, on by default: a shard "
                         "stops collecting below the score that k docs of the OTHER shards already reach; results are identical)")
    ap.add_argument("--exchange", action="store_true", help="(accepted for older scripts: the exchange is the default at N>1)")
    ap.add_argument("--shard-bounds", default="speculation", choices=["speculation", "exchange", "local"],
                    help="N>1 (or --emulate-world): how a shard learns a threshold beyond its own k-th score.  speculation "
                         "(default): every shard guesses the WHOLE search's k-th score from its own docs "
                         "(nrtgpu_search_bm25_shard_device_begin), the guesses are checked against the merged lists and failed "
                         "queries run again on every rank; exchange: the score-bound table in shared memory "
                         "(nrtgpu_exchange_open; the default of rounds 3-4); local: neither (a shard's own speculation only)")
    ap.add_argument("--emulate-peers", default="none", choices=["none", "final"],
                    help="with --emulate-world: open the bound exchange for that job and play the other ranks' rows: none = they "
                         "never publish (the shard prunes on its own, the pessimistic end), final = each publishes from the start "
                         "what it would hold after its own scan (the optimistic end; a real job lies between the two)")
    ap.add_argument("--packed", action="store_true",
                    help="compressed postings (NRTGPU_FLAG_PACKED_POSTINGS): one 32-bit word per posting in HBM.  A separately "
                         "reported configuration: the roofline's algorithmic bytes are then 4 per posting, not 9")
    ap.add_argument("--blocking-wait", action="store_true", help="NRTGPU_FLAG_BLOCKING_WAIT also at one rank (default: only for N > 1)")
    ap.add_argument("--no-prune", action="store_true",
                    help="A/B: exhaustive scan only (NRTGPU_FLAG_NO_PRUNE): every posting of every query term is streamed")
    ap.add_argument("--debug-same-gpu", action="store_true",
                    help="debug: run an N-rank job with every rank on GPU 0 (gloo, collectives staged through the host)")
    ap.add_argument("--all-to-all", action="store_true",
                    help="with --torch-collective: torch.distributed's all_to_all_single instead of its all-gather")
    ap.add_argument("--exchange-mode", default="alltoall", choices=["alltoall", "allgather"],
                    help="N>1, the library's exchange stage (nrtgpu_dist_exchange_merge): alltoall (default) = every rank receives the "
                         "other ranks' lists for ITS slice of the batch, merges and delivers only those (1/N of the bytes per xGMI link, "
                         "of the merge and of the host-side unpacking); allgather = BASELINE.json's north-star form: every rank merges "
                         "every query and holds every answer")
    ap.add_argument("--sync-submit", action="store_true",
                    help="N>1: --host-threads threads each block in nrtgpu_search_bm25_batch_device_epoch (rounds 1-2) instead of ONE thread "
                         "submitting with nrtgpu_search_bm25_batch_device_begin and the exchange thread waiting (nrtgpu_pending_wait)")
    ap.add_argument("--submitters", type=int, default=0,
                    help="N>1 (begin / wait submission): threads that plan and enqueue steps side by side (0 = 2 when the host has >= 6 CPUs "
                         "per rank, else 1): a rank whose planning takes longer than its kernel is bound by ONE submitting thread")
    ap.add_argument("--planner-threads", type=int, default=0, help="planner threads per in-flight call (0 = what the box's CPUs allow)")
    ap.add_argument("--speculation-margin", type=float, default=-1.0,
                    help="the speculative thresholds' safety margin in standard deviations (nrtgpu_set_speculation; < 0: the library's default, 5; 0: off)")
    ap.add_argument("--closed-loop", default="1,8,64,512",
                    help="N=1, C3: after the batch line, closed loop with that many concurrent callers, one query per call through "
                         "nrtgpu_search_bm25_coalesced (comma list; empty = skip): qps / p50 / p99 per caller count")
    ap.add_argument("--closed-loop-ms", type=int, default=1500)
    ap.add_argument("--exhaustive-steps", type=int, default=8,
                    help="C3 / C2, one GPU: steps of the EXHAUSTIVE route (every posting streamed) timed after the main run for "
                         "roofline.exhaustive (0 = skip)")
    ap.add_argument("--c2-steps", type=int, default=200,
                    help="C3 line, one GPU: also time this many steps of BASELINE config 2 (1 M docs, 2-term, top-100) in the same run "
                         "(roofline.c2; 0 = skip)")
    ap.add_argument("--c5-steps", type=int, default=16,
                    help="C3 line, one GPU: also time this many 256-query batches of BASELINE config 5's shape at 5 M docs -- BM25 "
                         "recall-1000 + exact cosine rescore over 768-d vectors, top-100, fused on the device (roofline.c5; 0 = skip; "
                         "the 50 M-doc run: scripts/gpu_c5_hybrid.py)")
    ap.add_argument("--c4-steps", type=int, default=8,
                    help="C3 line: also run BASELINE config 4 (10 M x 768 exact kNN, 64 queries per pass) for that many passes -> roofline.c4 (0 = skip)")
    ap.add_argument("--no-sketch", action="store_true",
                    help="C4 A/B: NRTGPU_FLAG_NO_VECTOR_SKETCH -- no fp16 copy of the rows, the exact search nominates from the fp32 rows "
                         "(twice the bytes per pass); the answers are the same bits")
    ap.add_argument("--c4-seg-rows", type=int, default=2_500_000, help="C4: rows per segment (default 2.5 M: 4 segments of the 10 M)")
    ap.add_argument("--c4-callers", action="store_true", help="C4, one GPU: --host-threads callers take the steps in turn (A/B)")
    ap.add_argument("--no-verify", action="store_true", help="C4: skip the fp64 check of the device's answer over all rows")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: take the multi-GPU path (device-resident top-k -> exchange -> merge) even at world size 1")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="debug, 1 GPU: index only rank 0's docid range of an N-GPU job (per-rank step time at --gpus N)")
    ap.add_argument("--emulate-rank", type=int, default=0, help="with --emulate-world: which rank's shard (the last rank holds the small segments)")
    ap.add_argument("--doc-shards", type=int, default=0,
                    help="N > 1: the topology N = D doc-shards x R query-groups.  D ranks share the index by docid range and exchange their "
                         "top-k (one communicator per group of D consecutive ranks); the R = N / D groups hold the same index and take "
                         "DIFFERENT batches of the query set.  0 = N (every GPU a shard of ONE search: BASELINE.json's north-star form); "
                         "1 = N replicas, no exchange.  Sharding buys latency, replication throughput (DESIGN 7: every BASELINE config "
                         "fits one MI355X); value counts the queries of all groups")
    ap.add_argument("--shard-layout", default="index", choices=["index", "per_shard", "balanced"],
                    help="N>1: a rank owns the pieces of the index's segments inside its docid range (default); per_shard (rounds 1-2): "
                         "its range cut into a full set of tiered segments of its own; balanced: the index's small segments are dealt "
                         "out to the ranks and each rank is filled up from the big segments' docid space -- every rank holds 2-3 leaves "
                         "instead of the last rank holding all the small ones")
    ap.add_argument("--debug-k", type=int, default=0, help="debug: numHits override (what a shard costs at a smaller k)")
    ap.add_argument("--torch-collective", action="store_true",
                    help="N>1: exchange with torch.distributed's all-gather instead of the library's own RCCL stage "
                         "(nrtgpu_dist_allgather_merge, the default; falls back to torch by itself if RCCL cannot be bound)")
    return ap.parse_args()


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes
    expose 256 hardware threads but grant 16 CPUs; more threads than that only oversubscribe)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def thread_cpu_seconds():
    """utime + stime of every thread of this process, by tid (Linux): {tid: (comm, seconds)}."""
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                with open(f"/proc/self/task/{tid}/stat") as f:
                    st = f.read()
                comm = st[st.index("(") + 1: st.rindex(")")]
                fields = st[st.rindex(")") + 2:].split()
                out[int(tid)] = (comm, (int(fields[11]) + int(fields[12])) / tck)
            except (OSError, ValueError):
                continue
    except OSError:
        pass
    return out


def stdout_to_stderr(fn):
    """Runs fn() with file descriptor 1 pointing at stderr: RCCL prints a version banner on stdout when a communicator is
    created (through C stdio, so it would land AFTER the JSON line at exit); the bench's stdout carries the JSON line only."""
    import ctypes

    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        return fn()
    finally:
        sys.stdout.flush()
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def cpu_baseline(corpus, query_ranks, k, n_queries):
    """The CPU oracle (oracle/nrt_oracle.c) on the host cores: a reported baseline next to the GPU
    number, not a target.  Bounded sample of the same queries; the timed region is one C call
    (OpenMP over queries, one collector = one Lucene slice per query).  `value` is the dynamically
    pruned scorer (MaxScore family -- what Lucene runs for this query shape under TOP_SCORES,
    SURVEY 8d); the exhaustive scorer is timed beside it on a quarter of the sample."""
    from oracle import oracle

    oracle.build()
    cores = usable_cpus()
    n_queries = min(max(n_queries, 64 * cores), len(query_ranks))   # every thread busy for tens of queries (~15 s of CPU work)
    sample = [query_ranks[i].tolist() for i in range(n_queries)]
    pb = oracle.PreparedBatch(corpus, sample, k)              # weights + impacts: index / Weight time, untimed
    n_ex = max(1, n_queries // 4)
    pb_ex = oracle.PreparedBatch(corpus, sample[:n_ex], k)
    pb_ex.run(True, cores)                                     # warm: page in the postings, spin up the pool
    t0 = time.perf_counter()
    pruned = pb.run(True, cores)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    exact = pb_ex.run(False, cores)
    dt_ex = time.perf_counter() - t0
    if not ((pruned[0][:n_ex] == exact[0]).all() and (pruned[1][:n_ex] == exact[1]).all()):
        raise RuntimeError("oracle: pruned and exhaustive top-k differ")
    total_p = sum(int(corpus.doc_freq.get(int(t), 0)) for q in sample for t in q)
    return {"value": round(n_queries / dt, 2), "unit": "queries/s", "cores": cores, "kind": "port",
            "exhaustive_value": round(n_ex / dt_ex, 2),
            "postings_scored_frac": round(pruned[5] / max(1, total_p), 4),
            "sample": f"first {n_queries} queries of the same query set, MaxScore-pruned windowed scorer + heap "
                      f"(oracle/nrt_oracle.c, C + OpenMP; NOT JVM Lucene), {cores} threads, {dt:.2f}s; "
                      f"exhaustive scorer on the first {n_ex}: {dt_ex:.2f}s"}


def lucene_baseline(w, searcher, queries, mgr, n_queries):
    """The reference's own CPU path, when this box can run it: probe `java` / `javac` and lucene-core
    (LUCENE_JARS=<classpath>, or jars under /opt/lucene); if present dump the same corpus + queries
    (scripts/dump_corpus.py), compile and run bench/lucene/LuceneBaseline.java at 1 and all host threads, and diff its
    docids / score bits against the device's answers.  Returns a dict for cpu_baseline["lucene"]."""
    import glob
    import shutil
    import subprocess
    import tempfile

    java, javac = shutil.which("java"), shutil.which("javac")
    jars = os.environ.get("LUCENE_JARS") or ":".join(sorted(glob.glob("/opt/lucene/*.jar")))
    if not java or not javac or not jars:
        return {"available": False, "reason": "no JDK on this box" if not (java and javac) else "no lucene-core jar (LUCENE_JARS)"}
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_corpus

    import numpy as np

    tmp = tempfile.mkdtemp(prefix="lucene_baseline_")
    try:
        dump_corpus.dump(w, os.path.join(tmp, "dump"), n_queries)
        subprocess.check_call([javac, "-cp", jars, os.path.join(ROOT, "bench", "lucene", "LuceneBaseline.java"), "-d", os.path.join(tmp, "cls")])
        out = {}
        for threads in (usable_cpus(), 1):
            res = os.path.join(tmp, f"out_{threads}.json")
            subprocess.check_call([java, "-Xmx32g", "-cp", os.path.join(tmp, "cls") + ":" + jars, "LuceneBaseline", os.path.join(tmp, "dump"),
                                   os.path.join(tmp, "index"), str(threads), res, str(n_queries)])
            r = json.load(open(res))
            out[f"threads_{threads}"] = {k_: r[k_] for k_ in ("queries_per_s", "p50_ms", "p99_ms", "threads")}
            out.update(lucene=r["lucene"], java=r["java"], index_build_s=r["index_build_s"], segments=r["segments"])
        got = searcher.search_batch(queries[:n_queries], [mgr] * n_queries)
        same_docs = same_bits = same_rel = 0
        for g, e in zip(got, r["results"]):
            same_docs += g.docs.tolist() == e["docs"]
            same_bits += g.scores.view(np.uint32).tolist() == [b & 0xFFFFFFFF for b in e["score_bits"]]
            same_rel += bool(g.relation_gte) == bool(e["gte"])
        out.update(available=True, queries=n_queries, docids_identical=same_docs, score_bits_identical=same_bits, relation_identical=same_rel)
        return out
    except Exception as e:   # noqa: BLE001 -- a reported baseline must never take the bench line down
        return {"available": False, "reason": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_record(kernel, pruned, avg_launch_ms, algorithmic_bytes_per_launch, bytes_per_posting, bytes_per_posting_streamed, traffic,
                    traffic_note=None):
    """The `roofline` object of the BM25 bench line from measured numbers (pure: tests/test_bench_contract.py checks its arithmetic).
    achieved / frac: for the exhaustive scan the ALGORITHMIC rate (9 B per posting of the query's terms / launch time).
    The pruned kernel skips most of those bytes by design, so its algorithmic rate is an EFFECTIVE figure (may exceed the peak):
    it moves to effective_*, and achieved / frac are the PHYSICAL rate -- HBM bytes of the PMC profile / launch time."""
    t = avg_launch_ms * 1e-3
    algo = algorithmic_bytes_per_launch / t / 1e9 if t > 0 else 0.0
    phys = traffic / t / 1e9 if (traffic and t > 0) else None
    main = phys if (pruned and phys is not None) else algo
    return {
        "bound": "hbm", "kernel": kernel, "effective": bool(pruned),
        "achieved": round(main, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(main / HBM_PEAK_GBS, 4),
        "achieved_is": "physical (PMC traffic / launch time)" if (pruned and phys is not None) else "algorithmic bytes / launch time",
        "effective_achieved": round(algo, 1) if pruned else None,
        "effective_frac": round(algo / HBM_PEAK_GBS, 4) if pruned else None,
        "bytes_per_posting": bytes_per_posting,
        "algorithmic_bytes_per_launch": int(algorithmic_bytes_per_launch),
        "achieved_at_8B_per_posting": round(algo * bytes_per_posting_streamed / bytes_per_posting, 1),
        "frac_at_8B_per_posting": round(algo * bytes_per_posting_streamed / bytes_per_posting / HBM_PEAK_GBS, 4),
        "physical_achieved": round(phys, 1) if phys is not None else None,
        "physical_frac": round(phys / HBM_PEAK_GBS, 4) if phys is not None else None,
        "avg_launch_ms": round(avg_launch_ms, 4),
        "traffic": traffic,
        "traffic_source": (("static profile: profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE of this command, collected in a run of "
                            "its own; " + str(traffic_note) + ")") if traffic else None),
    }


def closed_loop(ctx, searcher, queries, mgr, callers, duration_ms):
    """SURVEY 8d's metric shape: C concurrent callers, each with ONE query in flight through nrtgpu_search_bm25_coalesced (the library
    merges them into device batches).  The callers are native threads of bench/loadgen (host-only tooling that knows nothing but
    include/nrtgpu.h): Python threads would measure the GIL.  -> {C: {qps, p50_ms, p99_ms, mean_batch}}"""
    import ctypes as C

    import numpy as np

    from nrtsearch_amd import _lib, build

    lg = C.CDLL(build.build_loadgen())
    lg.loadgen_closed_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p]
    L = _lib.load()
    fn = C.cast(L.nrtgpu_search_bm25_coalesced, C.c_void_p)
    m = searcher._marshal(queries, [mgr] * len(queries))
    out = {}
    for c in callers:
        res = np.zeros(4, dtype=np.float64)
        ctx.reset_stats()
        rc = lg.loadgen_closed_loop(fn, ctx._h, searcher._segs, searcher._bases, len(searcher.leaves), m.queries, len(queries), int(c),
                                    int(duration_ms), res.ctypes.data)
        if rc != 0:
            out[str(c)] = {"error": int(rc)}
            continue
        st = ctx.stats()
        out[str(c)] = {"qps": round(res[0] / res[1], 1), "p50_ms": round(res[2], 3), "p99_ms": round(res[3], 3),
                       "mean_batch": round(st["queries"] / max(1, st["batches"]), 1)}
    return out


FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md); v_mfma_f32_16x16x4_f32 is exact fp32
FP16_MFMA_PEAK_TFLOPS = 2500.0   # dense fp16 matrix peak of an MI355X (MI355X_MICROARCH.md; the 2:1 sparsity figure is never priced against)


def knn_roofline_record(rows_per_pass, dim, queries_per_pass, score_ms, sketched, score_launches_per_pass, second_passes, traffic=None,
                        traffic_source=None):
    """The `roofline` object of the exact-kNN line from measured numbers (pure: tests/test_bench_contract.py checks its arithmetic).
    A pass over the rows that nominates from the fp16 sketch STREAMS 2 bytes per element (steps of 32 dimensions padded to whole
    groups of four), half of SURVEY 8d's algorithmic fp32 bytes: `frac` is then the PHYSICAL fraction and the algorithmic rate
    stands beside it as effective_* (it may exceed the peak); from the fp32 rows the two coincide and the fp32 matrix-pipe
    fraction is reported as well."""
    t = score_ms * 1e-3
    algorithmic = rows_per_pass * dim * 4
    steps16 = ((dim + 31) // 32 + 3) // 4 * 4
    streamed = rows_per_pass * (steps16 * 64 if sketched else dim * 4)
    effective = algorithmic / t / 1e9 if t > 0 else 0.0
    achieved = streamed / t / 1e9 if t > 0 else 0.0
    tflops = 2.0 * rows_per_pass * dim * queries_per_pass / t / 1e12 if t > 0 else 0.0
    return {"bound": "hbm", "kernel": "knn_sketch_kernel" if sketched else "knn_score_kernel", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "achieved_is": ("physical: the fp16 sketch's bytes (2 per element) / launch time" if sketched
                            else "algorithmic bytes (fp32 rows) / launch time"),
            "effective": bool(sketched),
            "effective_achieved": round(effective, 1) if sketched else None,
            "effective_frac": round(effective / HBM_PEAK_GBS, 4) if sketched else None,
            "algorithmic_bytes_per_launch": int(algorithmic), "streamed_bytes_per_launch": int(streamed),
            "launch": "the nomination kernel's launches of one pass over the rows (<= 64 queries; a few rounds, theta tightens in between)",
            "score_launches_per_panel": round(score_launches_per_pass, 2),
            "second_passes": int(second_passes),
            "avg_launch_ms": round(score_ms, 4),
            # the matrix cores: useful multiply-adds (rows x dim x queries of the pass, x 2) / launch time against the dense peak of the
            # operand type -- fp16 for the sketch (v_mfma_f32_16x16x32_f16), fp32 for the fp32 rows (v_mfma_f32_16x16x4_f32).  The pass
            # is bound by the bytes it streams: at 64 queries per pass the matrix rate is what the HBM rate allows.
            "mfma_tflops": round(tflops, 2), "mfma_peak_tflops": FP16_MFMA_PEAK_TFLOPS if sketched else FP32_MFMA_PEAK_TFLOPS,
            "mfma_dtype": "f16" if sketched else "f32",
            "mfma_frac": round(tflops / (FP16_MFMA_PEAK_TFLOPS if sketched else FP32_MFMA_PEAK_TFLOPS), 4),
            "traffic": traffic, "traffic_source": traffic_source}


def knn_traffic_record(lib_id, q_per_pass):
    """HBM bytes per pass of the sketch kernel from profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE, a run of its own), only
    when the record was taken from THIS build's kernels."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        for rec in json.load(open(pmc)):
            if rec.get("workload") == "C4" and rec.get("kernel") == "knn_sketch_kernel" and rec.get("queries_per_pass") == int(q_per_pass):
                if rec.get("build_id") == lib_id:
                    return rec.get("hbm_bytes_per_launch"), rec.get("note")
                return None, f"profiles/pmc_traffic.json holds a C4 record of build {rec.get('build_id')}; the loaded library is {lib_id}: not reused"
    except Exception:
        pass
    return None, None


def run_c4(args, emit=True):
    """BASELINE.json config 4 at one GPU: N x 768 fp32 rows resident in HBM, exact (brute-force) cosine kNN top-100 --
    what KnnFloatVectorQuery / ExactVectorQuery compute, answered by nrtgpu_knn_exact.  A step = one call with
    --knn-queries queries (every <= 64 of them stream the rows once).  Roofline: HBM (N * dim * 4 bytes per pass), with the
    fp32 MFMA fraction beside it (at 64 queries per pass the matrix rate needed is as large as the HBM rate allows)."""
    import numpy as np
    import torch

    from nrtsearch_amd import api, build

    build.build()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    n_all, dim, k, Q = (args.docs or 10_000_000), 768, 100, max(1, args.knn_queries)
    # N > 1 (BASELINE config 4, 1 -> 8 GPUs): the rows are partitioned like the docid shards -- rank r holds rows
    # [r * n / N, (r + 1) * n / N) -- every rank scores its rows for every query, the per-rank top-k lists are exchanged and merged
    # inside the library (nrtgpu_dist_knn_exact).  --emulate-world N on one GPU: rank --emulate-rank's share, no exchange.
    shard_world, shard_rank = (args.emulate_world, args.emulate_rank) if (args.emulate_world > 1 and world == 1) else (world, rank)
    row_lo = n_all * shard_rank // shard_world
    n = n_all * (shard_rank + 1) // shard_world - row_lo
    seg_rows = max(16, args.c4_seg_rows)
    from nrtsearch_amd import _lib
    ctx = api.GpuContext(device_id=local_rank, max_batch=64, collect_timing=True,
                         flags=_lib.NRTGPU_FLAG_NO_VECTOR_SKETCH if args.no_sketch else 0)
    mode = api.EXCHANGE_ALLTOALL if args.exchange_mode == "alltoall" else api.EXCHANGE_ALLGATHER
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

        def _init():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            box = [api.GpuContext.dist_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctx.dist_init(world, rank, box[0])

        stdout_to_stderr(_init)
    t_build = time.perf_counter()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(777 + shard_rank)
    leaves, base, first_seg = [], 0, None
    # Parity at the BASELINE size, in this run: for three queries of the last timed panel the cosine score map of EVERY row is taken
    # in fp64 (torch on the device, chunk by chunk while the rows are generated) and the 100 best kept -- what the device's answer
    # is compared with below ("verify").  One GPU only (a rank of a job sees its rows only).
    qrng = np.random.Generator(np.random.PCG64(778))
    panels = [qrng.standard_normal((Q, dim), dtype=np.float32) for _ in range(4)]
    verify_q = sorted(set([0, Q // 2, Q - 1])) if (world == 1 and shard_world == 1 and not args.no_verify) else []
    last_panel = panels[(args.warmup + args.steps - 1) % len(panels)]
    vq64 = torch.from_numpy(last_panel[verify_q].astype(np.float64)).cuda() if verify_q else None
    v_best = None    # (scores fp64 [nq, <=2k], docs int64)
    while base < n:
        rows = min(seg_rows, n - base)
        dev_rows = torch.randn((rows, dim), generator=gen, device="cuda", dtype=torch.float32)
        if verify_q:
            cand_s, cand_d = [], []
            for a0 in range(0, rows, 500_000):
                blk = dev_rows[a0: a0 + 500_000].to(torch.float64)
                cos = (blk @ vq64.T) / (blk.norm(dim=1, keepdim=True) * vq64.norm(dim=1).unsqueeze(0))
                sc64 = torch.clamp((1.0 + cos) / 2.0, min=0.0).T            # VectorSimilarityFunction.COSINE's score map
                top = torch.topk(sc64, k=min(2 * k, sc64.shape[1]), dim=1)
                cand_s.append(top.values)
                cand_d.append(top.indices + (row_lo + base + a0))
            cs, cd = torch.cat(cand_s + ([v_best[0]] if v_best else []), dim=1), torch.cat(cand_d + ([v_best[1]] if v_best else []), dim=1)
            keep = torch.topk(cs, k=min(2 * k, cs.shape[1]), dim=1)
            v_best = (keep.values, torch.gather(cd, 1, keep.indices))
        host = dev_rows.cpu().numpy()
        del dev_rows
        g = api.GpuSegment(ctx, rows, row_lo + base)
        g.add_vectors(0, host)
        g.seal()
        leaves.append(g)
        if first_seg is None:
            first_seg = host[: min(rows, 1_000_000)].copy()
        del host
        base += rows
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics())
    t_build = time.perf_counter() - t_build
    lat = []

    def one(panel):
        return sr.dist_knn_exact(0, "cosine", panel, k, mode=mode) if world > 1 else sr.knn_exact(0, "cosine", panel, k)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one(panels[i % len(panels)])
    ctx.reset_stats()
    fence()
    # One caller by default: a step is one call, its latency the call's.  --c4-callers: --host-threads callers take the steps in
    # turn (the library runs one call's kernels at a time; a caller's staging and unpacking overlap the other's kernels) --
    # measured: no gain in throughput (the gap between two calls is ~0.1 ms of 3.6), twice the latency.
    n_thr = max(1, args.host_threads) if (world == 1 and args.c4_callers) else 1
    import gc
    gc.collect()
    gc.disable()   # the interpreter's cycle collector is not what is measured: one full collection (40-55 ms with torch loaded) otherwise lands in some step
    t0 = time.perf_counter()
    if n_thr == 1:
        for i in range(args.steps):
            ts = time.perf_counter()
            last = one(panels[(args.warmup + i) % len(panels)])
            lat.append(time.perf_counter() - ts)
    else:
        import threading
        results = {}

        def caller(tix):
            for i in range(tix, args.steps, n_thr):
                ts = time.perf_counter()
                r = one(panels[(args.warmup + i) % len(panels)])
                lat.append(time.perf_counter() - ts)
                if i == args.steps - 1:
                    results["last"] = r

        threads = [threading.Thread(target=caller, args=(t,)) for t in range(n_thr)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        last = results["last"]
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ctx.stats()
    # Two callers taking the steps in turn ("staging overlapped with the previous panel"): a caller's
    # staging, small selection / rescoring kernels, result copy and unpacking can overlap the other caller's stream of the rows.
    # Measured (round 6, 10 M x 768, 64 queries per pass; profiles/r06_c2_threads_c4_two_callers.log): 3.26 - 3.31 ms per pass call
    # against 3.41 - 3.48 with one caller (-5 %; one call saw none, the default line's short leg 2.99 - 3.24): the pass is
    # bandwidth-bound and two panels' streams of the rows share nothing, so what overlaps is only what lies between a call's device
    # time (2.98 ms: three nomination launches 2.77, three selections 0.12, rescoring 0.09) and the call -- launch gaps and the
    # host's staging.  Fewer launches per panel would take that out for ONE caller too: not built.
    # (Later in round 6 most of that "host's staging" turned out to be the Python mirror's per-call pointer set-up -- now kept per
    #  thread, api._topdocs_outputs: one caller 3.16 - 3.22 ms per pass call at 64 queries, profiles/r06_c4_wrapper.log.)
    # (Kernel times come from the one-caller loop above: with overlapping launches a launch's HIP-event time is no longer its own.)
    two = None
    if world == 1 and n_thr == 1 and args.steps >= 4 and not args.c4_callers:
        import threading
        gc.disable()
        t2 = time.perf_counter()

        def caller2(tix):
            for i in range(tix, args.steps, 2):
                one(panels[(args.warmup + i) % len(panels)])

        ths = [threading.Thread(target=caller2, args=(t,)) for t in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        fence()
        e2 = time.perf_counter() - t2
        gc.enable()
        two = {"callers": 2, "steps": args.steps, "ms_per_pass_call": round(e2 / args.steps * 1e3, 4), "queries_per_s": round(args.steps * Q / e2, 2)}
    n_panels = max(1, st["knn_panels"])
    score_ms = st["knn_score_ms"] / n_panels                    # knn_score_kernel launches of one panel (HIP events, its stream)
    q_per_panel = Q / max(1, (Q + 63) // 64)   # queries per pass over the rows (two 32-query panels on paired workgroups)
    # The pass over the rows nominates from the fp16 sketch when the segment keeps one (2 bytes per element, matrix-core operand
    # order): the kernel then READS half of SURVEY 8d's algorithmic bytes.  As for the pruned BM25 kernel the roofline fraction is the
    # PHYSICAL one (bytes the kernel streams / launch time) and the algorithmic figure stands beside it as effective_*.
    sketched = st["knn_sketch_launches"] > 0 and st["knn_sketch_launches"] == st["knn_score_launches"]
    out = {
        "metric": "queries/sec, exact kNN 10M x 768 fp32 cosine top-100" if n_all == 10_000_000 else f"queries/sec, exact kNN {n_all} x 768 fp32 cosine top-100",
        "value": round(args.steps * Q / elapsed, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "p50_latency_ms": round(statistics.median(lat) * 1e3, 4),
        "max_latency_ms": round(max(lat) * 1e3, 4), "slowest_step": int(np.argmax(lat)),
        # steps that took more than twice the median (step index, ms): a stall on the host or the device shows up here, not in p50
        "latency_outliers": [(int(i_), round(l_ * 1e3, 3)) for i_, l_ in enumerate(lat) if l_ > 2.0 * statistics.median(lat)][:16],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C4: {n_all} x {dim} fp32 rows, brute-force cosine top-{k}", "n_docs": n_all, "rows_per_gpu": n, "dim": dim, "k": k,
                   "queries_per_step": Q, "host_threads": n_thr, "segments_per_gpu": len(leaves), "corpus_build_s": round(t_build, 1),
                   "sharding": ("rows partitioned by docid range, 1 process per GPU, per-rank top-k exchanged inside the library "
                                f"(nrtgpu_dist_knn_exact, {'all-to-all' if mode == api.EXCHANGE_ALLTOALL else 'all-gather'})" if world > 1 else
                                (f"[emulating rank {shard_rank} of {shard_world}: its rows, no exchange]" if shard_world > 1 else "one GPU"))},
        "roofline": knn_roofline_record(st["knn_rows"] / n_panels, dim, q_per_panel, score_ms, sketched, st["knn_score_launches"] / n_panels,
                                        st["knn_second_passes"],
                                        *(knn_traffic_record(build.build_id(_lib.LIB_PATH), q_per_panel) if (sketched and world == 1 and shard_world == 1) else (None, None))),
    }
    out["roofline"]["build_id"] = build.build_id(_lib.LIB_PATH)
    if rank == 0 and world == 1 and shard_world == 1 and args.closed_loop and emit:
        # queries/s AND latency under concurrent clients: C native caller threads, ONE query per call through
        # nrtgpu_knn_exact_coalesced -- the library merges them into panels of up to 64 that share a pass over the rows
        import ctypes as C

        lg = C.CDLL(build.build_loadgen())
        lg.loadgen_closed_loop_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                               C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        fn = C.cast(_lib.load().nrtgpu_knn_exact_coalesced, C.c_void_p)
        qset = np.ascontiguousarray(np.concatenate(panels, axis=0)[:256], dtype=np.float32)
        out["closed_loop"] = {"entry": "nrtgpu_knn_exact_coalesced, one query per call, native caller threads (bench/loadgen)"}
        for c in [int(x) for x in args.closed_loop.split(",") if x.strip()]:
            res = np.zeros(4, dtype=np.float64)
            ctx.reset_stats()
            rc = lg.loadgen_closed_loop_knn(fn, ctx._h, sr._segs, sr._bases, len(sr.leaves), 0, 0, qset.ctypes.data, len(qset), dim, k, c,
                                            int(args.closed_loop_ms), res.ctypes.data)
            st_c = ctx.stats()
            out["closed_loop"][str(c)] = ({"error": int(rc)} if rc != 0 else
                                          {"qps": round(res[0] / res[1], 1), "p50_ms": round(res[2], 3), "p99_ms": round(res[3], 3),
                                           "mean_panel": round(res[0] / max(1, st_c["knn_panels"]), 1)})
    if verify_q and v_best is not None:
        # the device's top-k of the last timed panel against the fp64 ranking over all rows: scores within 2e-5 relative, a docid
        # that differs at its rank must be a near-tie of the fp64 doc there (the fp32 sums differ in their last bits)
        vs, vd = v_best[0].cpu().numpy(), v_best[1].cpu().numpy()
        ok, off_rank = True, 0
        for j, qi in enumerate(verify_q):
            order = np.lexsort((vd[j], -vs[j]))[:k]
            ref_d, ref_s = vd[j][order], vs[j][order]
            gd, gs = last[qi].docs, last[qi].scores
            ok = ok and len(gd) == k and last[qi].total_hits == n and np.allclose(gs, ref_s, rtol=2e-5, atol=2e-6)
            score_of = {int(d_): float(s_) for d_, s_ in zip(vd[j], vs[j])}
            for r in range(min(k, len(gd))):
                if gd[r] != ref_d[r]:
                    off_rank += 1
                    ok = ok and int(gd[r]) in score_of and abs(score_of[int(gd[r])] - ref_s[r]) <= 2e-5 * ref_s[r] + 2e-6
        out["verify"] = {"agrees_with_fp64": bool(ok), "queries": len(verify_q), "rows": n, "docids_off_rank_among_near_ties": off_rank,
                         "what": "device top-100 vs fp64 cosine score map over every row (torch, on the device), last timed panel"}
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        # the C restatement of ExactVectorQuery + collector on the host cores, bounded sample: first rows of segment 0
        from oracle import oracle

        oracle.build()
        cores = usable_cpus()
        nq_cpu = min(Q, 32)   # (round 6: 32 queries -- half a panel -- instead of 8)
        t1 = time.perf_counter()
        docs, scores, cnt = oracle.knn_exact(0, panels[(args.warmup + args.steps - 1) % len(panels)][:nq_cpu], first_seg, k, n_threads=cores)
        dt = time.perf_counter() - t1
        # The sample doubles as a check of the device's answer: a row of the sample that made the device's top-k over ALL rows is
        # certainly among the k best of the sample -- so it must be in the oracle's list, with the same score (1e-5 relative: the
        # summation orders differ); and the sample's best row that the device ranks at all must come in the oracle's order.
        ok, checked = True, 0
        for qi in range(nq_cpu):
            dev = {int(d_): float(s_) for d_, s_ in zip(last[qi].docs, last[qi].scores) if int(d_) < len(first_seg)}
            cpu = {int(d_): float(s_) for d_, s_ in zip(docs[qi][: cnt[qi]], scores[qi][: cnt[qi]])}
            for d_, s_ in dev.items():
                checked += 1
                ok = ok and d_ in cpu and abs(cpu[d_] - s_) <= 1e-5 * max(abs(s_), 1e-30) + 1e-7
            order = [d_ for d_ in docs[qi][: cnt[qi]].tolist() if d_ in dev]
            ok = ok and order == [int(d_) for d_ in last[qi].docs if int(d_) in dev]
        out["cpu_baseline"] = {"value": round(nq_cpu * (n / len(first_seg)) ** -1 / dt, 3), "unit": "queries/s", "cores": cores, "kind": "port",
                               "rows_per_s": round(nq_cpu * len(first_seg) / dt, 1),
                               "sample": f"{nq_cpu} queries x the first {len(first_seg)} rows (oracle/nrt_oracle.c nrt_oracle_knn_exact, scalar fp32 "
                                         f"left to right, C + OpenMP, {cores} threads, {dt:.2f}s); value = queries/s extrapolated to {n} rows",
                               "agrees_with_device": bool(ok) and checked > 0, "device_hits_checked": checked}
    out["two_callers"] = two
    if not emit:   # a leg of another workload's line (main: roofline.c4): hand the line back, leave nothing resident
        for g in leaves:
            g.release()
        ctx.close()
        return out
    if rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    if world > 1:
        dist.barrier()
        ctx.dist_close()
        dist.destroy_process_group()
    ctx.close()
    return out


def c2_leg(args, device, flags, planner_threads):
    """BASELINE config 2 inside the C3 line (every BASELINE config on the round's build, in the driver's
    line): 1 M docs, Zipf terms, 2-term BooleanQuery, top-100, 1024 queries per step, FOUR submitting threads (the host's share decides this size) -- otherwise the same loop as the
    headline's.  (bench.py --workload C2 is the full line.)"""
    import threading

    from nrtsearch_amd import api, synth, workload

    w = workload.C2
    B = args.batch
    n_distinct = max(B, (w.n_queries // B) * B)
    qr = synth.make_queries(n_distinct, w.n_terms, w.max_rank)
    corpus = workload.build_shard_corpus(w, qr, 1, 0)
    ctx = api.GpuContext(device_id=device, max_batch=B, collect_timing=True, flags=flags, host_threads=planner_threads)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    queries = workload.boolean_queries(qr)
    mgr = api.TopScoreDocCollectorManager(w.k)
    batches = [api.PreparedBatch(sr, queries[i: i + B], [mgr] * B) for i in range(0, n_distinct, B)]
    for pb in batches:
        pb.run()
    # (four submitting threads: at this size a step is the host's -- 0.31 ms of kernel under 0.36 ms of planning -- and two threads
    #  leave the device idle a third of the time: measured 1.96 M queries/s with two, 2.68 M with three, 2.99 M with four at 6.6 CPUs)
    n_thr = max(4, args.host_threads)

    def run(first, count):
        def worker(tix):
            for i in range(tix, count, n_thr):
                batches[(first + i) % len(batches)].run()
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    run(0, 4)
    ctx.reset_stats()
    t0 = time.perf_counter()
    run(4, args.c2_steps)
    dt = time.perf_counter() - t0
    st = ctx.stats()
    pruned = st["maxscore_ms"] > st["scan_ms"]
    launches = max(1, st["maxscore_launches"] if pruned else st["scan_launches"])
    k_ms = (st["maxscore_ms"] if pruned else st["scan_ms"]) / launches
    algo = (st["maxscore_postings"] if pruned else st["scan_postings"]) / launches * BYTES_PER_POSTING
    rate = algo / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    rec = {
        "workload": w.name, "kernel": "bm25_maxscore_kernel" if pruned else "bm25_scan_kernel", "steps": args.c2_steps, "batch_queries": B, "host_threads": n_thr,
        "queries_per_s": round(args.c2_steps * B / dt, 1), "ms_per_step": round(dt / args.c2_steps * 1e3, 4), "avg_launch_ms": round(k_ms, 4),
        "algorithmic_bytes_per_launch": int(algo), "effective_achieved": round(rate, 1), "effective_frac": round(rate / HBM_PEAK_GBS, 4),
        "unit": "GB/s", "peak": HBM_PEAK_GBS, "host_plan_ms_per_step": round(st["host_plan_ms"] / max(1, st["batches"]), 4),
        "device_bytes": int(sum(l.device_bytes for l in leaves)), "mean_postings_per_query": float(workload.postings_per_query(corpus.doc_freq, qr).mean()),
        "note": "effective = 9 B x the postings of the queries' terms / the kernel's average launch (the pruned kernel streams a few per cent of "
                "them); a step is host-bound at this size when ms_per_step exceeds avg_launch_ms",
    }
    for l in leaves:
        l.release()
    ctx.close()
    return rec


def c5_leg(args, device, planner_threads, n_docs=5_000_000, seg_docs=2_500_000, dim=768, B=256):
    """BASELINE config 5's shape inside the C3 line, at the 5 M-doc size the GPU suite checks against the oracle
    (tests/test_baseline_sizes_gpu.py::test_hybrid_c5_shape): BM25 recall-1000 over a 5-term disjunction -> exact cosine rescore of
    the 1000 recalled docs against their 768-d fp32 vectors -> top-100, first pass and tail fused on the device
    (nrtgpu_search_hybrid_batch; QueryRescore.java:40-57).  The vectors of every segment are one random block (the tail gathers
    1000 rows per query: their values do not matter for the timing).  50 M docs: scripts/gpu_c5_hybrid.py."""
    import ctypes as C

    import numpy as np
    import torch

    from nrtsearch_amd import _lib, api, synth

    t0 = time.perf_counter()
    qr = synth.make_queries(B, 5, 10000)
    ranks = sorted(set(int(r) for r in qr.reshape(-1)))
    lens = synth.doc_lengths(n_docs)
    norms_all = synth.int_to_byte4(lens)
    n_seg = (n_docs + seg_docs - 1) // seg_docs
    bases = np.minimum(np.arange(n_seg + 1, dtype=np.int64) * seg_docs, n_docs)
    per_docs = [[] for _ in range(n_seg)]
    per_freqs = [[] for _ in range(n_seg)]
    doc_freq = {}
    for r in ranks:
        d, f = synth.term_postings(n_docs, r)
        doc_freq[r] = int(len(d))
        cuts = np.searchsorted(d, bases)
        for s_ in range(n_seg):
            a, b = int(cuts[s_]), int(cuts[s_ + 1])
            per_docs[s_].append((d[a:b] - bases[s_]).astype(np.int32))
            per_freqs[s_].append(f[a:b])
    # (the block comes from the device's generator: 7.7 GB of normal variates take the host's tens of seconds)
    block = torch.randn((seg_docs, dim), dtype=torch.float32, device=f"cuda:{device}", generator=torch.Generator(device=f"cuda:{device}").manual_seed(7)).cpu().numpy()
    ctx = api.GpuContext(device_id=device, max_batch=B, collect_timing=True, host_threads=planner_threads)
    leaves = []
    for s_ in range(n_seg):
        counts = np.asarray([len(x) for x in per_docs[s_]], dtype=np.int64)
        g = api.GpuSegment(ctx, int(bases[s_ + 1] - bases[s_]), int(bases[s_]))
        g.add_field_norms(0, norms_all[bases[s_]: bases[s_ + 1]].copy())
        g.add_terms(0, np.asarray(ranks, dtype=np.int64), np.concatenate([[0], np.cumsum(counts)]).astype(np.int64),
                    np.ascontiguousarray(np.concatenate(per_docs[s_]), dtype=np.int32), np.ascontiguousarray(np.concatenate(per_freqs[s_]), dtype=np.int32))
        g.add_vectors(7, block[: int(bases[s_ + 1] - bases[s_])])
        g.seal()
        leaves.append(g)
    del block
    stats = api.IndexStatistics()
    stats.fields[0] = api.CollectionStatistics(n_docs, int(lens.astype(np.int64).sum()))
    for t_, df_ in doc_freq.items():
        stats.doc_freq[(0, int(t_))] = int(df_)
    sr = api.GpuIndexSearcher(ctx, leaves, stats)
    build_s = time.perf_counter() - t0
    queries = [api.BooleanQuery(tuple(api.TermQuery(0, int(t)) for t in row)) for row in qr]
    mgr = api.TopScoreDocCollectorManager(1000)
    qv = np.random.default_rng(8).standard_normal((B, dim), dtype=np.float32)
    L = _lib.load()
    m = sr._marshal(queries, [mgr] * B)
    outs = (_lib.TopDocs * B)()
    od = np.zeros((B, 1000), np.int32)
    os_ = np.zeros((B, 1000), np.float32)
    for qi in range(B):
        outs[qi].capacity = 1000
        outs[qi].docs = od[qi].ctypes.data_as(C.POINTER(C.c_int32))
        outs[qi].scores = os_[qi].ctypes.data_as(C.POINTER(C.c_float))

    def fused():
        _lib.check(L.nrtgpu_search_hybrid_batch(ctx._h, sr._segs, sr._bases, len(leaves), m.queries, B, 7, 0, qv.ctypes.data, dim, C.c_float(1.0), 1.0, 2.0,
                                                100, outs))

    def first_pass():
        _lib.check(L.nrtgpu_search_bm25_batch(ctx._h, sr._segs, sr._bases, len(leaves), m.queries, B, outs))

    fused()
    first_pass()
    ctx.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.c5_steps):
        fused()
    dt_f = (time.perf_counter() - t0) / args.c5_steps
    st = ctx.stats()
    t0 = time.perf_counter()
    for _ in range(args.c5_steps):
        first_pass()
    dt_1 = (time.perf_counter() - t0) / args.c5_steps
    # two callers taking the batches in turn (each with output arrays of its own): a caller's planning and unpacking run under the
    # other's kernels -- what a server's request threads do; the single caller's numbers above stay the leg's headline
    two = None
    if args.c5_steps >= 4:
        import threading

        def outputs():
            o = (_lib.TopDocs * B)()
            d_, s_ = np.zeros((B, 1000), np.int32), np.zeros((B, 1000), np.float32)
            for qi in range(B):
                o[qi].capacity = 1000
                o[qi].docs = C.cast(d_.ctypes.data + qi * 4000, C.POINTER(C.c_int32))
                o[qi].scores = C.cast(s_.ctypes.data + qi * 4000, C.POINTER(C.c_float))
            return o, d_, s_

        mine = [outputs(), outputs()]

        def caller(t):
            for _ in range(t, args.c5_steps, 2):
                _lib.check(L.nrtgpu_search_hybrid_batch(ctx._h, sr._segs, sr._bases, len(leaves), m.queries, B, 7, 0, qv.ctypes.data, dim, C.c_float(1.0),
                                                        1.0, 2.0, 100, mine[t][0]))

        t0 = time.perf_counter()
        ths = [threading.Thread(target=caller, args=(t,)) for t in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt_2 = (time.perf_counter() - t0) / args.c5_steps
        two = {"callers": 2, "ms_per_batch": round(dt_2 * 1e3, 3), "queries_per_s": round(B / dt_2, 1)}
    pruned = st["maxscore_ms"] > st["scan_ms"]
    k_ms = (st["maxscore_ms"] / max(1, st["maxscore_launches"])) if pruned else (st["scan_ms"] / max(1, st["scan_launches"]))
    ppq = float(np.mean([sum(doc_freq[int(t)] for t in row) for row in qr]))
    rec = {
        "workload": f"C5 shape: {n_docs // 1_000_000}M docs BM25 recall-1000 + {dim}-d exact cosine rescore top-100, fused on the device",
        "docs": n_docs, "dim": dim, "batch_queries": B, "batches": args.c5_steps,
        "queries_per_s": round(B / dt_f, 1), "ms_per_batch": round(dt_f * 1e3, 3), "first_pass_ms_per_batch": round(dt_1 * 1e3, 3),
        "tail_ms_per_batch": round((dt_f - dt_1) * 1e3, 3), "two_callers": two, "first_pass_kernel": "bm25_maxscore_kernel" if pruned else "bm25_scan_kernel",
        "first_pass_kernel_ms": round(k_ms, 4), "mean_postings_per_query": ppq,
        "effective_frac": round(9.0 * ppq * B / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else None,
        "rescore_gather_bytes_per_batch": B * 1000 * dim * 4, "device_bytes": int(sum(l.device_bytes for l in leaves)), "setup_s": round(build_s, 1),
        "note": "parity of this shape against the oracle: tests/test_baseline_sizes_gpu.py::test_hybrid_c5_shape; tail = fused call - first pass alone",
    }
    for l in leaves:
        l.release()
    ctx.close()
    return rec


def main():
    args = parse_args()
    if os.environ.get("NRTGPU_BENCH_WATCHDOG"):   # debug aid: every thread's Python stack on stderr after that many seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["NRTGPU_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # launched as plain `python bench.py --gpus N`: start the ranks ourselves, exactly as the driver would
            # (one process per GPU over RCCL, rendezvous on 127.0.0.1), and hand their output and exit code through
            import socket
            import subprocess

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.call(cmd))
        args.gpus = world
    if args.workload == "C4":
        return run_c4(args)
    # the topology (--doc-shards): D ranks per doc-shard group, R groups; rank = group * D + its place in the group
    from nrtsearch_amd import dist as nd

    try:
        D, R, group, grank = nd.topology(world, rank, args.doc_shards)
    except ValueError as e_:
        sys.exit(f"--doc-shards: {e_}")

    import torch  # first: its bundled HIP runtime must be the one libnrtgpu.so binds to
    import torch.distributed as dist

    if args.debug_same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.debug_same_gpu:
            stdout_to_stderr(lambda: (dist.init_process_group("gloo"), dist.barrier()))   # (gloo announces its connections on stdout)
        else:
            def _init():
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()   # the communicator is created here at the latest

            stdout_to_stderr(_init)

    # the doc-shard group's own process group (torch.distributed collectives of the exchange stay inside it); every rank creates
    # every group, in the same order
    pg = nd.doc_shard_groups(world, D) if world > 1 else None

    def all_gather(dst, src):
        if args.debug_same_gpu:   # gloo: stage through the host
            tmp = torch.empty(dst.shape, dtype=dst.dtype)
            dist.all_gather_into_tensor(tmp, src.cpu(), group=pg)
            dst.copy_(tmp)
        else:
            dist.all_gather_into_tensor(dst, src, group=pg)

    def all_to_all(dst, src):
        if args.debug_same_gpu:
            tmp = torch.empty(dst.shape, dtype=dst.dtype)
            dist.all_to_all_single(tmp, src.cpu(), group=pg)
            dst.copy_(tmp)
        else:
            dist.all_to_all_single(dst, src, group=pg)

    import numpy as np

    from nrtsearch_amd import _lib, api, build, synth, workload

    build.build()
    w = {"C2": workload.C2, "C3": workload.C3, "SMOKE": workload.SMOKE}[args.workload]
    if args.docs:
        w.n_docs = args.docs
    if args.debug_k:
        w.k = args.debug_k
    B = args.batch
    n_distinct = max(B, (w.n_queries // B) * B)
    qranks = synth.make_queries(n_distinct, w.n_terms, w.max_rank)
    t_build = time.perf_counter()
    # (a real job: this rank's docid shard is its place in its doc-shard group; one GPU emulating a rank of a larger job: that rank's)
    shard_world, shard_rank = (args.emulate_world, args.emulate_rank) if (args.emulate_world > 1 and world == 1) else (D, grank)
    corpus = workload.build_shard_corpus(w, qranks, shard_world, shard_rank, layout=args.shard_layout, variant=args.corpus_variant)
    t_build = time.perf_counter() - t_build

    flags = ((_lib.NRTGPU_FLAG_NO_PREFETCH if args.no_prefetch else 0) | (_lib.NRTGPU_FLAG_NO_PRUNE if args.no_prune else 0)
             | (_lib.NRTGPU_FLAG_PACKED_POSTINGS if args.packed else 0)
             # several ranks share this host's CPUs: their callers sleep on their results instead of spinning (the box grants
             # 16 CPUs; 8 ranks x (2 scan threads + the exchange thread) spinning would want 24)
             | (_lib.NRTGPU_FLAG_BLOCKING_WAIT if (shard_world > 1 or args.blocking_wait) else 0))
    # planner threads per in-flight call: what the box's CPUs allow once every rank has its submitting threads
    # (the node's ranks share the host; 4 is the library's default and enough at one rank)
    # (a rank of an N-GPU job has ONE submitting thread unless --sync-submit)
    if args.sync_submit or not (D > 1 or args.force_dist):
        submitters = max(1, args.host_threads)
    else:   # begin / wait: one submitting thread per rank, two where the host can afford them
        submitters = args.submitters if args.submitters > 0 else (2 if usable_cpus() >= 6 * max(world, shard_world) else 1)
        submitters = max(1, min(submitters, 2))   # (three result buffers are in flight: at most two steps being planned)
    planner_threads = args.planner_threads or max(1, min(4, usable_cpus() // max(1, max(world, shard_world) * submitters)))
    ctx = api.GpuContext(device_id=local_rank, max_batch=B, target_items=args.target_items, collect_timing=True, flags=flags,
                         host_threads=planner_threads)
    if args.speculation_margin >= 0.0:
        ctx.set_speculation(args.speculation_margin)
    if shard_world > 1:   # this shard's real share of the index (the "balanced" layout and the index's last shard are not 1 / N)
        ctx.set_shard_share(sum(s_.max_doc for s_ in corpus.segments), w.n_docs)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    searcher = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    queries = workload.boolean_queries(qranks)
    mgr = api.TopScoreDocCollectorManager(w.k)  # default totalHitsThreshold = 1000
    batches = [api.PreparedBatch(searcher, queries[i: i + B], [mgr] * B) for i in range(0, n_distinct, B)]
    ppq = workload.postings_per_query(corpus.doc_freq, qranks)   # index-global P per query

    k_stride = (w.k + 15) // 16 * 16
    use_dist = D > 1 or args.force_dist   # (D == 1 at N > 1: replicas -- every rank runs the one-GPU loop over ITS batches, nothing is exchanged)

    def batch_index(step):
        """Which batch of the query set this rank's doc-shard group runs at its step `step`: the groups interleave (group g takes
        batches g, g + R, ...), so that at any time the R groups work on R different batches."""
        return (step * R + group) % len(batches)
    exchange_name = None
    emu_exchange = False
    lib_mode = api.EXCHANGE_ALLGATHER
    peer_words = None   # --emulate-peers final: [batch] -> (shard_world, B) uint32 score bits the other ranks would publish
    shard_spec = use_dist and shard_world > 1 and args.shard_bounds == "speculation" and not args.no_prune and args.emulate_peers == "none"
    shard_spec_stat = {"queries": 0, "failed": 0, "reran_batches": 0}
    if D > 1 and world > 1 and not args.no_exchange and args.shard_bounds == "exchange":
        # cross-GPU bound exchange (include/nrtgpu.h): one shared-memory table per doc-shard group, opened by every rank of it
        import uuid
        box = [f"/nrtgpu_bench_{uuid.uuid4().hex[:16]}" if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        exchange_name = box[0] + f"_g{group}"
        try:
            ctx.exchange_open(exchange_name, D, grank)
        except Exception as e:   # a rank without the table only misses the pruning; results do not depend on it
            print(f"[rank {rank}] bound exchange unavailable: {e}", file=sys.stderr, flush=True)
        dist.barrier()
    elif world == 1 and shard_world > 1 and args.emulate_peers != "none":
        import uuid
        k2 = -(-w.k // (shard_world - 1))
        peer_words = [np.zeros((shard_world, B), dtype=np.uint64) for _ in batches]
        for r in range(shard_world):
            if r == shard_rank:
                continue
            c_r = workload.build_shard_corpus(w, qranks, shard_world, r, layout=args.shard_layout, variant=args.corpus_variant)
            ctx_r = api.GpuContext(device_id=local_rank, max_batch=B, flags=flags, host_threads=planner_threads)
            leaves_r = [api.GpuSegment.from_data(ctx_r, s) for s in c_r.segments]
            sr_r = api.GpuIndexSearcher(ctx_r, leaves_r, api.IndexStatistics.from_corpus(c_r))
            for bi in range(len(batches)):
                res = sr_r.search_batch(queries[bi * B: (bi + 1) * B], [mgr] * B)
                sc = np.array([float(t.scores[k2 - 1]) if len(t.scores) >= k2 else 0.0 for t in res], dtype=np.float32)
                peer_words[bi][r] = sc.view(np.uint32).astype(np.uint64)
            for l in leaves_r:
                l.release()
            ctx_r.close()
            del c_r
        exchange_name = f"/nrtgpu_bench_{uuid.uuid4().hex[:16]}"
        ctx.exchange_open(exchange_name, shard_world, shard_rank)
        peer_table = np.memmap("/dev/shm" + exchange_name, dtype=np.uint64, mode="r+", shape=(8, shard_world, B))
        peer_rows = [r for r in range(shard_world) if r != shard_rank]

    def play_peers(epoch):
        """The other ranks' rows of the exchange table for this epoch (slot epoch % 8, tag epoch + 1), written before the
        step is submitted; at most NB steps are in flight, so the slot's previous epoch is long finished."""
        if peer_words is not None:
            words = peer_words[epoch % len(batches)]
            peer_table[epoch % 8, peer_rows, :] = (np.uint64(epoch + 1) << np.uint64(32)) | words[peer_rows]
    NB = 3  # device result buffers in flight between the scan threads and the exchange thread
    lib_collective = False
    lib_collective_hung = False   # its setup thread never came back: leave through os._exit at the end
    if use_dist:
        bufs = [(torch.zeros((B, k_stride), dtype=torch.int64, device="cuda"),
                 torch.zeros((B,), dtype=torch.int32, device="cuda"),
                 torch.zeros((B,), dtype=torch.int64, device="cuda")) for _ in range(NB)]
        guess_bufs = [torch.zeros((B,), dtype=torch.int64, device="cuda") for _ in range(NB)]   # --shard-bounds speculation
        guess_host = torch.zeros((B,), dtype=torch.int64).pin_memory()
        # Default (BASELINE.json's north star): RCCL all-gather of every rank's top-k (rank r's rows are
        # [r * B, (r + 1) * B)), every rank merges everything and holds every answer.  --all-to-all splits the reduce
        # instead: rank r receives every rank's lists for ITS B / world queries (rows [j * B/world, (j + 1) * B/world)
        # came from rank j) and merges only those -- 1/world of the bytes and of the merge work; the exchange stage
        # is overlapped with the scans either way, so this changes latency, not throughput.
        split_reduce = (B % D == 0) and args.all_to_all and args.torch_collective
        if split_reduce and world > 1:   # probe the collective once; every rank must take the same path
            ok = 1
            try:
                probe = torch.zeros((D * 2,), dtype=torch.int64, device="cuda")
                all_to_all(torch.empty_like(probe), probe)
                torch.cuda.synchronize()
            except Exception as e:   # noqa: BLE001
                print(f"[rank {rank}] all_to_all_single unavailable ({e}); using all-gather", file=sys.stderr, flush=True)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if args.debug_same_gpu else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            split_reduce = bool(flag.item())
        rows = B if split_reduce else D * B
        g_keys = torch.zeros((rows, k_stride), dtype=torch.int64, device="cuda")
        g_cnt = torch.zeros((rows,), dtype=torch.int32, device="cuda")
        g_hits = torch.zeros((rows,), dtype=torch.int64, device="cuda")
        mq = B // D if split_reduce else B
        merger = api.PreparedMerge(ctx, D, mq, k_stride, [w.k] * mq, [api.TOTAL_HITS_THRESHOLD] * mq)
        # The exchange stage through the C ABI (what a JVM caller has): the library's own RCCL communicator, one grouped
        # all-gather + merge per batch.  Every rank must take the same path: agree on it once.
        lib_mode = api.EXCHANGE_ALLTOALL if (args.exchange_mode == "alltoall" and B % D == 0) else api.EXCHANGE_ALLGATHER
        # One GPU playing rank r of an N-GPU job (--emulate-world): the exchange stage does the work THAT rank would do -- the
        # merge of N lists for its slice of the batch (all-to-all) or for every query (all-gather), the result copy and the
        # unpacking -- with the other ranks' lists stood in for by copies of its own (the xGMI transfer itself, ~1 MB per link and
        # batch in the all-to-all form, is not in it).
        emu_exchange = world == 1 and shard_world > 1 and not args.torch_collective and B % shard_world == 0
        if emu_exchange:
            W_e = shard_world
            mq_e = B // W_e if lib_mode == api.EXCHANGE_ALLTOALL else B
            # per batch of the query set: the W_e lists as the exchange would leave them.  The OTHER ranks' lists are staged once,
            # before the warmup (below) -- in a real job they arrive over xGMI while this rank scans; list l of a batch is this
            # rank's own list of that batch with bits 27-29 of the docid word flipped by (l - rank) mod W_e, so the merge sees W_e
            # disjoint lists of the same shape and order -- and per step only this rank's own list is copied into its place.
            e_keys = [torch.zeros((W_e, mq_e, k_stride), dtype=torch.int64, device="cuda") for _ in batches]
            e_cnt = [torch.zeros((W_e, mq_e), dtype=torch.int32, device="cuda") for _ in batches]
            e_hits = [torch.zeros((W_e, mq_e), dtype=torch.int64, device="cuda") for _ in batches]
            e_tag = (((torch.arange(W_e, dtype=torch.int64, device="cuda") - shard_rank) % W_e) << 27).view(W_e, 1, 1)
            merger = api.PreparedMerge(ctx, W_e, mq_e, k_stride, [w.k] * mq_e, [api.TOTAL_HITS_THRESHOLD] * mq_e)
        # (NRTGPU_BENCH_DEBUG_LIB_COLLECTIVE=1: the library's collective under --debug-same-gpu too -- tests/test_bench_two_ranks_gpu.py
        #  binds it to tests/mockrccl, which carries messages between processes that share a GPU)
        debug_lib = args.debug_same_gpu and os.environ.get("NRTGPU_BENCH_DEBUG_LIB_COLLECTIVE") == "1"
        if not (args.torch_collective or (args.debug_same_gpu and not debug_lib) or split_reduce or emu_exchange):
            ok = 1
            try:
                # one communicator per doc-shard group: its first rank makes the id, every rank learns every group's
                ids = [None] * world
                mine = api.GpuContext.dist_unique_id() if grank == 0 else None
                if world > 1:
                    dist.all_gather_object(ids, mine)
                else:
                    ids = [mine]
                box = [ids[group * D]]
                # communicator + one exchange of the real shape, under a watchdog: a setup that never returns must cost
                # this run the library path, not the measurement (the ranks then agree on torch.distributed below)
                import threading as _th

                probe_err = []

                def _probe():
                    try:
                        stdout_to_stderr(lambda: ctx.dist_init(D, grank, box[0]))
                        keys0, cnt0, hits0 = bufs[0]
                        merger.run_dist(keys0.data_ptr(), cnt0.data_ptr(), hits0.data_ptr(), lib_mode)   # (zero counts: merges nothing)
                    except Exception as e_:   # noqa: BLE001
                        probe_err.append(e_)

                th = _th.Thread(target=_probe, daemon=True)
                th.start()
                th.join(timeout=float(os.environ.get("NRTGPU_BENCH_COLLECTIVE_TIMEOUT", "180")))
                if th.is_alive():
                    lib_collective_hung = True
                    raise RuntimeError("setup did not finish in time")
                if probe_err:
                    raise probe_err[0]
            except Exception as e:   # noqa: BLE001
                print(f"[rank {rank}] library collective unavailable ({e}); using torch.distributed", file=sys.stderr, flush=True)
                ok = 0
            if world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if args.debug_same_gpu else "cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            lib_collective = bool(ok)

    import threading
    from concurrent.futures import ThreadPoolExecutor

    last = {}
    lat = []
    stage = {"scan_call_s": 0.0, "wait_s": 0.0, "exchange_s": 0.0, "merge_call_s": 0.0, "steps": 0}   # multi-GPU path: where a step's time goes

    def run_steps(first, count, record):
        """`count` steps starting at batch index `first`.  The C ABI is thread-safe (one workspace + HIP
        stream per in-flight call, ctypes drops the GIL), so plan building of step i+1 overlaps the
        kernels of step i.  Multi-GPU: scan threads leave each rank's top-k in HBM; this thread issues
        the collectives in step order (RCCL all-gather over xGMI, or --all-to-all) and runs TopDocs.merge."""
        if not use_dist:
            n_thr = max(1, args.host_threads)

            def worker(tix):
                for i in range(tix, count, n_thr):
                    ts = time.perf_counter()
                    pb = batches[batch_index(first + i)]
                    pb.run()
                    if record:
                        lat.append(time.perf_counter() - ts)
                    last["td"] = pb

            if n_thr == 1:
                worker(0)
            else:
                threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
                for t in threads:
                    t.start()
                for t in threads:
                    t.join()
            return
        free = [threading.Semaphore(1) for _ in range(NB)]
        t_start = [0.0] * count
        # --shard-bounds speculation needs somebody who checks the guesses: the library's exchange (run_dist_checked) or the
        # emulated one below; with torch.distributed carrying the lists a shard speculates on its own list only
        spec_on = [shard_spec and (emu_exchange or lib_collective) and not args.sync_submit]

        def run_again(bi_, bad):
            """Queries whose guess failed the check against the merged list: every rank runs THOSE again without speculation -- a small
            launch of its own behind the batch, whose other answers stand -- and their lists are exchanged and merged once more
            (round 6; through round 5 the whole batch was run and merged again here: a stand-in that overstated the cost)."""
            shard_spec_stat["reran_batches"] += 1
            shard_spec_stat["reran_queries"] = shard_spec_stat.get("reran_queries", 0) + len(bad)
            nb_ = len(bad)
            q0 = bi_ * B
            pb2 = api.PreparedBatch(searcher, [queries[q0 + int(j)] for j in bad], [mgr] * nb_)
            tmp = (torch.zeros((nb_, k_stride), dtype=torch.int64, device="cuda"), torch.zeros((nb_,), dtype=torch.int32, device="cuda"),
                   torch.zeros((nb_,), dtype=torch.int64, device="cuda"))
            h = pb2.begin_shard_device(k_stride, tmp[0].data_ptr(), tmp[1].data_ptr(), tmp[2].data_ptr(), 0, 0)
            api.PreparedBatch.wait_device(h)
            return tmp

        pending = [None] * count

        def produce(i):
            b = i % NB
            free[b].acquire()          # the exchange thread has gathered this buffer's previous contents
            t_start[i] = time.perf_counter()
            keys, cnt, hits = bufs[b]
            pb = batches[batch_index(first + i)]
            play_peers(first + i)
            if spec_on[0]:   # this shard's thresholds: guesses at the whole search's k-th score, left in guess_bufs[b] for the check
                pending[i] = pb.begin_shard_device(k_stride, keys.data_ptr(), cnt.data_ptr(), hits.data_ptr(), shard_world,
                                                   guess_bufs[b].data_ptr())
            elif args.sync_submit:
                pb.run_device(k_stride, keys.data_ptr(), cnt.data_ptr(), hits.data_ptr(), epoch=(first + i) if exchange_name else -1)
            else:   # plan + enqueue only: the exchange thread waits for the results
                pending[i] = pb.begin_device(k_stride, keys.data_ptr(), cnt.data_ptr(), hits.data_ptr(), epoch=(first + i) if exchange_name else -1)
            if record:
                stage["scan_call_s"] += time.perf_counter() - t_start[i]
            return b

        def step_failed(exc):
            """A step of the exchange stage failed: the scan threads wait for buffers nobody will hand back, the other ranks for a
            collective this one will not issue -- say what happened and leave at once (the launcher ends the other ranks)."""
            import traceback
            print(f"[rank {rank}] the exchange stage failed at a step: {exc!r}", file=sys.stderr, flush=True)
            traceback.print_exc(file=sys.stderr)
            sys.stderr.flush()
            os._exit(1)

        with ThreadPoolExecutor(max_workers=max(1, args.host_threads) if args.sync_submit else submitters) as ex:  # FIFO: steps start in order
            futs = [ex.submit(produce, i) for i in range(count)]
            for i in range(count):
              try:
                b = futs[i].result()
                if pending[i] is not None:
                    tw0 = time.perf_counter()
                    api.PreparedBatch.wait_device(pending[i])
                    if record:
                        stage["wait_s"] += time.perf_counter() - tw0
                te0 = time.perf_counter()
                keys, cnt, hits = bufs[b]
                if emu_exchange:
                    # (a key is (score bits << 32) | ~docid and docids are unique across the shards of a real job: the stand-in
                    #  lists get distinct docids -- list l flips bits 27-29 of the docid word with l -- so the merge sees N disjoint
                    #  lists of the same shape and order; the selection kernels assume unique keys)
                    bi_ = batch_index(first + i)
                    e_keys[bi_][shard_rank].copy_(keys[:mq_e])
                    e_cnt[bi_][shard_rank].copy_(cnt[:mq_e])
                    e_hits[bi_][shard_rank].copy_(hits[:mq_e])
                    if spec_on[0]:
                        guess_host[:mq_e].copy_(guess_bufs[b][:mq_e], non_blocking=True)
                    torch.cuda.current_stream().synchronize()
                    free[b].release()
                    te1 = time.perf_counter()
                    merger.run(e_keys[bi_].data_ptr(), e_cnt[bi_].data_ptr(), e_hits[bi_].data_ptr())
                    if spec_on[0]:
                        # the check of nrtgpu_dist_exchange_merge_checked, done here because the lists did not travel: the k-th key of
                        # the merged list must reach the largest guess (the other ranks' guesses are stood in for by this one's)
                        g_ = guess_host[:mq_e].numpy().view(np.uint64)
                        bad = np.flatnonzero((g_ != 0) & (merger.kth_keys() < g_))
                        shard_spec_stat["queries"] += mq_e
                        shard_spec_stat["failed"] += len(bad)
                        batches[batch_index(first + i)].note_shard_speculation(mq_e, len(bad))
                        if len(bad):
                            # the failed queries alone: run again, their W_e lists stood in for as at the setup, merged on their own
                            tk, tc, th_ = run_again(bi_, bad)
                            nb_ = len(bad)
                            rk = torch.bitwise_xor(tk.unsqueeze(0), e_tag).contiguous()
                            rc = tc.unsqueeze(0).expand(W_e, nb_).contiguous()
                            rh = th_.unsqueeze(0).expand(W_e, nb_).contiguous()
                            torch.cuda.current_stream().synchronize()
                            api.PreparedMerge(ctx, W_e, nb_, k_stride, [w.k] * nb_, [api.TOTAL_HITS_THRESHOLD] * nb_).run(rk.data_ptr(), rc.data_ptr(), rh.data_ptr())
                    if record:
                        lat.append(time.perf_counter() - t_start[i])
                        stage["exchange_s"] += te1 - te0
                        stage["merge_call_s"] += time.perf_counter() - te1
                        stage["steps"] += 1
                    continue
                if lib_collective and spec_on[0]:
                    # exchange (the guesses ride along) + merge + the check of the guesses against the merged lists: the same
                    # verdicts on every rank; what failed is run again by every rank without speculation, gathered whole
                    bad = merger.run_dist_checked(keys.data_ptr(), cnt.data_ptr(), hits.data_ptr(), guess_bufs[b].data_ptr(), lib_mode)
                    free[b].release()
                    pb_ = batches[batch_index(first + i)]
                    shard_spec_stat["queries"] += B
                    shard_spec_stat["failed"] += len(bad)
                    pb_.note_shard_speculation(B, len(bad))
                    if len(bad):
                        shard_spec_stat["reran_batches"] += 1
                        q0 = batch_index(first + i) * B
                        last["rerun"] = searcher.dist_search_batch([queries[q0 + int(j)] for j in bad], [mgr] * len(bad),
                                                                   mode=api.EXCHANGE_ALLGATHER | api.EXCHANGE_NO_SPECULATION)
                    if record:
                        lat.append(time.perf_counter() - t_start[i])
                        stage["exchange_s"] += time.perf_counter() - te0
                        stage["steps"] += 1
                    continue
                if lib_collective:
                    merger.run_dist(keys.data_ptr(), cnt.data_ptr(), hits.data_ptr(), lib_mode)   # exchange + merge, synchronous
                    free[b].release()
                    if record:
                        lat.append(time.perf_counter() - t_start[i])
                        stage["exchange_s"] += time.perf_counter() - te0
                        stage["steps"] += 1
                    continue
                exchange = all_to_all if split_reduce else all_gather
                exchange(g_keys, keys) if (world > 1 and D > 1) else g_keys.copy_(keys)
                exchange(g_cnt, cnt) if (world > 1 and D > 1) else g_cnt.copy_(cnt)
                exchange(g_hits, hits) if (world > 1 and D > 1) else g_hits.copy_(hits)
                torch.cuda.current_stream().synchronize()   # only this stream: the next scan keeps running
                free[b].release()
                te1 = time.perf_counter()
                merger.run(g_keys.data_ptr(), g_cnt.data_ptr(), g_hits.data_ptr())
                if record:
                    lat.append(time.perf_counter() - t_start[i])
                    stage["exchange_s"] += te1 - te0
                    stage["merge_call_s"] += time.perf_counter() - te1
                    stage["steps"] += 1
              except Exception as exc_:   # noqa: BLE001
                step_failed(exc_)
        last["td"] = merger

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Setup, before the warmup: every batch of the query set is planned once, so that the resident per-leaf table of every query
    # term exists (the library builds a term's table the first time a query names it and keeps it -- index-side state like the
    # postings themselves, not a result).  Without this, --warmup 5 of a ten-batch query set leaves the first pass over batches
    # 5-9 inside the timed region paying for it (the driver's --steps 20: a quarter of its steps).
    if not use_dist:
        for pb in batches:
            pb.run()

    def stage_lists(bi_, keys, cnt, hits):
        """The emulated exchange's W_e lists of batch bi_ from this rank's own (see e_keys above)."""
        torch.bitwise_xor(keys[:mq_e].unsqueeze(0), e_tag, out=e_keys[bi_])
        e_cnt[bi_].copy_(cnt[:mq_e].unsqueeze(0).expand(W_e, mq_e))
        e_hits[bi_].copy_(hits[:mq_e].unsqueeze(0).expand(W_e, mq_e))

    if use_dist and emu_exchange:
        for bi_, pb in enumerate(batches):   # (outside the bound exchange: a shard's top-k is the same list with or without bounds)
            keys, cnt, hits = bufs[0]
            pb.run_device(k_stride, keys.data_ptr(), cnt.data_ptr(), hits.data_ptr(), epoch=-1)
            stage_lists(bi_, keys, cnt, hits)
        torch.cuda.synchronize()
    run_steps(0, args.warmup, False)
    ctx.reset_stats()
    import gc
    gc.collect()
    gc.disable()   # (as in run_c4: the interpreter's cycle collector is not what is measured)
    fence()
    n_thr = max(1, args.host_threads)
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    tc0 = thread_cpu_seconds()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps, True)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    tc1 = thread_cpu_seconds()
    # CPUs kept busy by kind of thread: this process's main thread (N > 1: the exchange stage), the library's helper threads
    # (planning, unpacking), everything else (the threads submitting steps, the runtimes' own)
    by_kind = {}
    for tid, (comm, sec) in tc1.items():
        kind = "main" if tid == os.getpid() else ("library helpers" if comm.startswith("nrtgpu-helper") else "submitting + runtime threads")
        by_kind[kind] = by_kind.get(kind, 0.0) + sec - tc0.get(tid, (comm, 0.0))[1]
    cpu_by_kind = {k_: round(v / max(elapsed, 1e-9), 2) for k_, v in by_kind.items()}
    host_cpu_busy = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / max(elapsed, 1e-9)   # CPUs this rank kept busy
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.debug_same_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = ctx.stats()
    n_q = args.steps * B * R   # (every doc-shard group ran `steps` batches of its own)
    qps = n_q / elapsed
    # Dominant kernel: whichever of the two scorers took more device time -- bm25_maxscore_kernel (dynamic pruning;
    # the default for this workload) or bm25_scan_kernel (exhaustive).  achieved = algorithmic bytes per launch / avg
    # launch time, measured with HIP events on the library's own stream (collect_timing).  Algorithmic bytes = 9 B x the
    # postings of the launch's query terms (the exhaustive-scan figure also normalises a pruned run, whose
    # "effective" rate may therefore exceed the peak; the physical rate = PMC traffic / launch time is reported beside it).
    pruned = st["maxscore_ms"] > st["scan_ms"]
    kernel = "bm25_maxscore_kernel" if pruned else "bm25_scan_kernel"
    launches = max(1, st["maxscore_launches"] if pruned else st["scan_launches"])
    scan_ms = (st["maxscore_ms"] if pruned else st["scan_ms"]) / launches
    bpp = 4 if args.packed else BYTES_PER_POSTING               # packed: the one word IS the posting (norm and freq inside its code)
    bpp_fused = 4 if args.packed else BYTES_PER_POSTING_FUSED
    bytes_per_launch = (st["maxscore_postings"] if pruned else st["scan_postings"]) / launches * bpp
    achieved = bytes_per_launch / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    # HBM traffic of the dominant kernel: a PMC profile of THIS build (rocprofv3 --pmc needs a run of its own, so the number is a
    # committed record) -- a record of another build's kernels says nothing about these: traffic stays null then
    traffic = None
    traffic_note = None
    lib_id = build.build_id(_lib.LIB_PATH)
    traffic_refused = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            recs = json.load(open(pmc))
            for rec in (recs if isinstance(recs, list) else [recs]):
                if (rec.get("workload") == args.workload and rec.get("batch") == B and rec.get("kernel") == kernel
                        and bool(rec.get("packed", False)) == bool(args.packed) and world == 1 and shard_world == 1):
                    if rec.get("build_id") is not None and rec.get("build_id") == lib_id:
                        traffic = rec.get("hbm_bytes_per_launch")
                        traffic_note = rec.get("note")
                    else:
                        traffic_refused = (f"profiles/pmc_traffic.json holds a record for this workload taken from build {rec.get('build_id')}; "
                                           f"the loaded library is build {lib_id}: not reused")
        except Exception:
            traffic = None
    out = {
        "metric": ("queries/sec, 10M-doc 5-term BM25 top-1000" if args.workload == "C3" else f"queries/sec, {w.name}") + (" (packed postings)" if args.packed else ""),
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "p50_latency_ms": round(statistics.median(lat) * 1e3, 4),
        "p99_latency_ms": round(float(np.percentile(np.asarray(lat), 99)) * 1e3, 4),   # (of `steps` batch calls: the slowest one at --steps 20)
        "latency_samples": len(lat),
        "max_latency_ms": round(max(lat) * 1e3, 4), "slowest_step": int(np.argmax(np.asarray(lat))),   # (a stall on the host or the device shows up here)
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "setup": "index resident in HBM; every batch of the query set planned once before the warmup (per-term leaf tables resident)",
        "config": {
            "workload": w.name + (" [packed postings: 4 B per posting in HBM]" if args.packed else ""),
            "device_bytes_per_gpu": int(sum(l.device_bytes for l in leaves)),
            "n_docs": w.n_docs, "terms_per_query": w.n_terms, "k": w.k, "batch_queries": B,
            "total_hits_threshold": api.TOTAL_HITS_THRESHOLD,
            "segments_per_gpu": len(corpus.segments),
            "topology": {"doc_shards": D, "query_groups": R,
                         "note": "N = D x R: D GPUs share the index by docid range and exchange their top-k; R such groups hold the same index and "
                                 "take different batches (value counts all groups' queries; latencies are one group's batch calls)"},
            "sharding": "contiguous docid ranges, 1 process per GPU" + ((", RCCL all-to-all of per-GPU top-k, each rank merges its slice of the batch" if split_reduce else
                                                                           (", RCCL all-to-all of per-GPU top-k, each rank merges and delivers its slice of the batch" if (lib_collective and lib_mode == api.EXCHANGE_ALLTOALL)
                                                                            else ", RCCL all-gather of per-GPU top-k + merge on every rank")) if use_dist else "")
                        + (f" (exchange stage emulated: merge of {shard_world} lists for {'this rank' + chr(39) + 's slice of the batch' if lib_mode == api.EXCHANGE_ALLTOALL else 'every query'}; the other ranks' lists staged before the warmup, the transfer itself not in it)" if (use_dist and emu_exchange) else "")
                        + ((f" (collective inside the library: nrtgpu_dist_exchange_merge, {'all-to-all' if lib_mode == api.EXCHANGE_ALLTOALL else 'all-gather'})"
                            if lib_collective else ("" if emu_exchange else " (collective: torch.distributed)")) if use_dist else "")
                        + (", score-bound exchange between shards" if exchange_name else "")
                        + (", shard-level speculative thresholds (guesses at the whole search's k-th score, checked against the merged lists)"
                           if (shard_spec and (emu_exchange or lib_collective) and not args.sync_submit) else "")
                        + (f" (the other ranks' rows played by this process: --emulate-peers {args.emulate_peers})" if peer_words is not None else "")
                        + (f" [emulating rank {shard_rank} of {shard_world}]" if (world == 1 and shard_world > 1) else "")
                        + (f" [{R} replicas, no exchange]" if (world > 1 and D == 1) else (f" [{R} query groups of {D} doc shards]" if R > 1 else "")),
            "mean_postings_per_query": float(ppq.mean()),
            "scan_items_per_step": (st["scan_items"] + st["maxscore_items"]) / max(1, st["batches"]),
            "dynamic_pruning": not args.no_prune,
            "speculation": ctx.spec_counters(),   # speculative thresholds of the MaxScore route: queries run under them / run again
            "shard_speculation": (shard_spec_stat if (shard_spec and use_dist) else None),   # ... checked against the MERGED lists (timed steps + warmup)
            "prefetch": not args.no_prefetch, "planner_threads": planner_threads, "host_cpus": usable_cpus(),
            "host_threads": n_thr, "host_cpus_busy": round(host_cpu_busy, 2), "host_cpus_busy_by_thread_kind": cpu_by_kind,
            "corpus_build_s": round(t_build, 1), "corpus_variant": args.corpus_variant,
            "dist_stage_ms": ({k_: round(v / max(1, stage["steps"]) * 1e3, 3) for k_, v in stage.items() if k_ != "steps"}
                              if use_dist else None),   # per step on this rank: scan call (per scan thread), exchange, merge call
        },
        "roofline": roofline_record(kernel, pruned, scan_ms, bytes_per_launch, bpp, bpp_fused, traffic, traffic_note),
    }
    out["roofline"].update({
        "build_id": lib_id,   # sha256 over the gfx950 machine code of the loaded library (nrtsearch_amd/build.py: build_id)
        "traffic_refused": traffic_refused if traffic is None else None,
        "accumulators": "fixed-point u64" if (pruned or st.get("fixed_point_launches", 0) == st["scan_launches"]) else "fp64",
        "other_scorer_ms_per_step": round((st["scan_ms"] if pruned else st["maxscore_ms"]) / max(1, st["batches"]), 4),
        "merge_ms_per_step": round(st["merge_ms"] / max(1, st["batches"]), 4),
        "host_plan_ms_per_step": round(st["host_plan_ms"] / max(1, st["batches"]), 4),
    })
    if rank == 0 and world == 1 and not use_dist and not args.no_prune and args.exhaustive_steps > 0 and args.workload in ("C3", "C2"):
        # The north star's ">= 40 % of the HBM-read roofline on the postings scan": the EXHAUSTIVE route over the same batches
        # (NRTGPU_FLAG_NO_PRUNE: every posting of every query term is streamed, nothing skipped), timed here in the same run --
        # algorithmic bytes (9 B per posting) / the scan kernel's average launch (HIP events on the library's stream).
        ctx_x = api.GpuContext(device_id=local_rank, max_batch=B, target_items=args.target_items, collect_timing=True,
                               flags=flags | _lib.NRTGPU_FLAG_NO_PRUNE, host_threads=planner_threads)
        leaves_x = [api.GpuSegment.from_data(ctx_x, s) for s in corpus.segments]
        sr_x = api.GpuIndexSearcher(ctx_x, leaves_x, api.IndexStatistics.from_corpus(corpus))
        bx = [api.PreparedBatch(sr_x, queries[i: i + B], [mgr] * B) for i in range(0, min(n_distinct, 4 * B), B)]
        for i in range(2):
            bx[i % len(bx)].run()
        ctx_x.reset_stats()
        tx0 = time.perf_counter()
        for i in range(args.exhaustive_steps):
            bx[i % len(bx)].run()
        tx = time.perf_counter() - tx0
        sx = ctx_x.stats()
        lx = max(1, sx["scan_launches"])
        x_ms = sx["scan_ms"] / lx
        x_bytes = sx["scan_postings"] / lx * bpp
        x_rate = x_bytes / (x_ms * 1e-3) / 1e9 if x_ms > 0 else 0.0
        # what the kernel physically streams: 8 B per posting (docid + score code: the norm byte was folded into the code at seal,
        # so SURVEY 8d's ninth byte is never read); the PMC record of this build beside it when there is one
        x_phys = sx["scan_postings"] / lx * bpp_fused / (x_ms * 1e-3) / 1e9 if x_ms > 0 else 0.0
        x_traffic = None
        try:
            for rec in json.load(open(pmc)):
                if (rec.get("workload") == args.workload and rec.get("batch") == B and rec.get("kernel") == "bm25_scan_kernel"
                        and bool(rec.get("packed", False)) == bool(args.packed) and rec.get("build_id") == lib_id):
                    x_traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            x_traffic = None
        out["roofline"]["exhaustive"] = {
            "physical_bytes_per_launch": int(sx["scan_postings"] / lx * bpp_fused), "physical_achieved": round(x_phys, 1),
            "physical_frac": round(x_phys / HBM_PEAK_GBS, 4),
            "traffic": x_traffic, "traffic_frac": (round(x_traffic / (x_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (x_traffic and x_ms > 0) else None),
            "kernel": "bm25_scan_kernel", "steps": args.exhaustive_steps, "avg_launch_ms": round(x_ms, 4),
            "algorithmic_bytes_per_launch": int(x_bytes), "achieved": round(x_rate, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "frac": round(x_rate / HBM_PEAK_GBS, 4), "queries_per_s": round(args.exhaustive_steps * B / tx, 1),
            "note": "every posting of the queries' terms streamed (NRTGPU_FLAG_NO_PRUNE), same index and batches, same run; achieved = "
                    "9 B x postings / the kernel's average launch; physical_* = the 8 B per posting the kernel streams (norm folded into "
                    "the code at seal) / the same launch time; traffic = PMC FETCH_SIZE x 2 (profiles/pmc_traffic.json) when the record is this build's",
        }
        for l in leaves_x:
            l.release()
        ctx_x.close()
    if rank == 0 and world == 1 and not use_dist and args.c2_steps > 0 and args.workload == "C3" and not args.docs and not args.no_prune:
        out["roofline"]["c2"] = c2_leg(args, local_rank, flags, planner_threads)
    if rank == 0 and world == 1 and not use_dist and args.c5_steps > 0 and args.workload == "C3" and not args.docs and not args.no_prune and not args.packed:
        out["roofline"]["c5"] = c5_leg(args, local_rank, planner_threads)
    if rank == 0 and world == 1 and not use_dist and args.c4_steps > 0 and args.workload == "C3" and not args.docs:
        # The other half of the path in the same line: BASELINE config 4 -- 10 M x 768 fp32 rows, exact
        # cosine top-100, 64 queries per pass -- for --c4-steps passes: the sketch kernel's physical HBM fraction, the matrix cores'
        # rate against the fp16 peak, the fp64 check over every row, and the C port on the host cores.  (bench.py --workload C4 is
        # the full line: closed loop, other panel widths.)
        c4_args = argparse.Namespace(**vars(args))
        c4_args.workload, c4_args.steps, c4_args.warmup, c4_args.knn_queries, c4_args.closed_loop = "C4", args.c4_steps, 2, 64, ""
        c4_args.no_sketch, c4_args.c4_callers, c4_args.no_verify, c4_args.emulate_world = False, False, False, 0
        c4 = run_c4(c4_args, emit=False)
        r4 = c4["roofline"]
        out["roofline"]["c4"] = {
            "workload": c4["config"]["workload"], "kernel": r4["kernel"], "passes": args.c4_steps, "queries_per_pass": 64,
            "queries_per_s": c4["value"], "ms_per_pass_call": c4["ms_per_step"], "avg_launch_ms": r4["avg_launch_ms"],
            "bound": "hbm", "achieved": r4["achieved"], "peak": r4["peak"], "unit": "GB/s", "frac": r4["frac"], "achieved_is": r4["achieved_is"],
            "streamed_bytes_per_launch": r4["streamed_bytes_per_launch"], "algorithmic_bytes_per_launch": r4["algorithmic_bytes_per_launch"],
            "effective_frac": r4["effective_frac"], "traffic": r4["traffic"], "traffic_source": r4["traffic_source"],
            "mfma_tflops": r4["mfma_tflops"], "mfma_peak_tflops": r4["mfma_peak_tflops"], "mfma_dtype": r4["mfma_dtype"], "mfma_frac": r4["mfma_frac"],
            "second_passes": r4["second_passes"], "verify": c4.get("verify"), "cpu_baseline": c4.get("cpu_baseline"),
            "two_callers": c4.get("two_callers"),   # the same passes with two callers taking turns: staging overlapped with the other's stream
            "corpus_build_s": c4["config"]["corpus_build_s"],
        }
    if rank == 0 and world == 1 and not use_dist and args.closed_loop and args.workload in ("C3", "C2"):
        # queries/s AND latency (BASELINE.json's metric): the closed loop of SURVEY 8d, same index, same query set
        callers = [int(x) for x in args.closed_loop.split(",") if x.strip()]
        out["closed_loop"] = closed_loop(ctx, searcher, queries, mgr, callers, args.closed_loop_ms)
        out["closed_loop"]["entry"] = "nrtgpu_search_bm25_coalesced, one query per call, native caller threads (bench/loadgen)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_queries > 0:
        out["cpu_baseline"] = cpu_baseline(corpus, qranks, w.k, args.cpu_queries)
        luc = lucene_baseline(w, searcher, queries, mgr, min(256, n_distinct))
        out["cpu_baseline"]["lucene"] = luc
        if luc.get("available"):   # the reference itself ran here: it is the baseline, the C port stays beside it
            cb = out["cpu_baseline"]
            cb["port_value"] = cb["value"]
            cb["value"] = luc[f"threads_{cb['cores']}"]["queries_per_s"]
            cb["kind"] = "reference"
            cb["sample"] = (f"first {luc['queries']} queries through JVM Lucene {luc['lucene']} (bench/lucene/LuceneBaseline.java), "
                            f"{cb['cores']} threads; C port on the same box: {cb['port_value']} queries/s")
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)   # whatever the runtimes still print at teardown (C stdio) goes to stderr: stdout stays the one JSON line
    if lib_collective_hung:   # a thread is stuck inside the collective library: no orderly teardown
        if world > 1:
            dist.barrier()
        os._exit(0)
    if world == 1 and exchange_name:
        ctx.exchange_close()
        os.unlink("/dev/shm" + exchange_name)
    if world > 1:
        dist.barrier()
        if exchange_name:
            ctx.exchange_close()
            if rank == 0 and os.path.exists("/dev/shm" + exchange_name):
                os.unlink("/dev/shm" + exchange_name)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
