#!/usr/bin/env python3
"""kNN exact search timing, config C4's shape: N x 768 fp32 rows (KNN_N, default 2M; C4 = 10M = 30.7 GB), cosine,
k = 100, rows spread over segments of <= KNN_SEG rows (default 2.5M; generated one segment at a time to bound
host memory)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nrtsearch_amd import api

n, dim = int(os.environ.get("KNN_N", 2_000_000)), 768
seg_rows = int(os.environ.get("KNN_SEG", 2_500_000))
rng = np.random.Generator(np.random.PCG64(777))
ctx = api.GpuContext(0, 64)
t0 = time.time()
leaves, base = [], 0
while base < n:
    rows = min(seg_rows, n - base)
    g = api.GpuSegment(ctx, rows, base)
    g.add_vectors(0, rng.standard_normal((rows, dim), dtype=np.float32))
    g.seal()
    leaves.append(g)
    base += rows
sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics())
queries = np.random.Generator(np.random.PCG64(778)).standard_normal((64, dim), dtype=np.float32)
print(json.dumps({"event": "upload", "n": n, "segments": len(leaves), "gen_upload_s": round(time.time() - t0, 1)}), flush=True)
for nq in (1, 8, 32, 64):
    sr.knn_exact(0, "cosine", queries[:nq], 100)
    t1 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        sr.knn_exact(0, "cosine", queries[:nq], 100)
    dt = (time.perf_counter() - t1) / reps
    passes = (nq + 31) // 32
    print(json.dumps({"event": "knn", "n": n, "queries": nq, "ms": round(dt * 1e3, 2), "qps": round(nq / dt, 1),
                      "GBps_alg": round(n * dim * 4 * passes / dt / 1e9, 1), "tflops": round(2.0 * n * dim * nq / dt / 1e12, 2)}), flush=True)
