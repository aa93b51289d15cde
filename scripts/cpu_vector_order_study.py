#!/usr/bin/env python3
"""What does "the oracle's bits" mean for exact vector search?  (CPU, no GPU)

The device returns, bit for bit, the scores of oracle/nrt_oracle.c's PINNED summation order: scalar, left to right, one
accumulator.  No Lucene build computes exactly that: its default DefaultVectorUtilSupport strides the dimension with four
accumulators (two for cosine), with or without fused multiply-add, and Panama builds use the SIMD width -- [Lucene-recall], not
in /root/reference, not runnable here.  All of these are fp32 evaluations of the same sums, so they agree to ~dim x 2^-24
relative; this script measures what that means at the C4 shape (768-d rows ~ Normal(0,1), PCG64(777); queries PCG64(778);
top-100): between the pinned order (0) and the restated Lucene orders (1: unrolled, 2: unrolled + fma)
  * how many of the top-100 DOCIDS differ as a set, and at how many ranks the docid differs,
  * the largest |score difference| among the hits and relative to the score.
    python scripts/cpu_vector_order_study.py [--rows 1000000] [--queries 16]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=16)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    oracle.build()
    rows = np.random.Generator(np.random.PCG64(777)).standard_normal((args.rows, args.dim), dtype=np.float32)
    queries = np.random.Generator(np.random.PCG64(778)).standard_normal((args.queries, args.dim), dtype=np.float32)
    unit_rows = rows / np.linalg.norm(rows, axis=1, keepdims=True).astype(np.float32)
    unit_q = queries / np.linalg.norm(queries, axis=1, keepdims=True).astype(np.float32)
    for name, sim, r, q in (("cosine", 0, rows, queries), ("dot_product (unit vectors)", 1, unit_rows, unit_q), ("l2_norm", 2, rows, queries),
                            ("max_inner_product", 3, rows, queries)):
        res = {}
        for order in (0, 1, 2):
            t0 = time.time()
            res[order] = oracle.knn_exact(sim, q, r, args.k, n_threads=args.threads, order=order)
            res[order] = res[order] + (time.time() - t0,)
        d0, s0 = res[0][0], res[0][1]
        for order in (1, 2):
            d1, s1 = res[order][0], res[order][1]
            set_diff = [len(set(d0[i].tolist()) ^ set(d1[i].tolist())) // 2 for i in range(args.queries)]
            rank_diff = [int((d0[i] != d1[i]).sum()) for i in range(args.queries)]
            # score differences of the docs both lists hold
            dmax, rmax = 0.0, 0.0
            for i in range(args.queries):
                m1 = {int(d): float(s) for d, s in zip(d1[i], s1[i])}
                for d, s in zip(d0[i], s0[i]):
                    if int(d) in m1:
                        dmax = max(dmax, abs(float(s) - m1[int(d)]))
                        rmax = max(rmax, abs(float(s) - m1[int(d)]) / max(abs(float(s)), 1e-30))
            print(json.dumps({"similarity": name, "rows": args.rows, "dim": args.dim, "queries": args.queries, "k": args.k,
                              "pinned_vs": {1: "lucene default, unrolled (recall)", 2: "lucene default, unrolled + fma (recall)"}[order],
                              "queries_whose_top_k_set_differs": int(sum(1 for x in set_diff if x)), "docids_swapped_in_or_out_total": int(sum(set_diff)),
                              "ranks_with_another_docid_total": int(sum(rank_diff)), "of_ranks": args.queries * args.k,
                              "max_abs_score_diff": dmax, "max_rel_score_diff": rmax,
                              "cpu_s": [round(res[0][3], 1), round(res[order][3], 1)]}), flush=True)


if __name__ == "__main__":
    main()
