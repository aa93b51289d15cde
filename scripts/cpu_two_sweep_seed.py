#!/usr/bin/env python3
"""CPU study for DESIGN.md §8 item 2 (no GPU): what theta would a first sweep over the RAREST clause(s) of a C3 query give, and how
much of the query is left to stream under it?  Per query (numpy, fp64 -- statistics only, not a parity tool):
  theta_final  = k-th best score over all docs;
  theta_1 / _2 = k-th best COMPLETE score among the docs that hold the rarest / one of the two rarest clauses (what the kernel
                 would know after sweeping those clauses over every window of the item, completing the docs by lookups);
  in_top       = share of the final top-k that such a sweep has already evaluated;
  ess(theta)   = share of the query's postings in clauses that are essential under theta (suffix sums of the clause bounds, as
                 bm25_maxscore_kernel decides it; bound of a clause = its weight, the score's supremum).
    python scripts/cpu_two_sweep_seed.py [n_queries=16] [workload=C3]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nrtsearch_amd import _lib, synth, workload   # noqa: E402


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    w = getattr(workload, sys.argv[2] if len(sys.argv) > 2 else "C3")
    N, k = w.n_docs, w.k
    qr = synth.make_queries(nq, w.n_terms, w.max_rank)
    lens = synth.doc_lengths(N)
    norms = synth.int_to_byte4(lens)
    avgdl = np.float32(int(lens.astype(np.int64).sum()) / N)
    import ctypes as C
    cache = np.zeros(256, dtype=np.float32)
    _lib.load().nrtgpu_bm25_norm_cache(C.c_float(float(avgdl)), C.c_float(1.2), C.c_float(0.75), cache.ctypes.data)
    post = {r: synth.term_postings(N, r) for r in sorted(set(int(x) for x in qr.reshape(-1)))}

    def essential_share(theta, terms, ws, df):
        order = sorted(terms, key=lambda r: -ws[r])          # heaviest (rarest) first, as the kernel orders them
        suf, ess = 0.0, 0
        sufs = []
        for r in reversed(order):
            suf += ws[r]
            sufs.append(suf)
        sufs = sufs[::-1]                                     # S_j
        for j, r in enumerate(order):
            if sufs[j] >= theta:
                ess += df[r]
        return ess / sum(df.values())

    rows = []
    for qi in range(nq):
        terms = [int(x) for x in qr[qi]]
        acc = np.zeros(N, dtype=np.float64)
        ws, df = {}, {}
        for r in terms:
            d, f = post[r]
            idf = np.float32(np.log(1 + (N - len(d) + 0.5) / (len(d) + 0.5)))
            ws[r], df[r] = float(idf), len(d)
            s = idf - idf / (np.float32(1) + f.astype(np.float32) * cache[norms[d]])
            np.add.at(acc, d, s.astype(np.float64))

        def kth(scores):
            nz = scores[scores > 0]
            return float(np.partition(nz, len(nz) - k)[len(nz) - k]) if len(nz) >= k else 0.0

        theta = kth(acc)
        order = sorted(terms, key=lambda r: -ws[r])
        top = np.argpartition(acc, N - k)[N - k:]
        out = {"df": [df[r] for r in order], "theta": round(theta, 3)}
        for n_first in (1, 2):
            seen = np.zeros(N, dtype=bool)
            for r in order[:n_first]:
                seen[post[r][0]] = True
            th_n = kth(np.where(seen, acc, 0.0))
            out[f"theta_{n_first}/theta"] = round(th_n / theta, 3) if theta > 0 else None
            out[f"in_top_{n_first}"] = round(float(seen[top].mean()), 3)
            out[f"ess_{n_first}"] = round(essential_share(th_n, terms, ws, df), 4)
            out[f"sweep_postings_{n_first}"] = round(sum(df[r] for r in order[:n_first]) / sum(df.values()), 4)
        out["ess_final"] = round(essential_share(theta, terms, ws, df), 4)
        rows.append(out)
        print(out, flush=True)
    keys = [k_ for k_ in rows[0] if k_ not in ("df",)]
    print("MEAN", {k_: round(float(np.mean([r[k_] for r in rows if r[k_] is not None])), 4) for k_ in keys})


main()
