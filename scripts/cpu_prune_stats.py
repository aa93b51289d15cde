#!/usr/bin/env python3
"""CPU study behind DESIGN.md's MaxScore section: how much of a C3 query is left to stream once theta is known.
For each query: exhaustive scores (numpy, fp64 -- statistics only), theta = k-th best; clauses sorted by upper bound
(= weight); non-essential = longest prefix with sum < theta (optionally only clauses dense enough for a direct map);
reports essential postings, docs surviving `partial + ne_ub >= theta`, and the sub-tile occupancy of the essential walk."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from nrtsearch_amd import synth, workload, api

def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dense_div = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    w = workload.C3
    N = w.n_docs
    qr = synth.make_queries(nq, w.n_terms, w.max_rank)
    lens = synth.doc_lengths(N)
    norms = synth.int_to_byte4(lens)
    stt = int(lens.astype(np.int64).sum())
    avgdl = np.float32(stt / N)
    # length table
    from oracle import oracle
    oracle.build()
    cache = api.BM25Similarity().norm_cache(api.CollectionStatistics(N, stt)) if False else None
    import ctypes as C
    from nrtsearch_amd import _lib
    cache = np.zeros(256, dtype=np.float32)
    _lib.load().nrtgpu_bm25_norm_cache(C.c_float(float(avgdl)), C.c_float(1.2), C.c_float(0.75), cache.ctypes.data)
    rows = []
    post = {}
    for r in sorted(set(int(x) for x in qr.reshape(-1))):
        post[r] = synth.term_postings(N, r)
    tot = dict(P=0, ess=0, surv=0, look=0, tiles_ess=0, tiles=0, pairs=0, ne_terms=0)
    for qi in range(nq):
        terms = [int(x) for x in qr[qi]]
        acc = np.zeros(N, dtype=np.float64)
        ws = {}
        sc = {}
        for r in terms:
            d, f = post[r]
            idf = np.float32(np.log(1 + (N - len(d) + 0.5) / (len(d) + 0.5)))
            ws[r] = float(idf)
            ninv = cache[norms[d]]
            s = idf - idf / (np.float32(1) + f.astype(np.float32) * ninv)
            sc[r] = s.astype(np.float64)
            np.add.at(acc, d, sc[r])
        P = sum(len(post[r][0]) for r in terms)
        nz = acc[acc > 0]
        theta = np.partition(nz, len(nz) - w.k)[len(nz) - w.k] if len(nz) > w.k else 0.0
        order = sorted(terms, key=lambda r: ws[r])
        ne, s = [], 0.0
        for r in order:
            if len(post[r][0]) * dense_div < N: continue   # not dense enough for a direct map: stays essential
            if s + ws[r] < theta:
                s += ws[r]; ne.append(r)
        ess = [r for r in terms if r not in ne]
        pe = sum(len(post[r][0]) for r in ess)
        part = np.zeros(N, dtype=np.float64)
        for r in ess: np.add.at(part, post[r][0], sc[r])
        edocs = part > 0
        surv = int(((part + s >= theta) & edocs).sum())
        tiles = (N + 1023) // 1024
        tcount = np.zeros(tiles, dtype=np.int64)
        pairs = np.zeros(tiles, dtype=np.int64)
        for r in ess:
            c = np.bincount(post[r][0] >> 10, minlength=tiles)
            tcount += c
            pairs += (c + 7 + 3) // 8   # ~ pairs incl. misalignment
        tot["P"] += P; tot["ess"] += pe; tot["surv"] += surv; tot["look"] += surv * len(ne)
        tot["tiles_ess"] += int((tcount > 0).sum()); tot["tiles"] += tiles; tot["pairs"] += int(pairs.sum()); tot["ne_terms"] += len(ne)
        rows.append((terms, round(float(theta), 3), [round(ws[r], 2) for r in order], ne, P, pe, int(edocs.sum()), surv))
    for r in rows[:12]: print(r)
    print({k: v / nq for k, v in tot.items()})
    print("essential fraction of postings", tot["ess"] / tot["P"], "survivor lookups per query", tot["look"] / nq,
          "tiles with essential postings", tot["tiles_ess"] / tot["tiles"], "pairs per tile", tot["pairs"] / tot["tiles"])

main()
