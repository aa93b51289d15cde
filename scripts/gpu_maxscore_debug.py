import os
#!/usr/bin/env python3
"""Debug aid for the MaxScore route: pruned vs exhaustive vs oracle on the tests' mid-size corpus, with a diff of
what is missing (rank, score, segment, tile)."""
import json, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NRTGPU_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nrtsearch_amd", "libnrtgpu_dev.so"))   # instrumented kernels: the development library (include/nrtgpu_dev.h)
from nrtsearch_amd import _lib, api, synth
from oracle import oracle

def bq(terms):
    cl = tuple(api.TermQuery(0, int(t)) for t in terms)
    return cl[0] if len(cl) == 1 else api.BooleanQuery(cl)

def main():
    oracle.build()
    ranks = [1, 2, 3, 5, 8, 13, 40, 100, 333, 1000, 5000, 9999]
    corpus = synth.build_corpus(300_000, ranks, n_segments=4, delete_fraction=0.02)
    bases = [s.doc_base for s in corpus.segments]
    shapes = [[1], [100], [1, 2], [1, 9999], [5, 40, 1000], [1, 2, 3, 5, 8], [13, 40, 100, 333, 1000], [1, 100, 1000, 5000, 9999]]
    INT_MAX = 2**31 - 1
    SHAPES = [[1], [100], [5000], [1, 2], [1, 9999], [333, 1000], [5, 40, 1000], [1, 2, 3, 5, 8], [13, 40, 100, 333, 1000],
              [1, 100, 1000, 5000, 9999], [2, 3, 5000], [1, 2, 3, 5, 8, 13, 40, 100, 333, 1000, 5000, 9999], [8, 8, 40],
              [9999, 5000], [3, 13, 333, 9999]]
    combos = [(10, 1000), (100, 1000), (1000, 1000), (1, 0), (37, 50), (100, INT_MAX)]
    if len(sys.argv) > 1:
        combos = [combos[int(x)] for x in sys.argv[1].split(",")]
    ctx = api.GpuContext(0, max_batch=1024)
    leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
    sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
    qs, mg, meta = [], [], []
    for terms in SHAPES:
        for k, thr in combos:
            qs.append(bq(terms)); mg.append(api.TopScoreDocCollectorManager(k, total_hits_threshold=thr)); meta.append((terms, k, thr))
    got = sr.search_batch(qs, mg)
    nbad = 0
    for (terms, k, thr), g in zip(meta, got):
        d, s, tot, gte = oracle.search_bm25(corpus, terms, k, total_hits_threshold=thr)
        ok = g.docs.tolist() == d.tolist() and g.scores.view(np.uint32).tolist() == s.view(np.uint32).tolist()
        if not ok:
            nbad += 1
            gs = set(g.docs.tolist())
            miss = [x for x in d.tolist() if x not in gs]
            info = []
            for x in miss[:8]:
                r = d.tolist().index(x)
                si = max(i for i, b in enumerate(bases) if b <= x)
                info.append(dict(doc=x, rank=r, score=float(s[r]), seg=si, tile=(x - bases[si]) // 1024))
            print(json.dumps(dict(test_batch=True, terms=terms, k=k, thr=thr, n_got=len(g.docs), n_missing=len(miss), missing=info,
                                  kth=float(s[-1]) if len(s) else None, got_gte=g.relation_gte, got_total=g.total_hits, exp_total=tot)))
    print(json.dumps(dict(test_batch=True, queries=len(qs), bad=nbad, stats=ctx.stats())))
    for l in leaves: l.release()
    ctx.close()
    for flags, label in []:
        ctx = api.GpuContext(0, max_batch=1024, flags=flags | (_lib.NRTGPU_FLAG_PROFILE if flags == 0 else 0))
        leaves = [api.GpuSegment.from_data(ctx, s) for s in corpus.segments]
        sr = api.GpuIndexSearcher(ctx, leaves, api.IndexStatistics.from_corpus(corpus))
        for mode in ("single", "batch"):
            for k in (10, 1000):
                qs = [bq(t) for t in shapes]
                mg = [api.TopScoreDocCollectorManager(k)] * len(qs)
                ctx.reset_stats()
                t0 = time.time()
                got = [sr.search(q, m) for q, m in zip(qs, mg)] if mode == "single" else sr.search_batch(qs * 8, mg * 8)[:len(qs)]
                dt = time.time() - t0
                for terms, g in zip(shapes, got):
                    d, s, tot, gte = oracle.search_bm25(corpus, terms, k)
                    ok = g.docs.tolist() == d.tolist() and g.scores.view(np.uint32).tolist() == s.view(np.uint32).tolist()
                    if not ok:
                        miss = [x for x in d.tolist() if x not in set(g.docs.tolist())]
                        extra = [x for x in g.docs.tolist() if x not in set(d.tolist())]
                        info = []
                        for x in miss[:6]:
                            r = d.tolist().index(x)
                            si = max(i for i, b in enumerate(bases) if b <= x)
                            info.append(dict(doc=x, rank=r, score=float(s[r]), seg=si, local=x - bases[si], tile=(x - bases[si]) // 1024))
                        print(json.dumps(dict(label=label, mode=mode, k=k, terms=terms, n_got=len(g.docs), n_exp=len(d), n_missing=len(miss),
                                              n_extra=len(extra), missing=info, kth=float(s[-1]), got_total=g.total_hits, exp_total=tot)))
                print(json.dumps(dict(label=label, mode=mode, k=k, secs=round(dt, 3), stats={k_: v for k_, v in ctx.stats().items() if "maxscore" in k_ or k_ in ("scan_items",)},
                                      prof=ctx.maxscore_profile() if flags == 0 else None)))
        for l in leaves: l.release()
        ctx.close()
main()
