#!/usr/bin/env python3
"""Dump a synthetic corpus + query set (nrtsearch_amd/synth.py) in the little-endian layout
bench/lucene/LuceneBaseline.java reads, so that JVM Lucene indexes exactly the postings / lengths / docids the
device searches.  usage: dump_corpus.py <workload C2|C3|SMOKE> <out-dir> [n_queries]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nrtsearch_amd import synth, workload  # noqa: E402


def dump(w, out_dir, n_queries):
    os.makedirs(out_dir, exist_ok=True)
    qr = synth.make_queries(n_queries, w.n_terms, w.max_rank)
    ranks = sorted(set(int(r) for r in qr.reshape(-1)))
    lens = synth.doc_lengths(w.n_docs)
    sizes = synth.tiered_segment_sizes(w.n_docs, w.segments_per_shard)
    offs = [0]
    with open(os.path.join(out_dir, "docids.i32"), "wb") as fd, open(os.path.join(out_dir, "freqs.i32"), "wb") as ff:
        for r in ranks:
            d, f = synth.term_postings(w.n_docs, r)
            d.astype("<i4").tofile(fd)
            f.astype("<i4").tofile(ff)
            offs.append(offs[-1] + len(d))
    lens.astype("<i4").tofile(os.path.join(out_dir, "lengths.i32"))
    np.asarray(ranks, dtype="<i8").tofile(os.path.join(out_dir, "terms.i64"))
    np.asarray(offs, dtype="<i8").tofile(os.path.join(out_dir, "offsets.i64"))
    qr.astype("<i8").tofile(os.path.join(out_dir, "queries.i64"))
    with open(os.path.join(out_dir, "meta.txt"), "w") as f:
        f.write(f"{w.n_docs} {len(ranks)} {n_queries} {w.n_terms} {w.k} {len(sizes)}\n" + " ".join(str(s) for s in sizes) + "\n")
    return qr


if __name__ == "__main__":
    w = getattr(workload, sys.argv[1])
    dump(w, sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1024)
    print("dumped", w.name, "to", sys.argv[2])
