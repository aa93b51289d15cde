#!/usr/bin/env python3
"""Dump a synthetic corpus + query set (nrtsearch_amd/synth.py) in the little-endian layout
bench/lucene/LuceneBaseline.java reads, so that JVM Lucene indexes exactly the postings / lengths / docids the
device searches.  usage: dump_corpus.py <workload C2|C3|SMOKE> <out-dir> [n_queries]
       dump_corpus.py --fixture <out-dir>     the 20 k-doc fixture + query shapes bench/lucene/LuceneGolden.java turns into
                                              tests/golden/lucene_shapes.json (scripts/make_lucene_goldens.sh)"""
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


# ---- the fixture of the JVM goldens: every score structure SURVEY 8(c) lists as [Lucene-recall] only ----
FIXTURE = workload.Workload("lucene golden fixture: 20k docs", 20_000, 4, 50, 8, 3, max_rank=400)
FIXTURE_RANKS = [1, 2, 5, 9, 20, 45, 90, 200, 380]


def fixture_filter(n_docs):
    """The doc set of the FILTER / MUST_NOT clause ("flt:1"): every third doc and a run of 700 docs."""
    m = (np.arange(n_docs) % 3 == 0)
    m[5000:5700] = True
    return m


def fixture_shapes():
    """(shape, k, totalHitsThreshold, param, term ids): one line of shapes.txt each.  Thresholds on both sides of the hit counts, so
    that EQUAL_TO and GREATER_THAN_OR_EQUAL_TO relations both occur; lengths above 40 tokens are the corpus's own (norm bytes the
    reference's fixtures never reach)."""
    r = FIXTURE_RANKS
    out = []
    for k, thr in ((10, 1000), (50, 20), (50, 2**31 - 1)):
        out += [("should", k, thr, 0.0, [r[4]]), ("should", k, thr, 0.0, [r[0], r[5]]), ("should", k, thr, 0.0, [r[1], r[3], r[6], r[8]]),
                ("should", k, thr, 0.0, [r[0], r[1], r[2], r[4], r[7]]),
                ("dismax", k, thr, 0.0, [r[1], r[4], r[7]]), ("dismax", k, thr, 0.3, [r[1], r[4], r[7]]), ("dismax", k, thr, 1.0, [r[2], r[5]]),
                ("must_should", k, thr, 1, [r[5], r[1], r[3]]), ("must_should", k, thr, 2, [r[2], r[4], r[0], r[6]]),
                ("must_should", k, thr, 3, [r[0], r[1], r[3]]),
                ("msm", k, thr, 2, [r[0], r[2], r[4], r[6]]), ("msm", k, thr, 3, [r[0], r[1], r[2], r[3], r[5]]),
                ("filter", k, thr, 0.0, [r[1], r[5], r[7]]), ("must_not", k, thr, 0.0, [r[0], r[4], r[8]]),
                ("boost", k, thr, 2.5, [r[3], r[1]])]
    return out


def dump_fixture(out_dir):
    """The fixture in LuceneBaseline's dump layout + filter.u8 + shapes.txt (bench/lucene/LuceneGolden.java)."""
    w = FIXTURE
    os.makedirs(out_dir, exist_ok=True)
    lens = synth.doc_lengths(w.n_docs)
    sizes = synth.tiered_segment_sizes(w.n_docs, w.segments_per_shard)
    offs = [0]
    with open(os.path.join(out_dir, "docids.i32"), "wb") as fd, open(os.path.join(out_dir, "freqs.i32"), "wb") as ff:
        for r in FIXTURE_RANKS:
            d, f = synth.term_postings(w.n_docs, r)
            d.astype("<i4").tofile(fd)
            f.astype("<i4").tofile(ff)
            offs.append(offs[-1] + len(d))
    lens.astype("<i4").tofile(os.path.join(out_dir, "lengths.i32"))
    np.asarray(FIXTURE_RANKS, dtype="<i8").tofile(os.path.join(out_dir, "terms.i64"))
    np.asarray(offs, dtype="<i8").tofile(os.path.join(out_dir, "offsets.i64"))
    np.zeros((0,), dtype="<i8").tofile(os.path.join(out_dir, "queries.i64"))
    fixture_filter(w.n_docs).astype(np.uint8).tofile(os.path.join(out_dir, "filter.u8"))
    shapes = fixture_shapes()
    with open(os.path.join(out_dir, "shapes.txt"), "w") as f:
        f.write("# shape k totalHitsThreshold param n_terms term ids (scripts/dump_corpus.py: fixture_shapes)\n")
        for shape, k, thr, param, terms in shapes:
            f.write(f"{shape} {k} {thr} {param} {len(terms)} " + " ".join(str(t) for t in terms) + "\n")
    with open(os.path.join(out_dir, "meta.txt"), "w") as f:
        f.write(f"{w.n_docs} {len(FIXTURE_RANKS)} 0 {w.n_terms} {w.k} {len(sizes)}\n" + " ".join(str(s_) for s_ in sizes) + "\n")
    return shapes


def fixture_corpus():
    """The same fixture as the oracle / the device see it: (corpus, per-segment filter words (uint64, Lucene's FixedBitSet layout))."""
    w = FIXTURE
    corpus = synth.build_corpus(w.n_docs, FIXTURE_RANKS, n_segments=w.segments_per_shard)
    m = fixture_filter(w.n_docs)
    words = []
    for seg in corpus.segments:
        bits = m[seg.doc_base: seg.doc_base + seg.max_doc]
        pad = (-len(bits)) % 64
        words.append(np.packbits(np.concatenate([bits, np.zeros(pad, dtype=bool)]).astype(np.uint8), bitorder="little").view(np.uint64).copy())
    return corpus, words


if __name__ == "__main__":
    if sys.argv[1] == "--fixture":
        dump_fixture(sys.argv[2])
        print("dumped the golden fixture to", sys.argv[2])
        sys.exit(0)
    w = getattr(workload, sys.argv[1])
    dump(w, sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1024)
    print("dumped", w.name, "to", sys.argv[2])
