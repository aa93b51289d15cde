"""Deterministic synthetic Zipf-term corpora and query sets.

This is workload tooling shared by tests and bench.py; it produces the *inputs* of the hot path
(what Lucene's PostingsEnum / NumericDocValues norms / FloatVectorValues would hand over at
segment-upload time), never results.

Corpus definition (one TEXT field):
  * doc lengths  len_d = clip(round(exp(Normal(ln 80, 0.6))), 4, 4000), PCG64(1234);
    norm byte = SmallFloat.intToByte4(len_d); sumTotalTermFreq = sum(len_d); docCount = N.
  * term of Zipf rank r >= 1 occurs in each doc independently with p = 1/(r+1) (so
    E[df] = N/(r+1), the s=1 Zipf profile of SURVEY 8d); realised by geometric gap sampling with
    PCG64(1234 ^ r) -- O(df) per term instead of an O(N) permutation -- and the realised df is what
    idf uses.  freq = min(255, Geometric(0.55)) from the same generator.
  * queries: n distinct ranks drawn log-uniformly from [1, max_rank], PCG64(4321).
  * segments: contiguous docid ranges following a tiered-merge profile (1/2, 1/4, 1/8, ...).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

FIELD_TEXT = 0

# ---- SmallFloat.intToByte4, vectorised (org.apache.lucene.util.SmallFloat) -------
_NUM_FREE_VALUES = 24


def int_to_byte4(lengths: np.ndarray) -> np.ndarray:
    """Vectorised SmallFloat.intToByte4 for non-negative int lengths -> uint8 norm bytes."""
    v = np.asarray(lengths, dtype=np.int64)
    out = v.copy()
    big = v >= _NUM_FREE_VALUES
    x = v[big] - _NUM_FREE_VALUES
    num_bits = np.zeros_like(x)
    nz = x > 0
    # floor(log2(x)) + 1 exactly for x < 2**53
    num_bits[nz] = np.floor(np.log2(x[nz].astype(np.float64))).astype(np.int64) + 1
    # guard against log2 rounding at exact powers of two
    num_bits = np.where((np.int64(1) << np.maximum(num_bits - 1, 0)) > x, num_bits - 1, num_bits)
    num_bits = np.where((np.int64(1) << num_bits) <= x, num_bits + 1, num_bits)
    sub = num_bits < 4
    shift = np.maximum(num_bits - 4, 0)
    enc = ((x >> shift) & 0x07) | ((shift + 1) << 3)
    enc = np.where(sub, x, enc)
    out[big] = _NUM_FREE_VALUES + enc
    return out.astype(np.uint8)


@dataclasses.dataclass
class SegmentData:
    """Columnar scoring data of one immutable segment (what gets uploaded to HBM)."""

    max_doc: int
    doc_base: int
    norms: np.ndarray                      # uint8[max_doc]
    term_ids: np.ndarray                   # int64[n_terms]   (term "hash" = Zipf rank here)
    offsets: np.ndarray                    # int64[n_terms+1]
    docids: np.ndarray                     # int32[P]  segment-local, ascending per term
    freqs: np.ndarray                      # int32[P]
    live_bits: Optional[np.ndarray] = None  # uint64[ceil(max_doc/64)] or None (all live)

    def postings(self, term_id: int) -> Tuple[np.ndarray, np.ndarray]:
        idx = np.searchsorted(self.term_ids, term_id)
        if idx >= len(self.term_ids) or self.term_ids[idx] != term_id:
            return self.docids[:0], self.freqs[:0]
        lo, hi = int(self.offsets[idx]), int(self.offsets[idx + 1])
        return self.docids[lo:hi], self.freqs[lo:hi]


@dataclasses.dataclass
class Corpus:
    n_docs: int
    doc_count: int                 # docs that have the field (== n_docs)
    sum_total_term_freq: int
    segments: List[SegmentData]
    doc_freq: Dict[int, int]       # index-global docFreq per term id (incl. deleted docs, as Lucene)

    @property
    def total_postings(self) -> int:
        return int(sum(len(s.docids) for s in self.segments))


def doc_lengths(n_docs: int, seed: int = 1234) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    ln = np.exp(rng.normal(np.log(80.0), 0.6, size=n_docs))
    return np.clip(np.rint(ln), 4, 4000).astype(np.int32)


def term_postings(n_docs: int, rank: int, seed: int = 1234) -> Tuple[np.ndarray, np.ndarray]:
    """Global (docids, freqs) of the term with Zipf rank `rank` (>= 1)."""
    p = 1.0 / (rank + 1.0)
    rng = np.random.Generator(np.random.PCG64(seed ^ rank))
    exp = n_docs * p
    m = int(exp + 6.0 * np.sqrt(exp) + 16)
    gaps = rng.geometric(p, size=m).astype(np.int64)
    pos = np.cumsum(gaps) - 1
    while pos[-1] < n_docs:  # practically never
        more = rng.geometric(p, size=m).astype(np.int64)
        pos = np.concatenate([pos, pos[-1] + np.cumsum(more)])
    docids = pos[pos < n_docs].astype(np.int32)
    freqs = np.minimum(255, rng.geometric(0.55, size=len(docids))).astype(np.int32)
    return docids, freqs


# ---- corpora whose docids are NOT independent draws --------------------------------------------------------------
# SURVEY 8d's corpus draws every posting list uniformly over [0, N): a workgroup's doc windows are then a random sample of the
# query's docs BY CONSTRUCTION, which is what the MaxScore route's speculative thresholds assume (maxscore.hip: ms_compact).  Real
# Lucene docids are insertion-ordered: terms come in bursts (time-clustered vocabulary), and an index may be sorted by something
# the score follows.  Two variants for that; statistics (df, doc lengths) keep the same profile:
#   "clustered": the docid axis is cut into CLUSTER_EPOCHS epochs; a term's density in an epoch is p x a weight drawn from a
#                heavy-tailed law (Gamma(0.35), mean 1, per term) -- a few epochs hold most of a term's postings.
#   "sorted"   : the same postings as the i.i.d. corpus, but docs are numbered by length, shortest first (an index sorted by a
#                field the BM25 score follows: a posting in a short doc scores more) -- scores fall along the docid axis.
CLUSTER_EPOCHS = 96


def term_postings_clustered(n_docs: int, rank: int, seed: int = 1234) -> Tuple[np.ndarray, np.ndarray]:
    """Global (docids, freqs) of the term of Zipf rank `rank` in the "clustered" corpus: E[df] = N / (rank + 1) as in
    term_postings, but the density follows a per-term burst profile over CLUSTER_EPOCHS epochs of the docid axis."""
    p = 1.0 / (rank + 1.0)
    rng = np.random.Generator(np.random.PCG64((seed * 7919) ^ (rank * 104729 + 17)))
    wts = rng.gamma(0.35, 1.0 / 0.35, size=CLUSTER_EPOCHS)
    wts *= CLUSTER_EPOCHS / wts.sum()                       # mean 1 over the epochs
    dens = np.minimum(p * wts, 0.95)
    edges = np.linspace(0, n_docs, CLUSTER_EPOCHS + 1).astype(np.int64)
    parts = []
    for e in range(CLUSTER_EPOCHS):
        lo, hi = int(edges[e]), int(edges[e + 1])
        pe = float(dens[e])
        if hi <= lo or pe < 1e-9:   # (an epoch the term practically skips; also keeps the geometric draws inside int64)
            continue
        exp = (hi - lo) * pe
        m = int(exp + 6.0 * np.sqrt(exp) + 16)
        pos = np.cumsum(rng.geometric(pe, size=m).astype(np.int64)) - 1
        while pos[-1] < hi - lo:  # practically never
            pos = np.concatenate([pos, pos[-1] + np.cumsum(rng.geometric(pe, size=m).astype(np.int64))])
        parts.append(lo + pos[pos < hi - lo])
    docids = (np.concatenate(parts) if parts else np.zeros(0, np.int64)).astype(np.int32)
    freqs = np.minimum(255, rng.geometric(0.55, size=len(docids))).astype(np.int32)
    return docids, freqs


def corpus_variant_arrays(n_docs: int, variant: str, seed: int = 1234):
    """-> (doc lengths in docid order, postings(rank) -> (docids, freqs)) of a corpus variant: "iid" (SURVEY 8d), "clustered",
    "sorted" (see above)."""
    lens = doc_lengths(n_docs, seed)
    if variant in ("", "iid"):
        return lens, lambda r: term_postings(n_docs, r, seed)
    if variant == "clustered":
        return lens, lambda r: term_postings_clustered(n_docs, r, seed)
    if variant == "sorted":
        order = np.argsort(lens, kind="stable")              # new docid -> old docid, shortest doc first
        new_of_old = np.empty(n_docs, dtype=np.int64)
        new_of_old[order] = np.arange(n_docs, dtype=np.int64)

        def postings(r):
            d, f = term_postings(n_docs, r, seed)
            nd = new_of_old[d]
            o = np.argsort(nd, kind="stable")
            return nd[o].astype(np.int32), f[o]

        return lens[order], postings
    raise ValueError(f"unknown corpus variant {variant!r}")


def tiered_segment_sizes(n_docs: int, n_segments: int) -> List[int]:
    """Tiered-merge profile: N/2, N/4, ... with the remainder in the last."""
    if n_segments <= 1:
        return [n_docs]
    sizes = []
    left = n_docs
    for _ in range(n_segments - 1):
        s = max(1, left // 2)
        sizes.append(s)
        left -= s
    sizes.append(left)
    return [s for s in sizes if s > 0]


def make_queries(n_queries: int, n_terms: int, max_rank: int = 10000, seed: int = 4321) -> np.ndarray:
    """int64[n_queries, n_terms] of distinct ranks, log-uniform in [1, max_rank]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((n_queries, n_terms), dtype=np.int64)
    for q in range(n_queries):
        ranks: List[int] = []
        while len(ranks) < n_terms:
            r = int(np.floor(np.exp(rng.uniform(0.0, np.log(max_rank + 1.0)))))
            r = min(max(r, 1), max_rank)
            if r not in ranks:
                ranks.append(r)
        out[q] = ranks
    return out


def build_corpus(
    n_docs: int,
    ranks: Sequence[int],
    n_segments: int = 1,
    delete_fraction: float = 0.0,
    seed: int = 1234,
    variant: str = "iid",
) -> Corpus:
    """Materialise the postings of `ranks` only (the query set's terms), split into segments.  variant: corpus_variant_arrays."""
    ranks = sorted(set(int(r) for r in ranks))
    lens, postings_of = corpus_variant_arrays(n_docs, variant, seed)
    norms_all = int_to_byte4(lens)
    sizes = tiered_segment_sizes(n_docs, n_segments)
    bases = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

    per_seg_docs: List[List[np.ndarray]] = [[] for _ in sizes]
    per_seg_freqs: List[List[np.ndarray]] = [[] for _ in sizes]
    per_seg_counts: List[List[int]] = [[] for _ in sizes]
    doc_freq: Dict[int, int] = {}
    for r in ranks:
        d, f = postings_of(r)
        doc_freq[r] = int(len(d))
        cuts = np.searchsorted(d, bases)
        for s in range(len(sizes)):
            lo, hi = int(cuts[s]), int(cuts[s + 1])
            per_seg_docs[s].append((d[lo:hi] - bases[s]).astype(np.int32))
            per_seg_freqs[s].append(f[lo:hi])
            per_seg_counts[s].append(hi - lo)

    segments: List[SegmentData] = []
    del_rng = np.random.Generator(np.random.PCG64(seed + 99))
    for s, size in enumerate(sizes):
        counts = np.asarray(per_seg_counts[s], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        docids = np.concatenate(per_seg_docs[s]) if ranks else np.zeros(0, np.int32)
        freqs = np.concatenate(per_seg_freqs[s]) if ranks else np.zeros(0, np.int32)
        live = None
        if delete_fraction > 0.0:
            alive = del_rng.random(size) >= delete_fraction
            padded = np.zeros(((size + 63) // 64) * 64, dtype=bool)
            padded[:size] = alive
            live = np.packbits(padded.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1)
        segments.append(
            SegmentData(
                max_doc=int(size),
                doc_base=int(bases[s]),
                norms=norms_all[bases[s]: bases[s] + size].copy(),
                term_ids=np.asarray(ranks, dtype=np.int64),
                offsets=offsets,
                docids=np.ascontiguousarray(docids, dtype=np.int32),
                freqs=np.ascontiguousarray(freqs, dtype=np.int32),
                live_bits=live,
            )
        )
    return Corpus(
        n_docs=n_docs,
        doc_count=n_docs,
        sum_total_term_freq=int(lens.astype(np.int64).sum()),
        segments=segments,
        doc_freq=doc_freq,
    )


def make_vectors(n: int, dim: int, seed: int = 777, normalize: bool = False) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    v = rng.standard_normal((n, dim), dtype=np.float32)
    if normalize:
        v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    return v


def random_mask(max_doc: int, density: float, seed: int) -> np.ndarray:
    """Doc set of a synthetic non-scoring clause as uint64 words (bit d set = doc d matches), PCG64(seed)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bits = np.zeros(((max_doc + 63) // 64) * 64, dtype=bool)
    bits[:max_doc] = rng.random(max_doc) < density
    return np.packbits(bits.reshape(-1, 64), axis=1, bitorder="little").view(np.uint64).reshape(-1)


def accept_words(seg: SegmentData, filter_words: Optional[np.ndarray] = None,
                 must_not_words: Optional[np.ndarray] = None) -> np.ndarray:
    """liveDocs & FILTER & ~MUST_NOT of one segment as uint64 words (what the bulk scorer accepts)."""
    n = (seg.max_doc + 63) // 64
    w = seg.live_bits.copy() if seg.live_bits is not None else np.full(n, ~np.uint64(0), dtype=np.uint64)
    if filter_words is not None:
        w &= filter_words
    if must_not_words is not None:
        w &= ~must_not_words
    return w
