"""Benchmark / smoke workloads: the BASELINE.json configurations as code.

C3 (north star): 10M docs, 5-term SHOULD disjunction, BM25, top-1000, default totalHitsThreshold.
A rank of an N-GPU job owns a contiguous docid range of the index (large segments are split by
docid range) cut into tiered segments; statistics stay index-global.
"""
from __future__ import annotations

import dataclasses
from typing import List, Sequence

import numpy as np

from . import api, synth


@dataclasses.dataclass
class Workload:
    name: str
    n_docs: int
    n_terms: int
    k: int
    n_queries: int          # size of the query set the batches cycle through
    segments_per_shard: int
    max_rank: int = 10000


C2 = Workload("C2: 1M docs, Zipf terms, 2-term BooleanQuery BM25 top-100", 1_000_000, 2, 100, 4096, 6)
C3 = Workload("C3: 10M docs, 5-term disjunction BM25 top-1000", 10_000_000, 5, 1000, 10240, 10)   # 10 000 queries, ten 1024-query batches
SMOKE = Workload("smoke: 200k docs, 3-term disjunction top-100", 200_000, 3, 100, 8, 3, max_rank=2000)


def shard_range(n_docs: int, world: int, rank: int):
    """Contiguous docid range of `rank` (tile-aligned so no doc tile straddles two GPUs)."""
    tile = 1024
    per = ((n_docs + world - 1) // world + tile - 1) // tile * tile
    lo = min(n_docs, rank * per)
    hi = min(n_docs, lo + per)
    return lo, hi


def shard_leaves(w: Workload, world: int, rank: int, layout: str = "index"):
    """The leaves of `rank`'s shard as (lo, hi, sizes): its docid range and the sizes of its leaves, in docid order.
    layout "index" (default): the INDEX has `segments_per_shard` tiered segments (what one GPU holds at world 1) and a rank
    owns the pieces of them that fall into its docid range -- large segments are cut by docid range like a
    LeafReaderContextPartition, so most ranks of an 8-GPU job hold one piece of a big segment and the last one the small
    segments.  layout "per_shard" (rounds 1-2): every rank's range is cut into `segments_per_shard` tiered segments of its own."""
    lo, hi = shard_range(w.n_docs, world, rank)
    if world > 1 and layout == "index":
        # a range boundary that falls next to a segment boundary (within 1 % of a shard) is moved onto it: no slivers
        edges = np.cumsum(synth.tiered_segment_sizes(w.n_docs, w.segments_per_shard))[:-1]
        per = shard_range(w.n_docs, world, 0)[1]

        def snap(x):
            if 0 < x < w.n_docs and len(edges):
                e = int(edges[np.argmin(np.abs(edges - x))])
                if abs(e - x) <= per // 100:
                    return e
            return x

        lo, hi = snap(lo), snap(hi)
    if hi <= lo:
        return lo, hi, []
    if world <= 1 or layout == "per_shard":
        return lo, hi, synth.tiered_segment_sizes(hi - lo, w.segments_per_shard)
    sizes, base = [], 0
    for g in synth.tiered_segment_sizes(w.n_docs, w.segments_per_shard):
        a, b = max(base, lo), min(base + g, hi)
        if b > a:
            sizes.append(b - a)
        base += g
    return lo, hi, sizes


def shard_pieces(w: Workload, world: int, rank: int, layout: str = "index"):
    """The leaves of `rank`'s shard as a list of (doc_base, size), in docid order.  "index" / "per_shard": shard_leaves (one
    contiguous docid range per rank).  "balanced": the index's small segments (under half a shard) are dealt out to the ranks one by
    one, largest first, and every rank is then filled up to its share from the docid space of the big segments (cut at 1024-doc
    boundaries, like a LeafReaderContextPartition): every rank holds the same number of docs in one or two big pieces plus one or
    two small segments, instead of the last rank holding all the small ones (whose per-leaf planning made it the slowest)."""
    if world <= 1 or layout != "balanced":
        lo, hi, sizes = shard_leaves(w, world, rank, layout)
        out, base = [], lo
        for g in sizes:
            out.append((int(base), int(g)))
            base += g
        return out
    sizes = synth.tiered_segment_sizes(w.n_docs, w.segments_per_shard)
    bases = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    share = -(-w.n_docs // world)
    small = sorted([(int(g), int(b)) for g, b in zip(sizes, bases) if g < share // 2], reverse=True)
    big = [(int(b), int(g)) for g, b in zip(sizes, bases) if g >= share // 2]
    pieces = [[] for _ in range(world)]
    load = [0] * world
    for i, (g, b) in enumerate(small):       # round robin in size order: the loads stay within one small segment of each other
        r = i % world
        pieces[r].append((b, g))
        load[r] += g
    bi, off = 0, 0                           # the big segments' docid space, handed out rank by rank
    for r in range(world):
        need = share - load[r] if r < world - 1 else 1 << 62
        while need > 0 and bi < len(big):
            b, g = big[bi]
            take = min(g - off, -(-need // 1024) * 1024)   # a cut inside a segment falls on a 1024-doc boundary
            pieces[r].append((b + off, take))
            load[r] += take
            need -= take
            off += take
            if off == g:
                bi, off = bi + 1, 0
    return sorted(pieces[rank])


def build_shard_corpus(w: Workload, queries: np.ndarray, world: int = 1, rank: int = 0, seed: int = 1234, layout: str = "index",
                       variant: str = "iid"):
    """Corpus restricted to this rank's leaves, with index-global statistics.  variant: "iid" (every posting list an
    independent uniform draw), "clustered" (terms in docid bursts), "sorted" (docs numbered by length) -- synth.corpus_variant_arrays."""
    ranks = sorted(set(int(r) for r in queries.reshape(-1)))
    lens, postings_of = synth.corpus_variant_arrays(w.n_docs, variant, seed)
    norms_all = synth.int_to_byte4(lens)
    pieces = shard_pieces(w, world, rank, layout)
    sizes = [g for _, g in pieces]
    bases = np.asarray([b for b, _ in pieces], dtype=np.int64)
    ends = bases + np.asarray(sizes, dtype=np.int64)
    per_docs: List[List[np.ndarray]] = [[] for _ in sizes]
    per_freqs: List[List[np.ndarray]] = [[] for _ in sizes]
    doc_freq = {}
    for r in ranks:
        d, f = postings_of(r)
        doc_freq[r] = int(len(d))
        lo_cut, hi_cut = np.searchsorted(d, bases), np.searchsorted(d, ends)
        for s in range(len(sizes)):
            a, b = int(lo_cut[s]), int(hi_cut[s])
            per_docs[s].append((d[a:b] - bases[s]).astype(np.int32))
            per_freqs[s].append(f[a:b])
    segments = []
    for s, size in enumerate(sizes):
        counts = np.asarray([len(x) for x in per_docs[s]], dtype=np.int64)
        segments.append(synth.SegmentData(
            max_doc=int(size), doc_base=int(bases[s]), norms=norms_all[bases[s]: bases[s] + size].copy(),
            term_ids=np.asarray(ranks, dtype=np.int64),
            offsets=np.concatenate([[0], np.cumsum(counts)]).astype(np.int64),
            docids=np.ascontiguousarray(np.concatenate(per_docs[s]), dtype=np.int32),
            freqs=np.ascontiguousarray(np.concatenate(per_freqs[s]), dtype=np.int32)))
    return synth.Corpus(n_docs=w.n_docs, doc_count=w.n_docs, sum_total_term_freq=int(lens.astype(np.int64).sum()),
                        segments=segments, doc_freq=doc_freq)


def boolean_queries(query_ranks: np.ndarray) -> List[api.Query]:
    out = []
    for row in query_ranks:
        cl = tuple(api.TermQuery(0, int(t)) for t in row)
        out.append(cl[0] if len(cl) == 1 else api.BooleanQuery(cl))
    return out


def postings_per_query(corpus_doc_freq, query_ranks: np.ndarray) -> np.ndarray:
    return np.asarray([sum(corpus_doc_freq[int(t)] for t in row) for row in query_ranks], dtype=np.int64)
