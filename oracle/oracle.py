"""ctypes front-end of the CPU oracle (oracle/nrt_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package nrtsearch_amd never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnrt_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "nrt_oracle.c")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


class _Term(C.Structure):
    _fields_ = [
        ("docids", C.c_void_p),
        ("freqs", C.c_void_p),
        ("n", C.c_int64),
        ("weight", C.c_float),
        ("norms", C.c_void_p),
        ("cache", C.c_void_p),
    ]


class _Leaf(C.Structure):
    _fields_ = [
        ("max_doc", C.c_int32),
        ("doc_base", C.c_int32),
        ("live_bits", C.c_void_p),
        ("n_terms", C.c_int32),
        ("terms", C.c_void_p),
        ("block_max", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.nrt_oracle_int_to_byte4.restype = C.c_int32
        L.nrt_oracle_int_to_byte4.argtypes = [C.c_int32]
        L.nrt_oracle_byte4_to_int.restype = C.c_int32
        L.nrt_oracle_byte4_to_int.argtypes = [C.c_int32]
        L.nrt_oracle_bm25_idf.restype = C.c_float
        L.nrt_oracle_bm25_idf.argtypes = [C.c_int64, C.c_int64]
        L.nrt_oracle_bm25_avgdl.restype = C.c_float
        L.nrt_oracle_bm25_avgdl.argtypes = [C.c_int64, C.c_int64]
        L.nrt_oracle_bm25_norm_cache.restype = None
        L.nrt_oracle_bm25_norm_cache.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.nrt_oracle_bm25_score.restype = C.c_float
        L.nrt_oracle_bm25_score.argtypes = [C.c_float, C.c_float, C.c_float]
        L.nrt_oracle_collector_new.restype = C.c_void_p
        L.nrt_oracle_collector_new.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32]
        L.nrt_oracle_collector_free.restype = None
        L.nrt_oracle_collector_free.argtypes = [C.c_void_p]
        L.nrt_oracle_collector_set_leaf.restype = None
        L.nrt_oracle_collector_set_leaf.argtypes = [C.c_void_p, C.c_int32]
        L.nrt_oracle_collector_collect.restype = None
        L.nrt_oracle_collector_collect.argtypes = [C.c_void_p, C.c_int32, C.c_float]
        L.nrt_oracle_collector_topdocs.restype = C.c_int32
        L.nrt_oracle_collector_topdocs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nrt_oracle_search_segment.restype = None
        L.nrt_oracle_search_segment.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.nrt_oracle_search_segment_msm.restype = None
        L.nrt_oracle_search_segment_msm.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                                    C.c_void_p]
        L.nrt_oracle_search_segment_dismax.restype = None
        L.nrt_oracle_search_segment_dismax.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_float,
                                                       C.c_void_p]
        L.nrt_oracle_search_segment_reqopt.restype = None
        L.nrt_oracle_search_segment_reqopt.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                                       C.c_void_p]
        L.nrt_oracle_block_max.restype = None
        L.nrt_oracle_block_max.argtypes = [C.c_void_p, C.c_void_p]
        L.nrt_oracle_search_segment_maxscore.restype = None
        L.nrt_oracle_search_segment_maxscore.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.nrt_oracle_search_batch.restype = None
        L.nrt_oracle_search_batch.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
        L.nrt_oracle_topdocs_merge.restype = C.c_int32
        L.nrt_oracle_topdocs_merge.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nrt_oracle_vector_score.restype = C.c_float
        L.nrt_oracle_vector_score.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        L.nrt_oracle_knn_exact.restype = None
        L.nrt_oracle_knn_exact.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                           C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nrt_oracle_rescore_combine.restype = C.c_float
        L.nrt_oracle_rescore_combine.argtypes = [C.c_float, C.c_int32, C.c_float, C.c_double, C.c_double]
        _lib = L
    return _lib


# ---- scalar helpers ---------------------------------------------------------------------------
def int_to_byte4(i: int) -> int:
    return lib().nrt_oracle_int_to_byte4(int(i))


def byte4_to_int(b: int) -> int:
    return lib().nrt_oracle_byte4_to_int(int(b))


def bm25_idf(doc_count: int, doc_freq: int) -> np.float32:
    return np.float32(lib().nrt_oracle_bm25_idf(int(doc_count), int(doc_freq)))


def bm25_avgdl(sum_ttf: int, doc_count: int) -> np.float32:
    return np.float32(lib().nrt_oracle_bm25_avgdl(int(sum_ttf), int(doc_count)))


def bm25_norm_cache(avgdl: float, k1: float = 1.2, b: float = 0.75) -> np.ndarray:
    out = np.zeros(256, dtype=np.float32)
    lib().nrt_oracle_bm25_norm_cache(C.c_float(avgdl), C.c_float(k1), C.c_float(b), out.ctypes.data)
    return out


def bm25_score(weight: float, freq: float, norm_inverse: float) -> np.float32:
    return np.float32(lib().nrt_oracle_bm25_score(C.c_float(weight), C.c_float(freq), C.c_float(norm_inverse)))


def vector_score(sim: int, q: np.ndarray, v: np.ndarray) -> np.float32:
    q = np.ascontiguousarray(q, dtype=np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    return np.float32(lib().nrt_oracle_vector_score(int(sim), q.ctypes.data, v.ctypes.data, int(q.shape[0])))


def vector_score_order(order: int, sim: int, q: np.ndarray, v: np.ndarray) -> np.float32:
    """vector_score under summation order `order` (nrt_oracle.c: 0 pinned scalar; 1 / 2 [Lucene-recall] DefaultVectorUtilSupport)."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    L = lib()
    L.nrt_oracle_vector_score_order.restype = C.c_float
    L.nrt_oracle_vector_score_order.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    return np.float32(L.nrt_oracle_vector_score_order(int(order), int(sim), q.ctypes.data, v.ctypes.data, int(q.shape[0])))


def knn_exact(sim: int, queries: np.ndarray, vecs: np.ndarray, k: int, live_words: Optional[np.ndarray] = None,
              doc_base: int = 0, boost: float = 1.0, n_threads: int = 1, order: int = 0):
    """ExactVectorQuery + top-k collector over one matrix of rows (row r = doc doc_base + r).
    -> (docs [n_q, k] int32, scores [n_q, k] float32, n [n_q]).  order: the summation order (0 = the pinned one)."""
    queries = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
    vecs = np.ascontiguousarray(vecs, dtype=np.float32)
    n_q, dim = queries.shape
    docs = np.zeros((n_q, k), dtype=np.int32)
    scores = np.zeros((n_q, k), dtype=np.float32)
    cnt = np.zeros(n_q, dtype=np.int32)
    lw = None if live_words is None else np.ascontiguousarray(live_words, dtype=np.uint64)
    L = lib()
    L.nrt_oracle_knn_exact_order.restype = None
    L.nrt_oracle_knn_exact_order.argtypes = [C.c_int32] + list(L.nrt_oracle_knn_exact.argtypes)
    L.nrt_oracle_knn_exact_order(int(order), int(sim), queries.ctypes.data, n_q, vecs.ctypes.data, int(vecs.shape[0]), int(dim),
                                 None if lw is None else lw.ctypes.data, int(doc_base), C.c_float(boost), int(k), int(n_threads),
                                 docs.ctypes.data, scores.ctypes.data, cnt.ctypes.data)
    return docs, scores, cnt


def rescore_combine(first: float, matched: bool, second: float, qw: float, rw: float) -> np.float32:
    return np.float32(lib().nrt_oracle_rescore_combine(C.c_float(first), int(matched), C.c_float(second), qw, rw))


# ---- collector --------------------------------------------------------------------------------
class Collector:
    """LazyQueueTopScoreDocCollector(numHits, after, totalHitsThreshold) driven one doc at a time."""

    def __init__(self, num_hits: int, after: Optional[Tuple[int, float]] = None, total_hits_threshold: int = 1000):
        self.num_hits = num_hits
        has_after = after is not None
        self._h = lib().nrt_oracle_collector_new(
            num_hits, int(has_after), int(after[0]) if has_after else 0,
            C.c_float(after[1] if has_after else 0.0), int(total_hits_threshold))
        if not self._h:
            raise ValueError("numHits must be > 0 and totalHitsThreshold >= 0")

    def set_leaf(self, doc_base: int) -> None:
        lib().nrt_oracle_collector_set_leaf(self._h, int(doc_base))

    def collect(self, leaf_doc: int, score: float) -> None:
        lib().nrt_oracle_collector_collect(self._h, int(leaf_doc), C.c_float(score))

    def topdocs(self):
        docs = np.zeros(self.num_hits, dtype=np.int32)
        scores = np.zeros(self.num_hits, dtype=np.float32)
        total = C.c_int64(0)
        gte = C.c_int32(0)
        n = lib().nrt_oracle_collector_topdocs(self._h, docs.ctypes.data, scores.ctypes.data, C.byref(total), C.byref(gte))
        return docs[:n].copy(), scores[:n].copy(), int(total.value), bool(gte.value)

    def close(self) -> None:
        if self._h:
            lib().nrt_oracle_collector_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def topdocs_merge(top_n: int, lists: Sequence[Tuple[np.ndarray, np.ndarray]]):
    lens = np.asarray([len(d) for d, _ in lists], dtype=np.int32)
    docs = np.concatenate([np.asarray(d, np.int32) for d, _ in lists]) if lists else np.zeros(0, np.int32)
    scores = np.concatenate([np.asarray(s, np.float32) for _, s in lists]) if lists else np.zeros(0, np.float32)
    od = np.zeros(max(top_n, 1), np.int32)
    os_ = np.zeros(max(top_n, 1), np.float32)
    n = lib().nrt_oracle_topdocs_merge(int(top_n), len(lists), lens.ctypes.data, docs.ctypes.data, scores.ctypes.data,
                                       od.ctypes.data, os_.ctypes.data)
    return od[:n].copy(), os_[:n].copy()


# ---- slices ----------------------------------------------------------------------------------------
DEFAULT_SLICING = (250_000, 5)   # the reference's sliceMaxDocs / sliceMaxSegments defaults


def leaf_slices(max_docs: Sequence[int], doc_bases: Sequence[int], slice_max_docs: int = 250_000,
                slice_max_segments: int = 5) -> List[List[int]]:
    """MyIndexSearcher.slices(leaves, sliceMaxDocs, sliceMaxSegments)
    (/root/reference/src/main/java/com/yelp/nrtsearch/server/search/MyIndexSearcher.java:163-208): leaves sorted by
    maxDoc descending (stable); a leaf above sliceMaxDocs is a slice of its own; the others are packed into the
    current group until it holds sliceMaxSegments leaves or more than sliceMaxDocs docs; each slice's leaves in
    docBase order.  Returns lists of leaf indices."""
    order = sorted(range(len(max_docs)), key=lambda i: -int(max_docs[i]))
    groups: List[List[int]] = []
    cur, doc_sum = None, 0
    for i in order:
        if max_docs[i] > slice_max_docs:
            groups.append([i])
            continue
        if cur is None:
            groups.append([i])
            cur = len(groups) - 1
        else:
            groups[cur].append(i)
        doc_sum += int(max_docs[i])
        if len(groups[cur]) >= slice_max_segments or doc_sum > slice_max_docs:
            cur, doc_sum = None, 0
    for g in groups:
        g.sort(key=lambda i: int(doc_bases[i]))
    return groups


def corpus_slices(corpus, slicing=DEFAULT_SLICING) -> List[List[int]]:
    if slicing is None:
        return [list(range(len(corpus.segments)))]
    return leaf_slices([s.max_doc for s in corpus.segments], [s.doc_base for s in corpus.segments], slicing[0], slicing[1])


# ---- whole-index search: one collector per slice, TopDocs.merge (LazyQueueTopScoreDocCollectorManager.java:137-144) ---
def bm25_query_stats(corpus, term_ids: Sequence[int], boosts: Optional[Sequence[float]] = None,
                     k1: float = 1.2, b: float = 0.75):
    """Index-global CollectionStatistics/TermStatistics -> (weights float32[n], cache float32[256])."""
    avgdl = bm25_avgdl(corpus.sum_total_term_freq, corpus.doc_count)
    cache = bm25_norm_cache(float(avgdl), k1, b)
    weights = np.zeros(len(term_ids), dtype=np.float32)
    for i, t in enumerate(term_ids):
        df = corpus.doc_freq.get(int(t), 0)
        idf = bm25_idf(corpus.doc_count, max(df, 0)) if df > 0 else np.float32(0.0)
        boost = np.float32(1.0 if boosts is None else boosts[i])
        weights[i] = np.float32(boost * idf)
    return weights, cache


_block_max_cache: dict = {}


def _block_max(corpus, si: int, term_id: int, term_struct, key_extra) -> np.ndarray:
    """Per-128-posting-block max score of one clause (index-time impacts in Lucene; cached here)."""
    key = (id(corpus), si, int(term_id), key_extra)
    bm = _block_max_cache.get(key)
    if bm is None:
        bm = np.zeros((int(term_struct.n) + 127) // 128, dtype=np.float32)
        lib().nrt_oracle_block_max(C.byref(term_struct), bm.ctypes.data)
        _block_max_cache[key] = bm
    return bm


def search_bm25(corpus, term_ids: Sequence[int], k: int, boosts: Optional[Sequence[float]] = None,
                after: Optional[Tuple[int, float]] = None, total_hits_threshold: int = 1000,
                segments: Optional[Sequence[int]] = None, omit_norms: bool = False, omit_freqs: bool = False,
                maxscore: bool = False, stats: Optional[dict] = None,
                accept: Optional[Sequence[Optional[np.ndarray]]] = None, min_should_match: int = 0,
                slicing=DEFAULT_SLICING, dismax: Optional[float] = None, must: Optional[Sequence[bool]] = None):
    """IndexSearcher.search(BooleanQuery(SHOULD TermQuery...), TopScoreDocCollectorManager(k, after, thr)): one
    collector per slice of the searcher (corpus_slices; each visits its leaves in docBase order), reduced like
    LazyQueueTopScoreDocCollectorManager.reduce: TopDocs.merge of the slices' hits, totalHits summed, relation
    GREATER_THAN_OR_EQUAL_TO if any slice's is.  `segments` (an explicit leaf list) or slicing=None: ONE collector.
    maxscore=True runs the dynamically pruned scorer (same top-k, totalHits a lower bound);
    stats["postings_scored"] then accumulates the postings it touched.
    accept[si] (uint64 words) replaces leaf si's liveDocs as the acceptDocs handed to the bulk scorer:
    liveDocs & FILTER doc set & ~MUST_NOT doc set -- what BooleanWeight's conjunction of a FILTER clause
    with the SHOULD disjunction (minimumNumberShouldMatch = 1) and its ReqExclScorer let through; such
    clauses add nothing to the score.
    dismax = tie breaker: the clauses are the disjuncts of a DisjunctionMaxQuery instead (best clause + tie x the others).
    must[i]: clause i is a MUST clause, the others SHOULD (minimumNumberShouldMatch 0): ReqOptSumScorer's float sum of two sums;
    all of them MUST: the conjunction == the disjunction with minimumNumberShouldMatch = n."""
    if must is not None and all(must):
        must, min_should_match = None, len(term_ids)
    if must is not None and not any(must):
        must = None
    if segments is None and slicing is not None:
        groups = corpus_slices(corpus, slicing)
        if len(groups) > 1:
            parts = [search_bm25(corpus, term_ids, k, boosts, after, total_hits_threshold, g, omit_norms, omit_freqs, maxscore,
                                 stats, accept, min_should_match, None, dismax, must) for g in groups]
            docs, scores = topdocs_merge(k, [(p[0], p[1]) for p in parts])
            return docs, scores, int(sum(p[2] for p in parts)), bool(any(p[3] for p in parts))
    weights, cache = bm25_query_stats(corpus, term_ids, boosts)
    col = Collector(k, after, total_hits_threshold)
    seg_ids = range(len(corpus.segments)) if segments is None else segments
    keep = []
    scored = C.c_int64(0)
    for si in seg_ids:
        seg = corpus.segments[si]
        arr = (_Term * max(len(term_ids), 1))()
        n_present = 0
        present_ids = []
        present_req = []
        lacks_must = False
        for i, t in enumerate(term_ids):
            d, f = seg.postings(int(t))
            if len(d) == 0:
                lacks_must = lacks_must or (must is not None and bool(must[i]))
                continue
            present_req.append(1 if (must is not None and must[i]) else 0)
            d = np.ascontiguousarray(d, dtype=np.int32)
            f = np.ascontiguousarray(f, dtype=np.int32)
            keep.append((d, f))
            arr[n_present].docids = d.ctypes.data
            arr[n_present].freqs = None if omit_freqs else f.ctypes.data
            arr[n_present].n = len(d)
            arr[n_present].weight = float(weights[i])
            arr[n_present].norms = None if omit_norms else seg.norms.ctypes.data
            arr[n_present].cache = cache.ctypes.data
            present_ids.append(int(t))
            n_present += 1
        live = seg.live_bits.ctypes.data if seg.live_bits is not None else None
        if accept is not None and accept[si] is not None:
            acc_bits = np.ascontiguousarray(accept[si], dtype=np.uint64)
            keep.append(acc_bits)
            live = acc_bits.ctypes.data
        if must is not None:
            if not lacks_must:   # (a leaf without one of the MUST terms has no hits)
                rq = (C.c_uint8 * max(n_present, 1))(*present_req)
                lib().nrt_oracle_search_segment_reqopt(seg.max_doc, seg.doc_base, live, n_present, C.byref(arr), rq, col._h)
        elif maxscore:
            bms = [_block_max(corpus, si, present_ids[j], arr[j],
                              (float(arr[j].weight), omit_norms, omit_freqs)) for j in range(n_present)]
            ptrs = (C.c_void_p * max(n_present, 1))(*[b.ctypes.data for b in bms])
            lib().nrt_oracle_search_segment_maxscore(seg.max_doc, seg.doc_base, live, n_present, C.byref(arr),
                                                     C.byref(ptrs), col._h, C.byref(scored))
        elif dismax is not None:
            lib().nrt_oracle_search_segment_dismax(seg.max_doc, seg.doc_base, live, n_present, C.byref(arr), C.c_float(dismax), col._h)
        elif min_should_match > 1:
            lib().nrt_oracle_search_segment_msm(seg.max_doc, seg.doc_base, live, n_present, C.byref(arr),
                                                int(min_should_match), col._h)
        else:
            lib().nrt_oracle_search_segment(seg.max_doc, seg.doc_base, live, n_present, C.byref(arr), col._h)
    if stats is not None:
        stats["postings_scored"] = stats.get("postings_scored", 0) + int(scored.value)
    res = col.topdocs()
    col.close()
    return res


class PreparedBatch:
    """Queries resolved to per-leaf clause arrays once (Weight creation + impacts, untimed), so that
    run() is a single C call: n_queries searches over n_threads OpenMP threads."""

    def __init__(self, corpus, queries: Sequence[Sequence[int]], k: int, total_hits_threshold: int = 1000, slicing=None):
        """slicing=None: one collector per query over all leaves (the timed CPU baseline); else one collector per
        (query, slice) and run() reduces them per query like the reference's CollectorManager."""
        self.k, self.thr, self.n_user = int(k), int(total_hits_threshold), len(queries)
        self._keep = []
        leaves = []
        offsets = [0]
        groups = corpus_slices(corpus, slicing)
        self._groups_per_query = len(groups)
        for term_ids, group in [(t, g) for t in queries for g in groups]:
            weights, cache = bm25_query_stats(corpus, term_ids)
            self._keep.append(cache)
            for si in group:
                seg = corpus.segments[si]
                arr = (_Term * max(len(term_ids), 1))()
                bms = []
                n_present = 0
                for i, t in enumerate(term_ids):
                    d, f = seg.postings(int(t))
                    if len(d) == 0:
                        continue
                    arr[n_present].docids = d.ctypes.data
                    arr[n_present].freqs = f.ctypes.data
                    arr[n_present].n = len(d)
                    arr[n_present].weight = float(weights[i])
                    arr[n_present].norms = seg.norms.ctypes.data
                    arr[n_present].cache = cache.ctypes.data
                    bms.append(_block_max(corpus, si, int(t), arr[n_present], (float(weights[i]), False, False)))
                    n_present += 1
                ptrs = (C.c_void_p * max(n_present, 1))(*[b.ctypes.data for b in bms])
                self._keep += [arr, ptrs]
                lf = _Leaf()
                lf.max_doc, lf.doc_base = seg.max_doc, seg.doc_base
                lf.live_bits = seg.live_bits.ctypes.data if seg.live_bits is not None else None
                lf.n_terms = n_present
                lf.terms = C.cast(arr, C.c_void_p)
                lf.block_max = C.cast(ptrs, C.c_void_p)
                leaves.append(lf)
            offsets.append(len(leaves))
        self._corpus = corpus
        self._leaves = (_Leaf * max(len(leaves), 1))(*leaves)
        self._offsets = np.asarray(offsets, dtype=np.int64)
        self.nq = self.n_user * self._groups_per_query

    def run(self, maxscore: bool, n_threads: int):
        docs = np.zeros((self.nq, self.k), np.int32)
        scores = np.zeros((self.nq, self.k), np.float32)
        n = np.zeros(self.nq, np.int32)
        total = np.zeros(self.nq, np.int64)
        gte = np.zeros(self.nq, np.int32)
        scored = C.c_int64(0)
        lib().nrt_oracle_search_batch(self.nq, self._offsets.ctypes.data, C.byref(self._leaves), self.k, self.thr,
                                      int(maxscore), int(n_threads), docs.ctypes.data, scores.ctypes.data,
                                      n.ctypes.data, total.ctypes.data, gte.ctypes.data, C.byref(scored))
        g = self._groups_per_query
        if g == 1:
            return docs, scores, n, total, gte, int(scored.value)
        md = np.zeros((self.n_user, self.k), np.int32)
        ms = np.zeros((self.n_user, self.k), np.float32)
        mn = np.zeros(self.n_user, np.int32)
        for q in range(self.n_user):   # CollectorManager.reduce: TopDocs.merge, totals summed, GTE if any slice's is
            d_, s_ = topdocs_merge(self.k, [(docs[q * g + j, : n[q * g + j]], scores[q * g + j, : n[q * g + j]]) for j in range(g)])
            mn[q] = len(d_)
            md[q, : len(d_)] = d_
            ms[q, : len(d_)] = s_
        return (md, ms, mn, total.reshape(self.n_user, g).sum(axis=1), gte.reshape(self.n_user, g).max(axis=1), int(scored.value))
