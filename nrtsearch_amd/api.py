"""Host-side mirror of the reference's operator surface for the accelerated path.

Names and argument meaning follow Lucene / nrtsearch so the parity tests read like the reference's
own tests:

  reference                                                         here
  ---------------------------------------------------------------   ---------------------------
  org.apache.lucene.search.TermQuery / BoostQuery / BooleanQuery     TermQuery / BoostQuery / BooleanQuery
  BM25Similarity.scorer(boost, collectionStats, termStats)           BM25Similarity.scorer
  TopScoreDocCollectorManager(numHits, after, totalHitsThreshold)    TopScoreDocCollectorManager
     (S/search/collectors/RelevanceCollector.java:63-68)
  IndexSearcher.search(Query, CollectorManager)                      GpuIndexSearcher.search
     (S/handler/SearchHandler.java:1412-1413)
  TopDocs / TotalHits.Relation                                       TopDocs

Everything numeric happens behind the C ABI (include/nrtgpu.h) on the GPU; this module only
marshals.  It has no CPU implementation of the query path.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from ._lib import NrtGpuError  # noqa: F401  (re-export)

INT_MAX = 2**31 - 1
TOTAL_HITS_THRESHOLD = 1000  # S/search/SearchRequestProcessor.java:102


# ---- queries (only the shapes eligible for the device route) ------------------------
EXCHANGE_ALLGATHER, EXCHANGE_ALLTOALL = 0, 1   # nrtgpu.h: NRTGPU_EXCHANGE_*
EXCHANGE_NO_SPECULATION = 0x100               # ... or-ed into dist_search_batch's mode: no shard-level guesses


@dataclasses.dataclass(frozen=True)
class TermQuery:
    field: int
    term: int  # injective 64-bit id of the term (the Java shim hashes the BytesRef)


@dataclasses.dataclass(frozen=True)
class BoostQuery:
    query: TermQuery
    boost: float


@dataclasses.dataclass(frozen=True)
class MaskFilter:
    """A non-scoring clause whose per-leaf doc set is resident as a mask (GpuSegment.set_mask): what a
    FILTER / MUST_NOT clause is once LRUQueryCache holds its DocIdSet."""

    mask_id: int


@dataclasses.dataclass(frozen=True)
class BooleanQuery:
    """SHOULD disjunction of term clauses, optionally narrowed by one FILTER and one MUST_NOT clause
    (S/query/QueryNodeMapper.java:257-283)."""

    should: Tuple[Union[TermQuery, BoostQuery], ...] = ()
    minimum_number_should_match: int = 0
    filter: Tuple[MaskFilter, ...] = ()
    must_not: Tuple[MaskFilter, ...] = ()
    must: Tuple[Union[TermQuery, BoostQuery], ...] = ()   # MatchQuery with operator MUST (QueryNodeMapper.java:369-373)


@dataclasses.dataclass(frozen=True)
class DisjunctionMaxQuery:
    """DisjunctionMaxQuery over (boosted) term queries (S/query/QueryNodeMapper.java:350-358): a doc scores its best
    disjunct plus tie_breaker_multiplier x the others (0..1)."""

    disjuncts: Tuple[Union[TermQuery, BoostQuery], ...] = ()
    tie_breaker_multiplier: float = 0.0


Query = Union[TermQuery, BoostQuery, BooleanQuery, DisjunctionMaxQuery]


class UnsupportedQuery(Exception):
    """The rewritten query is not eligible for the device route: run the CPU (Lucene) path."""


@dataclasses.dataclass
class ScoreDoc:
    doc: int
    score: float


@dataclasses.dataclass
class TopScoreDocCollectorManager:
    num_hits: int
    after: Optional[ScoreDoc] = None
    total_hits_threshold: int = TOTAL_HITS_THRESHOLD
    # Scorable.setMinCompetitiveScore fed from outside this searcher (other shards of the same search)
    min_competitive_score: float = 0.0


@dataclasses.dataclass
class TopDocs:
    docs: np.ndarray            # int32 global docids
    scores: np.ndarray          # float32
    total_hits: int
    relation_gte: bool          # True == TotalHits.Relation.GREATER_THAN_OR_EQUAL_TO


_OUT_ARRAYS = threading.local()


def _topdocs_outputs(nq: int, k: int):
    """(outs, docs, scores) for a call that returns nq lists of up to k hits: an array of nrtgpu_topdocs whose docs / scores pointers
    address the rows of two numpy arrays.  Built once per thread and shape and used again -- 2 x nq pointer objects through
    numpy's ctypes bridge cost 0.3 ms at 64 queries, a tenth of an exact vector search's pass; the caller copies what it returns."""
    cache = getattr(_OUT_ARRAYS, "by_shape", None)
    if cache is None:
        cache = _OUT_ARRAYS.by_shape = {}
    got = cache.get((nq, k))
    if got is None:
        outs = (_lib.TopDocs * nq)()
        docs = np.zeros((nq, k), dtype=np.int32)
        scores = np.zeros((nq, k), dtype=np.float32)
        d0, s0 = docs.ctypes.data, scores.ctypes.data
        for qi in range(nq):
            outs[qi].capacity = k
            outs[qi].docs = C.cast(d0 + qi * k * 4, C.POINTER(C.c_int32))
            outs[qi].scores = C.cast(s0 + qi * k * 4, C.POINTER(C.c_float))
        if len(cache) >= 8:
            cache.clear()
        got = cache[(nq, k)] = (outs, docs, scores)
    outs = got[0]
    for qi in range(nq):
        outs[qi].n_hits = 0
        outs[qi].total_hits = 0
    return got


def _topdocs_lists(outs, docs, scores, n: int, none_if_negative: bool = False):
    """The call's answers as TopDocs: ONE copy of each array (the arrays of _topdocs_outputs are used again), a row's hits as a view of it."""
    dc, sc = docs.copy(), scores.copy()
    res = []
    for qi in range(n):
        o = outs[qi]
        th = int(o.total_hits)
        if none_if_negative and th < 0:
            res.append(None)
            continue
        m = o.n_hits
        res.append(TopDocs(dc[qi, :m], sc[qi, :m], th, bool(o.total_hits_is_lower_bound)))
    return res


# ---- statistics / similarity ---------------------------------------------------------------------
@dataclasses.dataclass
class CollectionStatistics:
    doc_count: int
    sum_total_term_freq: int


class IndexStatistics:
    """Index-global statistics the searcher reads from the top-level reader."""

    def __init__(self):
        self.fields: Dict[int, CollectionStatistics] = {}
        self.doc_freq: Dict[Tuple[int, int], int] = {}

    @classmethod
    def from_corpus(cls, corpus, field: int = 0) -> "IndexStatistics":
        st = cls()
        st.fields[field] = CollectionStatistics(corpus.doc_count, corpus.sum_total_term_freq)
        for t, df in corpus.doc_freq.items():
            st.doc_freq[(field, int(t))] = int(df)
        return st


class BM25Similarity:
    """Default similarity of the reference (S/similarity/SimilarityCreator.java:33,41)."""

    def __init__(self, k1: float = 1.2, b: float = 0.75):
        self.k1, self.b = float(k1), float(b)

    def idf(self, doc_freq: int, doc_count: int) -> np.float32:
        return np.float32(_lib.load().nrtgpu_bm25_idf(int(doc_count), int(doc_freq)))

    def avgdl(self, cs: CollectionStatistics) -> np.float32:
        return np.float32(_lib.load().nrtgpu_bm25_avgdl(int(cs.sum_total_term_freq), int(cs.doc_count)))

    def norm_cache(self, cs: CollectionStatistics) -> np.ndarray:
        out = np.zeros(256, dtype=np.float32)
        _lib.load().nrtgpu_bm25_norm_cache(C.c_float(float(self.avgdl(cs))), C.c_float(self.k1), C.c_float(self.b),
                                          out.ctypes.data)
        return out

    def scorer(self, boost: float, cs: CollectionStatistics, doc_freq: int) -> Tuple[np.float32, np.ndarray]:
        """-> (weight = boost * idf, normInverse cache[256])."""
        idf = self.idf(doc_freq, cs.doc_count)
        return np.float32(np.float32(boost) * idf), self.norm_cache(cs)


# ---- device context / segment store --------------------------------------------------------------
class GpuContext:
    def __init__(self, device_id: int = 0, max_batch: int = 1024, target_items: int = 0,
                 collect_timing: bool = False, flags: int = 0, host_threads: int = 0, lookup_budget_pct: int = 0):
        L = _lib.load()
        if os.environ.get("NRTGPU_PACKED_POSTINGS", "") not in ("", "0"):   # run anything (the whole test suite) on the packed layout
            flags |= _lib.NRTGPU_FLAG_PACKED_POSTINGS
        self.flags = flags
        if lookup_budget_pct == 0 and os.environ.get("NRTGPU_TEST_LOOKUP_BUDGET_PCT"):   # run anything on another lookup budget (tests, A/B scripts)
            lookup_budget_pct = int(os.environ["NRTGPU_TEST_LOOKUP_BUDGET_PCT"])
        cfg = _lib.Config(device_id, max_batch, target_items, int(collect_timing), flags, host_threads, int(lookup_budget_pct), 0)
        h = C.c_void_p()
        _lib.check(L.nrtgpu_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.max_batch = max_batch

    def set_slicing(self, slice_max_docs: int = 250_000, slice_max_segments: int = 5, virtual_shards: int = 1) -> None:
        """MyIndexSearcher.SlicingParams of the searcher this context serves (TotalHits.relation is decided per slice)."""
        _lib.check(_lib.load().nrtgpu_set_slicing(self._h, int(slice_max_docs), int(slice_max_segments), int(virtual_shards)))

    @staticmethod
    def dist_unique_id() -> bytes:
        """An ncclUniqueId (128 bytes) for dist_init: made on one rank, handed to the others by the deployment."""
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().nrtgpu_dist_unique_id(buf))
        return bytes(buf.raw)

    def dist_init(self, world: int, rank: int, unique_id: bytes) -> None:
        """Joins the RCCL communicator of a `world`-GPU search (one process per GPU): the collective lives in the library."""
        _lib.check(_lib.load().nrtgpu_dist_init(self._h, int(world), int(rank), C.c_char_p(unique_id)))

    def dist_close(self) -> None:
        _lib.load().nrtgpu_dist_close(self._h)

    def exchange_open(self, shm_name: str, world: int, rank: int) -> None:
        """Cross-GPU bound exchange (include/nrtgpu.h); synchronise the ranks once before the first search."""
        _lib.check(_lib.load().nrtgpu_exchange_open(self._h, shm_name.encode(), int(world), int(rank)))

    def exchange_close(self) -> None:
        _lib.load().nrtgpu_exchange_close(self._h)

    def stats(self) -> dict:
        st = _lib.Stats()
        _lib.check(_lib.load().nrtgpu_get_stats(self._h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in _lib.Stats._fields_}

    def scan_profile(self) -> dict:
        out = np.zeros(16, dtype=np.float64)
        _lib.check(_lib.load().nrtgpu_get_scan_profile(self._h, out.ctypes.data))
        # counters of wave 0 of every item, summed over the items since reset_stats (NRTGPU_FLAG_PROFILE)
        names = ["prologue_cycles", "rendezvous_wait_cycles", "rendezvous_cycles", "walk_cycles", "epilogue_cycles",
                 "rendezvous", "compactions", "subtiles", "rz_select_cycles", "sparse_subtiles", "rz_keep_cycles",
                 "rz_publish_append_cycles", "candidate_subtiles", "maybe_subtiles", "last_wave_finish_cycles",
                 "first_wave_finish_cycles"]
        return dict(zip(names, out.tolist()))

    def maxscore_profile(self) -> dict:
        out = np.zeros(16, dtype=np.float64)
        _lib.check(_lib.load().nrtgpu_get_maxscore_profile(self._h, out.ctypes.data))
        names = ["windows", "compactions", "chunks", "postings_streamed", "postings_surviving", "docs_evaluated", "lookups",
                 "candidates", "prologue_cycles", "item_cycles", "waves_meeting_cycles", "waves_idle_cycles", "waves_part_prologue_cycles",
                 "waves_walk_cycles", "last_wave_out_cycles", "epilogue_cycles"]
        return dict(zip(names, out.tolist()))

    @staticmethod
    def set_thread_deadline(seconds_from_now: Optional[float]) -> None:
        """The calling thread's deadline for the search calls it makes from now on (nrtgpu_set_thread_deadline_ns): None = no deadline."""
        L = _lib.load()
        L.nrtgpu_set_thread_deadline_ns(0 if seconds_from_now is None else int(L.nrtgpu_monotonic_ns() + seconds_from_now * 1e9))

    @staticmethod
    def last_diagnostics() -> dict:
        """What the calling thread's last completed search call cost (nrtgpu_last_diagnostics)."""
        d = _lib.Diagnostics()
        _lib.check(_lib.load().nrtgpu_last_diagnostics(C.byref(d)))
        return {n: getattr(d, n) for n, _ in _lib.Diagnostics._fields_ if n != "reserved"}

    def reset_stats(self) -> None:
        _lib.load().nrtgpu_reset_stats(self._h)

    def maxscore_item_walls(self):
        """(walls[n_slots, 8] = start, end (100 MHz ticks), item, windows, round begin, CU, round, workgroup; n_items) of the last
        instrumented MaxScore launch (nrtgpu_get_maxscore_item_walls): rows >= the call's items are helper sessions."""
        L = _lib.load()
        n_items = C.c_int64(0)
        n = int(L.nrtgpu_get_maxscore_item_walls(self._h, None, 0, C.byref(n_items)))
        out = np.zeros((max(n, 0), 8), dtype=np.uint64)
        if n > 0:
            L.nrtgpu_get_maxscore_item_walls(self._h, out.ctypes.data, n, C.byref(n_items))
        return out, int(n_items.value)

    @staticmethod
    def set_thread_slices(slice_of_leaf: Optional[Sequence[int]]) -> None:
        """Partial residency (nrtgpu_set_thread_slices): the calling thread's next searches run over a SUBSET of the searcher's leaves
        and count their hits by the whole searcher's slices -- slice_of_leaf[i] = the slice of the call's i-th leaf; None clears."""
        if slice_of_leaf is None:
            _lib.check(_lib.load().nrtgpu_set_thread_slices(None, 0))
        else:
            a = np.ascontiguousarray(slice_of_leaf, dtype=np.int32)
            _lib.check(_lib.load().nrtgpu_set_thread_slices(a.ctypes.data, len(a)))

    def set_speculation(self, margin: float) -> None:
        """Speculative thresholds of the MaxScore route (nrtgpu_set_speculation): the guess's safety margin in standard deviations; 0 = off."""
        _lib.check(_lib.load().nrtgpu_set_speculation(self._h, C.c_float(float(margin))))

    def set_shard_share(self, shard_docs: int, index_docs: int) -> None:
        """This context's share of a sharded index (nrtgpu_set_shard_share): the shard-level guesses then count the other shards'
        docs by it instead of assuming equal shards; (0, 0): equal shards."""
        _lib.check(_lib.load().nrtgpu_set_shard_share(self._h, int(shard_docs), int(index_docs)))

    def spec_counters(self) -> dict:
        """Speculative thresholds of the MaxScore route (nrtgpu_stats.spec_*): queries run under them since nrtgpu_set_speculation,
        queries run again, whether the library has switched them off for this context."""
        st = self.stats()
        return {"queries": int(st["spec_queries"]), "reruns": int(st["spec_reruns"]), "switched_off": bool(st["spec_disabled"]),
                "scattered": bool(st["spec_scattered"])}

    def debug_live_segments(self) -> int:
        """Segment handles of this context (uploads and forks) not freed yet (nrtgpu_debug_live_segments)."""
        return int(_lib.load().nrtgpu_debug_live_segments(self._h))

    def debug_hold_coalescers(self, hold: bool) -> None:
        """Test hook (nrtgpu_debug_hold_coalescers): while held, coalescer leaders leave only with a full batch / panel."""
        _lib.check(_lib.load().nrtgpu_debug_hold_coalescers(self._h, 1 if hold else 0))

    def debug_coalescer_pending(self, which: int) -> int:
        """Requests parked in a coalescer: 0 = nrtgpu_search_bm25_coalesced, 1 = nrtgpu_knn_exact_coalesced."""
        return int(_lib.load().nrtgpu_debug_coalescer_pending(self._h, int(which)))

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().nrtgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuSegment:
    """HBM-resident columnar replica of one immutable segment."""

    def __init__(self, ctx: GpuContext, max_doc: int, doc_base: int = 0):
        self.ctx, self.max_doc, self.doc_base = ctx, int(max_doc), int(doc_base)
        h = C.c_void_p()
        _lib.check(_lib.load().nrtgpu_segment_begin(ctx._h, self.max_doc, 0, C.byref(h)))
        self._h = h

    @classmethod
    def from_data(cls, ctx: GpuContext, seg, field: int = 0, omit_norms: bool = False,
                  omit_freqs: bool = False) -> "GpuSegment":
        """Upload a synth.SegmentData (what the Java side reads through PostingsEnum / norms)."""
        g = cls(ctx, seg.max_doc, seg.doc_base)
        g.add_field_norms(field, None if omit_norms else seg.norms)
        g.add_terms(field, seg.term_ids, seg.offsets, seg.docids, None if omit_freqs else seg.freqs)
        g.seal()
        if seg.live_bits is not None:
            g.set_live_docs(seg.live_bits)
        return g

    def add_field_norms(self, field: int, norms: Optional[np.ndarray]) -> None:
        p = None
        if norms is not None:
            norms = np.ascontiguousarray(norms, dtype=np.uint8)
            assert norms.shape[0] == self.max_doc
            p = norms.ctypes.data
        _lib.check(_lib.load().nrtgpu_segment_add_field_norms(self._h, int(field), p))

    def add_terms(self, field: int, term_ids, offsets, docids, freqs) -> None:
        term_ids = np.ascontiguousarray(term_ids, dtype=np.int64)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        docids = np.ascontiguousarray(docids, dtype=np.int32)
        fp = None
        if freqs is not None:
            freqs = np.ascontiguousarray(freqs, dtype=np.int32)
            fp = freqs.ctypes.data
        _lib.check(_lib.load().nrtgpu_segment_add_terms(self._h, int(field), len(term_ids), term_ids.ctypes.data,
                                                        offsets.ctypes.data, docids.ctypes.data, fp))

    def add_vectors(self, field: int, vectors: np.ndarray, ord_to_doc: Optional[np.ndarray] = None) -> None:
        vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        op = None
        if ord_to_doc is not None:
            ord_to_doc = np.ascontiguousarray(ord_to_doc, dtype=np.int32)
            op = ord_to_doc.ctypes.data
        _lib.check(_lib.load().nrtgpu_segment_add_vectors(self._h, int(field), vectors.shape[1], vectors.shape[0],
                                                          op, vectors.ctypes.data))

    def seal(self) -> None:
        _lib.check(_lib.load().nrtgpu_segment_seal(self._h))

    def set_live_docs(self, bits: Optional[np.ndarray]) -> None:
        if bits is None:
            _lib.check(_lib.load().nrtgpu_segment_set_live_docs(self._h, None, 0))
            return
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        _lib.check(_lib.load().nrtgpu_segment_set_live_docs(self._h, bits.ctypes.data, len(bits)))

    def fork(self, live_bits: Optional[np.ndarray]) -> "GpuSegment":
        """A new reader version of this sealed segment: shares its data, carries its own liveDocs (nrtgpu_segment_fork)."""
        g = GpuSegment.__new__(GpuSegment)
        g.ctx, g.max_doc, g.doc_base = self.ctx, self.max_doc, self.doc_base
        h = C.c_void_p()
        if live_bits is None:
            _lib.check(_lib.load().nrtgpu_segment_fork(self._h, None, 0, C.byref(h)))
        else:
            live_bits = np.ascontiguousarray(live_bits, dtype=np.uint64)
            _lib.check(_lib.load().nrtgpu_segment_fork(self._h, live_bits.ctypes.data, len(live_bits), C.byref(h)))
        g._h = h
        return g

    def set_mask(self, mask_id: int, bits: Optional[np.ndarray]) -> None:
        """Doc set of a non-scoring clause (uint64 words, bit d = doc d matches); None drops it."""
        if bits is None:
            _lib.check(_lib.load().nrtgpu_segment_set_mask(self._h, int(mask_id), None, 0))
            return
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        _lib.check(_lib.load().nrtgpu_segment_set_mask(self._h, int(mask_id), bits.ctypes.data, len(bits)))

    @property
    def device_bytes(self) -> int:
        return int(_lib.load().nrtgpu_segment_device_bytes(self._h))

    def release(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().nrtgpu_segment_release(self._h)
            self._h = None


# ---- searcher ------------------------------------------------------------------------------------
def _unsupported(msg: str):
    raise UnsupportedQuery(msg)


def _flatten(query: Query) -> Tuple[List[Tuple[int, int, float, int]], int, List[int], List[int], int, float]:
    """Eligibility predicate of SURVEY 8b on the rewritten query -> [(field, term, boost, occur: 0 SHOULD / 1 MUST)], msm,
    filter mask ids, must_not mask ids (any number of FILTER / MUST_NOT clauses: the library combines their masks at plan
    time), disjunction_max (1: best clause + tie breaker x the others instead of the sum), tie breaker."""
    def one(q, occur: int = 0) -> Tuple[int, int, float, int]:
        if isinstance(q, TermQuery):
            return (q.field, q.term, 1.0, occur)
        if isinstance(q, BoostQuery) and isinstance(q.query, TermQuery):
            return (q.query.field, q.query.term, float(q.boost), occur)
        raise UnsupportedQuery(f"clause {q!r} is not a (boosted) TermQuery")

    def dismax(q: DisjunctionMaxQuery):
        if not 0.0 <= q.tie_breaker_multiplier <= 1.0:
            raise UnsupportedQuery("DisjunctionMaxQuery with a tie breaker outside [0, 1]")
        if not q.disjuncts:
            raise UnsupportedQuery("empty DisjunctionMaxQuery")
        return [one(c) for c in q.disjuncts]

    def masks(q: BooleanQuery) -> Tuple[List[int], List[int]]:
        if any(not isinstance(c, MaskFilter) or c.mask_id <= 0 for c in q.filter + q.must_not):
            raise UnsupportedQuery("FILTER / MUST_NOT clauses must be resident masks")
        if len(q.filter) > _lib.NRTGPU_MAX_MASKS or len(q.must_not) > _lib.NRTGPU_MAX_MASKS:
            raise UnsupportedQuery(f"more than {_lib.NRTGPU_MAX_MASKS} FILTER or MUST_NOT clauses")
        return [c.mask_id for c in q.filter], [c.mask_id for c in q.must_not]

    if isinstance(query, DisjunctionMaxQuery):
        return dismax(query), 0, [], [], 1, float(query.tie_breaker_multiplier)
    if isinstance(query, BooleanQuery) and len(query.must) == 1 and isinstance(query.must[0], DisjunctionMaxQuery):
        # "+dismax #filter -must_not": one scoring clause, the masks add nothing to the score
        if query.should:
            raise UnsupportedQuery("a DisjunctionMaxQuery next to SHOULD clauses")
        f, mn = masks(query)
        return dismax(query.must[0]), 0, f, mn, 1, float(query.must[0].tie_breaker_multiplier)
    if isinstance(query, BooleanQuery):
        f, mn = masks(query)
        if query.must:
            # a conjunction of term clauses matches the docs all of them match and sums all their scores
            # (ConjunctionScorer: double sum, one cast): the disjunction with minimumNumberShouldMatch = n
            if query.should:
                # ReqOptSumScorer returns (float)required + (float)optional -- two separately rounded sums added in float
                # [Lucene-recall]: the clauses carry their occur, the kernel a second accumulator (plan.h: kMsSecReqOpt)
                if query.minimum_number_should_match > 0:   # (Lucene then scores the SHOULD part as one more required scorer)
                    raise UnsupportedQuery("MUST clauses next to minimumNumberShouldMatch > 0")
                return [one(c, 1) for c in query.must] + [one(c, 0) for c in query.should], 0, f, mn, 0, 0.0
            return [one(c) for c in query.must], len(query.must), f, mn, 0, 0.0
        if not query.should:
            raise UnsupportedQuery("empty BooleanQuery")
        if query.filter and query.minimum_number_should_match < 1:
            # with a FILTER clause Lucene makes the SHOULD clauses optional: filter-only docs would be hits of score 0
            raise UnsupportedQuery("FILTER with minimumNumberShouldMatch = 0")
        return [one(c) for c in query.should], query.minimum_number_should_match, f, mn, 0, 0.0
    return [one(query)], 0, [], [], 0, 0.0


class _Marshalled:
    """Keeps the ctypes arrays of a batch alive for the duration of the call."""

    def __init__(self, n: int):
        self.queries = (_lib.Bm25Query * n)()
        self.keep: list = []


class GpuIndexSearcher:
    """IndexSearcher over GPU-resident leaves: search(query, collectorManager) -> TopDocs."""

    def __init__(self, ctx: GpuContext, leaves: Sequence[GpuSegment], stats: IndexStatistics,
                 similarity: Optional[BM25Similarity] = None):
        self.ctx, self.leaves, self.stats = ctx, list(leaves), stats
        self.similarity = similarity or BM25Similarity()
        n = len(self.leaves)
        self._segs = (C.c_void_p * max(n, 1))(*[l._h for l in self.leaves])
        self._bases = (C.c_int32 * max(n, 1))(*[l.doc_base for l in self.leaves])
        self._cache_memo: Dict[int, np.ndarray] = {}

    def _norm_cache(self, field: int) -> np.ndarray:
        c = self._cache_memo.get(field)
        if c is None:
            c = self.similarity.norm_cache(self.stats.fields[field])
            self._cache_memo[field] = c
        return c

    def _marshal(self, queries: Sequence[Query], managers: Sequence[TopScoreDocCollectorManager]) -> _Marshalled:
        m = _Marshalled(len(queries))
        for qi, (query, mgr) in enumerate(zip(queries, managers)):
            clauses, msm, filters, must_nots, dis_max, tie_breaker = _flatten(query)
            fields: List[int] = []
            terms = (_lib.Term * len(clauses))()
            for ti, (field, term, boost, occur) in enumerate(clauses):
                if field not in self.stats.fields:
                    raise UnsupportedQuery(f"no statistics for field {field}")
                if field not in fields:
                    fields.append(field)
                df = self.stats.doc_freq.get((field, term), 0)
                cs = self.stats.fields[field]
                w = np.float32(np.float32(boost) * self.similarity.idf(df, cs.doc_count)) if df > 0 else np.float32(0)
                terms[ti] = _lib.Term(field, fields.index(field), term, float(w), int(occur))
            cache = np.concatenate([self._norm_cache(f) for f in fields]).astype(np.float32)
            m.keep += [terms, cache]
            q = m.queries[qi]
            q.n_terms = len(clauses)
            q.terms = terms
            q.n_caches = len(fields)
            q.norm_cache = cache.ctypes.data_as(C.POINTER(C.c_float))
            q.k = int(mgr.num_hits)
            q.total_hits_threshold = int(mgr.total_hits_threshold)
            q.has_after = int(mgr.after is not None)
            q.after_doc = int(mgr.after.doc) if mgr.after is not None else 0
            q.after_score = float(mgr.after.score) if mgr.after is not None else 0.0
            q.min_should_match = int(msm)
            q.min_competitive_score = float(mgr.min_competitive_score)
            q.filter_mask = int(filters[0]) if filters else 0
            q.must_not_mask = int(must_nots[0]) if must_nots else 0
            q.disjunction_max = int(dis_max)
            q.tie_breaker = float(np.float32(tie_breaker))
            for ids, n_name, p_name in ((filters[1:], "n_more_filters", "more_filters"), (must_nots[1:], "n_more_must_not", "more_must_not")):
                if ids:   # further FILTER / MUST_NOT clauses: the library ANDs / AND-NOTs their masks at plan time
                    arr = (C.c_int32 * len(ids))(*[int(x) for x in ids])
                    m.keep.append(arr)
                    setattr(q, n_name, len(ids))
                    setattr(q, p_name, arr)
        return m

    def search_batch(self, queries: Sequence[Query], managers: Sequence[TopScoreDocCollectorManager]) -> List[TopDocs]:
        n = len(queries)
        m = self._marshal(queries, managers)
        outs = (_lib.TopDocs * n)()
        bufs = []
        for qi, mgr in enumerate(managers):
            cap = max(int(mgr.num_hits), 1)
            d = np.zeros(cap, dtype=np.int32)
            s = np.zeros(cap, dtype=np.float32)
            bufs.append((d, s))
            outs[qi].capacity = cap
            outs[qi].docs = d.ctypes.data_as(C.POINTER(C.c_int32))
            outs[qi].scores = s.ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(_lib.load().nrtgpu_search_bm25_batch(self.ctx._h, self._segs, self._bases, len(self.leaves),
                                                        m.queries, n, outs))
        res = []
        for qi in range(n):
            nh = outs[qi].n_hits
            d, s = bufs[qi]
            res.append(TopDocs(d[:nh].copy(), s[:nh].copy(), int(outs[qi].total_hits),
                               bool(outs[qi].total_hits_is_lower_bound)))
        return res

    def search(self, query: Query, manager: TopScoreDocCollectorManager) -> TopDocs:
        return self.search_batch([query], [manager])[0]

    def dist_search_batch(self, queries: Sequence[Query], managers: Sequence[TopScoreDocCollectorManager],
                          mode: int = EXCHANGE_ALLGATHER) -> List[Optional[TopDocs]]:
        """The multi-GPU search through nrtgpu_dist_search_bm25_batch_mode: this rank's leaves, RCCL exchange + merge inside
        the library (GpuContext.dist_init first).  EXCHANGE_ALLGATHER: every rank gets every answer; EXCHANGE_ALLTOALL: this
        rank gets the answers of its slice of the batch, None for the others."""
        n = len(queries)
        m = self._marshal(queries, managers)
        outs = (_lib.TopDocs * n)()
        bufs = []
        for qi, mgr in enumerate(managers):
            cap = max(int(mgr.num_hits), 1)
            d = np.zeros(cap, dtype=np.int32)
            s = np.zeros(cap, dtype=np.float32)
            bufs.append((d, s))
            outs[qi].capacity = cap
            outs[qi].docs = d.ctypes.data_as(C.POINTER(C.c_int32))
            outs[qi].scores = s.ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(_lib.load().nrtgpu_dist_search_bm25_batch_mode(self.ctx._h, self._segs, self._bases, len(self.leaves), m.queries, n,
                                                                  int(mode), outs))
        return [None if outs[qi].total_hits < 0 else
                TopDocs(bufs[qi][0][: outs[qi].n_hits].copy(), bufs[qi][1][: outs[qi].n_hits].copy(), int(outs[qi].total_hits),
                        bool(outs[qi].total_hits_is_lower_bound)) for qi in range(n)]

    def dist_knn_exact(self, field: int, similarity: str, queries: np.ndarray, k: int, boost: float = 1.0,
                       mode: int = EXCHANGE_ALLGATHER) -> List[Optional[TopDocs]]:
        """Exact vector search over a row-partitioned field (nrtgpu_dist_knn_exact): this rank's leaves, exchange + merge
        inside the library; every rank passes the same queries."""
        queries = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        if similarity == "normalized_cosine":
            queries = np.ascontiguousarray(queries / np.linalg.norm(queries, axis=1, keepdims=True).astype(np.float32), dtype=np.float32)
        nq, dim = queries.shape
        outs, docs, scores = _topdocs_outputs(nq, int(k))
        _lib.check(_lib.load().nrtgpu_dist_knn_exact(self.ctx._h, self._segs, self._bases, len(self.leaves), int(field),
                                                     self.SIMILARITY[similarity], queries.ctypes.data, nq, dim, int(k),
                                                     C.c_float(boost), int(mode), outs))
        return _topdocs_lists(outs, docs, scores, nq, none_if_negative=True)

    def supported(self, query: Query, manager: TopScoreDocCollectorManager) -> bool:
        """The eligibility predicate alone (nrtgpu_query_supported): would the device route take this query?"""
        m = self._marshal([query], [manager])
        rc = _lib.load().nrtgpu_query_supported(self.ctx._h, self._segs, len(self.leaves), m.queries)
        if rc == _lib.NRTGPU_OK:
            return True
        if rc == _lib.NRTGPU_ERR_UNSUPPORTED:
            return False
        _lib.check(rc)
        return False

    def search_coalesced(self, query: Query, manager: TopScoreDocCollectorManager) -> TopDocs:
        """Blocking single search meant to be called from many threads at once: the library merges
        concurrent callers into device batches (nrtgpu_search_bm25_coalesced)."""
        m = self._marshal([query], [manager])
        cap = max(int(manager.num_hits), 1)
        d = np.zeros(cap, dtype=np.int32)
        s = np.zeros(cap, dtype=np.float32)
        out = _lib.TopDocs()
        out.capacity = cap
        out.docs = d.ctypes.data_as(C.POINTER(C.c_int32))
        out.scores = s.ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(_lib.load().nrtgpu_search_bm25_coalesced(self.ctx._h, self._segs, self._bases, len(self.leaves),
                                                            m.queries, C.byref(out)))
        return TopDocs(d[: out.n_hits].copy(), s[: out.n_hits].copy(), int(out.total_hits), bool(out.total_hits_is_lower_bound))

    # ---- vectors: ExactFloatVectorQuery / vector rescorer ------------------------------------------
    SIMILARITY = {"cosine": 0, "dot_product": 1, "normalized_cosine": 1, "l2_norm": 2, "max_inner_product": 3}

    def knn_exact(self, field: int, similarity: str, queries: np.ndarray, k: int, boost: float = 1.0) -> List[TopDocs]:
        """Brute-force exact vector search over every doc with a vector (S/query/vector/ExactVectorQuery.java)."""
        queries = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        if similarity == "normalized_cosine":  # unit-normalised query + dot product (VectorFieldDef.java:568-573)
            queries = queries / np.linalg.norm(queries, axis=1, keepdims=True).astype(np.float32)
            queries = np.ascontiguousarray(queries, dtype=np.float32)
        nq, dim = queries.shape
        outs, docs, scores = _topdocs_outputs(nq, int(k))
        _lib.check(_lib.load().nrtgpu_knn_exact(self.ctx._h, self._segs, self._bases, len(self.leaves), int(field),
                                                self.SIMILARITY[similarity], queries.ctypes.data, nq, dim, int(k),
                                                C.c_float(boost), outs))
        return _topdocs_lists(outs, docs, scores, nq)

    def knn_exact_coalesced(self, field: int, similarity: str, query: np.ndarray, k: int, boost: float = 1.0) -> TopDocs:
        """What a request thread calls with ONE exact vector query: concurrent callers are merged into panels of up to 64 queries
        that share a pass over the rows (nrtgpu_knn_exact_coalesced)."""
        query = np.ascontiguousarray(query, dtype=np.float32).reshape(-1)
        if similarity == "normalized_cosine":
            query = np.ascontiguousarray(query / np.linalg.norm(query).astype(np.float32), dtype=np.float32)
        out = _lib.TopDocs()
        docs = np.zeros(k, dtype=np.int32)
        scores = np.zeros(k, dtype=np.float32)
        out.capacity = k
        out.docs = docs.ctypes.data_as(C.POINTER(C.c_int32))
        out.scores = scores.ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(_lib.load().nrtgpu_knn_exact_coalesced(self.ctx._h, self._segs, self._bases, len(self.leaves), int(field),
                                                          self.SIMILARITY[similarity], query.ctypes.data, int(query.shape[0]), int(k),
                                                          C.c_float(boost), C.byref(out)))
        return TopDocs(docs[: out.n_hits].copy(), scores[: out.n_hits].copy(), int(out.total_hits), bool(out.total_hits_is_lower_bound))

    def knn_exact_relation(self, field: int, k: int, total_hits_threshold: int = TOTAL_HITS_THRESHOLD) -> bool:
        """TotalHits.relation of an exact vector query over these leaves by the reference's per-slice rule: True = GREATER_THAN_OR_EQUAL_TO
        (nrtgpu_knn_exact_relation; host only)."""
        rc = _lib.load().nrtgpu_knn_exact_relation(self.ctx._h, self._segs, self._bases, len(self.leaves), int(field), int(k), int(total_hits_threshold))
        if rc < 0:
            _lib.check(rc)
        return bool(rc)

    def knn_search(self, field: int, similarity: str, queries: np.ndarray, k: int, boost: float = 1.0,
                   filter: Optional[MaskFilter] = None, min_score: float = 0.0) -> List[TopDocs]:
        """The `knn` request path (KnnQuery with filter and similarity threshold) answered by exact search."""
        queries = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        if similarity == "normalized_cosine":
            queries = np.ascontiguousarray(queries / np.linalg.norm(queries, axis=1, keepdims=True).astype(np.float32), dtype=np.float32)
        nq, dim = queries.shape
        outs, docs, scores = _topdocs_outputs(nq, int(k))
        _lib.check(_lib.load().nrtgpu_knn_search(self.ctx._h, self._segs, self._bases, len(self.leaves), int(field),
                                                 self.SIMILARITY[similarity], queries.ctypes.data, nq, dim, int(k),
                                                 C.c_float(boost), int(filter.mask_id) if filter else 0,
                                                 C.c_float(min_score), outs))
        return _topdocs_lists(outs, docs, scores, nq)

    def rescore_vectors(self, hits: TopDocs, field: int, similarity: str, query: np.ndarray, window: int,
                        query_weight: float = 1.0, rescore_weight: float = 1.0, boost: float = 1.0) -> TopDocs:
        """RescoreOperation.rescore with a QueryRescore whose rescoreQuery is an exact vector query
        (S/rescore/QueryRescore.java:40-57)."""
        query = np.ascontiguousarray(query, dtype=np.float32)
        d = np.ascontiguousarray(hits.docs, dtype=np.int32)
        s = np.ascontiguousarray(hits.scores, dtype=np.float32)
        out = _lib.TopDocs()
        od = np.zeros(max(window, 1), dtype=np.int32)
        os_ = np.zeros(max(window, 1), dtype=np.float32)
        out.capacity = window
        out.docs = od.ctypes.data_as(C.POINTER(C.c_int32))
        out.scores = os_.ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(_lib.load().nrtgpu_rescore_vectors(self.ctx._h, self._segs, self._bases, len(self.leaves), int(field),
                                                      self.SIMILARITY[similarity], query.ctypes.data, len(query),
                                                      C.c_float(boost), d.ctypes.data, s.ctypes.data, len(d),
                                                      float(query_weight), float(rescore_weight), int(window),
                                                      C.byref(out)))
        return TopDocs(od[: out.n_hits].copy(), os_[: out.n_hits].copy(), hits.total_hits, hits.relation_gte)

    def search_hybrid_batch(self, queries: Sequence[Query], managers: Sequence[TopScoreDocCollectorManager], field: int,
                            similarity: str, query_vectors: np.ndarray, window: int, query_weight: float = 1.0,
                            rescore_weight: float = 1.0, boost: float = 1.0) -> List[TopDocs]:
        """search() followed by the vector rescorer for every query, fused on the device (config C5)."""
        n = len(queries)
        qv = np.ascontiguousarray(np.atleast_2d(query_vectors), dtype=np.float32)
        if qv.shape[0] != n:
            raise ValueError("one query vector per query")
        m = self._marshal(queries, managers)
        if window >= 1:
            outs, docs, scores = _topdocs_outputs(n, int(window))
        else:
            outs = (_lib.TopDocs * n)()
            docs = np.zeros((n, 1), dtype=np.int32)
            scores = np.zeros((n, 1), dtype=np.float32)
            for qi in range(n):
                outs[qi].capacity = window
                outs[qi].docs = docs[qi].ctypes.data_as(C.POINTER(C.c_int32))
                outs[qi].scores = scores[qi].ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(_lib.load().nrtgpu_search_hybrid_batch(
            self.ctx._h, self._segs, self._bases, len(self.leaves), m.queries, n, int(field), self.SIMILARITY[similarity],
            qv.ctypes.data, qv.shape[1], C.c_float(boost), float(query_weight), float(rescore_weight), int(window), outs))
        return _topdocs_lists(outs, docs, scores, n)


    def dist_search_hybrid_batch(self, queries: Sequence[Query], managers: Sequence[TopScoreDocCollectorManager], field: int,
                                 similarity: str, query_vectors: np.ndarray, window: int, query_weight: float = 1.0,
                                 rescore_weight: float = 1.0, boost: float = 1.0, mode: int = EXCHANGE_ALLGATHER) -> List[Optional[TopDocs]]:
        """The hybrid over docid-range shards (nrtgpu_dist_search_hybrid_batch): this rank's leaves; every rank passes the same
        queries and query vectors."""
        n = len(queries)
        qv = np.ascontiguousarray(np.atleast_2d(query_vectors), dtype=np.float32)
        if qv.shape[0] != n:
            raise ValueError("one query vector per query")
        m = self._marshal(queries, managers)
        outs = (_lib.TopDocs * n)()
        docs = np.zeros((n, max(window, 1)), dtype=np.int32)
        scores = np.zeros((n, max(window, 1)), dtype=np.float32)
        for qi in range(n):
            outs[qi].capacity = window
            outs[qi].docs = docs[qi].ctypes.data_as(C.POINTER(C.c_int32))
            outs[qi].scores = scores[qi].ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(_lib.load().nrtgpu_dist_search_hybrid_batch(
            self.ctx._h, self._segs, self._bases, len(self.leaves), m.queries, n, int(field), self.SIMILARITY[similarity],
            qv.ctypes.data, qv.shape[1], C.c_float(boost), float(query_weight), float(rescore_weight), int(window), int(mode), outs))
        return [None if outs[qi].total_hits < 0 else
                TopDocs(docs[qi, : outs[qi].n_hits].copy(), scores[qi, : outs[qi].n_hits].copy(), int(outs[qi].total_hits),
                        bool(outs[qi].total_hits_is_lower_bound)) for qi in range(n)]


# ---- pre-marshalled batches (bench / serving loop: no Python work inside the timed region) --------
class PreparedBatch:
    """A batch of queries marshalled once; `run()` is a single C-ABI call."""

    def __init__(self, searcher: GpuIndexSearcher, queries: Sequence[Query],
                 managers: Sequence[TopScoreDocCollectorManager]):
        self.searcher = searcher
        self.n = len(queries)
        self._m = searcher._marshal(queries, managers)
        self.k = max(int(m.num_hits) for m in managers)
        self._outs = (_lib.TopDocs * self.n)()
        self.docs = np.zeros((self.n, self.k), dtype=np.int32)
        self.scores = np.zeros((self.n, self.k), dtype=np.float32)
        for qi in range(self.n):
            self._outs[qi].capacity = self.k
            self._outs[qi].docs = self.docs[qi].ctypes.data_as(C.POINTER(C.c_int32))
            self._outs[qi].scores = self.scores[qi].ctypes.data_as(C.POINTER(C.c_float))

    def run(self) -> None:
        s = self.searcher
        _lib.check(_lib.load().nrtgpu_search_bm25_batch(s.ctx._h, s._segs, s._bases, len(s.leaves),
                                                        self._m.queries, self.n, self._outs))

    def run_device(self, k_stride: int, d_keys: int, d_counts: int, d_hits: int, epoch: int = -1) -> None:
        """Results stay in HBM (device pointers as ints) for the RCCL all-gather.  epoch >= 0: take part in
        the context's cross-GPU bound exchange (GpuContext.exchange_open) as that batch number."""
        s = self.searcher
        _lib.check(_lib.load().nrtgpu_search_bm25_batch_device_epoch(
            s.ctx._h, s._segs, s._bases, len(s.leaves), self._m.queries, self.n, int(k_stride),
            C.c_void_p(d_keys), C.c_void_p(d_counts), C.c_void_p(d_hits), int(epoch)))

    def begin_device(self, k_stride: int, d_keys: int, d_counts: int, d_hits: int, epoch: int = -1) -> int:
        """run_device in two halves (nrtgpu_search_bm25_batch_device_begin): plans and enqueues, returns a pending handle at
        once; wait_device(handle) blocks until the results are complete in HBM."""
        s = self.searcher
        h = C.c_void_p()
        _lib.check(_lib.load().nrtgpu_search_bm25_batch_device_begin(
            s.ctx._h, s._segs, s._bases, len(s.leaves), self._m.queries, self.n, int(k_stride),
            C.c_void_p(d_keys), C.c_void_p(d_counts), C.c_void_p(d_hits), int(epoch), C.byref(h)))
        return h.value

    def begin_shard_device(self, k_stride: int, d_keys: int, d_counts: int, d_hits: int, spec_world: int, d_guess: int) -> int:
        """begin_device for ONE SHARD of a search that `spec_world` GPUs share (nrtgpu_search_bm25_shard_device_begin): the
        speculative thresholds are guesses at the whole search's k-th score, the largest per query is left in d_guess (n x u64 in
        HBM) for the check against the merged list (PreparedMerge.run_dist_checked).  spec_world 0: no speculation."""
        s = self.searcher
        h = C.c_void_p()
        _lib.check(_lib.load().nrtgpu_search_bm25_shard_device_begin(
            s.ctx._h, s._segs, s._bases, len(s.leaves), self._m.queries, self.n, int(k_stride),
            C.c_void_p(d_keys), C.c_void_p(d_counts), C.c_void_p(d_hits), int(spec_world), C.c_void_p(d_guess or None), C.byref(h)))
        return h.value

    def note_shard_speculation(self, n_queries: int, n_failed: int) -> None:
        s = self.searcher
        _lib.check(_lib.load().nrtgpu_note_shard_speculation(s.ctx._h, s._segs, len(s.leaves), int(n_queries), int(n_failed)))

    @staticmethod
    def wait_device(handle: int) -> None:
        _lib.check(_lib.load().nrtgpu_pending_wait(C.c_void_p(handle)))

    def topdocs(self, qi: int) -> TopDocs:
        o = self._outs[qi]
        return TopDocs(self.docs[qi, : o.n_hits].copy(), self.scores[qi, : o.n_hits].copy(), int(o.total_hits),
                       bool(o.total_hits_is_lower_bound))


class PreparedMerge:
    """TopDocs.merge of all-gathered per-GPU results, output arrays marshalled once; `run()` is a
    single C-ABI call (nrtgpu_merge_topk_device).  List l of query q is row l * n_queries + q of the
    gathered arrays (all_gather_into_tensor concatenation order)."""

    def __init__(self, ctx: GpuContext, n_lists: int, n_queries: int, k_stride: int, ks: Sequence[int],
                 thresholds: Sequence[int]):
        self.ctx, self.n_lists, self.n, self.k_stride = ctx, int(n_lists), int(n_queries), int(k_stride)
        self._ks = np.ascontiguousarray(ks, dtype=np.int32)
        self._thr = np.ascontiguousarray(thresholds, dtype=np.int32)
        kmax = int(self._ks.max())
        self._outs = (_lib.TopDocs * self.n)()
        self.docs = np.zeros((self.n, kmax), dtype=np.int32)
        self.scores = np.zeros((self.n, kmax), dtype=np.float32)
        for qi in range(self.n):
            self._outs[qi].capacity = kmax
            self._outs[qi].docs = self.docs[qi].ctypes.data_as(C.POINTER(C.c_int32))
            self._outs[qi].scores = self.scores[qi].ctypes.data_as(C.POINTER(C.c_float))

    def run(self, d_keys: int, d_counts: int, d_hits: int) -> None:
        _lib.check(_lib.load().nrtgpu_merge_topk_device(
            self.ctx._h, self.n_lists, self.n, self.k_stride, C.c_void_p(d_keys), C.c_void_p(d_counts),
            C.c_void_p(d_hits), self._ks.ctypes.data, self._thr.ctypes.data, self._outs))

    def run_dist(self, d_keys: int, d_counts: int, d_hits: int, mode: int = EXCHANGE_ALLGATHER) -> None:
        """This rank's device-resident shard results -> RCCL exchange inside the library -> merge
        (nrtgpu_dist_exchange_merge; GpuContext.dist_init first; n_lists must equal the world size).  EXCHANGE_ALLTOALL:
        only this rank's slice of the batch is merged and delivered (owned(qi))."""
        _lib.check(_lib.load().nrtgpu_dist_exchange_merge(
            self.ctx._h, self.n, self.k_stride, C.c_void_p(d_keys), C.c_void_p(d_counts), C.c_void_p(d_hits),
            self._ks.ctypes.data, self._thr.ctypes.data, int(mode), self._outs))

    def run_dist_checked(self, d_keys: int, d_counts: int, d_hits: int, d_guess: int, mode: int = EXCHANGE_ALLGATHER) -> np.ndarray:
        """run_dist with the shards' speculative thresholds (PreparedBatch.begin_shard_device's d_guess) checked against the
        merged lists (nrtgpu_dist_exchange_merge_checked) -> the indices of the queries EVERY rank has to run again without
        speculation (the same on every rank)."""
        failed = np.zeros(self.n, dtype=np.uint8)
        n_failed = C.c_int32(0)
        _lib.check(_lib.load().nrtgpu_dist_exchange_merge_checked(
            self.ctx._h, self.n, self.k_stride, C.c_void_p(d_keys), C.c_void_p(d_counts), C.c_void_p(d_hits), C.c_void_p(d_guess),
            self._ks.ctypes.data, self._thr.ctypes.data, int(mode), self._outs, failed.ctypes.data, C.byref(n_failed)))
        return np.flatnonzero(failed) if n_failed.value else np.zeros(0, dtype=np.int64)

    def kth_keys(self) -> np.ndarray:
        """Per query the key -- (score bits << 32) | (0xFFFFFFFF - docid), as the device packs it (csrc/plan.h: pack_key) -- of
        rank k of the merged list, 0 where the list is shorter: what a shard's guess is checked against (bench.py's emulated
        exchange does the check itself; the library's is nrtgpu_dist_exchange_merge_checked)."""
        n_hits = np.frombuffer(self._outs, dtype=np.int32).reshape(self.n, C.sizeof(_lib.TopDocs) // 4)[:, 0]
        rows = np.arange(self.n)
        col = self._ks.astype(np.int64) - 1
        keys = (self.scores.view(np.uint32)[rows, col].astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - self.docs[rows, col].astype(np.uint64))
        return np.where(n_hits >= self._ks, keys, np.uint64(0))

    def owned(self, qi: int) -> bool:
        return self._outs[qi].total_hits >= 0

    def topdocs(self, qi: int) -> TopDocs:
        o = self._outs[qi]
        return TopDocs(self.docs[qi, : o.n_hits].copy(), self.scores[qi, : o.n_hits].copy(), int(o.total_hits),
                       bool(o.total_hits_is_lower_bound))


def merge_topk_device(ctx: GpuContext, n_lists: int, n_queries: int, k_stride: int, d_keys: int, d_counts: int,
                      d_hits: int, ks: Sequence[int], thresholds: Sequence[int]) -> List[TopDocs]:
    """TopDocs.merge of all-gathered per-GPU results (device pointers as ints)."""
    pm = PreparedMerge(ctx, n_lists, n_queries, k_stride, ks, thresholds)
    pm.run(d_keys, d_counts, d_hits)
    return [pm.topdocs(qi) for qi in range(n_queries)]


def slices(max_docs: Sequence[int], num_docs: Optional[Sequence[int]] = None, virtual_shards: int = 1,
           slice_max_docs: int = 250_000, slice_max_segments: int = 5):
    """MyIndexSearcher.slices / slicesForShards (S/search/MyIndexSearcher.java:79-208).
    -> (slices as lists of leaf indices in the reference's slice order, shard_of_leaf)."""
    n = len(max_docs)
    md = np.ascontiguousarray(max_docs, dtype=np.int32)
    nd = np.ascontiguousarray(num_docs if num_docs is not None else max_docs, dtype=np.int32)
    sl = np.full(max(n, 1), -1, dtype=np.int32)
    sh = np.full(max(n, 1), -1, dtype=np.int32)
    ns = _lib.load().nrtgpu_slices(n, md.ctypes.data, nd.ctypes.data, None, int(virtual_shards), int(slice_max_docs),
                                   int(slice_max_segments), sl.ctypes.data, sh.ctypes.data)
    if ns < 0:
        _lib.check(ns)
    out: List[List[int]] = [[] for _ in range(ns)]
    for i in range(n):
        out[sl[i]].append(i)
    return out, sh[:n].tolist()


def blend(retriever_docs: Sequence[np.ndarray], retriever_scores: Optional[Sequence[np.ndarray]] = None,
          boosts: Optional[Sequence[float]] = None, mode: str = "rrf", k: int = 60, start_hit: int = 0, top_hits: int = 10) -> TopDocs:
    """BlenderOperation.blend through the library (nrtgpu_blend): weighted RRF or score order, ties in the reference's order."""
    n = len(retriever_docs)
    docs = [np.ascontiguousarray(d, dtype=np.int32) for d in retriever_docs]
    scs = [np.ascontiguousarray(s_, dtype=np.float32) for s_ in retriever_scores] if retriever_scores is not None else None
    dp = (C.c_void_p * max(n, 1))(*[d.ctypes.data for d in docs])
    sp = (C.c_void_p * max(n, 1))(*[s_.ctypes.data for s_ in scs]) if scs is not None else None
    counts = np.asarray([len(d) for d in docs], dtype=np.int32)
    b = np.ascontiguousarray(boosts, dtype=np.float32) if boosts is not None else None
    cap = max(int(top_hits), 1)
    od, os_ = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
    out = _lib.TopDocs()
    out.capacity = cap
    out.docs = od.ctypes.data_as(C.POINTER(C.c_int32))
    out.scores = os_.ctypes.data_as(C.POINTER(C.c_float))
    _lib.check(_lib.load().nrtgpu_blend(n, dp, sp, counts.ctypes.data, b.ctypes.data if b is not None else None,
                                        0 if mode == "rrf" else 1, int(k), int(start_hit), int(top_hits), C.byref(out)))
    return TopDocs(od[: out.n_hits].copy(), os_[: out.n_hits].copy(), int(out.total_hits), bool(out.total_hits_is_lower_bound))


# ---- blenders (multi-retriever; O(k) host work that stays in Java in the reference) ----------------
def weighted_rrf_blend(retriever_hits: Sequence[np.ndarray], boosts: Optional[Sequence[float]] = None, k: int = 60,
                       start_hit: int = 0, top_hits: int = 10) -> TopDocs:
    """WeightedRrfBlenderOperation.mergeHits + BlenderOperation.sortAndPaginate
    (S/search/multiretriever/blender/operation/WeightedRrfBlenderOperation.java:53-78,
    .../score/WeightedRRFScoreDoc.java:62,75): score(doc) = sum over retrievers of boost / (k + rank),
    rank 1-based, accumulated in fp32 in retriever declaration order.  Equal scores are returned in
    docid order here (the reference leaves ties to its heap's order)."""
    if k < 1:
        raise ValueError(f"k must be >= 1, got: {k}")
    if top_hits == 0 or start_hit > top_hits:
        return TopDocs(np.zeros(0, np.int32), np.zeros(0, np.float32), 0, True)
    merged: Dict[int, np.float32] = {}
    for ri, docs in enumerate(retriever_hits):
        w = np.float32(1.0 if boosts is None else boosts[ri])
        for i, doc in enumerate(np.asarray(docs).tolist()):
            contrib = np.float32(w / np.float32(k + i + 1))
            merged[doc] = np.float32(merged[doc] + contrib) if doc in merged else contrib
    order = sorted(merged.items(), key=lambda kv: (-float(kv[1]), kv[0]))[: min(top_hits, len(merged))]
    page = order[min(start_hit, len(order)):]
    return TopDocs(np.asarray([d for d, _ in page], np.int32), np.asarray([s for _, s in page], np.float32),
                   len(merged), True)
