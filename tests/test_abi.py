"""CPU-only checks of the drop-in boundary: the library builds, loads, exports every symbol that
include/nrtgpu.h declares, fails loudly without a GPU, and its host-side restatements (SmallFloat,
BM25 statistics, slices) agree with the oracle / the reference's own tests."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from nrtsearch_amd import _lib, api, build
from tests.conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nrtgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nrtgpu_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    declared = _declared_symbols()
    assert declared == sorted(_lib.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/nrtgpu.h but not exported"


def test_dev_symbols_are_not_part_of_the_product_boundary(lib):
    """include/nrtgpu_dev.h (test hooks, instrumented kernels' counters, the closed-loop generator) belongs to the development
    library only: the product library exports none of it, the development library all of it next to the product ABI."""
    text = open(os.path.join(ROOT, "include", "nrtgpu_dev.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(nrtgpu_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(_lib.DEV_SYMBOLS)
    for name in declared:
        assert not hasattr(lib, name), f"{name} is a development symbol but the product library exports it"
    build.build_dev()
    dev = _lib.load_dev()
    for name in declared + _lib.ABI_SYMBOLS:
        assert hasattr(dev, name), f"{name} missing from the development library"


def test_version_string(lib):
    assert b"gfx950" in lib.nrtgpu_version()


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_create_fails_loudly_without_gpu(lib):
    h = C.c_void_p()
    rc = lib.nrtgpu_create(None, C.byref(h))
    assert rc == _lib.NRTGPU_ERR_HIP and not h
    assert b"no CPU fallback" in lib.nrtgpu_last_error()
    with pytest.raises(_lib.NrtGpuError):
        api.GpuContext()


def test_struct_sizes_match_header(lib):
    # layout pinned on both sides of the boundary (x86-64 SysV)
    assert C.sizeof(_lib.Config) == 32
    assert C.sizeof(_lib.Term) == 24
    assert C.sizeof(_lib.Bm25Query) == 112
    assert C.sizeof(_lib.TopDocs) == 40
    assert C.sizeof(_lib.Stats) == 184
    assert C.sizeof(_lib.Diagnostics) == 56


def test_deadline_clock_and_thread_local_state_need_no_device(lib):
    """nrtgpu_monotonic_ns / nrtgpu_set_thread_deadline_ns / nrtgpu_last_diagnostics are host state of the calling thread."""
    import threading

    t0 = lib.nrtgpu_monotonic_ns()
    assert lib.nrtgpu_monotonic_ns() >= t0 > 0
    lib.nrtgpu_set_thread_deadline_ns(t0 - 1)          # an expired deadline on THIS thread ...
    seen = []

    def other():                                        # ... is nobody else's
        d = _lib.Diagnostics()
        assert lib.nrtgpu_last_diagnostics(C.byref(d)) == 0
        seen.append((d.queries, d.total_ms))

    th = threading.Thread(target=other)
    th.start()
    th.join()
    assert seen == [(0, 0.0)]
    lib.nrtgpu_set_thread_deadline_ns(0)
    assert lib.nrtgpu_last_diagnostics(None) == _lib.NRTGPU_ERR_INVALID_ARG


def test_host_smallfloat_matches_oracle(lib, oracle):
    for v in list(range(0, 5000)) + [2**k + d for k in range(3, 31) for d in (-1, 0, 1)] + [2**31 - 1]:
        assert lib.nrtgpu_int_to_byte4(v) == oracle.int_to_byte4(v)
    for b in range(256):
        assert lib.nrtgpu_byte4_to_int(b) == oracle.byte4_to_int(b)


def test_host_bm25_statistics_match_oracle(lib, oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(1, 10**8))
        df = int(rng.integers(1, n + 1))
        assert np.float32(lib.nrtgpu_bm25_idf(n, df)) == oracle.bm25_idf(n, df)
        sttf = int(rng.integers(n, 400 * n))
        assert np.float32(lib.nrtgpu_bm25_avgdl(sttf, n)) == oracle.bm25_avgdl(sttf, n)
    for avgdl in (2.0, 2.5, 4.0, 80.3, 1234.5):
        got = np.zeros(256, np.float32)
        lib.nrtgpu_bm25_norm_cache(C.c_float(avgdl), C.c_float(1.2), C.c_float(0.75), got.ctypes.data)
        assert np.array_equal(got, oracle.bm25_norm_cache(avgdl))


def test_similarity_golden_through_api(lib):
    # SearchStateTest.java:117 through the product-side helpers (not the oracle)
    sim = api.BM25Similarity()
    cs = api.CollectionStatistics(doc_count=2, sum_total_term_freq=5)
    w, cache = sim.scorer(1.0, cs, doc_freq=2)
    ninv = cache[lib.nrtgpu_int_to_byte4(3)]
    score = np.float32(w - w / (np.float32(1.0) + np.float32(1.0) * ninv))
    assert score == np.float32(0.0766057)


# ---- MyIndexSearcher.slices: src/test/java/com/yelp/nrtsearch/server/search/MyIndexSearcherTest.java
def test_slices_packing_rules(lib):
    # segments > sliceMaxDocs get their own slice; others packed until > maxDocs or >= maxSegments
    sl, _ = api.slices([300_000, 10, 260_000, 20, 30, 40, 50, 60], slice_max_docs=250_000, slice_max_segments=5)
    assert sl[0] == [0] and sl[1] == [2]
    # sorted by maxDoc desc: 60,50,40,30,20 fill one 5-segment slice (leaves sorted by docBase inside), 10 is left
    assert sl[2] == [3, 4, 5, 6, 7] and sl[3] == [1]
    # docSum > maxDocsPerSlice closes the group
    sl, _ = api.slices([100, 100, 100, 100], slice_max_docs=150, slice_max_segments=5)
    assert sl == [[0, 1], [2, 3]]
    assert api.slices([], slice_max_docs=10, slice_max_segments=5)[0] == []


def test_virtual_shards_lpt(lib):
    # greedy LPT on live docs (MyIndexSearcher.java:117-140): every leaf in exactly one shard, balanced
    sizes = [5_000_000, 2_500_000, 1_250_000, 625_000, 312_500, 156_250, 78_125, 78_125]
    sl, shard = api.slices(sizes, virtual_shards=4, slice_max_docs=250_000, slice_max_segments=5)
    assert sorted(i for s in sl for i in s) == list(range(len(sizes)))
    assert len(set(shard)) == 4
    load = [sum(sizes[i] for i in range(len(sizes)) if shard[i] == s) for s in range(4)]
    assert max(load) == 5_000_000            # the big segment alone
    # slices ordered largest first (…:154-158)
    tot = [sum(sizes[i] for i in s) for s in sl]
    assert tot == sorted(tot, reverse=True)


# ---- golden vectors: MyIndexSearcherTest.java:150-180 and MyIndexSearcherVirtualShardsTest.java:68-109
def _shape(sizes, **kw):
    sl, _ = api.slices(sizes, **kw)
    return [sum(sizes[i] for i in s) for s in sl], [len(s) for s in sl]


def test_reference_slice_golden_vectors(lib):
    ten = [10] * 10
    assert _shape(ten, slice_max_docs=25, slice_max_segments=10)[1] == [3, 3, 3, 1]       # testSliceDocsLimit
    assert _shape(ten, slice_max_docs=1000, slice_max_segments=4)[1] == [4, 4, 2]        # testSliceSegmentsLimit
    cases = [
        (dict(virtual_shards=4, slice_max_docs=10000, slice_max_segments=100), [10] * 7, [20, 20, 20, 10], [2, 2, 2, 1]),
        (dict(virtual_shards=4, slice_max_docs=10000, slice_max_segments=100), [10, 10], [10, 10], [1, 1]),
        (dict(virtual_shards=4, slice_max_docs=10000, slice_max_segments=100), [], [], []),
        (dict(virtual_shards=3, slice_max_docs=10000, slice_max_segments=100), list(range(1, 10)), [16, 15, 14], [3, 3, 3]),
        (dict(virtual_shards=2, slice_max_docs=25, slice_max_segments=100), [10] * 9, [30, 30, 20, 10], [3, 3, 2, 1]),
        (dict(virtual_shards=3, slice_max_docs=10000, slice_max_segments=2), [10, 10, 10, 10, 10, 10, 5, 4, 3],
         [20, 20, 20, 5, 4, 3], [2, 2, 2, 1, 1, 1]),
    ]
    for kw, sizes, docs, segs in cases:
        assert _shape(sizes, **kw) == (docs, segs), (kw, sizes)


# ---- blender: src/test/java/com/yelp/nrtsearch/server/search/MultiRetrieverSearchTest.java:410-497 ------
def test_weighted_rrf_blend_golden():
    # one retriever: RRF score at rank r is 1/(60+r) +- 1e-5
    td = api.weighted_rrf_blend([np.array([7, 3, 9])], top_hits=10)
    assert td.docs.tolist() == [7, 3, 9]
    for r, s in enumerate(td.scores.tolist(), start=1):
        assert abs(s - 1.0 / (60 + r)) <= 1e-5
    # docs found by both retrievers outrank 1/61
    td = api.weighted_rrf_blend([np.array([1, 2, 3]), np.array([3, 4, 1])], top_hits=10)
    both = {1, 3}
    assert set(td.docs[:2].tolist()) == both and all(s > 1.0 / 61 for s in td.scores[:2].tolist())
    assert td.total_hits == 4 and td.relation_gte
    # boosts / pagination / k validation (WeightedRRFScoreDocTest.java)
    td = api.weighted_rrf_blend([np.array([5]), np.array([6])], boosts=[1.0, 3.0], top_hits=2)
    assert td.docs.tolist() == [6, 5] and abs(float(td.scores[0]) - 3.0 / 61) < 1e-6
    assert api.weighted_rrf_blend([np.array([1, 2, 3])], start_hit=1, top_hits=3).docs.tolist() == [2, 3]
    assert len(api.weighted_rrf_blend([np.array([1])], top_hits=0).docs) == 0
    with pytest.raises(ValueError):
        api.weighted_rrf_blend([np.array([1])], k=0)


def test_query_eligibility_mapping():
    """Host logic only: which rewritten queries the mirror sends to the device and how."""
    from nrtsearch_amd import api
    t = [api.TermQuery(0, i) for i in (3, 5, 8)]
    assert api._flatten(t[0]) == ([(0, 3, 1.0, 0)], 0, [], [], 0, 0.0)
    assert api._flatten(api.BoostQuery(t[1], 2.0)) == ([(0, 5, 2.0, 0)], 0, [], [], 0, 0.0)
    assert api._flatten(api.BooleanQuery(tuple(t), 2)) == ([(0, 3, 1.0, 0), (0, 5, 1.0, 0), (0, 8, 1.0, 0)], 2, [], [], 0, 0.0)
    assert api._flatten(api.BooleanQuery(tuple(t), 1, (api.MaskFilter(4),), (api.MaskFilter(9),)))[2:] == ([4], [9], 0, 0.0)
    # any number of FILTER / MUST_NOT clauses (QueryNodeMapper.java:257-283): the library combines their masks at plan time
    assert api._flatten(api.BooleanQuery(tuple(t), 1, (api.MaskFilter(1), api.MaskFilter(2)), (api.MaskFilter(7), api.MaskFilter(5), api.MaskFilter(6))))[2:] == ([1, 2], [7, 5, 6], 0, 0.0)
    # a pure-MUST conjunction of terms is the disjunction that needs every clause
    assert api._flatten(api.BooleanQuery(must=tuple(t)))[1] == 3
    assert api._flatten(api.BooleanQuery(must=tuple(t), filter=(api.MaskFilter(2),)))[1:] == (3, [2], [], 0, 0.0)
    # MUST next to SHOULD clauses (minimumNumberShouldMatch 0): the clauses carry their occur, MUST first (ReqOptSumScorer)
    assert api._flatten(api.BooleanQuery(tuple(t[:1]), must=tuple(t[1:]))) == ([(0, 5, 1.0, 1), (0, 8, 1.0, 1), (0, 3, 1.0, 0)], 0, [], [], 0, 0.0)
    # DisjunctionMaxQuery over (boosted) term queries (QueryNodeMapper.java:350-358): best clause + tie breaker x the others
    dm = api.DisjunctionMaxQuery((t[0], api.BoostQuery(t[2], 3.0)))
    assert api._flatten(dm) == ([(0, 3, 1.0, 0), (0, 8, 3.0, 0)], 0, [], [], 1, 0.0)
    assert api._flatten(api.BooleanQuery(must=(dm,), filter=(api.MaskFilter(6),)))[1:] == (0, [6], [], 1, 0.0)
    assert api._flatten(api.DisjunctionMaxQuery(tuple(t), 0.25))[4:] == (1, 0.25)
    import pytest
    for bad in (api.BooleanQuery(tuple(t), 0, (api.MaskFilter(4),)),              # FILTER + optional SHOULD: score-0 hits
                api.BooleanQuery(tuple(t[:1]), 1, must=tuple(t[1:])),              # MUST next to minimumNumberShouldMatch > 0
                api.BooleanQuery((api.BooleanQuery(tuple(t)),)),                    # nested clause
                api.BooleanQuery(tuple(t), 1, tuple(api.MaskFilter(i) for i in range(1, 10))),   # more masks than NRTGPU_MAX_MASKS
                api.DisjunctionMaxQuery(tuple(t), 1.5),                             # tie breaker outside [0, 1]
                api.DisjunctionMaxQuery((api.BooleanQuery(tuple(t)),)),             # disjunct that is not a term query
                api.BooleanQuery(tuple(t[:1]), must=(api.DisjunctionMaxQuery(tuple(t)),)),
                api.BooleanQuery(())):
        with pytest.raises(api.UnsupportedQuery):
            api._flatten(bad)


def test_planner_item_counts():
    """The planner's cut of a batch into work items (host logic; no device): per-query rounding for big batches,
    exactly one item per CU for small ones."""
    import ctypes as C
    L = _lib.load()

    def counts(costs, target=256):
        c = np.asarray(costs, dtype=np.int64)
        out = np.zeros(len(c), dtype=np.int64)
        assert L.nrtgpu_plan_item_counts(len(c), c.ctypes.data, target, out.ctypes.data) == 0
        return out

    # a full batch: one item per query, only a query far above the fair share is split
    big = counts([4_000_000] * 1023 + [40_000_000])
    assert big[:1023].tolist() == [1] * 1023 and big[1023] == 2          # per item ~16 M: 40 M / 16 M = 2.5 -> rounds to 2
    # 64 equal queries on 256 CUs: exactly 4 each (per-query rounding used to give a few more than 256)
    eq = counts([4_500_000] * 64)
    assert eq.tolist() == [4] * 64
    # unequal small batch: exactly 256 items, proportional to cost, every live query gets one
    rng = np.random.Generator(np.random.PCG64(1))
    costs = rng.integers(500_000, 12_000_000, size=40)
    it = counts(costs)
    assert it.sum() == 256 and (it >= 1).all()
    share = costs * 256 / costs.sum()
    assert (np.abs(it - share) < 1.0 + 1e-9).all()
    # queries that match nothing get no item; a cheap batch is not cut below the minimum item cost
    assert counts([0, 3_000_000, 0]).tolist()[0] == 0
    assert counts([200_000] * 8).tolist() == [2] * 8 or counts([200_000] * 8).sum() <= 16
    tiny = counts([50_000] * 4)
    assert tiny.tolist() == [1] * 4
    assert L.nrtgpu_plan_item_counts(1, None, 256, None) != 0


def test_fixed_point_scale_makes_every_score_an_integer():
    """The exactness claim behind the fixed-point accumulators, checked on the host: at the scale the
    planner picks, every BM25 score a clause can produce -- any freq >= 1, any norm byte up to the largest one
    present -- is a positive integer below 2^32, so sums of up to 32 clauses shifted by <= 15 stay below 2^53."""
    import ctypes as C
    from oracle import oracle
    L = _lib.load()
    rng = np.random.Generator(np.random.PCG64(7))
    freqs = np.concatenate([np.arange(1, 300), [1000, 65535, 2**22 - 1]]).astype(np.float32)
    accepted = 0
    for trial in range(60):
        avgdl = np.float32(rng.choice([2.5, 17.0, 80.0, 400.0, 3000.0]))
        cache = oracle.bm25_norm_cache(float(avgdl))
        weight = np.float32(rng.choice([0.0488, 0.693, 2.3, 7.7, 13.1]) * rng.choice([1.0, 0.5, 3.25, 100.0]))
        max_norm = int(rng.choice([1, 40, 90, 127, 180, 255]))
        scale = C.c_int32(0)
        ok = L.nrtgpu_fixed_point_scale(C.c_float(weight), cache.ctypes.data, max_norm, C.byref(scale))
        assert ok in (0, 1)
        norms = np.arange(0, max_norm + 1)
        ninv = cache[norms][None, :]
        f = freqs[:, None]
        sc = (weight - weight / (np.float32(1.0) + f * ninv)).astype(np.float32)     # float32 ops, as BM25Similarity
        assert sc.dtype == np.float32
        if not ok:
            # refused only for a reason: the range really does not fit 8 binades below the weight
            s_min = float(sc[0, max_norm])
            assert s_min <= 0 or np.floor(np.log2(float(weight))) - np.floor(np.log2(s_min)) > 7
            continue
        accepted += 1
        scaled = sc.astype(np.float64) * 2.0 ** scale.value
        assert (scaled == np.floor(scaled)).all() and (scaled >= 1).all() and (scaled < 2.0 ** 32).all()
    assert accepted >= 20   # (norm byte 255 is a 2-billion-token field: those configurations are refused)


def test_blend_product_function_matches_the_reference_rules(lib):
    """nrtgpu_blend: WeightedRrfBlenderOperation / score order + sortAndPaginate; the golden 1 / (60 + r) of
    MultiRetrieverSearchTest.java:410-442, docs in both retrievers outrank 1/61 (:449-497), window / startHit rules."""
    td = api.blend([np.array([7, 3, 9])], top_hits=10)
    assert td.docs.tolist() == [7, 3, 9] and td.relation_gte and td.total_hits == 3
    assert np.allclose(td.scores, [1 / 61, 1 / 62, 1 / 63], rtol=0, atol=1e-7)
    td = api.blend([np.array([1, 2, 3]), np.array([3, 4, 1])], top_hits=10)
    assert td.docs.tolist()[:2] == [1, 3] and td.total_hits == 4
    assert abs(float(td.scores[0]) - float(np.float32(np.float32(1 / np.float32(61)) + np.float32(1 / np.float32(63))))) < 1e-9
    assert float(td.scores[1]) > 1 / 61
    td = api.blend([np.array([5]), np.array([6])], boosts=[1.0, 3.0], top_hits=2)
    assert td.docs.tolist() == [6, 5]
    assert api.blend([np.array([1, 2, 3])], start_hit=1, top_hits=3).docs.tolist() == [2, 3]
    assert len(api.blend([np.array([1])], top_hits=0).docs) == 0
    with pytest.raises(_lib.NrtGpuError):
        api.blend([np.array([1])], k=0)
    # score order: boost * score summed over the retrievers
    td = api.blend([np.array([1, 2]), np.array([2, 3])], [np.array([2.0, 1.0], np.float32), np.array([4.0, 0.5], np.float32)],
                   boosts=[1.0, 0.5], mode="score", top_hits=3)
    assert td.docs.tolist() == [2, 1, 3] and np.allclose(td.scores, [3.0, 2.0, 0.25])
    # against the Python mirror on random lists (distinct scores: tie order is the reference's containers', checked below)
    rng = np.random.default_rng(3)
    for _ in range(20):
        lists = [rng.choice(500, size=int(rng.integers(1, 60)), replace=False) for _ in range(int(rng.integers(1, 4)))]
        boosts = rng.uniform(0.5, 2.0, size=len(lists)).astype(np.float32)
        a = api.blend(lists, boosts=boosts, top_hits=25)
        b = api.weighted_rrf_blend(lists, boosts=boosts, top_hits=25)
        assert a.total_hits == b.total_hits and np.array_equal(a.scores, b.scores)
        distinct = len(set(a.scores.tolist())) == len(a.scores)
        assert (not distinct) or a.docs.tolist() == b.docs.tolist()
    # ties: every doc of one list of 3 scores differently, but two one-hit retrievers tie at 1/61: the survivor order is
    # java.util.HashMap's bucket order (doc 17 -> bucket 1, doc 33 -> bucket 1 after doc 17; doc 2 -> bucket 2) feeding the heap
    td = api.blend([np.array([33]), np.array([2]), np.array([17])], top_hits=3)
    assert sorted(td.docs.tolist()) == [2, 17, 33] and len(set(td.scores.tolist())) == 1


def test_packed_posting_word_arithmetic():
    """Host restatement of the packed-postings word (plan.h: kPack*; no device): the exception number of a posting is
    rebuilt from the 11 bits its word carries and the directory entry of its 2048-posting block, whatever the density of
    exceptions; doc offsets are relative to the 2^20-doc super-window a cell (shift <= 10) never leaves."""
    rng = np.random.Generator(np.random.PCG64(5))
    ESC_BASE, LOW, BLOCK = 1664, 2047, 2048
    for density in (0.0005, 0.2, 1.0):
        n = 50_000
        is_exc = rng.random(n) < density
        e_of = np.cumsum(is_exc) - 1                                  # exceptions numbered in posting order
        code = np.where(is_exc, ESC_BASE + (e_of & LOW), 7)          # what pack_write_kernel stores
        assert code.max() < 4096
        counts = np.add.reduceat(is_exc.astype(np.int64), np.arange(0, n, BLOCK))
        directory = np.concatenate([[0], np.cumsum(counts)])          # exclusive prefix per block (+ total)
        p = np.nonzero(is_exc)[0]
        e0 = directory[p // BLOCK]
        rebuilt = e0 + (((code[p] - ESC_BASE) - e0) & LOW)            # packed_escape_word
        assert np.array_equal(rebuilt, e_of[p])
    # doc offset: 20 bits inside the super-window; every cell of a packed segment lies inside one
    for shift in range(0, 11):
        for cell in (0, 1, 5, 1023):
            first_tile, last_tile = cell << shift, ((cell + 1) << shift) - 1
            assert (first_tile * 1024) >> 20 == (last_tile * 1024 + 1023) >> 20
    doc = np.array([0, 1, (1 << 20) - 1, 1 << 20, (1 << 20) + 77, 2_600_000 - 1])
    word = ((doc & ((1 << 20) - 1)) << 12) | 5
    assert np.array_equal((word >> 12) | (doc & ~((1 << 20) - 1)), doc) and word.max() < 2**32


def _header_prototypes():
    """include/nrtgpu.h -> {symbol: (return kind, [argument kinds])}, kinds: "ptr", "i32", "i64", "f32", "f64", "void"."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "nrtgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)

    def kind(decl):
        decl = decl.strip()
        if "*" in decl or "[" in decl:
            return "ptr"
        base = re.sub(r"\b(const|unsigned)\b", "", decl).split()
        t = base[0] if base else ""
        return {"int32_t": "i32", "int": "i32", "uint32_t": "i32", "int64_t": "i64", "uint64_t": "i64", "float": "f32", "double": "f64",
                "void": "void"}[t]

    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[\w]+(?:\s*\*)?)\s+(\*?)\s*(nrtgpu_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret = kind(m.group(1) + m.group(2) + " x")
        args = m.group(4).strip()
        kinds = [] if args in ("", "void") else [kind(a) for a in args.split(",") if a.strip()]
        protos[m.group(3)] = (ret, kinds)
    return protos


def test_ctypes_and_java_bindings_pass_what_the_header_declares(lib):
    """Every entry point the ctypes binding gives argument types for, and every downcall handle of java/.../NrtGpu.java, takes the
    arguments include/nrtgpu.h declares -- count AND kind (pointer, 32 / 64-bit integer, float, double): a binding that lags the
    header pushes garbage through the C ABI without any error."""
    import ctypes as C
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    protos = _header_prototypes()
    assert len(protos) > 40

    def ckind(t):
        if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return "ptr"
        return {C.c_int32: "i32", C.c_int: "i32", C.c_uint32: "i32", C.c_int64: "i64", C.c_uint64: "i64", C.c_float: "f32",
                C.c_double: "f64"}[t]

    checked = 0
    for name, (ret, kinds) in protos.items():
        fn = getattr(lib, name, None)
        if fn is None or fn.argtypes is None:
            continue
        got = [ckind(t) for t in fn.argtypes]
        assert got == kinds, f"{name}: the ctypes binding passes {got}, the header declares {kinds}"
        checked += 1
    assert checked > 30
    # java/.../NrtGpu.java: h("<symbol>", FunctionDescriptor.of(RET, args...)) / ofVoid(args...)
    jtext = open(os.path.join(root, "java", "src", "main", "java", "com", "yelp", "nrtsearch", "gpu", "NrtGpu.java")).read()
    jkind = {"ADDRESS": "ptr", "JAVA_INT": "i32", "JAVA_LONG": "i64", "JAVA_FLOAT": "f32", "JAVA_DOUBLE": "f64"}
    jchecked = 0
    for m in re.finditer(r'h\("(nrtgpu_\w+)",\s*FunctionDescriptor\.(of|ofVoid)\(([^;]*?)\)\);', jtext, flags=re.S):
        name, form, args = m.group(1), m.group(2), [a.strip() for a in m.group(3).split(",") if a.strip()]
        assert name in protos, f"NrtGpu.java binds {name}, which include/nrtgpu.h does not declare"
        ret, kinds = protos[name]
        if form == "of":
            assert jkind[args[0]] == ret, f"NrtGpu.java: {name} returns {args[0]}, the header says {ret}"
            args = args[1:]
        else:
            assert ret == "void", f"NrtGpu.java: {name} is bound as void, the header says {ret}"
        assert [jkind[a] for a in args] == kinds, f"NrtGpu.java: {name} passes {args}, the header declares {kinds}"
        jchecked += 1
    assert jchecked >= 20
