"""ctypes binding of libnrtgpu.so -- the same C ABI (include/nrtgpu.h) the Java JNI/FFM shim binds.

Fails loudly: there is no Python / CPU implementation behind these calls.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NRTGPU_LIB_PATH") or os.path.join(_HERE, "libnrtgpu.so")  # override: A/B builds of the kernels

NRTGPU_OK = 0
NRTGPU_ERR_INVALID_ARG = -1
NRTGPU_ERR_HIP = -2
NRTGPU_ERR_OOM = -3
NRTGPU_ERR_UNSUPPORTED = -4
NRTGPU_ERR_STATE = -5
NRTGPU_ERR_TIMEOUT = -6
NRTGPU_MAX_K = 1024
NRTGPU_MAX_TERMS = 32
NRTGPU_TILE_DOCS = 1024
NRTGPU_FLAG_NO_PREFETCH = 1
NRTGPU_FLAG_NO_FIXED_POINT = 2
NRTGPU_FLAG_NO_MASK_VARIANT = 4
NRTGPU_FLAG_NO_LIVE_FOLD = 8
NRTGPU_FLAG_NO_PRUNE = 16
NRTGPU_FLAG_PACKED_POSTINGS = 32
NRTGPU_FLAG_BLOCKING_WAIT = 64
NRTGPU_FLAG_NO_VECTOR_SKETCH = 128
NRTGPU_FLAG_PROFILE = 7 << 8   # include/nrtgpu_dev.h: the development library only
NRTGPU_MAX_MASKS = 8

# every symbol include/nrtgpu.h declares (tests/test_abi.py checks the header against this list)
ABI_SYMBOLS = [
    "nrtgpu_version", "nrtgpu_last_error", "nrtgpu_create", "nrtgpu_destroy",
    "nrtgpu_segment_begin", "nrtgpu_segment_add_field_norms", "nrtgpu_segment_add_terms",
    "nrtgpu_segment_add_vectors", "nrtgpu_segment_seal", "nrtgpu_segment_set_live_docs",
    "nrtgpu_segment_set_mask", "nrtgpu_segment_release", "nrtgpu_segment_device_bytes",
    "nrtgpu_search_bm25", "nrtgpu_search_bm25_batch", "nrtgpu_search_bm25_batch_device",
    "nrtgpu_search_bm25_batch_device_epoch", "nrtgpu_exchange_open", "nrtgpu_exchange_close",
    "nrtgpu_search_bm25_coalesced", "nrtgpu_set_coalescing", "nrtgpu_query_supported",
    "nrtgpu_merge_topk_device", "nrtgpu_knn_exact", "nrtgpu_knn_exact_coalesced", "nrtgpu_knn_search", "nrtgpu_rescore_vectors", "nrtgpu_search_hybrid_batch",
    "nrtgpu_int_to_byte4", "nrtgpu_byte4_to_int", "nrtgpu_bm25_idf", "nrtgpu_bm25_avgdl",
    "nrtgpu_bm25_norm_cache", "nrtgpu_slices", "nrtgpu_plan_item_counts", "nrtgpu_fixed_point_scale", "nrtgpu_get_stats", "nrtgpu_reset_stats",
    "nrtgpu_set_slicing",
    "nrtgpu_blend", "nrtgpu_dist_unique_id", "nrtgpu_dist_init", "nrtgpu_dist_search_bm25_batch", "nrtgpu_dist_allgather_merge", "nrtgpu_segment_fork",
    "nrtgpu_search_bm25_batch_device_begin", "nrtgpu_pending_wait", "nrtgpu_set_thread_deadline_ns", "nrtgpu_monotonic_ns", "nrtgpu_last_diagnostics", "nrtgpu_dist_close", "nrtgpu_dist_owned_range", "nrtgpu_dist_search_bm25_batch_mode", "nrtgpu_dist_exchange_merge", "nrtgpu_dist_exchange_merge_checked", "nrtgpu_search_bm25_shard_device_begin", "nrtgpu_note_shard_speculation", "nrtgpu_dist_knn_exact", "nrtgpu_dist_search_hybrid_batch",
    "nrtgpu_knn_exact_relation", "nrtgpu_set_speculation", "nrtgpu_set_shard_share", "nrtgpu_set_thread_slices",
]
# what include/nrtgpu_dev.h adds: test hooks and measurement helpers of the development library (libnrtgpu_dev.so) only
DEV_SYMBOLS = [
    "nrtgpu_bench_closed_loop", "nrtgpu_debug_hold_coalescers", "nrtgpu_debug_coalescer_pending", "nrtgpu_debug_live_segments",
    "nrtgpu_debug_spec_counters", "nrtgpu_get_scan_profile", "nrtgpu_get_maxscore_profile", "nrtgpu_get_maxscore_item_walls",
]
DEV_LIB_PATH = os.path.join(_HERE, "libnrtgpu_dev.so")


class Config(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("max_batch", C.c_int32), ("target_items", C.c_int32),
                ("collect_timing", C.c_int32), ("flags", C.c_int32), ("host_threads", C.c_int32),
                ("lookup_budget_pct", C.c_int32), ("reserved", C.c_int32)]


class Term(C.Structure):
    _fields_ = [("field_id", C.c_int32), ("cache_slot", C.c_int32), ("term_hash", C.c_int64),
                ("weight", C.c_float), ("occur", C.c_int32)]   # occur: 0 SHOULD, 1 MUST


class Bm25Query(C.Structure):
    _fields_ = [("n_terms", C.c_int32), ("terms", C.POINTER(Term)), ("n_caches", C.c_int32),
                ("norm_cache", C.POINTER(C.c_float)), ("k", C.c_int32), ("total_hits_threshold", C.c_int32),
                ("has_after", C.c_int32), ("after_doc", C.c_int32), ("after_score", C.c_float),
                ("min_should_match", C.c_int32), ("min_competitive_score", C.c_float), ("filter_mask", C.c_int32),
                ("must_not_mask", C.c_int32), ("disjunction_max", C.c_int32),
                ("n_more_filters", C.c_int32), ("more_filters", C.POINTER(C.c_int32)),
                ("n_more_must_not", C.c_int32), ("more_must_not", C.POINTER(C.c_int32)),
                ("tie_breaker", C.c_float), ("reserved", C.c_int32)]


class TopDocs(C.Structure):
    _fields_ = [("n_hits", C.c_int32), ("capacity", C.c_int32), ("docs", C.POINTER(C.c_int32)),
                ("scores", C.POINTER(C.c_float)), ("total_hits", C.c_int64),
                ("total_hits_is_lower_bound", C.c_int32)]


class Diagnostics(C.Structure):   # nrtgpu_diagnostics
    _fields_ = [("total_ms", C.c_double), ("plan_ms", C.c_double), ("queue_ms", C.c_double), ("device_ms", C.c_double),
                ("postings", C.c_int64), ("queries", C.c_int32), ("items_maxscore", C.c_int32), ("items_scan", C.c_int32),
                ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("batches", C.c_int64), ("queries", C.c_int64), ("scan_launches", C.c_int64),
                ("scan_ms", C.c_double), ("scan_postings", C.c_int64), ("scan_items", C.c_int64),
                ("merge_ms", C.c_double), ("host_plan_ms", C.c_double), ("fixed_point_launches", C.c_int64),
                ("maxscore_launches", C.c_int64), ("maxscore_ms", C.c_double), ("maxscore_postings", C.c_int64),
                ("maxscore_items", C.c_int64), ("knn_panels", C.c_int64), ("knn_score_launches", C.c_int64),
                ("knn_score_ms", C.c_double), ("knn_rows", C.c_int64), ("knn_second_passes", C.c_int64), ("knn_sketch_launches", C.c_int64),
                ("spec_queries", C.c_int64), ("spec_reruns", C.c_int64), ("spec_disabled", C.c_int64), ("spec_scattered", C.c_int64)]


class NrtGpuError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"nrtgpu error {code}: {message}")
        self.code = code


_lib = None
_dev = None


def load() -> C.CDLL:
    """dlopen the in-tree library; raises if it has not been built (python -m nrtsearch_amd.build)."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def load_dev() -> C.CDLL:
    """The development library (include/nrtgpu_dev.h: the product sources + test hooks, instrumented kernels, experiment knobs).
    tests/conftest.py's `dev_lib` fixture makes it the library api.py talks to for the tests that need a hook."""
    global _dev
    if _dev is None:
        _dev = _open(DEV_LIB_PATH)
    return _dev


def _open(path: str) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m nrtsearch_amd.build` "
            "(hipcc, gfx950). There is no Python/CPU fallback for the query path.")
    L = C.CDLL(path)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.nrtgpu_version.restype = C.c_char_p
    L.nrtgpu_last_error.restype = C.c_char_p
    L.nrtgpu_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.nrtgpu_destroy.argtypes = [vp]
    L.nrtgpu_destroy.restype = None
    L.nrtgpu_segment_begin.argtypes = [vp, i32, i32, C.POINTER(vp)]
    L.nrtgpu_segment_add_field_norms.argtypes = [vp, i32, vp]
    L.nrtgpu_segment_add_terms.argtypes = [vp, i32, i64, vp, vp, vp, vp]
    L.nrtgpu_segment_add_vectors.argtypes = [vp, i32, i32, i32, vp, vp]
    L.nrtgpu_segment_seal.argtypes = [vp]
    L.nrtgpu_segment_set_live_docs.argtypes = [vp, vp, i32]
    L.nrtgpu_segment_set_mask.argtypes = [vp, i32, vp, i32]
    L.nrtgpu_segment_release.argtypes = [vp]
    L.nrtgpu_segment_release.restype = None
    L.nrtgpu_segment_device_bytes.argtypes = [vp]
    L.nrtgpu_segment_device_bytes.restype = i64
    L.nrtgpu_search_bm25.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), C.POINTER(TopDocs)]
    L.nrtgpu_search_bm25_batch.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), i32, C.POINTER(TopDocs)]
    L.nrtgpu_search_bm25_batch_device.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), i32, i32, vp, vp, vp]
    L.nrtgpu_search_bm25_batch_device_epoch.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), i32, i32, vp, vp, vp, i64]
    L.nrtgpu_search_bm25_coalesced.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), C.POINTER(TopDocs)]
    L.nrtgpu_set_coalescing.argtypes = [vp, i32]
    if hasattr(L, "nrtgpu_bench_closed_loop"):   # development build only (include/nrtgpu_dev.h)
        L.nrtgpu_bench_closed_loop.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), i32, i32, i32, vp]
    L.nrtgpu_query_supported.argtypes = [vp, vp, i32, C.POINTER(Bm25Query)]
    L.nrtgpu_exchange_open.argtypes = [vp, C.c_char_p, i32, i32]
    L.nrtgpu_exchange_close.argtypes = [vp]
    L.nrtgpu_exchange_close.restype = None
    L.nrtgpu_merge_topk_device.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, C.POINTER(TopDocs)]
    L.nrtgpu_knn_exact.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, i32, i32, f32, C.POINTER(TopDocs)]
    L.nrtgpu_knn_exact_coalesced.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, i32, f32, C.POINTER(TopDocs)]
    L.nrtgpu_knn_exact_coalesced.restype = i32
    L.nrtgpu_knn_search.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, i32, i32, f32, i32, f32, C.POINTER(TopDocs)]
    L.nrtgpu_rescore_vectors.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, f32, vp, vp, i32, C.c_double, C.c_double, i32,
                                         C.POINTER(TopDocs)]
    L.nrtgpu_search_hybrid_batch.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), i32, i32, i32, vp, i32, f32,
                                             C.c_double, C.c_double, i32, C.POINTER(TopDocs)]
    L.nrtgpu_int_to_byte4.argtypes = [i32]
    L.nrtgpu_byte4_to_int.argtypes = [i32]
    L.nrtgpu_bm25_idf.argtypes = [i64, i64]
    L.nrtgpu_bm25_idf.restype = f32
    L.nrtgpu_bm25_avgdl.argtypes = [i64, i64]
    L.nrtgpu_bm25_avgdl.restype = f32
    L.nrtgpu_bm25_norm_cache.argtypes = [f32, f32, f32, vp]
    L.nrtgpu_bm25_norm_cache.restype = None
    L.nrtgpu_plan_item_counts.argtypes = [i32, vp, i32, vp]
    L.nrtgpu_fixed_point_scale.argtypes = [f32, vp, i32, vp]
    L.nrtgpu_slices.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp, vp]
    L.nrtgpu_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.nrtgpu_reset_stats.argtypes = [vp]
    L.nrtgpu_set_thread_slices.argtypes = [vp, i32]
    L.nrtgpu_set_speculation.argtypes = [vp, C.c_float]
    L.nrtgpu_set_speculation.restype = C.c_int
    L.nrtgpu_set_shard_share.argtypes = [vp, C.c_int64, C.c_int64]
    L.nrtgpu_set_shard_share.restype = C.c_int
    L.nrtgpu_knn_exact_relation.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.nrtgpu_reset_stats.restype = None
    if hasattr(L, "nrtgpu_debug_hold_coalescers"):   # development build only (include/nrtgpu_dev.h)
        L.nrtgpu_debug_hold_coalescers.argtypes = [vp, C.c_int32]
        L.nrtgpu_debug_live_segments.argtypes = [vp]
        L.nrtgpu_debug_live_segments.restype = C.c_int64
        L.nrtgpu_debug_spec_counters.argtypes = [vp, vp]
        L.nrtgpu_debug_spec_counters.restype = C.c_int
        L.nrtgpu_get_maxscore_item_walls.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int64)]
        L.nrtgpu_get_maxscore_item_walls.restype = C.c_int64
        L.nrtgpu_debug_coalescer_pending.argtypes = [vp, C.c_int32]
        L.nrtgpu_get_scan_profile.argtypes = [vp, vp]
        L.nrtgpu_get_maxscore_profile.argtypes = [vp, vp]
    L.nrtgpu_set_slicing.argtypes = [vp, i32, i32, i32]
    L.nrtgpu_blend.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, C.POINTER(TopDocs)]
    L.nrtgpu_dist_unique_id.argtypes = [vp]
    L.nrtgpu_dist_init.argtypes = [vp, i32, i32, vp]
    L.nrtgpu_dist_search_bm25_batch.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), i32, C.POINTER(TopDocs)]
    L.nrtgpu_dist_allgather_merge.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, C.POINTER(TopDocs)]
    L.nrtgpu_segment_fork.argtypes = [vp, vp, i32, C.POINTER(vp)]
    L.nrtgpu_search_bm25_batch_device_begin.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), i32, i32, vp, vp, vp, C.c_int64, vp]
    L.nrtgpu_pending_wait.argtypes = [vp]
    L.nrtgpu_set_thread_deadline_ns.argtypes = [C.c_int64]
    L.nrtgpu_set_thread_deadline_ns.restype = None
    L.nrtgpu_monotonic_ns.argtypes = []
    L.nrtgpu_monotonic_ns.restype = C.c_int64
    L.nrtgpu_last_diagnostics.argtypes = [vp]
    L.nrtgpu_dist_owned_range.argtypes = [vp, i32, i32, vp, vp]
    L.nrtgpu_dist_search_bm25_batch_mode.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), i32, i32, C.POINTER(TopDocs)]
    L.nrtgpu_dist_exchange_merge.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, i32, C.POINTER(TopDocs)]
    L.nrtgpu_dist_exchange_merge_checked.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, C.POINTER(TopDocs), vp, vp]
    L.nrtgpu_search_bm25_shard_device_begin.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), i32, i32, vp, vp, vp, i32, vp, vp]
    L.nrtgpu_note_shard_speculation.argtypes = [vp, vp, i32, i32, i32]
    L.nrtgpu_dist_knn_exact.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, i32, i32, C.c_float, i32, C.POINTER(TopDocs)]
    L.nrtgpu_dist_search_hybrid_batch.argtypes = [vp, vp, vp, i32, C.POINTER(Bm25Query), i32, i32, i32, vp, i32, C.c_float, C.c_double, C.c_double,
                                                  i32, i32, C.POINTER(TopDocs)]
    L.nrtgpu_dist_close.argtypes = [vp]
    L.nrtgpu_dist_close.restype = None
    return L


def check(rc: int) -> None:
    if rc != NRTGPU_OK:
        raise NrtGpuError(rc, load().nrtgpu_last_error().decode("utf-8", "replace"))
