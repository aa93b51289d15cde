package com.yelp.nrtsearch.gpu;

import static java.lang.foreign.ValueLayout.*;

import java.io.IOException;
import java.lang.foreign.*;
import java.lang.invoke.MethodHandle;

/**
 * FFM binding of include/nrtgpu.h (x86-64 SysV).  No logic: one downcall handle per ABI function the plugin uses and the
 * struct layouts.  Field OFFSETS are never typed by hand: the shim's accessors go through {@link NrtGpuLayouts}, generated from
 * the header by scripts/gen_java_layouts.py (gcc's offsetof), and tests/test_java_layouts.py checks these StructLayouts member by
 * member against the same numbers.  NOT COMPILED in the image this repository is developed in (no JDK).
 */
final class NrtGpu {
  private NrtGpu() {}

  static final int OK = 0, ERR_INVALID_ARG = -1, ERR_HIP = -2, ERR_OOM = -3, ERR_UNSUPPORTED = -4, ERR_STATE = -5, ERR_TIMEOUT = -6;
  static final int MAX_K = 1024, MAX_TERMS = 32, MAX_MASKS = 8;

  private static final Linker L = Linker.nativeLinker();
  private static final SymbolLookup LIB =
      SymbolLookup.libraryLookup(System.getProperty("nrtgpu.library", "libnrtgpu.so"), Arena.global());

  private static MethodHandle h(String name, FunctionDescriptor d) {
    return L.downcallHandle(LIB.find(name).orElseThrow(() -> new UnsatisfiedLinkError(name)), d);
  }

  // typedef struct { int32 field_id, cache_slot; int64 term_hash; float weight; int32 occur; } nrtgpu_term;
  static final StructLayout TERM =
      MemoryLayout.structLayout(JAVA_INT.withName("field_id"), JAVA_INT.withName("cache_slot"), JAVA_LONG.withName("term_hash"),
          JAVA_FLOAT.withName("weight"), JAVA_INT.withName("occur"));
  // nrtgpu_bm25_query
  static final StructLayout QUERY =
      MemoryLayout.structLayout(JAVA_INT.withName("n_terms"), MemoryLayout.paddingLayout(4), ADDRESS.withName("terms"),
          JAVA_INT.withName("n_caches"), MemoryLayout.paddingLayout(4), ADDRESS.withName("norm_cache"), JAVA_INT.withName("k"),
          JAVA_INT.withName("total_hits_threshold"), JAVA_INT.withName("has_after"), JAVA_INT.withName("after_doc"),
          JAVA_FLOAT.withName("after_score"), JAVA_INT.withName("min_should_match"), JAVA_FLOAT.withName("min_competitive_score"),
          JAVA_INT.withName("filter_mask"), JAVA_INT.withName("must_not_mask"), JAVA_INT.withName("disjunction_max"),
          JAVA_INT.withName("n_more_filters"), MemoryLayout.paddingLayout(4), ADDRESS.withName("more_filters"),
          JAVA_INT.withName("n_more_must_not"), MemoryLayout.paddingLayout(4), ADDRESS.withName("more_must_not"),
          JAVA_FLOAT.withName("tie_breaker"), JAVA_INT.withName("reserved"));
  // nrtgpu_topdocs
  static final StructLayout TOPDOCS =
      MemoryLayout.structLayout(JAVA_INT.withName("n_hits"), JAVA_INT.withName("capacity"), ADDRESS.withName("docs"),
          ADDRESS.withName("scores"), JAVA_LONG.withName("total_hits"), JAVA_INT.withName("total_hits_is_lower_bound"),
          MemoryLayout.paddingLayout(4));
  // nrtgpu_config
  static final StructLayout CONFIG =
      MemoryLayout.structLayout(JAVA_INT.withName("device_id"), JAVA_INT.withName("max_batch"), JAVA_INT.withName("target_items"),
          JAVA_INT.withName("collect_timing"), JAVA_INT.withName("flags"), JAVA_INT.withName("host_threads"),
          JAVA_INT.withName("lookup_budget_pct"), JAVA_INT.withName("reserved"));

  // nrtgpu_diagnostics
  static final StructLayout DIAGNOSTICS =
      MemoryLayout.structLayout(JAVA_DOUBLE.withName("total_ms"), JAVA_DOUBLE.withName("plan_ms"), JAVA_DOUBLE.withName("queue_ms"),
          JAVA_DOUBLE.withName("device_ms"), JAVA_LONG.withName("postings"), JAVA_INT.withName("queries"),
          JAVA_INT.withName("items_maxscore"), JAVA_INT.withName("items_scan"), JAVA_INT.withName("reserved"));

  /** The request thread's deadline (absolute, on nrtgpu_monotonic_ns' clock; 0 = none) and what its last call cost. */
  static final MethodHandle SET_DEADLINE = h("nrtgpu_set_thread_deadline_ns", FunctionDescriptor.ofVoid(JAVA_LONG));
  static final MethodHandle MONOTONIC_NS = h("nrtgpu_monotonic_ns", FunctionDescriptor.of(JAVA_LONG));
  static final MethodHandle LAST_DIAGNOSTICS = h("nrtgpu_last_diagnostics", FunctionDescriptor.of(JAVA_INT, ADDRESS));

  static final MethodHandle CREATE = h("nrtgpu_create", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));
  static final MethodHandle DESTROY = h("nrtgpu_destroy", FunctionDescriptor.ofVoid(ADDRESS));
  static final MethodHandle LAST_ERROR = h("nrtgpu_last_error", FunctionDescriptor.of(ADDRESS));
  static final MethodHandle SET_SLICING = h("nrtgpu_set_slicing", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT));
  /** Partial residency: the calling thread's next searches run over a subset of the searcher's leaves, counted by the whole searcher's slices. */
  static final MethodHandle SET_THREAD_SLICES = h("nrtgpu_set_thread_slices", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT));
  static final MethodHandle SEG_BEGIN = h("nrtgpu_segment_begin", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS));
  static final MethodHandle ADD_NORMS = h("nrtgpu_segment_add_field_norms", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, ADDRESS));
  static final MethodHandle ADD_TERMS =
      h("nrtgpu_segment_add_terms", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
  static final MethodHandle ADD_VECTORS =
      h("nrtgpu_segment_add_vectors", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS));
  static final MethodHandle SEAL = h("nrtgpu_segment_seal", FunctionDescriptor.of(JAVA_INT, ADDRESS));
  static final MethodHandle SET_LIVE = h("nrtgpu_segment_set_live_docs", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT));
  static final MethodHandle FORK = h("nrtgpu_segment_fork", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
  static final MethodHandle SET_MASK = h("nrtgpu_segment_set_mask", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT));
  static final MethodHandle RELEASE = h("nrtgpu_segment_release", FunctionDescriptor.ofVoid(ADDRESS));
  static final MethodHandle SUPPORTED = h("nrtgpu_query_supported", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
  /** What a request thread calls: one query, blocks; concurrent callers are merged into device batches inside the library. */
  static final MethodHandle SEARCH1 =
      h("nrtgpu_search_bm25_coalesced", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS));
  static final MethodHandle SEARCH_BATCH =
      h("nrtgpu_search_bm25_batch", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS));
  /**
   * ExactVectorQuery: every doc with a vector scored, the k best (the oracle-order fp32 similarity of each hit).  One query per
   * call, blocks; concurrent request threads are merged into panels of up to 64 queries that share a pass over the rows.
   */
  static final MethodHandle KNN_EXACT1 = h("nrtgpu_knn_exact_coalesced", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT,
      JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_FLOAT, ADDRESS));
  /** TotalHits.relation of an exact vector query by the reference's per-slice rule (host only): 1 = GREATER_THAN_OR_EQUAL_TO. */
  static final MethodHandle KNN_RELATION = h("nrtgpu_knn_exact_relation", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT,
      JAVA_INT, JAVA_INT, JAVA_INT));
  static final MethodHandle KNN_SEARCH = h("nrtgpu_knn_search", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT,
      JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_FLOAT, JAVA_INT, JAVA_FLOAT, ADDRESS));
  static final MethodHandle RESCORE = h("nrtgpu_rescore_vectors", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_INT,
      JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, JAVA_FLOAT, ADDRESS, ADDRESS, JAVA_INT, JAVA_DOUBLE, JAVA_DOUBLE, JAVA_INT, ADDRESS));

  /** Speculative thresholds of the MaxScore route (results stay exact: a query whose guess fails is run again inside the call):
   *  the guess's safety margin in standard deviations; 0 switches them off for the context -- e.g. a live setting for an index
   *  sorted by a field the score follows (the library also gives up by itself when too many guesses fail). */
  static final MethodHandle SET_SPECULATION = h("nrtgpu_set_speculation", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_FLOAT));
  // one process per GPU over virtual shards: reader.numDocs() of this shard's leaves against the whole searcher's
  static final MethodHandle SET_SHARD_SHARE = h("nrtgpu_set_shard_share", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, JAVA_LONG));

  static String lastError() {
    try {
      MemorySegment p = (MemorySegment) LAST_ERROR.invokeExact();
      return p.reinterpret(4096).getString(0);
    } catch (Throwable t) {
      return "nrtgpu_last_error failed: " + t;
    }
  }

  /** NRTGPU_ERR_* -> the exceptions the reference throws at the same places (INTEGRATION.md, error mapping). */
  static void check(int rc) throws IOException {
    switch (rc) {
      case OK -> {}
      case ERR_INVALID_ARG -> throw new IllegalArgumentException(lastError());   // LazyQueueTopScoreDocCollectorManager.java:90-98
      case ERR_STATE -> throw new IllegalStateException(lastError());
      default -> throw new IOException(lastError());                               // -> SearchHandlerException -> gRPC INTERNAL
    }
  }
}
