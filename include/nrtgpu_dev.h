/*
 * nrtgpu_dev.h -- test hooks and measurement helpers of the development build (-DNRTGPU_DEV, libnrtgpu_dev.so).  NOT part
 * of the drop-in boundary (include/nrtgpu.h -- the header the Java binding mirrors): the product library exports none of
 * this, compiles no instrumented kernel, reads no experiment knob from the environment (runtime_internal.h: dev_env_*), and
 * rejects every non-zero value of nrtgpu_config.flags bits 8-11 (7 = instrumented kernels, same results; kernels.hip:
 * variants 1-4 and 6 drop work to time the rest and return wrong results).  The GPU tests that need a hook run against the
 * development library (tests/conftest.py: dev_lib) -- the same sources with these few functions added.
 */
#ifndef NRTGPU_DEV_H
#define NRTGPU_DEV_H
#include "nrtgpu.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Closed-loop load generator (C concurrent clients): `clients` native threads each issue one query at a
 * time through nrtgpu_search_bm25_coalesced for duration_ms, cycling through `queries`.
 * out4 = {completed queries, elapsed seconds, p50 latency ms, p99 latency ms}. */
int  nrtgpu_bench_closed_loop(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                              const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t clients, int32_t duration_ms,
                              double* out4);
/* TEST HOOKS for the two coalescers (this one and nrtgpu_knn_exact_coalesced).  hold != 0: no leader leaves with less than a
 * full batch (max_batch queries) / panel (64 queries) until the hold is released -- a test queues a known set of callers behind
 * it, waits until nrtgpu_debug_coalescer_pending (which: 0 = BM25, 1 = exact vector search) reports all of them, releases,
 * and may then assert the batches formed BY CONSTRUCTION (what a batch is must not depend on the host's speed).  Not for
 * production callers: a held coalescer parks every request. */
int  nrtgpu_debug_hold_coalescers(nrtgpu_ctx* ctx, int32_t hold);
int  nrtgpu_debug_coalescer_pending(nrtgpu_ctx* ctx, int32_t which);
/* TEST HOOK: segment handles of the context (uploads and forks) that have not been freed yet.  nrtgpu_segment_release under
 * running searches defers the free to the last of them: this count is how a test observes that it happened. */
int64_t nrtgpu_debug_live_segments(nrtgpu_ctx* ctx);
/* Speculative thresholds (nrtgpu_set_speculation): out3 = {queries run under speculation, queries whose guess failed the
 * merge's check and were run again, 1 once the library has switched speculation off for this context} -- the three
 * nrtgpu_stats.spec_* values on their own. */
int  nrtgpu_debug_spec_counters(nrtgpu_ctx* ctx, int64_t* out3);

#define NRTGPU_FLAG_PROFILE (7 << 8)  /* instrumented kernels (same results): per-item phase cycle and event counters
                                       * (nrtgpu_get_scan_profile, nrtgpu_get_maxscore_profile).  Bits 8-11 hold no other
                                       * value in the product library: nrtgpu_create rejects them */
/* sums over all items since the last reset, instrumented kernel only (wave 0 of each workgroup;
 * cycles = shader clock): [0] prologue cycles, [1] cycles waiting at the rendezvous barrier,
 * [2] rendezvous cycles incl. that wait, [3] walk cycles, [4] epilogue cycles, [5] rendezvous,
 * [6] compactions, [7] sub-tiles, [8] rendezvous: selection cycles, [9] sparse sub-tiles (collected
 * through the postings), [10] rendezvous: keep cycles, [11] rendezvous: publish + append cycles, [12] sub-tiles with candidates,
 * [13] sub-tiles with a possibly competitive doc, [14] last wave's finish cycle, [15] first wave's */
int  nrtgpu_get_scan_profile(nrtgpu_ctx* ctx, double* out16);
/* the same flag, items of the MaxScore route; sums over items since the last reset: [0] doc windows walked, [1] top-k
 * compactions, [2] posting chunks (512 postings), [3] postings streamed, [4] postings whose bound reached theta,
 * [5] docs evaluated, [6] lookups in later clauses, [7] candidates collected; shader-clock cycles: [8] item prologue,
 * [9] whole item, [10] waves in meetings (waiting + compaction), [11] waves out of windows waiting for the item's end,
 * [12] waves in part prologues, [13] waves walking windows, [14] the item's last wave running out of windows, [15] item
 * epilogue ([10]-[13]: summed over the item's 12 waves) */
int  nrtgpu_get_maxscore_profile(nrtgpu_ctx* ctx, double* out16);
/* the same flag: WHEN the pieces of the last MaxScore launch ran.  Per output slot eight words -- {start, end} on the device's
 * 100 MHz wall clock, the item worked on, the doc windows walked, when the workgroup's round began (persistent workgroups choose
 * work round after round), the CU (XCC << 8 | SE, SH, CU), the round, the workgroup; a slot nobody used is all zeros.  The first
 * *n_items slots are the items' owners; slots beyond the call's items (MaxScore + scan) are HELPERS: workgroups that shared the
 * windows of an item someone else owns.  Returns the number of slots (<= cap_slots are written); the makespan of the
 * launch against its balanced load is max(end) - min(start) vs sum(end - start) / CUs. */
int64_t nrtgpu_get_maxscore_item_walls(nrtgpu_ctx* ctx, uint64_t* out, int64_t cap_slots, int64_t* n_items);

#ifdef __cplusplus
}
#endif
#endif /* NRTGPU_DEV_H */
