/*
 * nrtgpu_dev.h -- measurement helpers of the development build (-DNRTGPU_DEV, libnrtgpu_dev.so).  NOT part of the
 * drop-in boundary (include/nrtgpu.h): the product library exports none of this, and rejects the timing-ablation
 * values of nrtgpu_config.flags bits 8-11 (kernels.hip: variants 1-4 and 6 drop work to time the rest and return
 * wrong results).
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
#ifdef __cplusplus
}
#endif
#endif /* NRTGPU_DEV_H */
