// loadgen.cpp -- closed-loop load generator for bench.py ("C concurrent clients"): measurement tooling OUTSIDE the
// product library.  It knows nothing but the C ABI (include/nrtgpu.h): `clients` native threads each issue ONE query at a time
// through the entry point it is handed -- nrtgpu_search_bm25_coalesced of the product library bench.py has loaded -- for
// duration_ms, cycling through `queries`.  Python threads would measure the GIL.
//   out4 = {completed queries, elapsed seconds, p50 latency ms, p99 latency ms}; returns the first non-zero status of a client.
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/nrtgpu.h"

typedef int (*coalesced_fn)(nrtgpu_ctx*, const nrtgpu_seg* const*, const int32_t*, int32_t, const nrtgpu_bm25_query*, nrtgpu_topdocs*);

extern "C" int loadgen_closed_loop(void* search_coalesced, nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                   int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t clients,
                                   int32_t duration_ms, double* out4) {
  if (!search_coalesced || !ctx || !queries || !out4 || n_queries <= 0 || clients <= 0 || duration_ms <= 0) return NRTGPU_ERR_INVALID_ARG;
  const coalesced_fn search = (coalesced_fn)search_coalesced;
  std::vector<std::vector<float>> lat((size_t)clients);
  std::vector<int> rcs((size_t)clients, 0);
  const auto t_begin = std::chrono::steady_clock::now();
  const auto t_stop = t_begin + std::chrono::milliseconds(duration_ms);
  int32_t kmax = 1;
  for (int i = 0; i < n_queries; ++i) kmax = std::max(kmax, queries[i].k);
  auto client = [&](int c) {
    std::vector<int32_t> docs((size_t)kmax);
    std::vector<float> scores((size_t)kmax);
    size_t i = (size_t)c * 7919u;
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      if (t0 >= t_stop) break;
      nrtgpu_topdocs o{};
      o.capacity = kmax;
      o.docs = docs.data();
      o.scores = scores.data();
      const int rc = search(ctx, segs, doc_bases, n_segs, &queries[i % (size_t)n_queries], &o);
      if (rc != 0) {
        rcs[(size_t)c] = rc;
        break;
      }
      lat[(size_t)c].push_back(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
      ++i;
    }
  };
  std::vector<std::thread> pool;
  for (int c = 0; c < clients; ++c) pool.emplace_back(client, c);
  for (auto& t : pool) t.join();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  for (int c = 0; c < clients; ++c)
    if (rcs[(size_t)c] != 0) return rcs[(size_t)c];
  std::vector<float> all;
  for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
  std::sort(all.begin(), all.end());
  out4[0] = (double)all.size();
  out4[1] = secs;
  out4[2] = all.empty() ? 0.0 : all[all.size() / 2];
  out4[3] = all.empty() ? 0.0 : all[(size_t)((double)all.size() * 0.99)];
  return 0;
}

// The same for exact vector searches: `clients` threads, each ONE query at a time through nrtgpu_knn_exact_coalesced, cycling
// through n_queries rows of `queries` (row-major, dim floats each).
typedef int (*knn_coalesced_fn)(nrtgpu_ctx*, const nrtgpu_seg* const*, const int32_t*, int32_t, int32_t, int32_t, const float*, int32_t, int32_t,
                                float, nrtgpu_topdocs*);

extern "C" int loadgen_closed_loop_knn(void* knn_coalesced, nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                       int32_t n_segs, int32_t field_id, int32_t similarity, const float* queries, int32_t n_queries,
                                       int32_t dim, int32_t k, int32_t clients, int32_t duration_ms, double* out4) {
  if (!knn_coalesced || !ctx || !queries || !out4 || n_queries <= 0 || clients <= 0 || duration_ms <= 0 || k <= 0) return NRTGPU_ERR_INVALID_ARG;
  const knn_coalesced_fn search = (knn_coalesced_fn)knn_coalesced;
  std::vector<std::vector<float>> lat((size_t)clients);
  std::vector<int> rcs((size_t)clients, 0);
  const auto t_begin = std::chrono::steady_clock::now();
  const auto t_stop = t_begin + std::chrono::milliseconds(duration_ms);
  auto client = [&](int c) {
    std::vector<int32_t> docs((size_t)k);
    std::vector<float> scores((size_t)k);
    size_t i = (size_t)c * 7919u;
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      if (t0 >= t_stop) break;
      nrtgpu_topdocs o{};
      o.capacity = k;
      o.docs = docs.data();
      o.scores = scores.data();
      const int rc = search(ctx, segs, doc_bases, n_segs, field_id, similarity, queries + (i % (size_t)n_queries) * (size_t)dim, dim, k, 1.0f, &o);
      if (rc != 0) {
        rcs[(size_t)c] = rc;
        break;
      }
      lat[(size_t)c].push_back(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
      ++i;
    }
  };
  std::vector<std::thread> pool;
  for (int c = 0; c < clients; ++c) pool.emplace_back(client, c);
  for (auto& t : pool) t.join();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  for (int c = 0; c < clients; ++c)
    if (rcs[(size_t)c] != 0) return rcs[(size_t)c];
  std::vector<float> all;
  for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
  std::sort(all.begin(), all.end());
  out4[0] = (double)all.size();
  out4[1] = secs;
  out4[2] = all.empty() ? 0.0 : all[all.size() / 2];
  out4[3] = all.empty() ? 0.0 : all[(size_t)((double)all.size() * 0.99)];
  return 0;
}
