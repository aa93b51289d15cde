// devtools.cpp -- measurement helpers that are NOT part of the product ABI: compiled into the library only with
// -DNRTGPU_DEV (python -m nrtsearch_amd.build --dev writes libnrtgpu_dev.so); declared in include/nrtgpu_dev.h.
#include "runtime_internal.h"
#include "../../include/nrtgpu_dev.h"

// Closed-loop load generator (diagnostics; SURVEY 8d's "C concurrent clients"): `clients` native threads each
// issue one query at a time through nrtgpu_search_bm25_coalesced for duration_ms, cycling through `queries`.
// out[0] = completed queries, out[1] = seconds, out[2] = p50 latency ms, out[3] = p99 latency ms.
extern "C" int nrtgpu_bench_closed_loop(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                        int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                        int32_t clients, int32_t duration_ms, double* out4) {
  if (!ctx || !queries || !out4 || n_queries <= 0 || clients <= 0 || duration_ms <= 0)
    return fail(NRTGPU_ERR_INVALID_ARG, "bad closed-loop arguments");
  std::vector<std::vector<float>> lat((size_t)clients);
  std::vector<int> rcs((size_t)clients, 0);
  std::vector<std::string> errs((size_t)clients);
  const auto t_begin = std::chrono::steady_clock::now();
  const auto t_stop = t_begin + std::chrono::milliseconds(duration_ms);
  auto client = [&](int c) {
    int32_t kmax = 1;
    for (int i = 0; i < n_queries; ++i) kmax = std::max(kmax, queries[i].k);
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
      const int rc = nrtgpu_search_bm25_coalesced(ctx, segs, doc_bases, n_segs, &queries[i % (size_t)n_queries], &o);
      if (rc != 0) {
        rcs[(size_t)c] = rc;
        errs[(size_t)c] = g_last_error;
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
    if (rcs[(size_t)c] != 0) return fail(rcs[(size_t)c], "client %d: %s", c, errs[(size_t)c].c_str());
  std::vector<float> all;
  for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
  std::sort(all.begin(), all.end());
  out4[0] = (double)all.size();
  out4[1] = secs;
  out4[2] = all.empty() ? 0.0 : all[all.size() / 2];
  out4[3] = all.empty() ? 0.0 : all[(size_t)((double)all.size() * 0.99)];
  return NRTGPU_OK;
}

