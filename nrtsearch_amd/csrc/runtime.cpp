// runtime.cpp -- core of the host runtime behind the C ABI of include/nrtgpu.h (compiled with hipcc): errors,
// the device context (one per process/GPU), the per-call workspaces ("slots": stream + pinned staging + device
// scratch), statistics, the cross-GPU exchange table and the host-side restatements the shim may call.
// The segment store, the planner, the search entry points and the vector entry points are segment.cpp,
// planner.cpp, search.cpp and vectors.cpp (shared declarations: runtime_internal.h).
// There is deliberately no CPU execution path here: without a gfx950 device nrtgpu_create fails.
#include "runtime_internal.h"

namespace nrtgpu {
namespace rt {
// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
thread_local std::string g_last_error;
thread_local int64_t g_deadline_ns = 0;
thread_local nrtgpu_diagnostics g_diag{};
thread_local std::vector<int32_t> g_thread_slices;
int64_t monotonic_ns() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000ll + (int64_t)ts.tv_nsec;
}

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}


double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static const int kSlots = 4;

int acquire_slot(nrtgpu_ctx* ctx, Slot** out) {
  std::unique_lock<std::mutex> lk(ctx->mu);
  for (;;) {
    for (auto& s : ctx->slots) {
      if (!s->busy) {
        s->busy = true;
        *out = s.get();
        return 0;
      }
    }
    ctx->cv.wait(lk);
  }
}
void release_slot(nrtgpu_ctx* ctx, Slot* s) {
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    s->busy = false;
  }
  ctx->cv.notify_one();
}

}  // namespace rt
}  // namespace nrtgpu

// ------------------------------------------------------------------------------------------------
// ABI: context
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// helper threads
// ------------------------------------------------------------------------------------------------
WorkPool::WorkPool(int helpers) {
  for (int i = 0; i < helpers; ++i)
    threads_.emplace_back([this] {
      (void)pthread_setname_np(pthread_self(), "nrtgpu-helper");   // (shows up in /proc/<pid>/task/<tid>/comm: who burns the CPUs)
      loop();
    });
}
WorkPool::~WorkPool() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
  }
  cv_.notify_all();
  for (auto& t : threads_) t.join();
}
void WorkPool::loop() {
  for (;;) {
    std::shared_ptr<Job> job;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] {
        if (stop_) return true;
        for (auto& j : jobs_)
          if (j->next.load(std::memory_order_relaxed) < j->n) return true;
        return false;
      });
      if (stop_) return;
      for (auto& j : jobs_)
        if (j->next.load(std::memory_order_relaxed) < j->n) {
          job = j;
          break;
        }
    }
    if (!job) continue;
    for (;;) {
      const int i = job->next.fetch_add(1, std::memory_order_acq_rel);
      if (i >= job->n) break;
      (*job->fn)(i);
      job->done.fetch_add(1, std::memory_order_acq_rel);
    }
  }
}
void WorkPool::run(int n, const std::function<void(int)>& fn) {
  if (n <= 0) return;
  if (n == 1 || threads_.empty()) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  auto job = std::make_shared<Job>();
  job->fn = &fn;
  job->n = n;
  {
    std::lock_guard<std::mutex> lk(mu_);
    jobs_.push_back(job);
  }
  cv_.notify_all();
  for (;;) {  // the caller works too
    const int i = job->next.fetch_add(1, std::memory_order_acq_rel);
    if (i >= n) break;
    fn(i);
    job->done.fetch_add(1, std::memory_order_acq_rel);
  }
  while (job->done.load(std::memory_order_acquire) < n) std::this_thread::yield();  // helpers finishing their last chunk
  std::lock_guard<std::mutex> lk(mu_);
  jobs_.erase(std::find(jobs_.begin(), jobs_.end(), job));
}

Launcher::Launcher(int device) {
  th = std::thread([this, device] {
    (void)hipSetDevice(device);
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;   // (stop, and nothing left to enqueue)
        f = std::move(q.front());
        q.pop_front();
      }
      f();
    }
  });
}
Launcher::~Launcher() {
  {
    std::lock_guard<std::mutex> lk(mu);
    stop = true;
  }
  cv.notify_all();
  if (th.joinable()) th.join();
}
void Launcher::push(std::function<void()> f) {
  {
    std::lock_guard<std::mutex> lk(mu);
    q.push_back(std::move(f));
  }
  cv.notify_one();
}

extern "C" const char* nrtgpu_version(void) { return "nrtgpu 0.1 (gfx950)"; }
extern "C" const char* nrtgpu_last_error(void) { return g_last_error.c_str(); }
extern "C" void nrtgpu_set_thread_deadline_ns(int64_t deadline_ns) { g_deadline_ns = deadline_ns; }
extern "C" int64_t nrtgpu_monotonic_ns(void) { return monotonic_ns(); }
extern "C" int nrtgpu_last_diagnostics(nrtgpu_diagnostics* out) {
  if (!out) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  *out = g_diag;
  return NRTGPU_OK;
}

extern "C" int nrtgpu_create(const nrtgpu_config* cfg, nrtgpu_ctx** out) {
  if (!out) return fail(NRTGPU_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  nrtgpu_config c{};
  if (cfg) c = *cfg;
  if (c.max_batch <= 0) c.max_batch = 1024;
#ifndef NRTGPU_DEV
  {
    const int variant = (c.flags >> 8) & 15;  // 7 = instrumented kernels (include/nrtgpu_dev.h); the rest are timing ablations with wrong results
    if (variant != 0)
      return fail(NRTGPU_ERR_INVALID_ARG, "flags: kernel variant %d exists only in the development build (-DNRTGPU_DEV)", variant);
  }
#endif
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0)
    return fail(NRTGPU_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (c.device_id < 0 || c.device_id >= n_dev) return fail(NRTGPU_ERR_INVALID_ARG, "device_id %d out of range [0,%d)", c.device_id, n_dev);
  HIP_TRY(hipSetDevice(c.device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, c.device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(NRTGPU_ERR_HIP, "device %d is %s; kernels are built for gfx950 only", c.device_id, prop.gcnArchName);
  auto ctx = std::make_unique<nrtgpu_ctx>();
  ctx->cfg = c;
  ctx->device = c.device_id;
  ctx->n_cus = prop.multiProcessorCount;
  for (int i = 0; i < kSlots; ++i) {
    auto s = std::make_unique<Slot>();
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&s->ev0));
    HIP_TRY(hipEventCreate(&s->ev1));
    HIP_TRY(hipEventCreate(&s->ev2));
    HIP_TRY(hipEventCreate(&s->ev3));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_turn, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_wait, hipEventBlockingSync | hipEventDisableTiming));
    ctx->slots.push_back(std::move(s));
  }
  ctx->pool = std::make_unique<WorkPool>(std::max(0, (c.host_threads > 0 ? c.host_threads : 4) - 1));
  *out = ctx.release();
  return NRTGPU_OK;
}

extern "C" void nrtgpu_exchange_close(nrtgpu_ctx* ctx) {
  if (!ctx || !ctx->xch_host) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  (void)hipHostUnregister(ctx->xch_host);
  (void)munmap(ctx->xch_host, ctx->xch_bytes);
  ctx->xch_host = nullptr;
  ctx->xch_dev = nullptr;
  ctx->xch_bytes = 0;
  ctx->xch_world = 0;
}

extern "C" int nrtgpu_exchange_open(nrtgpu_ctx* ctx, const char* shm_name, int32_t world, int32_t rank) {
  if (!ctx || !shm_name || !shm_name[0]) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (world < 2 || world > 64 || rank < 0 || rank >= world) return fail(NRTGPU_ERR_INVALID_ARG, "bad world %d / rank %d", world, rank);
  if (ctx->xch_host) return fail(NRTGPU_ERR_STATE, "an exchange is already open on this context");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t stride = (size_t)ctx->cfg.max_batch;
  const size_t bytes = ((size_t)kExchangeSlots * (size_t)world * stride * 8 + 4095) & ~(size_t)4095;
  const int fd = shm_open(shm_name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return fail(NRTGPU_ERR_HIP, "shm_open(%s) failed: %s", shm_name, strerror(errno));
  if (ftruncate(fd, (off_t)bytes) != 0) {
    close(fd);
    return fail(NRTGPU_ERR_HIP, "ftruncate(%s, %zu) failed: %s", shm_name, bytes, strerror(errno));
  }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail(NRTGPU_ERR_HIP, "mmap(%s) failed: %s", shm_name, strerror(errno));
  // my own rows start silent (tag 0 never matches an epoch tag); the other ranks clear theirs
  unsigned long long* t = (unsigned long long*)p;
  for (int sl = 0; sl < kExchangeSlots; ++sl)
    memset(t + ((size_t)sl * world + rank) * stride, 0, stride * 8);
  hipError_t e = hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  void* dp = nullptr;
  if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, p, 0);
  if (e != hipSuccess) {
    (void)munmap(p, bytes);
    return fail(NRTGPU_ERR_HIP, "mapping the exchange table into the GPU failed: %s", hipGetErrorString(e));
  }
  ctx->xch_host = p;
  ctx->xch_dev = (unsigned long long*)dp;
  ctx->xch_bytes = bytes;
  ctx->xch_world = world;
  ctx->xch_rank = rank;
  return NRTGPU_OK;
}

extern "C" void nrtgpu_destroy(nrtgpu_ctx* ctx) {
  if (!ctx) return;
  ctx->launcher.reset();   // (joins: nothing is enqueued behind the streams' last synchronisation below)
  nrtgpu_dist_close(ctx);
  nrtgpu_exchange_close(ctx);
  (void)hipSetDevice(ctx->device);
  for (auto& s : ctx->slots) {
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    s->h_plan.release();
    s->d_plan.release();
    s->d_work.release();
    s->d_aux.release();
    s->h_aux.release();
    s->h_out.release();
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->ev2) (void)hipEventDestroy(s->ev2);
    if (s->ev3) (void)hipEventDestroy(s->ev3);
    if (s->ev_turn) (void)hipEventDestroy(s->ev_turn);
    if (s->ev_wait) (void)hipEventDestroy(s->ev_wait);
    for (hipEvent_t e : s->round_ev) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
  }
  delete ctx;
}

extern "C" int nrtgpu_set_slicing(nrtgpu_ctx* ctx, int32_t slice_max_docs, int32_t slice_max_segments, int32_t virtual_shards) {
  if (!ctx || slice_max_docs < 0 || slice_max_segments <= 0 || virtual_shards <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad slicing parameters");
  ctx->slice_max_docs = slice_max_docs;
  ctx->slice_max_segments = slice_max_segments;
  ctx->virtual_shards = virtual_shards;
  return NRTGPU_OK;
}

extern "C" int nrtgpu_set_thread_slices(const int32_t* slice_of_leaf, int32_t n_leaves) {
  if (n_leaves < 0 || (n_leaves > 0 && !slice_of_leaf)) return fail(NRTGPU_ERR_INVALID_ARG, "bad thread slices");
  for (int32_t i = 0; i < n_leaves; ++i)
    if (slice_of_leaf[i] < 0) return fail(NRTGPU_ERR_INVALID_ARG, "slice_of_leaf[%d] = %d: slice numbers are >= 0", i, slice_of_leaf[i]);
  g_thread_slices.assign(slice_of_leaf, slice_of_leaf + n_leaves);
  return NRTGPU_OK;
}

extern "C" int nrtgpu_get_stats(nrtgpu_ctx* ctx, nrtgpu_stats* out) {
  if (!ctx || !out) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  *out = ctx->stats;
  out->spec_queries = ctx->spec_queries.load(std::memory_order_relaxed);
  out->spec_reruns = ctx->spec_reruns.load(std::memory_order_relaxed);
  out->spec_disabled = ctx->spec_off.load(std::memory_order_relaxed);
  out->spec_scattered = ctx->spec_scattered.load(std::memory_order_relaxed);
  return NRTGPU_OK;
}
extern "C" void nrtgpu_reset_stats(nrtgpu_ctx* ctx) {
  if (!ctx) return;
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  ctx->stats = nrtgpu_stats{};
  for (double& p : ctx->prof) p = 0;
  for (double& p : ctx->ms_prof) p = 0;
}
#ifdef NRTGPU_DEV   // include/nrtgpu_dev.h: the instrumented kernels' counters
extern "C" int nrtgpu_get_scan_profile(nrtgpu_ctx* ctx, double* out16) {
  if (!ctx || !out16) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  for (int i = 0; i < 16; ++i) out16[i] = ctx->prof[i];
  return NRTGPU_OK;
}

extern "C" int nrtgpu_get_maxscore_profile(nrtgpu_ctx* ctx, double* out16) {
  if (!ctx || !out16) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  for (int i = 0; i < 16; ++i) out16[i] = ctx->ms_prof[i];
  return NRTGPU_OK;
}

extern "C" int64_t nrtgpu_get_maxscore_item_walls(nrtgpu_ctx* ctx, uint64_t* out, int64_t cap_slots, int64_t* n_items) {
  if (!ctx) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  const int64_t n = (int64_t)(ctx->last_walls.size() / 8);
  if (n_items) *n_items = ctx->last_walls_items;
  if (out)
    for (int64_t i = 0; i < std::min(n, cap_slots) * 8; ++i) out[i] = ctx->last_walls[(size_t)i];
  return n;
}
#endif

// ------------------------------------------------------------------------------------------------
// ABI: host-side restatements
// ------------------------------------------------------------------------------------------------
extern "C" int32_t nrtgpu_int_to_byte4(int32_t length) { return hostmath::int_to_byte4(length); }
extern "C" int32_t nrtgpu_byte4_to_int(int32_t b) { return hostmath::byte4_to_int(b); }
extern "C" float nrtgpu_bm25_idf(int64_t doc_count, int64_t doc_freq) { return hostmath::bm25_idf(doc_count, doc_freq); }
extern "C" float nrtgpu_bm25_avgdl(int64_t sttf, int64_t doc_count) { return hostmath::bm25_avgdl(sttf, doc_count); }
extern "C" void nrtgpu_bm25_norm_cache(float avgdl, float k1, float b, float* out256) { hostmath::bm25_norm_cache(avgdl, k1, b, out256); }

extern "C" int nrtgpu_fixed_point_scale(float weight, const float* norm_cache256, int32_t max_norm, int32_t* out_scale) {
  if (!norm_cache256 || !out_scale || max_norm < 0 || max_norm > 255) return fail(NRTGPU_ERR_INVALID_ARG, "bad fixed_point_scale arguments");
  return fixed_scale_of_term(weight, norm_cache256, (uint32_t)max_norm, out_scale) ? 1 : 0;
}

extern "C" int nrtgpu_blend(int32_t n_retrievers, const int32_t* const* docs, const float* const* scores, const int32_t* counts,
                            const float* boosts, int32_t mode, int32_t rank_constant, int32_t start_hit, int32_t top_hits,
                            nrtgpu_topdocs* out) {
  if (n_retrievers < 0 || (n_retrievers > 0 && (!docs || !counts)) || !out || top_hits < 0 || start_hit < 0 || (mode != 0 && mode != 1))
    return fail(NRTGPU_ERR_INVALID_ARG, "bad blend arguments");
  if (mode == 0 && rank_constant < 1) return fail(NRTGPU_ERR_INVALID_ARG, "k must be >= 1, got: %d", rank_constant);  // WeightedRRFScoreDoc.java:62
  if (mode == 1 && n_retrievers > 0 && !scores) return fail(NRTGPU_ERR_INVALID_ARG, "score-order blend needs the retrievers' scores");
  for (int32_t r = 0; r < n_retrievers; ++r)
    if (counts[r] < 0 || (counts[r] > 0 && (!docs[r] || (mode == 1 && !scores[r])))) return fail(NRTGPU_ERR_INVALID_ARG, "bad retriever %d", r);
  std::vector<hostmath::BlendHit> page;
  const int64_t total = hostmath::blend_hits(n_retrievers, docs, scores, counts, boosts, rank_constant, mode == 0, start_hit, top_hits, &page);
  const int32_t cap = out->capacity > 0 ? out->capacity : top_hits;
  const int32_t m = std::min<int32_t>((int32_t)page.size(), cap);
  for (int32_t i = 0; i < m; ++i) {
    if (out->docs) out->docs[i] = page[(size_t)i].doc;
    if (out->scores) out->scores[i] = page[(size_t)i].score;
  }
  out->n_hits = m;
  out->total_hits = total;
  out->total_hits_is_lower_bound = 1;   // BlenderOperation.java: always GREATER_THAN_OR_EQUAL_TO
  return NRTGPU_OK;
}

extern "C" int nrtgpu_plan_item_counts(int32_t n_queries, const int64_t* query_costs, int32_t target_items, int64_t* out_items) {
  if (n_queries < 0 || (n_queries > 0 && (!query_costs || !out_items)) || target_items <= 0)
    return fail(NRTGPU_ERR_INVALID_ARG, "bad plan_item_counts arguments");
  hostmath::plan_item_counts(query_costs, n_queries, target_items, (int64_t)1 << 17, out_items);
  return NRTGPU_OK;
}

extern "C" int32_t nrtgpu_slices(int32_t n_leaves, const int32_t* max_docs, const int32_t* num_docs, const int32_t* doc_bases,
                                 int32_t virtual_shards, int32_t slice_max_docs, int32_t slice_max_segments,
                                 int32_t* slice_of_leaf, int32_t* shard_of_leaf) {
  if (n_leaves < 0 || (n_leaves > 0 && (!max_docs || !slice_of_leaf))) return fail(NRTGPU_ERR_INVALID_ARG, "bad leaf arrays");
  std::vector<hostmath::LeafInfo> all((size_t)n_leaves);
  int32_t base = 0;
  for (int32_t i = 0; i < n_leaves; ++i) {
    all[(size_t)i] = {i, max_docs[i], num_docs ? num_docs[i] : max_docs[i], doc_bases ? doc_bases[i] : base};
    base += max_docs[i];
  }
  std::vector<std::vector<int32_t>> sl;
  std::vector<int32_t> shard;
  if (virtual_shards > 1) {
    sl = hostmath::slices_for_shards(all, virtual_shards, slice_max_docs, slice_max_segments, &shard);
  } else {
    sl = hostmath::slices(all, slice_max_docs, slice_max_segments, all);
  }
  for (size_t s = 0; s < sl.size(); ++s)
    for (int32_t li : sl[s]) slice_of_leaf[li] = (int32_t)s;
  if (shard_of_leaf)
    for (int32_t i = 0; i < n_leaves; ++i) shard_of_leaf[i] = virtual_shards > 1 ? shard[(size_t)i] : 0;
  return (int32_t)sl.size();
}
