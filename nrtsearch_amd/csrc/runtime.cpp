// runtime.cpp -- host runtime behind the C ABI of include/nrtgpu.h (compiled with hipcc).
//
// Owns: the device context (one per process/GPU), the segment store (read-only columnar replica of
// each Lucene segment's scoring data in HBM), batch launch plans, the per-call workspaces
// ("slots": stream + pinned staging + device scratch) and the result unpacking.
// There is deliberately no CPU execution path here: without a gfx950 device nrtgpu_create fails.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <climits>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/nrtgpu.h"
#include "host_math.h"
#include "plan.h"

namespace nrtgpu {
void launch_bm25_scan(hipStream_t stream, bool fixed_point, bool pipelined, int ablation, uint32_t n_items, const DItem* items,
                      const DPart* parts, const DTerm* terms, const DQuery* queries, const float* caches, unsigned long long* theta_g,
                      unsigned long long* quant_g, const DExchange* xch, uint64_t* item_keys, uint32_t* item_counts,
                      uint64_t* item_hits, uint32_t k_stride, uint64_t* item_prof);
void launch_merge_topk(hipStream_t stream, uint32_t n_queries, const uint64_t* in_keys, const uint32_t* in_counts,
                       const uint64_t* in_hits, const uint32_t* list_idx, const uint32_t* q_base,
                       const uint32_t* q_nlists, uint32_t k_stride_in, const uint32_t* q_k, uint64_t* out_keys,
                       uint32_t* out_counts, uint64_t* out_hits, uint32_t k_stride_out);
void launch_fold_norms(hipStream_t stream, const uint32_t* docids, const uint32_t* freqs, const uint8_t* norms,
                       uint32_t* fnorm, uint64_t n, uint32_t* overflow);
void launch_apply_live(hipStream_t stream, const uint32_t* docids, uint32_t* fnorm, uint64_t n, const uint64_t* live);
void launch_knn_row_norms(hipStream_t st, const float* vecs, int32_t dim, int64_t n, float* norm2);
int launch_knn_score(hipStream_t st, uint32_t blocks, const float* vecs, const float* vnorm2, const int32_t* ord_to_doc,
                     const uint64_t* live_bits, int32_t dim, int64_t row_begin, int64_t row_end, int32_t doc_base,
                     const float* qpanel, const float* qnorm2, int32_t n_q, int32_t sim, float boost,
                     const unsigned long long* theta, uint64_t* cand, uint32_t* cand_cnt, uint32_t cap);
void launch_knn_select(hipStream_t st, uint32_t n_q, uint64_t* topk, uint32_t* topk_cnt, uint32_t k_stride, uint32_t k,
                       const uint64_t* cand, uint32_t* cand_cnt, uint32_t cap, unsigned long long* theta,
                       uint32_t* overflow);
void launch_rescore_vectors(hipStream_t st, const float* vecs, const float* vnorm2, int32_t dim, const float* query,
                            float qnorm2, int32_t sim, float boost, const int64_t* vec_row, const float* first_scores,
                            int32_t n, double qw, double rw, float* out_scores);
void launch_hybrid_rescore(hipStream_t st, uint32_t n_queries, const uint64_t* first_keys, const uint32_t* first_counts,
                           uint32_t k_stride, const DVecSeg* segs, int32_t n_segs, int32_t dim, const float* qvecs,
                           const float* qnorm2, int32_t sim, float boost, double qw, double rw, uint32_t window,
                           uint64_t* out_keys, uint32_t* out_counts, uint32_t w_stride);
}  // namespace nrtgpu

using namespace nrtgpu;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(_e == hipErrorOutOfMemory ? NRTGPU_ERR_OOM : NRTGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                  hipGetErrorString(_e), __FILE__, __LINE__);                                      \
  } while (0)

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------
// growable device / pinned buffers
// ------------------------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

// ------------------------------------------------------------------------------------------------
// segment store
// ------------------------------------------------------------------------------------------------
struct TermEntry {
  uint32_t group;      // which upload group holds the columns
  uint64_t start;      // first posting in the group's columns
  uint32_t count;
  uint32_t shift;      // doc-range cell = tile >> shift
  uint64_t cell_start; // first entry of the term's cell table inside the group's table buffer
};

struct TermGroup {
  uint32_t* d_docids = nullptr;
  uint32_t* d_freqs = nullptr;   // raw freq column, only between add_terms and seal (nullptr => freq == 1)
  uint32_t* d_fnorm = nullptr;   // score-code column (same allocation as d_docids), filled at seal
  bool folded = false;
  uint32_t* d_cells = nullptr;   // concatenated per-term cell tables
  bool has_freqs = false;
  uint64_t n_postings = 0;
};

// Read-only open-addressing view of a field's term dictionary (built at seal): the planner does
// one lookup per (query clause, leaf), ~50k per batch, so a probe should touch one cache line.
struct FlatDict {
  struct Cell { int64_t key; uint32_t idx; uint32_t used; };
  std::vector<Cell> cells;
  std::vector<TermEntry> entries;
  uint32_t shift = 64;
  static inline uint64_t mix(int64_t k) { return (uint64_t)k * 0x9E3779B97F4A7C15ull; }
  void build(const std::unordered_map<int64_t, TermEntry>& d) {
    size_t cap = 16;
    uint32_t bits = 4;
    while (cap < d.size() * 2 + 2) { cap <<= 1; ++bits; }
    cells.assign(cap, Cell{0, 0, 0});
    entries.clear();
    entries.reserve(d.size());
    shift = 64 - bits;
    for (const auto& kv : d) {
      size_t h = (size_t)(mix(kv.first) >> shift);
      while (cells[h].used) h = (h + 1) & (cap - 1);
      cells[h] = Cell{kv.first, (uint32_t)entries.size(), 1u};
      entries.push_back(kv.second);
    }
  }
  inline const TermEntry* find(int64_t key) const {
    if (cells.empty()) return nullptr;
    const size_t mask = cells.size() - 1;
    size_t h = (size_t)(mix(key) >> shift);
    for (;;) {
      const Cell& c = cells[h];
      if (!c.used) return nullptr;
      if (c.key == key) return &entries[c.idx];
      h = (h + 1) & mask;
    }
  }
};

struct FieldData {
  uint8_t* d_norms = nullptr;    // nullptr => norms omitted
  uint32_t max_norm = 1;         // largest norm byte of the field in this segment (longest doc); 1 when omitted
  std::unordered_map<int64_t, TermEntry> dict;   // build-time (duplicate detection); searches use `flat`
  FlatDict flat;
  std::vector<TermGroup> groups;
  float* d_vectors = nullptr;
  float* d_vnorm2 = nullptr;            // |v|^2 per row (cosine / euclidean)
  int32_t* d_ord_to_doc = nullptr;
  std::vector<int32_t> h_ord_to_doc;     // host copy: docid -> row lookups of the rescore path
  int32_t dim = 0, n_vec = 0;
};

struct nrtgpu_ctx;
struct nrtgpu_seg {
  nrtgpu_ctx* ctx = nullptr;
  int32_t max_doc = 0;
  uint32_t n_tiles = 0;
  bool sealed = false;
  std::map<int32_t, FieldData> fields;
  uint64_t* d_live = nullptr;
  int64_t device_bytes = 0;
  // FILTER / MUST_NOT clauses as doc-set masks: host copies of the registered masks and of liveDocs,
  // and the combined accept sets (live & filter & ~must_not) the scan reads, built on first use
  std::vector<uint64_t> h_live;                      // empty = all live
  bool live_folded = false;  // the posting columns carry the current liveDocs (apply_live_kernel): the scan needs no mask for them
  // searches hold this shared from planning until their kernels have finished; set_live_docs / set_mask take it
  // exclusively, so a reader-version change never rewrites columns or masks under a running scan
  mutable std::shared_mutex content_mu;
  mutable std::atomic<int> content_writers{0};  // pending exclusive owners: new searches let them go first (no writer starvation)
  std::map<int32_t, std::vector<uint64_t>> masks;
  mutable std::mutex accept_mu;
  mutable std::map<std::pair<int32_t, int32_t>, uint64_t*> accept;
};

// Exclusive ownership of a segment's content (liveDocs, masks, the posting columns' liveness coding).
struct SegWriteLock {
  nrtgpu_seg* seg;
  explicit SegWriteLock(nrtgpu_seg* s);
  ~SegWriteLock();
  SegWriteLock(const SegWriteLock&) = delete;
  SegWriteLock& operator=(const SegWriteLock&) = delete;
};

// Shared locks on the content of every (distinct) segment of a call, taken in address order.
struct SegReadLocks {
  std::vector<const nrtgpu_seg*> held;
  SegReadLocks(const nrtgpu_seg* const* segs, int32_t n) {
    for (int32_t i = 0; i < n; ++i)
      if (segs && segs[i]) held.push_back(segs[i]);
    std::sort(held.begin(), held.end());
    held.erase(std::unique(held.begin(), held.end()), held.end());
    for (const nrtgpu_seg* s : held) {
      while (s->content_writers.load(std::memory_order_acquire) > 0) std::this_thread::yield();
      s->content_mu.lock_shared();
    }
  }
  ~SegReadLocks() {
    for (const nrtgpu_seg* s : held) s->content_mu.unlock_shared();
  }
  SegReadLocks(const SegReadLocks&) = delete;
  SegReadLocks& operator=(const SegReadLocks&) = delete;
};

SegWriteLock::SegWriteLock(nrtgpu_seg* s) : seg(s) {
  seg->content_writers.fetch_add(1, std::memory_order_acq_rel);
  seg->content_mu.lock();
}
SegWriteLock::~SegWriteLock() {
  seg->content_mu.unlock();
  seg->content_writers.fetch_sub(1, std::memory_order_acq_rel);
}

static const size_t kMaskPadBytes = 256;  // doc-set masks are readable one sub-tile (128 bytes) past max_doc

static void drop_accept_sets(nrtgpu_seg* seg) {
  std::lock_guard<std::mutex> lk(seg->accept_mu);
  for (auto& kv : seg->accept) {
    (void)hipFree(kv.second);
    seg->device_bytes -= (int64_t)((seg->max_doc + 63) / 64) * 8;
  }
  seg->accept.clear();
}

// ------------------------------------------------------------------------------------------------
// per-call workspace
// ------------------------------------------------------------------------------------------------
struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  PinBuf h_plan;     // host staging of the plan blob
  DevBuf d_plan;     // device copy
  DevBuf d_work;     // theta + item outputs + merge outputs
  DevBuf d_aux;      // hybrid tail: leaf table, query vectors, rescored windows
  PinBuf h_aux;
  PinBuf h_out;      // merged results on the host
  bool busy = false;
};

struct nrtgpu_ctx {
  nrtgpu_config cfg{};
  int device = 0;
  int n_cus = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::unique_ptr<Slot>> slots;
  std::mutex gpu_mu;    // device execution of one batch at a time: a scan kernel wants the whole GPU,
                        // overlapping two only stretches both (host-side planning/unpacking still overlap)
  std::mutex stats_mu;
  nrtgpu_stats stats{};
  double prof[16] = {0};
  // request coalescing (nrtgpu_search_bm25_coalesced)
  std::mutex co_mu;
  std::condition_variable co_cv;
  std::vector<struct CoRequest*> co_pending;  // waiting for a leader
  struct CoRequest* co_leader = nullptr;      // the caller lingering for / about to run the next batch
  int co_inflight = 0;                        // coalesced batches executing right now
  int co_inflight_queries = 0;                // ... and how many queries they hold
  int co_last_batch = 0;                      // size of the batch formed last (a lone caller does not linger)
  int32_t co_linger_us = 150;
  // cross-GPU bound exchange (nrtgpu_exchange_open)
  void* xch_host = nullptr;                 // mmap of the shared table
  unsigned long long* xch_dev = nullptr;    // the same memory as the GPU sees it
  size_t xch_bytes = 0;
  int32_t xch_world = 0, xch_rank = 0;
};

static const int kSlots = 4;

static int acquire_slot(nrtgpu_ctx* ctx, Slot** out) {
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
static void release_slot(nrtgpu_ctx* ctx, Slot* s) {
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    s->busy = false;
  }
  ctx->cv.notify_one();
}

// ------------------------------------------------------------------------------------------------
// ABI: context
// ------------------------------------------------------------------------------------------------
extern "C" const char* nrtgpu_version(void) { return "nrtgpu 0.1 (gfx950)"; }
extern "C" const char* nrtgpu_last_error(void) { return g_last_error.c_str(); }

extern "C" int nrtgpu_create(const nrtgpu_config* cfg, nrtgpu_ctx** out) {
  if (!out) return fail(NRTGPU_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  nrtgpu_config c{};
  if (cfg) c = *cfg;
  if (c.max_batch <= 0) c.max_batch = 1024;
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
    ctx->slots.push_back(std::move(s));
  }
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
    if (s->stream) (void)hipStreamDestroy(s->stream);
  }
  delete ctx;
}

extern "C" int nrtgpu_get_stats(nrtgpu_ctx* ctx, nrtgpu_stats* out) {
  if (!ctx || !out) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  *out = ctx->stats;
  return NRTGPU_OK;
}
extern "C" void nrtgpu_reset_stats(nrtgpu_ctx* ctx) {
  if (!ctx) return;
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  ctx->stats = nrtgpu_stats{};
  for (double& p : ctx->prof) p = 0;
}
extern "C" int nrtgpu_get_scan_profile(nrtgpu_ctx* ctx, double* out16) {
  if (!ctx || !out16) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  for (int i = 0; i < 16; ++i) out16[i] = ctx->prof[i];
  return NRTGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// ABI: segment lifecycle
// ------------------------------------------------------------------------------------------------
static int dev_alloc(nrtgpu_seg* seg, void** p, size_t bytes) {
  HIP_TRY(hipMalloc(p, bytes));
  seg->device_bytes += (int64_t)bytes;
  return 0;
}

extern "C" int nrtgpu_segment_begin(nrtgpu_ctx* ctx, int32_t max_doc, int32_t /*device_hint*/, nrtgpu_seg** out) {
  if (!ctx || !out) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (max_doc <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "max_doc must be > 0, got %d", max_doc);
  auto* seg = new nrtgpu_seg();
  seg->ctx = ctx;
  seg->max_doc = max_doc;
  seg->n_tiles = (uint32_t)(((int64_t)max_doc + kTileDocs - 1) / kTileDocs);
  *out = seg;
  return NRTGPU_OK;
}

extern "C" int nrtgpu_segment_add_field_norms(nrtgpu_seg* seg, int32_t field_id, const uint8_t* norm_bytes) {
  if (!seg) return fail(NRTGPU_ERR_INVALID_ARG, "seg is NULL");
  if (seg->sealed) return fail(NRTGPU_ERR_STATE, "segment already sealed");
  HIP_TRY(hipSetDevice(seg->ctx->device));
  FieldData& f = seg->fields[field_id];
  if (f.d_norms) {
    (void)hipFree(f.d_norms);
    f.d_norms = nullptr;
  }
  if (!norm_bytes) return NRTGPU_OK;  // norms omitted: norm value 1 everywhere
  void* p = nullptr;
  if (int rc = dev_alloc(seg, &p, (size_t)seg->max_doc + 64)) return rc;
  f.d_norms = (uint8_t*)p;
  HIP_TRY(hipMemcpy(f.d_norms, norm_bytes, (size_t)seg->max_doc, hipMemcpyHostToDevice));
  uint32_t mx = 0;
  for (int32_t d = 0; d < seg->max_doc; ++d) mx = std::max<uint32_t>(mx, norm_bytes[d]);
  f.max_norm = mx;
  return NRTGPU_OK;
}

extern "C" int nrtgpu_segment_add_terms(nrtgpu_seg* seg, int32_t field_id, int64_t n_terms, const int64_t* term_hash,
                                        const int64_t* offsets, const int32_t* docids, const int32_t* freqs) {
  if (!seg) return fail(NRTGPU_ERR_INVALID_ARG, "seg is NULL");
  if (seg->sealed) return fail(NRTGPU_ERR_STATE, "segment already sealed");
  if (n_terms < 0 || (n_terms > 0 && (!term_hash || !offsets))) return fail(NRTGPU_ERR_INVALID_ARG, "bad term arrays");
  if (n_terms == 0) return NRTGPU_OK;
  const int64_t total = offsets[n_terms];
  if (offsets[0] != 0 || total < 0) return fail(NRTGPU_ERR_INVALID_ARG, "offsets must start at 0 and be non-negative");
  if (total > 0 && !docids) return fail(NRTGPU_ERR_INVALID_ARG, "docids is NULL");
  HIP_TRY(hipSetDevice(seg->ctx->device));
  FieldData& f = seg->fields[field_id];

  // doc-range cell tables: per term, posting offset at each cell boundary (cell = 2^shift tiles).
  // Dense terms get one cell per tile; sparse terms coarser cells so a table never exceeds ~1/8
  // of the term's postings.
  std::vector<uint32_t> cells;
  std::vector<TermEntry> entries((size_t)n_terms);
  for (int64_t t = 0; t < n_terms; ++t) {
    const int64_t lo = offsets[t], hi = offsets[t + 1];
    if (hi < lo || hi > total) return fail(NRTGPU_ERR_INVALID_ARG, "offsets not monotone at term %lld", (long long)t);
    const int64_t cnt = hi - lo;
    if (cnt > 0xFFFFFFFFll) return fail(NRTGPU_ERR_UNSUPPORTED, "term with more than 2^32 postings");
    uint32_t shift = 0;
    const uint64_t budget = std::max<int64_t>(1, cnt / 8);
    while (((uint64_t)(seg->n_tiles - 1) >> shift) + 1 > budget && shift < 31) ++shift;
    const uint32_t n_cells = (uint32_t)(((uint64_t)(seg->n_tiles - 1) >> shift) + 1);
    TermEntry& e = entries[(size_t)t];
    e.group = (uint32_t)f.groups.size();
    e.start = (uint64_t)lo;
    e.count = (uint32_t)cnt;
    e.shift = shift;
    e.cell_start = cells.size();
    const int64_t cell_docs = (int64_t)kTileDocs << shift;  // a cell covers 2^shift sub-tiles
    int64_t p = lo;
    int32_t prev = -1;
    for (uint32_t c = 0; c < n_cells; ++c) {
      cells.push_back((uint32_t)(p - lo));
      const int64_t bound = (int64_t)(c + 1) * cell_docs;  // first doc of the next cell
      while (p < hi && (int64_t)docids[p] < bound) {
        const int32_t d = docids[p];
        if (d <= prev || d >= seg->max_doc)
          return fail(NRTGPU_ERR_INVALID_ARG, "docids of term %lld not strictly ascending in [0,max_doc)", (long long)t);
        prev = d;
        ++p;
      }
    }
    if (p != hi) return fail(NRTGPU_ERR_INVALID_ARG, "docids of term %lld exceed max_doc", (long long)t);
    cells.push_back((uint32_t)cnt);
  }
  for (int64_t t = 0; t < n_terms; ++t)
    if (f.dict.count(term_hash[t])) return fail(NRTGPU_ERR_INVALID_ARG, "term %lld added twice to field %d", (long long)term_hash[t], field_id);

  TermGroup g;
  g.n_postings = (uint64_t)total;
  void* p = nullptr;
  // one allocation per upload group: [docid column | code column], each padded to a multiple of
  // 16 bytes plus 64 (16-byte group loads may run past the end); the kernel addresses the code column
  // as docid column + a per-term constant
  const size_t col_bytes = (((size_t)total * 4 + 15) & ~(size_t)15) + 64;
  if (int rc = dev_alloc(seg, &p, 2 * col_bytes)) return rc;
  g.d_docids = (uint32_t*)p;
  g.d_fnorm = (uint32_t*)((char*)p + col_bytes);
  HIP_TRY(hipMemset(p, 0, 2 * col_bytes));
  if (total) HIP_TRY(hipMemcpy(g.d_docids, docids, (size_t)total * 4, hipMemcpyHostToDevice));
  if (freqs) {
    if (int rc = dev_alloc(seg, &p, col_bytes)) return rc;
    g.d_freqs = (uint32_t*)p;
    g.has_freqs = true;
    HIP_TRY(hipMemset(g.d_freqs, 0, col_bytes));
    if (total) HIP_TRY(hipMemcpy(g.d_freqs, freqs, (size_t)total * 4, hipMemcpyHostToDevice));
  }
  if (int rc = dev_alloc(seg, &p, cells.size() * 4 + 64)) return rc;
  g.d_cells = (uint32_t*)p;
  HIP_TRY(hipMemcpy(g.d_cells, cells.data(), cells.size() * 4, hipMemcpyHostToDevice));
  f.groups.push_back(g);
  for (int64_t t = 0; t < n_terms; ++t) f.dict.emplace(term_hash[t], entries[(size_t)t]);
  return NRTGPU_OK;
}

extern "C" int nrtgpu_segment_add_vectors(nrtgpu_seg* seg, int32_t field_id, int32_t dim, int32_t n,
                                          const int32_t* ord_to_doc, const float* row_major) {
  if (!seg) return fail(NRTGPU_ERR_INVALID_ARG, "seg is NULL");
  if (seg->sealed) return fail(NRTGPU_ERR_STATE, "segment already sealed");
  if (dim <= 0 || n < 0 || (n > 0 && !row_major)) return fail(NRTGPU_ERR_INVALID_ARG, "bad vector arguments");
  if (n > seg->max_doc) return fail(NRTGPU_ERR_INVALID_ARG, "more vectors (%d) than docs (%d)", n, seg->max_doc);
  HIP_TRY(hipSetDevice(seg->ctx->device));
  FieldData& f = seg->fields[field_id];
  if (f.d_vectors) return fail(NRTGPU_ERR_STATE, "vectors of field %d already added", field_id);
  f.dim = dim;
  f.n_vec = n;
  if (n == 0) return NRTGPU_OK;
  void* p = nullptr;
  if (int rc = dev_alloc(seg, &p, (size_t)n * dim * 4 + 256)) return rc;
  f.d_vectors = (float*)p;
  HIP_TRY(hipMemcpy(f.d_vectors, row_major, (size_t)n * dim * 4, hipMemcpyHostToDevice));
  if (int rc = dev_alloc(seg, &p, (size_t)n * 4 + 64)) return rc;
  f.d_vnorm2 = (float*)p;
  launch_knn_row_norms(nullptr, f.d_vectors, dim, n, f.d_vnorm2);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  if (ord_to_doc) {
    for (int32_t i = 0; i < n; ++i)
      if (ord_to_doc[i] < 0 || ord_to_doc[i] >= seg->max_doc || (i > 0 && ord_to_doc[i] <= ord_to_doc[i - 1]))
        return fail(NRTGPU_ERR_INVALID_ARG, "ord_to_doc must be strictly ascending docids in [0,max_doc)");
    if (int rc = dev_alloc(seg, &p, (size_t)n * 4)) return rc;
    f.d_ord_to_doc = (int32_t*)p;
    HIP_TRY(hipMemcpy(f.d_ord_to_doc, ord_to_doc, (size_t)n * 4, hipMemcpyHostToDevice));
    f.h_ord_to_doc.assign(ord_to_doc, ord_to_doc + n);
  }
  return NRTGPU_OK;
}

static int fold_live_docs(nrtgpu_seg* seg);

extern "C" int nrtgpu_segment_seal(nrtgpu_seg* seg) {
  if (!seg) return fail(NRTGPU_ERR_INVALID_ARG, "seg is NULL");
  if (seg->sealed) return NRTGPU_OK;
  HIP_TRY(hipSetDevice(seg->ctx->device));
  // fold every posting's field-norm byte into its freq word: fnorm = (freq << 8) | norm
  uint32_t* d_overflow = nullptr;
  HIP_TRY(hipMalloc((void**)&d_overflow, 4));
  HIP_TRY(hipMemset(d_overflow, 0, 4));
  int rc = NRTGPU_OK;
  for (auto& kv : seg->fields) {
    FieldData& f = kv.second;
    for (auto& g : f.groups) {
      if (g.folded) continue;
      launch_fold_norms(nullptr, g.d_docids, g.d_freqs, f.d_norms, g.d_fnorm, g.n_postings, d_overflow);
      hipError_t e = hipGetLastError();
      g.folded = true;
      if (e != hipSuccess) {
        rc = fail(NRTGPU_ERR_HIP, "fold_norms failed: %s", hipGetErrorString(e));
        break;
      }
    }
    if (rc) break;
  }
  uint32_t overflow = 0;
  if (!rc) {
    hipError_t e = hipMemcpy(&overflow, d_overflow, 4, hipMemcpyDeviceToHost);  // also syncs the null stream
    if (e != hipSuccess) rc = fail(NRTGPU_ERR_HIP, "seal: %s", hipGetErrorString(e));
  }
  (void)hipFree(d_overflow);
  if (rc) return rc;
  if (overflow) return fail(NRTGPU_ERR_UNSUPPORTED, "a term frequency >= 2^22 does not fit the packed freq|norm column");
  for (auto& kv : seg->fields) kv.second.flat.build(kv.second.dict);
  for (auto& kv : seg->fields)
    for (auto& g : kv.second.groups)
      if (g.d_freqs) {  // raw freq column no longer needed
        (void)hipFree(g.d_freqs);
        g.d_freqs = nullptr;
        seg->device_bytes -= (int64_t)((((size_t)g.n_postings * 4 + 15) & ~(size_t)15) + 64);
      }
  seg->sealed = true;
  if (seg->d_live) return fold_live_docs(seg);  // liveDocs set before the seal
  return NRTGPU_OK;
}

// Re-code the posting columns for the segment's current liveDocs (kernels.hip: apply_live_kernel).  One pass
// over the segment's postings per reader version instead of a liveness test per matched doc per query.
static int fold_live_docs(nrtgpu_seg* seg) {
  seg->live_folded = false;
  if (seg->ctx->cfg.flags & NRTGPU_FLAG_NO_LIVE_FOLD) return NRTGPU_OK;
  if (!seg->sealed) return NRTGPU_OK;  // seal folds
  for (auto& kv : seg->fields)
    for (auto& g : kv.second.groups)
      launch_apply_live(nullptr, g.d_docids, g.d_fnorm, g.n_postings, seg->d_live);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  seg->live_folded = seg->d_live != nullptr;
  return NRTGPU_OK;
}

extern "C" int nrtgpu_segment_set_live_docs(nrtgpu_seg* seg, const uint64_t* bits, int32_t n_words) {
  if (!seg) return fail(NRTGPU_ERR_INVALID_ARG, "seg is NULL");
  SegWriteLock content(seg);  // waits for the searches running over this segment; later ones wait for it
  HIP_TRY(hipSetDevice(seg->ctx->device));
  const int32_t need = (seg->max_doc + 63) / 64;
  drop_accept_sets(seg);
  if (!bits) {
    if (seg->d_live) (void)hipFree(seg->d_live);
    seg->d_live = nullptr;
    seg->h_live.clear();
    return fold_live_docs(seg);
  }
  if (n_words < need) return fail(NRTGPU_ERR_INVALID_ARG, "live bits: %d words given, %d needed", n_words, need);
  seg->h_live.assign(bits, bits + need);
  if (!seg->d_live) {  // padded: the masked scan variant reads whole sub-tiles (128 bytes) of the mask
    void* p = nullptr;
    if (int rc = dev_alloc(seg, &p, (size_t)need * 8 + kMaskPadBytes)) return rc;
    seg->d_live = (uint64_t*)p;
    HIP_TRY(hipMemset((char*)p + (size_t)need * 8, 0, kMaskPadBytes));
  }
  HIP_TRY(hipMemcpy(seg->d_live, bits, (size_t)need * 8, hipMemcpyHostToDevice));
  return fold_live_docs(seg);
}

extern "C" int nrtgpu_segment_set_mask(nrtgpu_seg* seg, int32_t mask_id, const uint64_t* bits, int32_t n_words) {
  if (!seg) return fail(NRTGPU_ERR_INVALID_ARG, "seg is NULL");
  if (mask_id <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "mask id must be > 0, got %d", mask_id);
  SegWriteLock content(seg);  // waits for the searches running over this segment; later ones wait for it
  HIP_TRY(hipSetDevice(seg->ctx->device));
  const int32_t need = (seg->max_doc + 63) / 64;
  drop_accept_sets(seg);
  if (!bits) {
    seg->masks.erase(mask_id);
    return NRTGPU_OK;
  }
  if (n_words < need) return fail(NRTGPU_ERR_INVALID_ARG, "mask %d: %d words given, %d needed", mask_id, n_words, need);
  seg->masks[mask_id].assign(bits, bits + need);
  return NRTGPU_OK;
}

// The doc set a query's hits must lie in: liveDocs & FILTER mask & ~MUST_NOT mask, resident in HBM.
// (0, 0) is liveDocs itself.  Built and uploaded on first use, then shared by every query that names
// the same pair (the role LRUQueryCache plays for Lucene's non-scoring clauses).
static int accept_set_of(const nrtgpu_seg* seg, int32_t filter_mask, int32_t must_not_mask, const uint64_t** out) {
  if (filter_mask == 0 && must_not_mask == 0) {
    *out = seg->live_folded ? nullptr : seg->d_live;
    return 0;
  }
  std::lock_guard<std::mutex> lk(seg->accept_mu);
  const auto key = std::make_pair(filter_mask, must_not_mask);
  auto it = seg->accept.find(key);
  if (it != seg->accept.end()) {
    *out = it->second;
    return 0;
  }
  const std::vector<uint64_t>* f = nullptr;
  const std::vector<uint64_t>* mn = nullptr;
  if (filter_mask) {
    auto m = seg->masks.find(filter_mask);
    if (m == seg->masks.end()) return fail(NRTGPU_ERR_UNSUPPORTED, "filter mask %d is not resident on a segment", filter_mask);
    f = &m->second;
  }
  if (must_not_mask) {
    auto m = seg->masks.find(must_not_mask);
    if (m == seg->masks.end()) return fail(NRTGPU_ERR_UNSUPPORTED, "must_not mask %d is not resident on a segment", must_not_mask);
    mn = &m->second;
  }
  const size_t need = (size_t)(seg->max_doc + 63) / 64;
  std::vector<uint64_t> w(need + kMaskPadBytes / 8, 0ull);  // padded like liveDocs
  for (size_t i = 0; i < need; ++i) {
    uint64_t v = seg->h_live.empty() ? ~0ull : seg->h_live[i];
    if (f) v &= (*f)[i];
    if (mn) v &= ~(*mn)[i];
    w[i] = v;
  }
  void* p = nullptr;
  HIP_TRY(hipSetDevice(seg->ctx->device));
  if (hipMalloc(&p, w.size() * 8) != hipSuccess) return fail(NRTGPU_ERR_OOM, "hipMalloc(%zu) for an accept set failed", w.size() * 8);
  if (hipMemcpy(p, w.data(), w.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(p);
    return fail(NRTGPU_ERR_HIP, "upload of an accept set failed");
  }
  const_cast<nrtgpu_seg*>(seg)->device_bytes += (int64_t)need * 8;
  seg->accept[key] = (uint64_t*)p;
  *out = (uint64_t*)p;
  return 0;
}

extern "C" void nrtgpu_segment_release(nrtgpu_seg* seg) {
  if (!seg) return;
  (void)hipSetDevice(seg->ctx->device);
  drop_accept_sets(seg);
  for (auto& kv : seg->fields) {
    FieldData& f = kv.second;
    if (f.d_norms) (void)hipFree(f.d_norms);
    if (f.d_vectors) (void)hipFree(f.d_vectors);
    if (f.d_vnorm2) (void)hipFree(f.d_vnorm2);
    if (f.d_ord_to_doc) (void)hipFree(f.d_ord_to_doc);
    for (auto& g : f.groups) {
      if (g.d_docids) (void)hipFree(g.d_docids);
      if (g.d_freqs) (void)hipFree(g.d_freqs);
      if (g.d_cells) (void)hipFree(g.d_cells);
    }
  }
  if (seg->d_live) (void)hipFree(seg->d_live);
  delete seg;
}

extern "C" int64_t nrtgpu_segment_device_bytes(const nrtgpu_seg* seg) { return seg ? seg->device_bytes : 0; }

// ------------------------------------------------------------------------------------------------
// plan building
// ------------------------------------------------------------------------------------------------
struct HostPlan {
  std::vector<DQuery> queries;
  std::vector<DItem> items;
  std::vector<DPart> parts;
  std::vector<DTerm> terms;
  std::vector<float> caches;
  std::vector<uint32_t> list_idx;   // per query: item indices (merge input lists)
  std::vector<uint32_t> q_base, q_nlists, q_k;
  std::vector<uint64_t> theta_init;  // per query: key below which nothing is collected (min_competitive_score)
  uint32_t k_stride = 0;
  int64_t postings = 0;             // postings in the scanned term ranges (algorithmic work)
  bool fixed_point = false;         // every query of the batch passed the fixed-point range analysis
  bool clause_counting = false;     // some query has minimumNumberShouldMatch > 1: count-carrying kernel variant
  bool masked = false;              // some part reads a doc-set mask (liveDocs / FILTER / MUST_NOT)
};

static inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

static int validate_query(const nrtgpu_bm25_query& q, int qi) {
  // LazyQueueTopScoreDocCollectorManager.java:90-98
  if (q.k <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: numHits must be > 0; got %d", qi, q.k);
  if (q.total_hits_threshold < 0) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: totalHitsThreshold must be >= 0, got %d", qi, q.total_hits_threshold);
  if (q.k > NRTGPU_MAX_K) return fail(NRTGPU_ERR_UNSUPPORTED, "query %d: numHits %d > %d", qi, q.k, NRTGPU_MAX_K);
  if (q.n_terms <= 0 || !q.terms) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: no terms", qi);
  if (q.n_terms > NRTGPU_MAX_TERMS) return fail(NRTGPU_ERR_UNSUPPORTED, "query %d: %d clauses > %d", qi, q.n_terms, NRTGPU_MAX_TERMS);
  if (q.min_should_match < 0) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: minimumNumberShouldMatch %d", qi, q.min_should_match);
  if (q.n_caches <= 0 || !q.norm_cache) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: norm_cache missing", qi);
  if (q.n_caches > kLdsCaches) return fail(NRTGPU_ERR_UNSUPPORTED, "query %d: %d scored fields > %d", qi, q.n_caches, kLdsCaches);
  for (int t = 0; t < q.n_terms; ++t) {
    if (q.terms[t].cache_slot < 0 || q.terms[t].cache_slot >= q.n_caches)
      return fail(NRTGPU_ERR_INVALID_ARG, "query %d term %d: cache_slot out of range", qi, t);
    if (!(q.terms[t].weight >= 0.0f)) return fail(NRTGPU_ERR_INVALID_ARG, "query %d term %d: weight must be >= 0", qi, t);
  }
  if (!(q.min_competitive_score >= 0.0f)) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: min_competitive_score must be >= 0", qi);
  if (q.filter_mask < 0 || q.must_not_mask < 0) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: mask ids must be >= 0", qi);
  return 0;
}

// Cost model for cutting a query into work items: postings streamed + a per-tile constant for the
// accumulator sweep (in posting equivalents).
static const int64_t kTileCostPostings = 48;

struct QS { uint32_t term_begin, n_terms; int32_t seg; int64_t postings; };
struct QTabs { uint32_t n; float weight[kTabTerms]; uint32_t cache[kTabTerms]; int32_t scale[kTabTerms]; int32_t fx_E; };
static const int32_t kNoFixed = INT32_MIN;  // QTabs.fx_E: the query needs the fp64 accumulators

// Fixed-point eligibility of one query term: every score the term can produce in these
// segments must be a positive integer below 2^32 after scaling by 2^E_t.  Scores grow with freq and
// shrink with the norm byte, so the smallest one is score(1, largest norm byte present) and the
// weight bounds them from above.  Returns false when the range does not fit.
static bool fixed_scale_of_term(float weight, const float* cache256, uint32_t max_norm, int32_t* scale) {
  const float s_min = nrtgpu::hostmath::bm25_score(weight, 1.0f, cache256[max_norm & 255u]);
  if (!(s_min > 0.0f) || !std::isnormal(s_min) || !std::isnormal(weight)) return false;
  const int e_min = std::ilogb(s_min), e_w = std::ilogb(weight);
  if (e_w - e_min > 7) return false;  // 24 mantissa bits + 8 binades of range fill the 32-bit table entry
  *scale = 23 - e_min;
  return *scale > -64 && *scale < 64;
}
struct PlanPiece {
  std::vector<DTerm> terms;
  std::vector<float> caches;
  int64_t postings = 0, cost = 0;
};

// Pass 1 of the planner for queries [q_begin, q_end): one dictionary lookup per (clause, leaf); score
// tables go to the clauses with the most postings; terms of a (query, leaf) sorted densest first.
// Offsets (term_begin, cache offsets) are relative to the piece.
static void resolve_queries(const nrtgpu_seg* const* segs, int32_t n_segs, const nrtgpu_bm25_query* queries, int q_begin,
                            int q_end, PlanPiece& pc, std::vector<std::vector<QS>>& per_query,
                            std::vector<uint32_t>& cache_base, std::vector<QTabs>& qtabs) {
  std::vector<int64_t> term_total;
  std::vector<int32_t> tab_of_term, term_scale;
  std::vector<const TermEntry*> found;
  std::vector<const FieldData*> fld((size_t)n_segs, nullptr);
  std::vector<const FieldData*> found_field;
  int32_t fld_id = 0;
  bool fld_valid = false;
  size_t prev_cache_off = 0, prev_cache_len = 0;
  for (int qi = q_begin; qi < q_end; ++qi) {
    const nrtgpu_bm25_query& q = queries[qi];
    // consecutive queries over the same fields carry identical normInverse tables: keep one copy
    const size_t cache_len = (size_t)q.n_caches * 256;
    if (prev_cache_len == cache_len && memcmp(pc.caches.data() + prev_cache_off, q.norm_cache, cache_len * sizeof(float)) == 0) {
      cache_base[(size_t)qi] = (uint32_t)prev_cache_off;
    } else {
      prev_cache_off = pc.caches.size();
      prev_cache_len = cache_len;
      cache_base[(size_t)qi] = (uint32_t)prev_cache_off;
      pc.caches.insert(pc.caches.end(), q.norm_cache, q.norm_cache + cache_len);
    }
    found.assign((size_t)q.n_terms * (size_t)n_segs, nullptr);
    found_field.assign((size_t)q.n_terms * (size_t)n_segs, nullptr);
    term_total.assign((size_t)q.n_terms, 0);
    for (int t = 0; t < q.n_terms; ++t) {
      if (!fld_valid || fld_id != q.terms[t].field_id) {  // per-leaf field lookup hoisted out of the clause loop
        fld_id = q.terms[t].field_id;
        fld_valid = true;
        for (int si = 0; si < n_segs; ++si) {
          auto fit = segs[si]->fields.find(fld_id);
          fld[(size_t)si] = fit == segs[si]->fields.end() ? nullptr : &fit->second;
        }
      }
      for (int si = 0; si < n_segs; ++si) {
        const FieldData* f = fld[(size_t)si];
        if (!f) continue;
        const TermEntry* e = f->flat.find(q.terms[t].term_hash);
        if (!e || e->count == 0) continue;
        found[(size_t)t * n_segs + si] = e;
        found_field[(size_t)t * n_segs + si] = f;
        term_total[(size_t)t] += e->count;
      }
    }
    // fixed-point analysis: per clause the scale of its scores, per query the common scale
    term_scale.assign((size_t)q.n_terms, 0);
    bool fx_ok = true;
    int32_t fx_E = kNoFixed;
    for (int t = 0; t < q.n_terms && fx_ok; ++t) {
      if (term_total[(size_t)t] == 0) continue;  // matches nothing here
      uint32_t max_norm = 0;
      for (int si = 0; si < n_segs; ++si)
        if (const FieldData* f = found_field[(size_t)t * n_segs + si]) max_norm = std::max(max_norm, f->max_norm);
      int32_t sc = 0;
      fx_ok = fixed_scale_of_term(q.terms[t].weight, q.norm_cache + (size_t)q.terms[t].cache_slot * 256, max_norm, &sc);
      term_scale[(size_t)t] = sc;
      if (fx_ok) fx_E = std::max(fx_E, sc);
    }
    for (int t = 0; t < q.n_terms && fx_ok; ++t)  // 32-bit entries shifted into the common scale, summed over
      if (term_total[(size_t)t] != 0 && fx_E - term_scale[(size_t)t] > 15) fx_ok = false;  // <= 32 clauses: < 2^53
    if (!fx_ok) fx_E = kNoFixed;
    tab_of_term.assign((size_t)q.n_terms, -1);
    QTabs& qt_ = qtabs[(size_t)qi];
    qt_.n = 0;
    qt_.fx_E = fx_E;
    for (int r = 0; r < kTabTerms && r < q.n_terms; ++r) {
      int best = -1;
      for (int t = 0; t < q.n_terms; ++t)
        if (tab_of_term[(size_t)t] < 0 && term_total[(size_t)t] > 0 && (best < 0 || term_total[(size_t)t] > term_total[(size_t)best])) best = t;
      if (best < 0) break;
      tab_of_term[(size_t)best] = (int32_t)qt_.n;
      qt_.weight[qt_.n] = q.terms[best].weight;
      qt_.cache[qt_.n] = (uint32_t)q.terms[best].cache_slot;
      qt_.scale[qt_.n] = term_scale[(size_t)best];
      qt_.n++;
    }
    per_query[(size_t)qi].reserve((size_t)n_segs);
    for (int si = 0; si < n_segs; ++si) {
      const nrtgpu_seg* seg = segs[si];
      QS qs{(uint32_t)pc.terms.size(), 0, si, 0};
      for (int t = 0; t < q.n_terms; ++t) {
        const nrtgpu_term& qt = q.terms[t];
        const TermEntry* ep = found[(size_t)t * n_segs + si];
        if (!ep) continue;
        const TermEntry& e = *ep;
        const FieldData& f = *found_field[(size_t)t * n_segs + si];
        const TermGroup& g = f.groups[e.group];
        DTerm d{};
        d.docids = g.d_docids;
        d.fnorm = g.d_fnorm;
        d.cell_off = g.d_cells + e.cell_start;
        d.start = e.start;
        d.count = e.count;
        d.shift = e.shift;
        d.weight = qt.weight;
        d.cache_off = cache_base[(size_t)qi] + (uint32_t)qt.cache_slot * 256u;
        d.cache_slot = (uint32_t)qt.cache_slot;
        d.tab_slot = tab_of_term[(size_t)t] >= 0 ? (uint32_t)tab_of_term[(size_t)t] : 0xFFFFFFFFu;
        d.fx_scale = term_scale[(size_t)t];
        d.fx_shift = fx_ok ? (uint32_t)(fx_E - term_scale[(size_t)t]) : 0u;
        pc.terms.push_back(d);
        qs.n_terms++;
        qs.postings += e.count;
      }
      if (qs.n_terms > 0) {
        // densest clause first; stable insertion sort (a handful of clauses; std::stable_sort allocates per call)
        DTerm* tb = pc.terms.data() + qs.term_begin;
        for (uint32_t i = 1; i < qs.n_terms; ++i) {
          const DTerm key = tb[i];
          uint32_t j = i;
          for (; j > 0 && tb[j - 1].count < key.count; --j) tb[j] = tb[j - 1];
          tb[j] = key;
        }
        per_query[(size_t)qi].push_back(qs);
        pc.postings += qs.postings;
        pc.cost += qs.postings + (int64_t)seg->n_tiles * kTileCostPostings;
      }
    }
  }
}

static int build_plan(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                      const nrtgpu_bm25_query* queries, int32_t n_queries, HostPlan& hp) {
  uint32_t kmax = 1;
  for (int qi = 0; qi < n_queries; ++qi) {
    if (int rc = validate_query(queries[qi], qi)) return rc;
    kmax = std::max<uint32_t>(kmax, (uint32_t)queries[qi].k);
  }
  for (int si = 0; si < n_segs; ++si) {
    if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
    if (!segs[si]->sealed) return fail(NRTGPU_ERR_STATE, "segment %d is not sealed", si);
    if (segs[si]->ctx != ctx) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d belongs to another context", si);
  }
  hp.k_stride = round_up(kmax, 16);
  hp.queries.resize((size_t)n_queries);
  hp.q_k.resize((size_t)n_queries);
  hp.theta_init.resize((size_t)n_queries);
  for (int qi = 0; qi < n_queries; ++qi)  // lowest key with that score: a doc scoring exactly the bound still passes
    hp.theta_init[(size_t)qi] = queries[qi].min_competitive_score > 0.0f ? pack_key(queries[qi].min_competitive_score, 0xFFFFFFFFu) : 0ull;

  static const bool plan_trace = getenv("NRTGPU_PLAN_TRACE") != nullptr;  // debug aid: phase times on stderr
  const double tp0 = plan_trace ? now_ms() : 0.0;
  // pass 1: resolve terms per (query, segment), densest term first; remember posting counts.
  // Queries are independent here, so the batch is cut into contiguous chunks resolved by
  // cfg.host_threads planner threads and concatenated (offsets rebased) afterwards.
  std::vector<std::vector<QS>> per_query((size_t)n_queries);
  std::vector<uint32_t> cache_base((size_t)n_queries);
  std::vector<QTabs> qtabs((size_t)n_queries);
  int n_thr = ctx->cfg.host_threads > 0 ? ctx->cfg.host_threads : 4;
  n_thr = std::max(1, std::min(n_thr, n_queries / 64));
  std::vector<PlanPiece> pieces((size_t)n_thr);
  auto chunk_begin = [&](int t) { return (int)((int64_t)n_queries * t / n_thr); };
  auto work = [&](int t) {
    resolve_queries(segs, n_segs, queries, chunk_begin(t), chunk_begin(t + 1), pieces[(size_t)t], per_query, cache_base, qtabs);
  };
  {
    std::vector<std::thread> pool;
    for (int t = 1; t < n_thr; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
  }
  const double tp1 = plan_trace ? now_ms() : 0.0;
  int64_t total_postings = 0, total_cost = 0;
  {
    size_t nt = 0, nc = 0;
    for (const PlanPiece& pc : pieces) { nt += pc.terms.size(); nc += pc.caches.size(); }
    hp.terms.reserve(nt);
    hp.caches.reserve(nc);
  }
  for (int t = 0; t < n_thr; ++t) {
    PlanPiece& pc = pieces[(size_t)t];
    const uint32_t term_base = (uint32_t)hp.terms.size(), c_base = (uint32_t)hp.caches.size();
    for (DTerm& d : pc.terms) d.cache_off += c_base;
    hp.terms.insert(hp.terms.end(), pc.terms.begin(), pc.terms.end());
    hp.caches.insert(hp.caches.end(), pc.caches.begin(), pc.caches.end());
    for (int qi = chunk_begin(t); qi < chunk_begin(t + 1); ++qi) {
      cache_base[(size_t)qi] += c_base;
      for (QS& qs : per_query[(size_t)qi]) qs.term_begin += term_base;
    }
    total_postings += pc.postings;
    total_cost += pc.cost;
  }
  hp.postings = total_postings;
  hp.fixed_point = (ctx->cfg.flags & NRTGPU_FLAG_NO_FIXED_POINT) == 0;
  for (int qi = 0; qi < n_queries && hp.fixed_point; ++qi)
    if (!per_query[(size_t)qi].empty() && qtabs[(size_t)qi].fx_E == kNoFixed) hp.fixed_point = false;
  // minimumNumberShouldMatch > 1 (QueryNodeMapper.java:259-261): the clause count rides in the fixed-point
  // accumulator, so the whole batch must be in fixed-point mode; otherwise the caller runs Lucene's WANDScorer
  hp.clause_counting = false;
  for (int qi = 0; qi < n_queries; ++qi)
    if (queries[qi].min_should_match > 1) hp.clause_counting = true;
  if (hp.clause_counting && !hp.fixed_point)
    return fail(NRTGPU_ERR_UNSUPPORTED, "minimumNumberShouldMatch > 1 needs the fixed-point accumulators (weights of a query in "
                                        "this batch span too many binades, or NRTGPU_FLAG_NO_FIXED_POINT is set)");

  const double tp2 = plan_trace ? now_ms() : 0.0;
  // pass 2: cut every query's leaves (in docBase order) into items of roughly equal cost.  An item
  // may span several segments (like a LeafSlice) and a large segment may be cut by tile range.
  // Measured on MI355X (one workgroup per CU): every extra item of a query costs a cold
  // top-k start, so a query is cut only when it alone would take longer than its fair share of the
  // batch on one CU.  target_items == 0 => one share per CU.
  const int64_t target_items = ctx->cfg.target_items > 0 ? ctx->cfg.target_items : (int64_t)std::max(ctx->n_cus, 1);
  const int64_t min_item_cost = 1 << 17;
  const int64_t per_item = std::max<int64_t>(min_item_cost, total_cost / std::max<int64_t>(1, target_items));
  struct Pending { int64_t cost; uint32_t query; uint32_t part_begin, n_parts; uint32_t tiles; };
  std::vector<Pending> pend;
  std::vector<int64_t> q_costs((size_t)n_queries, 0), q_items((size_t)n_queries, 0);
  int64_t n_live = 0, n_items_total = 0;
  for (int qi = 0; qi < n_queries; ++qi) {
    for (const QS& qs : per_query[(size_t)qi]) q_costs[(size_t)qi] += qs.postings + (int64_t)segs[qs.seg]->n_tiles * kTileCostPostings;
    if (q_costs[(size_t)qi] == 0) continue;
    ++n_live;
    q_items[(size_t)qi] = std::max<int64_t>(1, (q_costs[(size_t)qi] + per_item / 2) / per_item);
    n_items_total += q_items[(size_t)qi];
  }
  // A small batch is cut into EXACTLY one item per CU: rounding each query on its own gives a few items more
  // than CUs, and near-equal items then run in two rounds with most CUs idle in the second (64 queries: 273
  // items on 256 CUs).  Largest-remainder apportionment of the CUs over the queries by cost.
  if (n_live > 0 && n_live * 2 <= target_items && n_items_total > target_items && total_cost >= target_items * min_item_cost) {
    std::vector<std::pair<double, int>> frac;
    int64_t given = 0;
    for (int qi = 0; qi < n_queries; ++qi) {
      if (q_costs[(size_t)qi] == 0) continue;
      const double share = (double)q_costs[(size_t)qi] * (double)target_items / (double)total_cost;
      q_items[(size_t)qi] = std::max<int64_t>(1, (int64_t)share);
      given += q_items[(size_t)qi];
      frac.emplace_back(share - std::floor(share), qi);
    }
    std::sort(frac.begin(), frac.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
    for (size_t i = 0; i < frac.size() && given < target_items; ++i, ++given) q_items[(size_t)frac[i].second]++;
  }
  for (int qi = 0; qi < n_queries; ++qi) {
    const int64_t q_cost = q_costs[(size_t)qi];
    if (q_cost == 0) continue;
    const int64_t n_it = q_items[(size_t)qi];
    const double budget = (double)q_cost / (double)n_it;
    Pending cur{0, (uint32_t)qi, (uint32_t)hp.parts.size(), 0, 0};
    double filled = 0.0;
    for (const QS& qs : per_query[(size_t)qi]) {
      const nrtgpu_seg* seg = segs[qs.seg];
      const double tile_cost = (double)qs.postings / (double)seg->n_tiles + (double)kTileCostPostings;
      const uint64_t* accept = nullptr;  // liveDocs, narrowed by the query's FILTER / MUST_NOT masks
      if (int rc = accept_set_of(seg, queries[qi].filter_mask, queries[qi].must_not_mask, &accept)) return rc;
      uint32_t tb = 0;
      while (tb < seg->n_tiles) {
        double room = budget - filled;
        uint32_t take = (uint32_t)std::max(1.0, std::floor(room / tile_cost + 0.5));
        take = std::min<uint32_t>(take, seg->n_tiles - tb);
        DPart p{};
        p.live_bits = accept;
        if (accept) hp.masked = true;
        p.term_begin = qs.term_begin;
        p.n_terms = qs.n_terms;
        p.tile_begin = tb;
        p.tile_end = tb + take;
        p.max_doc = (uint32_t)seg->max_doc;
        p.doc_base = doc_bases ? doc_bases[qs.seg] : 0;
        p.tile_offset = cur.tiles;
        hp.parts.push_back(p);
        cur.n_parts++;
        cur.tiles += take;
        cur.cost += (int64_t)(take * tile_cost);
        filled += take * tile_cost;
        tb += take;
        if (filled >= budget * 0.999) {  // item full: close it
          pend.push_back(cur);
          cur = Pending{0, (uint32_t)qi, (uint32_t)hp.parts.size(), 0, 0};
          filled = 0.0;
        }
      }
    }
    if (cur.n_parts > 0) {
      // A short remainder (the tile rounding of the items before it) does not become an item of its own: it would
      // finish without a single compaction, never publish its quantile, and with one peer silent the bound
      // exchange between the query's items never forms (kernels.hip: peers_bound).  It joins the item before it.
      if (!pend.empty() && pend.back().query == (uint32_t)qi && (double)cur.cost < 0.5 * budget &&
          pend.back().part_begin + pend.back().n_parts == cur.part_begin) {
        Pending& prev = pend.back();
        for (uint32_t pi2 = 0; pi2 < cur.n_parts; ++pi2) hp.parts[cur.part_begin + pi2].tile_offset += prev.tiles;
        prev.n_parts += cur.n_parts;
        prev.tiles += cur.tiles;
        prev.cost += cur.cost;
      } else {
        pend.push_back(cur);
      }
    }
  }
  // longest-processing-time-first launch order: the hardware dispatcher hands out workgroups in
  // index order, so big items start first and small ones fill the tail
  // (cf. slices ordered largest first, MyIndexSearcher.java:154-158)
  std::stable_sort(pend.begin(), pend.end(), [](const Pending& a, const Pending& b) { return a.cost > b.cost; });
  hp.items.resize(pend.size());
  std::vector<std::vector<uint32_t>> lists((size_t)n_queries);
  for (size_t i = 0; i < pend.size(); ++i) {
    DItem it{};
    it.query = pend[i].query;
    it.part_begin = pend[i].part_begin;
    it.n_parts = pend[i].n_parts;
    it.cache_off = cache_base[pend[i].query];
    it.n_caches = (uint32_t)queries[pend[i].query].n_caches;
    const QTabs& qt_ = qtabs[pend[i].query];
    it.n_tabs = qt_.n;
    it.fx_E = qt_.fx_E;
    for (uint32_t r = 0; r < qt_.n; ++r) {
      it.tab_weight[r] = qt_.weight[r];
      it.tab_cache[r] = qt_.cache[r];
      it.tab_scale[r] = qt_.scale[r];
    }
    it.peer_slot = (uint32_t)lists[pend[i].query].size();  // rebased by the query's list offset below
    hp.items[i] = it;
    lists[pend[i].query].push_back((uint32_t)i);
  }
  hp.q_base.resize((size_t)n_queries);
  hp.q_nlists.resize((size_t)n_queries);
  for (int qi = 0; qi < n_queries; ++qi) {
    const nrtgpu_bm25_query& q = queries[qi];
    hp.q_base[(size_t)qi] = (uint32_t)hp.list_idx.size();
    hp.q_nlists[(size_t)qi] = (uint32_t)lists[(size_t)qi].size();
    hp.list_idx.insert(hp.list_idx.end(), lists[(size_t)qi].begin(), lists[(size_t)qi].end());
    for (uint32_t ii : lists[(size_t)qi]) hp.items[ii].peer_slot += hp.q_base[(size_t)qi];
    hp.q_k[(size_t)qi] = (uint32_t)q.k;
    DQuery& dq = hp.queries[(size_t)qi];
    dq.k = (uint32_t)q.k;
    dq.has_after = q.has_after ? 1u : 0u;
    dq.after_doc = q.after_doc;
    dq.after_score = q.after_score;
    dq.item_begin = hp.q_base[(size_t)qi];
    dq.n_items = hp.q_nlists[(size_t)qi];
    dq.min_should_match = (uint32_t)std::max(q.min_should_match, 0);
  }
  if (plan_trace)
    fprintf(stderr, "[nrtgpu plan] %d queries: resolve %.3f ms (%d threads), concat %.3f, cut+items %.3f; %zu terms %zu parts %zu items\n",
            n_queries, tp1 - tp0, n_thr, tp2 - tp1, now_ms() - tp2, hp.terms.size(), hp.parts.size(), hp.items.size());
  return 0;
}

// layout helper: carve 256-byte aligned regions out of one blob
struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
};

struct DeviceRun {
  // device pointers valid until the slot is reused
  uint64_t* out_keys = nullptr;
  uint32_t* out_counts = nullptr;
  uint64_t* out_hits = nullptr;
  uint64_t* prof = nullptr;   // instrumented variant: 8 counters per item
  size_t n_items = 0;
};

// Enqueue plan upload + scan + merge on the slot's stream.  Merge output goes to (ext_keys,
// ext_counts, ext_hits) when given (device-resident variant), else into the slot's scratch.
// `gpu` (unlocked on entry) is taken only once the plan has reached the device: the upload of this
// batch overlaps the kernels of the batch another host thread has in flight.
static int enqueue_search(nrtgpu_ctx* ctx, Slot* slot, const HostPlan& hp, int32_t n_queries, uint32_t k_stride_out,
                          uint64_t* ext_keys, uint32_t* ext_counts, uint64_t* ext_hits, DeviceRun* run,
                          std::unique_lock<std::mutex>& gpu, int64_t epoch = -1) {
  const size_t n_items = hp.items.size();
  Carver pc;
  const size_t o_queries = pc.take(hp.queries.size() * sizeof(DQuery));
  const size_t o_items = pc.take(n_items * sizeof(DItem));
  const size_t o_parts = pc.take(hp.parts.size() * sizeof(DPart));
  const size_t o_terms = pc.take(hp.terms.size() * sizeof(DTerm));
  const size_t o_caches = pc.take(hp.caches.size() * sizeof(float));
  const size_t o_lidx = pc.take(hp.list_idx.size() * 4);
  const size_t o_qbase = pc.take(hp.q_base.size() * 4);
  const size_t o_qnl = pc.take(hp.q_nlists.size() * 4);
  const size_t o_qk = pc.take(hp.q_k.size() * 4);
  const size_t o_theta = pc.take(hp.theta_init.size() * 8);  // uploaded with the plan, then updated by the kernel
  const size_t o_quant = pc.take(hp.list_idx.size() * 8);    // per item: published quantile bound (zeros)
  const bool use_xch = epoch >= 0 && ctx->xch_dev != nullptr;
  const size_t o_xch = pc.take(use_xch ? sizeof(DExchange) : 0);
  const size_t plan_bytes = pc.off;
  if (int rc = slot->h_plan.reserve(plan_bytes)) return rc;
  if (int rc = slot->d_plan.reserve(plan_bytes)) return rc;
  char* hb = (char*)slot->h_plan.p;
  memcpy(hb + o_queries, hp.queries.data(), hp.queries.size() * sizeof(DQuery));
  if (n_items) memcpy(hb + o_items, hp.items.data(), n_items * sizeof(DItem));
  if (!hp.parts.empty()) memcpy(hb + o_parts, hp.parts.data(), hp.parts.size() * sizeof(DPart));
  if (!hp.terms.empty()) memcpy(hb + o_terms, hp.terms.data(), hp.terms.size() * sizeof(DTerm));
  memcpy(hb + o_caches, hp.caches.data(), hp.caches.size() * sizeof(float));
  if (!hp.list_idx.empty()) memcpy(hb + o_lidx, hp.list_idx.data(), hp.list_idx.size() * 4);
  memcpy(hb + o_qbase, hp.q_base.data(), hp.q_base.size() * 4);
  memcpy(hb + o_qnl, hp.q_nlists.data(), hp.q_nlists.size() * 4);
  memcpy(hb + o_qk, hp.q_k.data(), hp.q_k.size() * 4);
  memcpy(hb + o_theta, hp.theta_init.data(), hp.theta_init.size() * 8);
  memset(hb + o_quant, 0, hp.list_idx.size() * 8);
  if (use_xch) {
    DExchange x{};
    const size_t stride = (size_t)ctx->cfg.max_batch;
    x.slot = ctx->xch_dev + (size_t)(epoch % kExchangeSlots) * (size_t)ctx->xch_world * stride;
    x.world = (uint32_t)ctx->xch_world;
    x.rank = (uint32_t)ctx->xch_rank;
    x.stride = (uint32_t)stride;
    x.tag = (uint32_t)(epoch + 1);  // never 0
    if (x.tag == 0) x.tag = 1;
    memcpy(hb + o_xch, &x, sizeof(x));
  }

  Carver wc;
  const size_t o_ikeys = wc.take(n_items * (size_t)hp.k_stride * 8);
  const size_t o_icnt = wc.take(n_items * 4);
  const size_t o_ihits = wc.take(n_items * 8);
  const size_t o_okeys = wc.take((size_t)n_queries * k_stride_out * 8);
  const size_t o_ocnt = wc.take((size_t)n_queries * 4);
  const size_t o_ohits = wc.take((size_t)n_queries * 8);
  // kernel variant: clause counting (8), doc-set masks somewhere in the batch (9), else what the flags ask for
  const int flag_variant = (ctx->cfg.flags >> 8) & 15;
  const int ablation = hp.clause_counting ? 8 : ((hp.masked && flag_variant == 0 && !(ctx->cfg.flags & NRTGPU_FLAG_NO_MASK_VARIANT)) ? 9 : flag_variant);
  const size_t o_prof = wc.take(ablation == 7 ? n_items * 128 : 0);
  if (int rc = slot->d_work.reserve(wc.off)) return rc;
  char* db = (char*)slot->d_plan.p;
  char* wb = (char*)slot->d_work.p;

  hipStream_t st = slot->stream;
  HIP_TRY(hipMemcpyAsync(db, hb, plan_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  gpu.lock();
  const bool timing = ctx->cfg.collect_timing != 0;
  if (timing) HIP_TRY(hipEventRecord(slot->ev0, st));
  launch_bm25_scan(st, hp.fixed_point, (ctx->cfg.flags & NRTGPU_FLAG_NO_PREFETCH) == 0, ablation, (uint32_t)n_items, (const DItem*)(db + o_items), (const DPart*)(db + o_parts), (const DTerm*)(db + o_terms),
                   (const DQuery*)(db + o_queries), (const float*)(db + o_caches),
                   (unsigned long long*)(db + o_theta), (unsigned long long*)(db + o_quant),
                   use_xch ? (const DExchange*)(db + o_xch) : nullptr, (uint64_t*)(wb + o_ikeys), (uint32_t*)(wb + o_icnt),
                   (uint64_t*)(wb + o_ihits), hp.k_stride, ablation == 7 ? (uint64_t*)(wb + o_prof) : nullptr);
  if (timing) HIP_TRY(hipEventRecord(slot->ev1, st));
  uint64_t* okeys = ext_keys ? ext_keys : (uint64_t*)(wb + o_okeys);
  uint32_t* ocnt = ext_counts ? ext_counts : (uint32_t*)(wb + o_ocnt);
  uint64_t* ohits = ext_hits ? ext_hits : (uint64_t*)(wb + o_ohits);
  launch_merge_topk(st, (uint32_t)n_queries, (const uint64_t*)(wb + o_ikeys), (const uint32_t*)(wb + o_icnt),
                    (const uint64_t*)(wb + o_ihits), (const uint32_t*)(db + o_lidx), (const uint32_t*)(db + o_qbase),
                    (const uint32_t*)(db + o_qnl), hp.k_stride, (const uint32_t*)(db + o_qk), okeys, ocnt, ohits,
                    k_stride_out);
  if (timing) HIP_TRY(hipEventRecord(slot->ev2, st));
  HIP_TRY(hipGetLastError());
  run->out_keys = okeys;
  run->out_counts = ocnt;
  run->out_hits = ohits;
  run->prof = ablation == 7 ? (uint64_t*)(wb + o_prof) : nullptr;
  run->n_items = n_items;
  return 0;
}

static void account(nrtgpu_ctx* ctx, Slot* slot, const HostPlan& hp, int32_t n_queries, double plan_ms) {
  float scan_ms = 0.f, merge_ms = 0.f;
  if (ctx->cfg.collect_timing) {
    (void)hipEventElapsedTime(&scan_ms, slot->ev0, slot->ev1);
    (void)hipEventElapsedTime(&merge_ms, slot->ev1, slot->ev2);
  }
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  ctx->stats.batches += 1;
  ctx->stats.queries += n_queries;
  ctx->stats.scan_launches += hp.items.empty() ? 0 : 1;
  ctx->stats.fixed_point_launches += (!hp.items.empty() && hp.fixed_point) ? 1 : 0;
  ctx->stats.scan_ms += scan_ms;
  ctx->stats.merge_ms += merge_ms;
  ctx->stats.scan_postings += hp.postings;
  ctx->stats.scan_items += (int64_t)hp.items.size();
  ctx->stats.host_plan_ms += plan_ms;
}

// relation: GREATER_THAN_OR_EQUAL_TO exactly where LazyQueueTopScoreDocCollector would have started
// publishing a min competitive score: totalHits > max(threshold, numHits) and the queue is full
// (LazyQueueTopScoreDocCollector.java:176-199, …Manager.java:102).
static inline int32_t relation_gte(int64_t total_hits, int32_t n_hits, int32_t k, int32_t threshold) {
  const int64_t thr = std::max<int64_t>(threshold, k);
  return (total_hits > thr && n_hits == k) ? 1 : 0;
}

static void unpack_topdocs(const uint64_t* keys, uint32_t n, uint64_t hits, const int32_t k, const int32_t threshold,
                           nrtgpu_topdocs* out) {
  const int32_t cap = out->capacity > 0 ? out->capacity : k;
  const int32_t m = std::min<int32_t>((int32_t)n, cap);
  // two plain loops (vectorisable): doc = ~low word, score = high word reinterpreted
  if (int32_t* __restrict__ docs = out->docs)
    for (int32_t i = 0; i < m; ++i) docs[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)keys[i]);
  if (uint32_t* __restrict__ sc = (uint32_t*)out->scores)
    for (int32_t i = 0; i < m; ++i) sc[i] = (uint32_t)(keys[i] >> 32);
  out->n_hits = m;
  out->total_hits = (int64_t)hits;
  out->total_hits_is_lower_bound = relation_gte((int64_t)hits, (int32_t)n, k, threshold);
}

// ------------------------------------------------------------------------------------------------
// ABI: search
// ------------------------------------------------------------------------------------------------
extern "C" int nrtgpu_search_bm25_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                        int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                        nrtgpu_topdocs* out) {
  if (!ctx || !queries || !out || (n_segs > 0 && !segs)) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_queries <= 0 || n_segs < 0) return fail(NRTGPU_ERR_INVALID_ARG, "n_queries must be > 0");
  if (n_queries > ctx->cfg.max_batch) return fail(NRTGPU_ERR_INVALID_ARG, "batch of %d exceeds max_batch %d", n_queries, ctx->cfg.max_batch);
  HIP_TRY(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  HostPlan hp;
  for (int si = 0; si < n_segs; ++si)
    if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
  SegReadLocks content(segs, n_segs);  // until this call's kernels have finished
  if (int rc = build_plan(ctx, segs, doc_bases, n_segs, queries, n_queries, hp)) return rc;
  const double plan_ms = now_ms() - t0;

  Slot* slot = nullptr;
  acquire_slot(ctx, &slot);
  struct Guard { nrtgpu_ctx* c; Slot* s; ~Guard() { release_slot(c, s); } } guard{ctx, slot};
  DeviceRun run;
  const size_t kb = (size_t)n_queries * hp.k_stride * 8, cb = (size_t)n_queries * 4, hb = (size_t)n_queries * 8;
  Carver oc;
  const size_t o_k = oc.take(kb), o_c = oc.take(cb), o_h = oc.take(hb);
  if (int rc = slot->h_out.reserve(oc.off)) return rc;
  char* ho = (char*)slot->h_out.p;
  {
    std::unique_lock<std::mutex> gpu(ctx->gpu_mu, std::defer_lock);
    if (int rc = enqueue_search(ctx, slot, hp, n_queries, hp.k_stride, nullptr, nullptr, nullptr, &run, gpu)) return rc;
    HIP_TRY(hipStreamSynchronize(slot->stream));  // kernels done: the next batch may have the device ...
  }
  // ... while this one's results travel to the host
  HIP_TRY(hipMemcpyAsync(ho + o_k, run.out_keys, kb, hipMemcpyDeviceToHost, slot->stream));
  HIP_TRY(hipMemcpyAsync(ho + o_c, run.out_counts, cb, hipMemcpyDeviceToHost, slot->stream));
  HIP_TRY(hipMemcpyAsync(ho + o_h, run.out_hits, hb, hipMemcpyDeviceToHost, slot->stream));
  HIP_TRY(hipStreamSynchronize(slot->stream));
  const uint64_t* keys = (const uint64_t*)(ho + o_k);
  const uint32_t* cnts = (const uint32_t*)(ho + o_c);
  const uint64_t* hits = (const uint64_t*)(ho + o_h);
  for (int qi = 0; qi < n_queries; ++qi)
    unpack_topdocs(keys + (size_t)qi * hp.k_stride, cnts[qi], hits[qi], queries[qi].k, queries[qi].total_hits_threshold, &out[qi]);
  if (run.prof && run.n_items) {
    std::vector<uint64_t> hp_prof(run.n_items * 16);
    HIP_TRY(hipMemcpy(hp_prof.data(), run.prof, hp_prof.size() * 8, hipMemcpyDeviceToHost));
    std::lock_guard<std::mutex> lk(ctx->stats_mu);
    for (size_t i = 0; i < run.n_items; ++i)
      for (int j = 0; j < 16; ++j) ctx->prof[j] += (double)hp_prof[i * 16 + j];
  }
  account(ctx, slot, hp, n_queries, plan_ms);
  return NRTGPU_OK;
}

// Hybrid tail: BM25 recall -> exact-vector rescore -> window, one stream, no host round trip between
// the stages (RescoreTask.java:47-50 -> QueryRescore.java:39-57 applied to the hits of
// SearchHandler.java:1412-1413).  Same results as nrtgpu_search_bm25_batch followed per query by
// nrtgpu_rescore_vectors.
extern "C" int nrtgpu_search_hybrid_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                          int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                          int32_t field_id, int32_t sim, const float* query_vectors, int32_t dim, float boost,
                                          double query_weight, double rescore_weight, int32_t window, nrtgpu_topdocs* out) {
  if (!ctx || !queries || !out || !query_vectors || (n_segs > 0 && (!segs || !doc_bases)))
    return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_queries <= 0 || n_segs < 0) return fail(NRTGPU_ERR_INVALID_ARG, "n_queries must be > 0");
  if (n_queries > ctx->cfg.max_batch) return fail(NRTGPU_ERR_INVALID_ARG, "batch of %d exceeds max_batch %d", n_queries, ctx->cfg.max_batch);
  if (dim <= 0 || sim < 0 || sim > 3 || window <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad rescore arguments");
  if (!(query_weight >= 0.0) || !(rescore_weight >= 0.0) || !(boost >= 0.0f))
    return fail(NRTGPU_ERR_UNSUPPORTED, "hybrid tail: negative weights (combined scores must stay >= 0)");
  HIP_TRY(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  HostPlan hp;
  for (int si = 0; si < n_segs; ++si)
    if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
  SegReadLocks content(segs, n_segs);  // until this call's kernels have finished
  if (int rc = build_plan(ctx, segs, doc_bases, n_segs, queries, n_queries, hp)) return rc;
  const double plan_ms = now_ms() - t0;
  for (int si = 0; si < n_segs; ++si) {
    auto fit = segs[si]->fields.find(field_id);
    if (fit != segs[si]->fields.end() && fit->second.d_vectors && fit->second.dim != dim)
      return fail(NRTGPU_ERR_INVALID_ARG, "vector dimension mismatch");
  }
  Slot* slot = nullptr;
  acquire_slot(ctx, &slot);
  struct Guard { nrtgpu_ctx* c; Slot* s; ~Guard() { release_slot(c, s); } } guard{ctx, slot};
  hipStream_t st = slot->stream;
  const uint32_t w_stride = round_up((uint32_t)std::min<int32_t>(window, NRTGPU_MAX_K), 16);
  const size_t nq = (size_t)n_queries;
  Carver ac;
  const size_t o_segs = ac.take((size_t)std::max(n_segs, 1) * sizeof(DVecSeg)), o_qv = ac.take(nq * (size_t)dim * 4),
               o_qn = ac.take(nq * 4);
  const size_t in_bytes = ac.off;
  const size_t o_wk = ac.take(nq * w_stride * 8), o_wc = ac.take(nq * 4);
  if (int rc = slot->d_aux.reserve(ac.off)) return rc;
  const size_t kb = nq * w_stride * 8, cb = nq * 4, hb = nq * 8;
  Carver hc;
  const size_t oh_in = hc.take(in_bytes), oh_k = hc.take(kb), oh_c = hc.take(cb), oh_fc = hc.take(cb), oh_h = hc.take(hb);
  if (int rc = slot->h_aux.reserve(hc.off)) return rc;
  char* ha = (char*)slot->h_aux.p;
  char* da = (char*)slot->d_aux.p;
  DVecSeg* hs = (DVecSeg*)(ha + oh_in + o_segs);
  for (int si = 0; si < n_segs; ++si) {
    DVecSeg v{};
    auto fit = segs[si]->fields.find(field_id);
    if (fit != segs[si]->fields.end() && fit->second.d_vectors) {
      v.vecs = fit->second.d_vectors;
      v.vnorm2 = fit->second.d_vnorm2;
      v.ord_to_doc = fit->second.d_ord_to_doc;
      v.n_vec = fit->second.n_vec;
    }
    v.doc_base = doc_bases[si];
    v.max_doc = segs[si]->max_doc;
    hs[si] = v;
  }
  memcpy(ha + oh_in + o_qv, query_vectors, nq * (size_t)dim * 4);
  float* hqn = (float*)(ha + oh_in + o_qn);
  for (size_t q = 0; q < nq; ++q) {  // |q|^2 in the order nrtgpu_rescore_vectors uses
    const float* qv = query_vectors + q * (size_t)dim;
    float qn = 0.f;
    for (int d = 0; d < dim; ++d) {
      volatile float p2 = qv[d] * qv[d];
      qn = qn + p2;
    }
    hqn[q] = qn;
  }
  HIP_TRY(hipMemcpyAsync(da, ha + oh_in, in_bytes, hipMemcpyHostToDevice, st));
  DeviceRun run;
  {
    std::unique_lock<std::mutex> gpu(ctx->gpu_mu, std::defer_lock);
    if (int rc = enqueue_search(ctx, slot, hp, n_queries, hp.k_stride, nullptr, nullptr, nullptr, &run, gpu)) return rc;
    launch_hybrid_rescore(st, (uint32_t)n_queries, run.out_keys, run.out_counts, hp.k_stride, (const DVecSeg*)(da + o_segs), n_segs,
                          dim, (const float*)(da + o_qv), (const float*)(da + o_qn), sim, boost, query_weight, rescore_weight,
                          (uint32_t)window, (uint64_t*)(da + o_wk), (uint32_t*)(da + o_wc), w_stride);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
  }
  HIP_TRY(hipMemcpyAsync(ha + oh_k, da + o_wk, kb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ha + oh_c, da + o_wc, cb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ha + oh_fc, run.out_counts, cb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ha + oh_h, run.out_hits, hb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const uint64_t* keys = (const uint64_t*)(ha + oh_k);
  const uint32_t* cnts = (const uint32_t*)(ha + oh_c);
  const uint32_t* first_cnts = (const uint32_t*)(ha + oh_fc);
  const uint64_t* hits = (const uint64_t*)(ha + oh_h);
  for (int qi = 0; qi < n_queries; ++qi) {
    // QueryRescorer keeps the first pass's TotalHits; the window only trims the hits
    unpack_topdocs(keys + (size_t)qi * w_stride, cnts[qi], hits[qi], std::min<int32_t>(window, NRTGPU_MAX_K), queries[qi].total_hits_threshold, &out[qi]);
    out[qi].total_hits_is_lower_bound = relation_gte((int64_t)hits[qi], (int32_t)first_cnts[qi], queries[qi].k, queries[qi].total_hits_threshold);
  }
  account(ctx, slot, hp, n_queries, plan_ms);
  return NRTGPU_OK;
}

extern "C" int nrtgpu_search_bm25(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                  const nrtgpu_bm25_query* q, nrtgpu_topdocs* out) {
  return nrtgpu_search_bm25_batch(ctx, segs, doc_bases, n_segs, q, 1, out);
}

// ------------------------------------------------------------------------------------------------
// Request coalescing: concurrent single-query callers (the SEARCH pool's threads) are merged into
// device batches leader/follower style -- no extra thread.  The first caller to find no lingering
// leader becomes one: it waits co_linger_us (or until max_batch requests are pending), takes every
// pending request that searches the same leaves, runs them as one batch and wakes their callers.
// Later arrivals elect the next leader, so two batches are in flight and planning overlaps kernels.
// ------------------------------------------------------------------------------------------------
struct CoRequest {
  const nrtgpu_seg* const* segs;
  const int32_t* doc_bases;
  int32_t n_segs;
  const nrtgpu_bm25_query* q;
  nrtgpu_topdocs* out;
  int rc = 0;
  bool done = false;   // results (or the error) are in place
  bool lead = false;   // promoted: this caller lingers for and runs the next batch
  std::string err;
  // Every caller sleeps on its own condition variable AND its own mutex: a finished batch wakes hundreds of
  // callers, and if they all had to re-acquire the coalescer's lock to leave their wait (and again to submit
  // their next request) the lock handoffs alone would cost more than the batch's kernels.  done / lead are
  // written under `m`; the lingering leader is the one waiter that uses `cv` with the coalescer's lock.
  std::mutex m;
  std::condition_variable cv;
};

static bool same_leaves(const CoRequest* a, const CoRequest* b) {
  if (a->n_segs != b->n_segs) return false;
  if (a->n_segs == 0) return true;
  if (memcmp(a->segs, b->segs, (size_t)a->n_segs * sizeof(void*)) != 0) return false;
  if ((a->doc_bases == nullptr) != (b->doc_bases == nullptr)) return false;
  return !a->doc_bases || memcmp(a->doc_bases, b->doc_bases, (size_t)a->n_segs * 4) == 0;
}

extern "C" int nrtgpu_set_coalescing(nrtgpu_ctx* ctx, int32_t linger_us) {
  if (!ctx || linger_us < 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad coalescing arguments");
  std::lock_guard<std::mutex> lk(ctx->co_mu);
  ctx->co_linger_us = linger_us;
  return NRTGPU_OK;
}

extern "C" int nrtgpu_search_bm25_coalesced(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                            int32_t n_segs, const nrtgpu_bm25_query* q, nrtgpu_topdocs* out) {
  if (!ctx || !q || !out || (n_segs > 0 && !segs)) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_segs < 0) return fail(NRTGPU_ERR_INVALID_ARG, "n_segs must be >= 0");
  if (int rc = validate_query(*q, 0)) return rc;  // a bad request must not fail its batch mates
  if (q->min_should_match > 1)  // whether it can run depends on the whole batch (fixed-point mode): use the batch call
    return fail(NRTGPU_ERR_UNSUPPORTED, "minimumNumberShouldMatch > 1 is not coalesced");
  for (int si = 0; si < n_segs; ++si) {
    if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
    if (!segs[si]->sealed) return fail(NRTGPU_ERR_STATE, "segment %d is not sealed", si);
    if (segs[si]->ctx != ctx) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d belongs to another context", si);
  }
  CoRequest me{segs, doc_bases, n_segs, q, out};
  std::vector<CoRequest*> batch;
  {
    std::unique_lock<std::mutex> lk(ctx->co_mu);
    ctx->co_pending.push_back(&me);
    if (ctx->co_leader) {  // follower: the lingering leader takes this request (or a later one does)
      if ((int32_t)ctx->co_pending.size() >= ctx->cfg.max_batch) ctx->co_leader->cv.notify_one();
      lk.unlock();
      {
        std::unique_lock<std::mutex> mine(me.m);
        me.cv.wait(mine, [&] { return me.done || me.lead; });
      }
      if (me.done) {
        if (me.rc != 0) g_last_error = me.err;
        return me.rc;
      }
      lk.lock();  // promoted: continue as the leader
    } else {
      ctx->co_leader = &me;
    }
    // leader: linger for company, then leave when the device is idle.  While one batch is running a second one
    // leaves only if it is big enough to be worth overlapping (planning and copies of one then hide behind the
    // kernels of the other: >= kCoOverlapMin queries, or twice the running batch); a smaller one waits for the
    // running batch's callers to come back and join it -- below a few hundred queries device time per query
    // falls so steeply with the batch size that one cohort of C callers beats two alternating cohorts of C / 2
    // even with the device idle between its batches (measured: 64 callers 18.2 k -> 23.9 k queries/s).  Never
    // more than two in flight.  Woken by a full queue or a finishing batch.
    constexpr int32_t kCoOverlapMin = 192;
    // (a caller that was alone last time and is alone now does not linger: a single stream of requests pays
    // no batching latency)
    const bool alone = ctx->co_last_batch <= 1 && ctx->co_pending.size() == 1 && ctx->co_inflight == 0;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(alone ? 0 : ctx->co_linger_us);
    for (;;) {
      const int32_t waiting = (int32_t)ctx->co_pending.size();
      if (waiting >= ctx->cfg.max_batch) break;
      const bool late = std::chrono::steady_clock::now() >= deadline;
      if (late && (ctx->co_inflight == 0 ||
                   (ctx->co_inflight == 1 && (waiting >= 2 * ctx->co_inflight_queries || waiting >= kCoOverlapMin)))) break;
      if (late) me.cv.wait(lk);
      else me.cv.wait_until(lk, deadline);
    }
    // take my request and every pending one over the same leaves (up to max_batch).  A big cohort that finds the
    // device idle is cut in two, so that from now on the host work of one half (planning, copies, waking its
    // callers) hides behind the kernels of the other.
    int32_t cap = ctx->cfg.max_batch;
    if (ctx->co_inflight == 0 && (int32_t)ctx->co_pending.size() >= 2 * kCoOverlapMin && (int32_t)ctx->co_pending.size() < cap)
      cap = ((int32_t)ctx->co_pending.size() + 1) / 2;
    std::vector<CoRequest*> rest;
    batch.push_back(&me);
    for (CoRequest* r : ctx->co_pending) {
      if (r == &me) continue;
      if ((int32_t)batch.size() < cap && same_leaves(&me, r)) batch.push_back(r);
      else rest.push_back(r);
    }
    ctx->co_pending.swap(rest);
    ctx->co_leader = nullptr;
    if (!ctx->co_pending.empty()) {  // hand the lead to the oldest request left behind
      CoRequest* next = ctx->co_pending.front();
      ctx->co_leader = next;
      std::lock_guard<std::mutex> theirs(next->m);
      next->lead = true;
      next->cv.notify_one();
    }
    ctx->co_inflight++;
    ctx->co_inflight_queries += (int)batch.size();
    ctx->co_last_batch = (int)batch.size();
  }
  // run the batch outside the lock
  std::vector<nrtgpu_bm25_query> qs(batch.size());
  std::vector<nrtgpu_topdocs> outs(batch.size());
  for (size_t i = 0; i < batch.size(); ++i) {
    qs[i] = *batch[i]->q;
    outs[i] = *batch[i]->out;
  }
  const int rc = nrtgpu_search_bm25_batch(ctx, segs, doc_bases, n_segs, qs.data(), (int32_t)qs.size(), outs.data());
  const std::string err = rc ? g_last_error : std::string();
  for (size_t i = 0; i < batch.size(); ++i) {
    CoRequest* r = batch[i];
    if (r == &me) {
      if (rc == 0) *out = outs[i];
      continue;
    }
    // (notified under the request's own lock: the woken caller cannot return -- and free its request -- before
    // we are done with it, and it contends with nobody but us)
    std::lock_guard<std::mutex> theirs(r->m);
    if (rc == 0) *r->out = outs[i];
    r->rc = rc;
    if (rc != 0) r->err = err;
    r->done = true;
    r->cv.notify_one();
  }
  {
    std::lock_guard<std::mutex> lk(ctx->co_mu);
    ctx->co_inflight--;
    ctx->co_inflight_queries -= (int)batch.size();
    if (ctx->co_leader) ctx->co_leader->cv.notify_one();  // a lingering leader may be waiting for the device
  }
  if (rc != 0) g_last_error = err;
  return rc;
}

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

extern "C" int nrtgpu_search_bm25_batch_device(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                               int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                               int32_t k_stride, void* d_keys, void* d_counts, void* d_hits) {
  return nrtgpu_search_bm25_batch_device_epoch(ctx, segs, doc_bases, n_segs, queries, n_queries, k_stride, d_keys, d_counts,
                                               d_hits, -1);
}

extern "C" int nrtgpu_search_bm25_batch_device_epoch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                                     int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                                     int32_t k_stride, void* d_keys, void* d_counts, void* d_hits,
                                                     int64_t epoch) {
  if (!ctx || !queries || !d_keys || !d_counts || !d_hits || (n_segs > 0 && !segs)) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_queries <= 0 || n_segs < 0) return fail(NRTGPU_ERR_INVALID_ARG, "n_queries must be > 0");
  if (n_queries > ctx->cfg.max_batch) return fail(NRTGPU_ERR_INVALID_ARG, "batch of %d exceeds max_batch %d", n_queries, ctx->cfg.max_batch);
  HIP_TRY(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  HostPlan hp;
  for (int si = 0; si < n_segs; ++si)
    if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
  SegReadLocks content(segs, n_segs);  // until this call's kernels have finished
  if (int rc = build_plan(ctx, segs, doc_bases, n_segs, queries, n_queries, hp)) return rc;
  if (k_stride < (int32_t)hp.k_stride && k_stride < NRTGPU_MAX_K) {
    for (int qi = 0; qi < n_queries; ++qi)
      if (queries[qi].k > k_stride) return fail(NRTGPU_ERR_INVALID_ARG, "k_stride %d smaller than numHits %d", k_stride, queries[qi].k);
  }
  const double plan_ms = now_ms() - t0;
  Slot* slot = nullptr;
  acquire_slot(ctx, &slot);
  struct Guard { nrtgpu_ctx* c; Slot* s; ~Guard() { release_slot(c, s); } } guard{ctx, slot};
  DeviceRun run;
  {
    std::unique_lock<std::mutex> gpu(ctx->gpu_mu, std::defer_lock);
    if (int rc = enqueue_search(ctx, slot, hp, n_queries, (uint32_t)k_stride, (uint64_t*)d_keys, (uint32_t*)d_counts,
                                (uint64_t*)d_hits, &run, gpu, epoch))
      return rc;
    HIP_TRY(hipStreamSynchronize(slot->stream));
  }
  account(ctx, slot, hp, n_queries, plan_ms);
  return NRTGPU_OK;
}

extern "C" int nrtgpu_merge_topk_device(nrtgpu_ctx* ctx, int32_t n_lists, int32_t n_queries, int32_t k_stride,
                                        const void* d_keys_in, const void* d_counts_in, const void* d_hits_in,
                                        const int32_t* ks, const int32_t* total_hits_thresholds, nrtgpu_topdocs* out) {
  if (!ctx || !d_keys_in || !d_counts_in || !d_hits_in || !ks || !total_hits_thresholds || !out)
    return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_lists <= 0 || n_queries <= 0 || k_stride <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad sizes");
  HIP_TRY(hipSetDevice(ctx->device));
  for (int qi = 0; qi < n_queries; ++qi)
    if (ks[qi] <= 0 || ks[qi] > NRTGPU_MAX_K || ks[qi] > k_stride) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: bad k %d", qi, ks[qi]);
  Slot* slot = nullptr;
  acquire_slot(ctx, &slot);
  struct Guard { nrtgpu_ctx* c; Slot* s; ~Guard() { release_slot(c, s); } } guard{ctx, slot};
  // plan blob: list_idx (n_queries * n_lists), q_base, q_nlists, q_k
  const size_t nq = (size_t)n_queries, nl = (size_t)n_lists;
  Carver pc;
  const size_t o_lidx = pc.take(nq * nl * 4), o_qbase = pc.take(nq * 4), o_qnl = pc.take(nq * 4), o_qk = pc.take(nq * 4);
  if (int rc = slot->h_plan.reserve(pc.off)) return rc;
  if (int rc = slot->d_plan.reserve(pc.off)) return rc;
  char* hb = (char*)slot->h_plan.p;
  uint32_t* lidx = (uint32_t*)(hb + o_lidx);
  uint32_t* qbase = (uint32_t*)(hb + o_qbase);
  uint32_t* qnl = (uint32_t*)(hb + o_qnl);
  uint32_t* qk = (uint32_t*)(hb + o_qk);
  for (size_t q = 0; q < nq; ++q) {
    qbase[q] = (uint32_t)(q * nl);
    qnl[q] = (uint32_t)nl;
    qk[q] = (uint32_t)ks[q];
    for (size_t l = 0; l < nl; ++l) lidx[q * nl + l] = (uint32_t)(l * nq + q);
  }
  Carver wc;
  const size_t o_okeys = wc.take(nq * (size_t)k_stride * 8), o_ocnt = wc.take(nq * 4), o_ohits = wc.take(nq * 8);
  if (int rc = slot->d_work.reserve(wc.off)) return rc;
  if (int rc = slot->h_out.reserve(wc.off)) return rc;
  char* db = (char*)slot->d_plan.p;
  char* wb = (char*)slot->d_work.p;
  char* ho = (char*)slot->h_out.p;
  hipStream_t st = slot->stream;
  HIP_TRY(hipMemcpyAsync(db, hb, pc.off, hipMemcpyHostToDevice, st));
  launch_merge_topk(st, (uint32_t)n_queries, (const uint64_t*)d_keys_in, (const uint32_t*)d_counts_in,
                    (const uint64_t*)d_hits_in, (const uint32_t*)(db + o_lidx), (const uint32_t*)(db + o_qbase),
                    (const uint32_t*)(db + o_qnl), (uint32_t)k_stride, (const uint32_t*)(db + o_qk),
                    (uint64_t*)(wb + o_okeys), (uint32_t*)(wb + o_ocnt), (uint64_t*)(wb + o_ohits), (uint32_t)k_stride);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(ho, wb, wc.off, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const uint64_t* keys = (const uint64_t*)(ho + o_okeys);
  const uint32_t* cnts = (const uint32_t*)(ho + o_ocnt);
  const uint64_t* hits = (const uint64_t*)(ho + o_ohits);
  for (int qi = 0; qi < n_queries; ++qi)
    unpack_topdocs(keys + (size_t)qi * k_stride, cnts[qi], hits[qi], ks[qi], total_hits_thresholds[qi], &out[qi]);
  return NRTGPU_OK;
}


// ------------------------------------------------------------------------------------------------
// ABI: exact vector search / vector rescore
// ------------------------------------------------------------------------------------------------
static const uint32_t kKnnCap = 1u << 18;   // candidate keys per query and round (2 MiB)
static const int kKnnMaxQ = 32;

// Shared by the two vector entry points.  knn_request = false: ExactVectorQuery (every doc with a vector
// matches, boost inside the score).  knn_request = true: the `knn` request path -- pre-filter mask, score
// threshold on the unboosted score (MinThresholdQuery's MinScoreWrapper), boost applied afterwards,
// totalHits = the docs returned.
static int knn_impl(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                    int32_t field_id, int32_t sim, const float* queries, int32_t n_queries, int32_t dim,
                    int32_t k, float boost, bool knn_request, int32_t filter_mask, float min_score, nrtgpu_topdocs* out) {
  if (!ctx || !queries || !out || (n_segs > 0 && !segs)) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_queries <= 0 || k <= 0 || dim <= 0 || sim < 0 || sim > 3) return fail(NRTGPU_ERR_INVALID_ARG, "bad knn arguments");
  if (k > NRTGPU_MAX_K) return fail(NRTGPU_ERR_UNSUPPORTED, "k %d > %d", k, NRTGPU_MAX_K);
  if (dim % 16 != 0 || dim > 1280) return fail(NRTGPU_ERR_UNSUPPORTED, "vector dimension %d (device path needs a multiple of 16, <= 1280)", dim);
  HIP_TRY(hipSetDevice(ctx->device));
  for (int si = 0; si < n_segs; ++si)
    if (!segs[si]) return fail(NRTGPU_ERR_STATE, "segment %d missing or not sealed", si);
  SegReadLocks content(segs, n_segs);  // liveDocs / masks stay as they are until the kernels have finished
  for (int si = 0; si < n_segs; ++si) {
    if (!segs[si] || !segs[si]->sealed) return fail(NRTGPU_ERR_STATE, "segment %d missing or not sealed", si);
    auto fit = segs[si]->fields.find(field_id);
    if (fit != segs[si]->fields.end() && fit->second.d_vectors && fit->second.dim != dim)
      return fail(NRTGPU_ERR_INVALID_ARG, "segment %d: field %d has dimension %d, query has %d", si, field_id, fit->second.dim, dim);
  }
  const uint32_t k_stride = round_up((uint32_t)k, 16);
  Slot* slot = nullptr;
  acquire_slot(ctx, &slot);
  struct Guard { nrtgpu_ctx* c; Slot* s; ~Guard() { release_slot(c, s); } } guard{ctx, slot};
  std::lock_guard<std::mutex> gpu(ctx->gpu_mu);
  hipStream_t st = slot->stream;
  Carver wc;
  const size_t o_q = wc.take((size_t)kKnnMaxQ * dim * 4), o_qn = wc.take(kKnnMaxQ * 4), o_th = wc.take(kKnnMaxQ * 8);
  const size_t o_tk = wc.take((size_t)kKnnMaxQ * k_stride * 8), o_tc = wc.take(kKnnMaxQ * 4);
  const size_t o_cc = wc.take(kKnnMaxQ * 4), o_ov = wc.take(64), o_cd = wc.take((size_t)kKnnMaxQ * kKnnCap * 8);
  if (int rc = slot->d_work.reserve(wc.off)) return rc;
  if (int rc = slot->h_out.reserve((size_t)kKnnMaxQ * k_stride * 8 + kKnnMaxQ * 4)) return rc;
  char* wb = (char*)slot->d_work.p;
  std::vector<float> qn(kKnnMaxQ);
  for (int q0 = 0; q0 < n_queries; q0 += kKnnMaxQ) {
    const int nq = std::min(kKnnMaxQ, n_queries - q0);
    for (int q = 0; q < nq; ++q) {
      float s2 = 0.f;  // squareMagnitude of the query, fp32
      const float* qv = queries + (size_t)(q0 + q) * dim;
      for (int d = 0; d < dim; ++d) {
        volatile float p2 = qv[d] * qv[d];
        s2 = s2 + p2;
      }
      qn[(size_t)q] = s2;
    }
    HIP_TRY(hipMemcpyAsync(wb + o_q, queries + (size_t)q0 * dim, (size_t)nq * dim * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(wb + o_qn, qn.data(), (size_t)nq * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(wb + o_th, 0, wc.off - o_th > 0 ? (o_cd - o_th) : 0, st));  // theta, topk, counters
    if (knn_request && min_score > 0.0f) {  // start theta just below the lowest key of that score: score >= min_score passes
      std::vector<uint64_t> th0((size_t)nq, pack_key(min_score, 0xFFFFFFFFu) - 1ull);
      HIP_TRY(hipMemcpyAsync(wb + o_th, th0.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));  // th0 is a stack vector
    }
    int64_t total_vec = 0;
    for (int si = 0; si < n_segs; ++si) {
      const nrtgpu_seg* seg = segs[si];
      auto fit = seg->fields.find(field_id);
      if (fit == seg->fields.end() || !fit->second.d_vectors) continue;
      const FieldData& f = fit->second;
      total_vec += f.n_vec;
      const uint64_t* accept = seg->d_live;  // (vectors are not re-coded for liveDocs: always the mask)
      if (knn_request && filter_mask != 0)
        if (int rc = accept_set_of(seg, filter_mask, 0, &accept)) return rc;
      // rounds never exceed the candidate capacity, so a list cannot overflow; theta tightens between rounds
      int64_t r = 0, round = 1 << 16;
      while (r < f.n_vec) {
        const int64_t re = std::min<int64_t>(f.n_vec, r + std::min<int64_t>(round, kKnnCap));
        const uint32_t blocks = (uint32_t)std::min<int64_t>((re - r + 255) / 256, (int64_t)std::max(ctx->n_cus, 1));  // 256 rows per workgroup step
        const int e = launch_knn_score(st, blocks, f.d_vectors, f.d_vnorm2, f.d_ord_to_doc, accept, dim, r, re,
                                       doc_bases ? doc_bases[si] : 0, (const float*)(wb + o_q), (const float*)(wb + o_qn), nq,
                                       sim, knn_request ? 1.0f : boost, (const unsigned long long*)(wb + o_th), (uint64_t*)(wb + o_cd),
                                       (uint32_t*)(wb + o_cc), kKnnCap);
        if (e) return fail(NRTGPU_ERR_HIP, "knn_score launch: %s", hipGetErrorString((hipError_t)e));
        launch_knn_select(st, (uint32_t)nq, (uint64_t*)(wb + o_tk), (uint32_t*)(wb + o_tc), k_stride, (uint32_t)k,
                          (const uint64_t*)(wb + o_cd), (uint32_t*)(wb + o_cc), kKnnCap,
                          (unsigned long long*)(wb + o_th), (uint32_t*)(wb + o_ov));
        r = re;
        round = std::min<int64_t>(round * 4, kKnnCap);  // (unbounded growth overflowed after 24 rounds: > 6M rows hung)
      }
    }
    HIP_TRY(hipGetLastError());
    char* ho = (char*)slot->h_out.p;
    HIP_TRY(hipMemcpyAsync(ho, wb + o_tk, (size_t)nq * k_stride * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(ho + (size_t)kKnnMaxQ * k_stride * 8, wb + o_tc, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint64_t* keys = (const uint64_t*)ho;
    const uint32_t* cnts = (const uint32_t*)(ho + (size_t)kKnnMaxQ * k_stride * 8);
    for (int q = 0; q < nq; ++q) {
      nrtgpu_topdocs* o = &out[q0 + q];
      const int32_t cap = o->capacity > 0 ? o->capacity : k;
      const int32_t m = std::min<int32_t>((int32_t)cnts[q], cap);
      for (int32_t i = 0; i < m; ++i) {
        if (o->docs) o->docs[i] = (int32_t)key_doc(keys[(size_t)q * k_stride + i]);
        if (o->scores) o->scores[i] = key_score(keys[(size_t)q * k_stride + i]);
      }
      o->n_hits = m;
      o->total_hits = total_vec;   // every doc with a vector matches an exact vector query (deletes not subtracted)
      o->total_hits_is_lower_bound = 0;
      if (knn_request) {
        o->total_hits = m;  // the rewritten knn query matches exactly the docs it returns
        if (boost != 1.0f && o->scores) {
          for (int32_t i = 0; i < m; ++i) o->scores[i] = o->scores[i] * boost;
          // distinct scores can round to one product: restore (score desc, doc asc) among equals
          if (o->docs)
            for (int32_t i = 1; i < m; ++i)
              for (int32_t j = i; j > 0 && o->scores[j - 1] == o->scores[j] && o->docs[j - 1] > o->docs[j]; --j) std::swap(o->docs[j - 1], o->docs[j]);
        }
      }
    }
  }
  return NRTGPU_OK;
}

extern "C" int nrtgpu_knn_exact(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                int32_t field_id, int32_t sim, const float* queries, int32_t n_queries, int32_t dim,
                                int32_t k, float boost, nrtgpu_topdocs* out) {
  return knn_impl(ctx, segs, doc_bases, n_segs, field_id, sim, queries, n_queries, dim, k, boost, false, 0, 0.0f, out);
}

extern "C" int nrtgpu_knn_search(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                 int32_t field_id, int32_t sim, const float* queries, int32_t n_queries, int32_t dim,
                                 int32_t k, float boost, int32_t filter_mask, float min_score, nrtgpu_topdocs* out) {
  if (filter_mask < 0 || !(min_score >= 0.0f) || !(boost > 0.0f))
    return fail(NRTGPU_ERR_INVALID_ARG, "knn search: filter_mask >= 0, min_score >= 0 and boost > 0 expected");
  return knn_impl(ctx, segs, doc_bases, n_segs, field_id, sim, queries, n_queries, dim, k, boost, true, filter_mask, min_score, out);
}

extern "C" int nrtgpu_rescore_vectors(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                      int32_t field_id, int32_t sim, const float* query, int32_t dim, float boost,
                                      const int32_t* docs, const float* first_scores, int32_t n, double query_weight,
                                      double rescore_weight, int32_t window, nrtgpu_topdocs* out) {
  if (!ctx || !query || !out || (n > 0 && (!docs || !first_scores)) || (n_segs > 0 && (!segs || !doc_bases)))
    return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n < 0 || dim <= 0 || sim < 0 || sim > 3 || window <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad rescore arguments");
  HIP_TRY(hipSetDevice(ctx->device));
  float qn = 0.f;
  for (int d = 0; d < dim; ++d) {
    volatile float p2 = query[d] * query[d];
    qn = qn + p2;
  }
  // hits -> (segment, vector row); per segment one gather kernel
  std::vector<int> seg_of((size_t)n, -1);
  std::vector<int64_t> row_of((size_t)n, -1);
  for (int i = 0; i < n; ++i) {
    for (int si = 0; si < n_segs; ++si) {
      const int32_t local = docs[i] - doc_bases[si];
      if (local < 0 || local >= segs[si]->max_doc) continue;
      seg_of[(size_t)i] = si;
      auto fit = segs[si]->fields.find(field_id);
      if (fit == segs[si]->fields.end() || !fit->second.d_vectors) break;
      const FieldData& f = fit->second;
      if (f.dim != dim) return fail(NRTGPU_ERR_INVALID_ARG, "vector dimension mismatch");
      if (f.h_ord_to_doc.empty()) {
        if (local < f.n_vec) row_of[(size_t)i] = local;
      } else {
        auto it = std::lower_bound(f.h_ord_to_doc.begin(), f.h_ord_to_doc.end(), local);
        if (it != f.h_ord_to_doc.end() && *it == local) row_of[(size_t)i] = it - f.h_ord_to_doc.begin();
      }
      break;
    }
    if (seg_of[(size_t)i] < 0) return fail(NRTGPU_ERR_INVALID_ARG, "hit %d (doc %d) is outside every segment", i, docs[i]);
  }
  Slot* slot = nullptr;
  acquire_slot(ctx, &slot);
  struct Guard { nrtgpu_ctx* c; Slot* s; ~Guard() { release_slot(c, s); } } guard{ctx, slot};
  std::lock_guard<std::mutex> gpu(ctx->gpu_mu);
  hipStream_t st = slot->stream;
  Carver wc;
  const size_t o_q = wc.take((size_t)dim * 4), o_rows = wc.take((size_t)n * 8 + 8), o_first = wc.take((size_t)n * 4 + 4),
               o_out = wc.take((size_t)n * 4 + 4);
  if (int rc = slot->d_work.reserve(wc.off)) return rc;
  char* wb = (char*)slot->d_work.p;
  std::vector<float> combined((size_t)n);
  HIP_TRY(hipMemcpyAsync(wb + o_q, query, (size_t)dim * 4, hipMemcpyHostToDevice, st));
  for (int si = 0; si < n_segs; ++si) {
    std::vector<int> idx;
    for (int i = 0; i < n; ++i)
      if (seg_of[(size_t)i] == si) idx.push_back(i);
    if (idx.empty()) continue;
    auto fit = segs[si]->fields.find(field_id);
    const FieldData* f = (fit != segs[si]->fields.end() && fit->second.d_vectors) ? &fit->second : nullptr;
    std::vector<int64_t> rows(idx.size());
    std::vector<float> first(idx.size()), res(idx.size());
    for (size_t j = 0; j < idx.size(); ++j) {
      rows[j] = row_of[(size_t)idx[j]];
      first[j] = first_scores[idx[j]];
    }
    if (!f) {  // no vectors in this leaf: second pass matches nothing
      for (size_t j = 0; j < idx.size(); ++j) combined[(size_t)idx[j]] = (float)(query_weight * (double)first[j]);
      continue;
    }
    HIP_TRY(hipMemcpyAsync(wb + o_rows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(wb + o_first, first.data(), first.size() * 4, hipMemcpyHostToDevice, st));
    launch_rescore_vectors(st, f->d_vectors, f->d_vnorm2, dim, (const float*)(wb + o_q), qn, sim, boost,
                           (const int64_t*)(wb + o_rows), (const float*)(wb + o_first), (int32_t)idx.size(), query_weight,
                           rescore_weight, (float*)(wb + o_out));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(res.data(), wb + o_out, res.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t j = 0; j < idx.size(); ++j) combined[(size_t)idx[j]] = res[j];
  }
  // QueryRescorer: sort by (combined score desc, doc asc), keep the window
  std::vector<int> order((size_t)n);
  for (int i = 0; i < n; ++i) order[(size_t)i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    if (combined[(size_t)a] != combined[(size_t)b]) return combined[(size_t)a] > combined[(size_t)b];
    return docs[a] < docs[b];
  });
  const int32_t cap = out->capacity > 0 ? out->capacity : window;
  const int32_t m = std::min<int32_t>(std::min<int32_t>(n, window), cap);
  for (int32_t i = 0; i < m; ++i) {
    if (out->docs) out->docs[i] = docs[order[(size_t)i]];
    if (out->scores) out->scores[i] = combined[(size_t)order[(size_t)i]];
  }
  out->n_hits = m;
  out->total_hits = n;
  out->total_hits_is_lower_bound = 0;
  return NRTGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// ABI: host-side restatements
// ------------------------------------------------------------------------------------------------
extern "C" int32_t nrtgpu_int_to_byte4(int32_t length) { return hostmath::int_to_byte4(length); }
extern "C" int32_t nrtgpu_byte4_to_int(int32_t b) { return hostmath::byte4_to_int(b); }
extern "C" float nrtgpu_bm25_idf(int64_t doc_count, int64_t doc_freq) { return hostmath::bm25_idf(doc_count, doc_freq); }
extern "C" float nrtgpu_bm25_avgdl(int64_t sttf, int64_t doc_count) { return hostmath::bm25_avgdl(sttf, doc_count); }
extern "C" void nrtgpu_bm25_norm_cache(float avgdl, float k1, float b, float* out256) { hostmath::bm25_norm_cache(avgdl, k1, b, out256); }

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
