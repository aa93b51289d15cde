// runtime_internal.h -- shared declarations of the host runtime behind include/nrtgpu.h (not part of the ABI).
//
// The runtime is split by concern: runtime.cpp (errors, context, workspaces, statistics, exchange table, host
// helpers), segment.cpp (the segment store: upload, seal, liveDocs, masks), planner.cpp (queries -> launch
// plan), search.cpp (BM25 entry points, hybrid tail, request coalescing, merge), vectors.cpp (exact kNN,
// vector rescoring).  Types the ABI names opaquely (nrtgpu_ctx, nrtgpu_seg) live in the global namespace;
// everything else in nrtgpu::rt.
#pragma once
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
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/nrtgpu.h"
#ifdef NRTGPU_DEV
#include "../../include/nrtgpu_dev.h"
#endif
#include "host_math.h"
#include "plan.h"

namespace nrtgpu {
void launch_bm25_scan(hipStream_t stream, bool fixed_point, bool pipelined, bool packed, int ablation, uint32_t n_items, const DItem* items,
                      const DPart* parts, const DTerm* terms, const DQuery* queries, const float* caches, unsigned long long* theta_g,
                      unsigned long long* quant_g, const DExchange* xch, uint32_t* slice_sum, uint64_t* item_keys, uint32_t* item_counts,
                      uint64_t* item_hits, uint32_t k_stride, uint64_t* item_prof);
void launch_bm25_maxscore(hipStream_t stream, bool profile, bool packed, int shapes, const MsArgs& args, const MsArgs* args_d);
void launch_term_frontier(hipStream_t stream, const uint32_t* fnorm, const uint64_t* t_start, const uint32_t* t_count,
                          const uint64_t* t_look, const uint32_t* t_meta, const void* look_base, uint32_t n_terms, DTermAux* out);
void launch_term_bits(hipStream_t stream, const uint32_t* docids, const uint64_t* t_start, const uint32_t* t_count, const uint64_t* t_look,
                      const uint32_t* which, uint32_t n_which, uint32_t max_count, void* look_base);
void launch_term_cells(hipStream_t stream, const uint32_t* docids, const uint64_t* t_start, const uint32_t* t_count, const uint64_t* t_look,
                       const uint32_t* t_meta, const uint32_t* which, uint32_t n_which, uint32_t max_cells, uint32_t max_doc, void* look_base);
void launch_expand_terms(hipStream_t stream, const DQExpand* qx, const DQTerm* qterms, const uint32_t* out_begin, uint32_t n_queries,
                         uint32_t n_leaves, DTerm* out);
void launch_slice_relation(hipStream_t stream, const uint32_t* slice_sum, const DQuery* queries, uint32_t n_slices, uint64_t* out_hits,
                           uint32_t n);
void launch_patch_hits(hipStream_t stream, const uint64_t* lower, uint64_t* hits, uint32_t n);
void launch_merge_topk(hipStream_t stream, uint32_t n_queries, const uint64_t* in_keys, const uint32_t* in_counts,
                       const uint64_t* in_hits, const uint32_t* list_idx, const uint32_t* q_base,
                       const uint32_t* q_nlists, uint32_t k_stride_in, const uint32_t* q_k, uint64_t* out_keys,
                       uint32_t* out_counts, uint64_t* out_hits, uint32_t k_stride_out, const uint32_t* help_query = nullptr,
                       uint32_t n_help = 0, uint32_t help_slot_base = 0, const unsigned long long* spec_g = nullptr);
void launch_fold_norms(hipStream_t stream, const uint32_t* docids, const uint32_t* freqs, const uint8_t* norms,
                       uint32_t* fnorm, uint64_t n, uint32_t* overflow);
void launch_pack_count(hipStream_t stream, const uint32_t* fnorm, uint64_t n, uint32_t n_blocks, uint32_t* counts);
void launch_pack_write(hipStream_t stream, const uint32_t* docids, const uint32_t* fnorm, uint64_t n, uint32_t n_blocks,
                       const uint32_t* dir, uint32_t* exceptions, uint32_t* packed);
void launch_apply_live(hipStream_t stream, const uint32_t* docids, uint32_t* fnorm, uint64_t n, const uint64_t* live);
void launch_knn_row_norms(hipStream_t st, const float* vecs, int32_t dim, int64_t n, float* norm2);
int launch_knn_score(hipStream_t st, uint32_t blocks, const float* vecs, const float* vnorm2, const int32_t* ord_to_doc,
                     const uint64_t* live_bits, int32_t dim, int64_t row_begin, int64_t row_end, int32_t doc_base,
                     const float* qpanel, const float* qnorm2, int32_t n_q, int32_t sim, float boost,
                     const unsigned long long* theta, uint64_t* cand, uint32_t* cand_cnt, uint32_t cap, int32_t append_only = 0);
void launch_knn_select(hipStream_t st, uint32_t n_q, uint64_t* topk, uint32_t* topk_cnt, uint32_t k_stride, uint32_t k,
                       const uint64_t* cand, uint32_t* cand_cnt, uint32_t cap, unsigned long long* theta,
                       uint32_t* overflow);
void launch_knn_refine_select(hipStream_t st, uint32_t n_q, uint64_t* topk, uint32_t* topk_cnt, uint32_t k_stride, uint32_t k,
                              const uint64_t* cand, uint32_t* cand_cnt, uint32_t cap, unsigned long long* theta, uint32_t* overflow,
                              const DVecSeg* segs, int32_t n_segs, int32_t dim, int32_t sim, const float* qpanel, const float* qnorm2,
                              float boost, const float* ebound, float erel, float min_score, uint32_t k_int, int32_t certify,
                              uint32_t* cert);
void launch_knn_norm_max(hipStream_t st, const float* norm2, int64_t n, uint32_t* out_bits);
void launch_knn_norm_min(hipStream_t st, const float* norm2, int64_t n, uint32_t* out_bits);
void launch_knn_absmax(hipStream_t st, const float* vecs, int64_t n_elems, uint32_t* out_bits);
size_t knn_sketch_bytes(int32_t dim, int64_t n);
size_t knn_sketch_lds_bytes(int32_t dim, int32_t n_q);
bool knn_sketch_fits(int32_t dim, int32_t n_q);   // the sketch kernel's LDS: panel + queue + its static tables within 160 KB
void launch_knn_panel_fp16(hipStream_t st, const float* qpanel, const float* qscale, int32_t dim, int32_t n_q, void* panel16);
void launch_knn_sketch_build(hipStream_t st, const float* vecs, int32_t dim, int64_t n, float scale, void* sketch);
int launch_knn_sketch(hipStream_t st, uint32_t blocks, const DKnnLeaf* leaves, int32_t n_leaves, int32_t dim, int64_t tile_begin,
                      int64_t tile_end, const void* panel16, const float* qnorm2, const float* qscale, int32_t n_q, int32_t sim,
                      float boost, const unsigned long long* theta, uint64_t* cand, uint32_t* cand_cnt, uint32_t cap, int32_t append_only = 0);
void launch_rescore_vectors(hipStream_t st, const float* vecs, const float* vnorm2, int32_t dim, const float* query,
                            float qnorm2, int32_t sim, float boost, const int64_t* vec_row, const float* first_scores,
                            int32_t n, double qw, double rw, float* out_scores);
void launch_hybrid_rescore(hipStream_t st, uint32_t n_queries, const uint64_t* first_keys, const uint32_t* first_counts,
                           uint32_t k_stride, const DVecSeg* segs, int32_t n_segs, int32_t dim, const float* qvecs,
                           const float* qnorm2, int32_t sim, float boost, double qw, double rw, uint32_t window,
                           uint64_t* out_keys, uint32_t* out_counts, uint32_t w_stride, int32_t drop_foreign = 0);
void launch_hybrid_hits(hipStream_t st, const uint64_t* first_hits, const uint32_t* first_counts, const uint32_t* q_k, int32_t carries,
                        uint64_t* out_hits, uint32_t n);
}  // namespace nrtgpu

namespace nrtgpu {
namespace rt {

// ---- errors (runtime.cpp) ----------------------------------------------------------------------
extern thread_local std::string g_last_error;
extern thread_local int64_t g_deadline_ns;           // nrtgpu_set_thread_deadline_ns: 0 = none
extern thread_local nrtgpu_diagnostics g_diag;       // nrtgpu_last_diagnostics
extern thread_local std::vector<int32_t> g_thread_slices;   // nrtgpu_set_thread_slices: the caller's slice of every leaf of its next calls
int64_t monotonic_ns();
inline bool deadline_passed(int64_t deadline_ns) { return deadline_ns != 0 && monotonic_ns() >= deadline_ns; }
#define NRT_CHECK_DEADLINE(what)                                                                             \
  do {                                                                                                       \
    if (deadline_passed(g_deadline_ns)) return fail(NRTGPU_ERR_TIMEOUT, "deadline passed %s", what);        \
  } while (0)
int fail(int code, const char* fmt, ...);
double now_ms();
// Experiment knobs: the DEVELOPMENT build (-DNRTGPU_DEV, libnrtgpu_dev.so) reads them from the environment for A/B runs; the
// product library is configured through nrtgpu_config and the nrtgpu_set_* calls only, and always takes the default.
inline long dev_env_int(const char* name, long dflt) {
#ifdef NRTGPU_DEV
  if (const char* e = getenv(name)) return atol(e);
#else
  (void)name;
#endif
  return dflt;
}
inline const char* dev_env_str(const char* name, const char* dflt) {
#ifdef NRTGPU_DEV
  if (const char* e = getenv(name)) return e;
#else
  (void)name;
#endif
  return dflt;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(_e == hipErrorOutOfMemory ? NRTGPU_ERR_OOM : NRTGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                  hipGetErrorString(_e), __FILE__, __LINE__);                                      \
  } while (0)

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
  uint32_t aux_idx;    // the term's DTermAux record inside the group's d_aux (written at seal)
};

struct TermGroup {
  uint32_t* d_docids = nullptr;
  uint32_t* d_freqs = nullptr;   // raw freq column, only between add_terms and seal (nullptr => freq == 1)
  uint32_t* d_fnorm = nullptr;   // score-code column (same allocation as d_docids), filled at seal
  bool folded = false;
  bool packed = false;           // NRTGPU_FLAG_PACKED_POSTINGS, after the seal: d_docids = packed column, d_fnorm = d_dict
  uint32_t* d_dict = nullptr;    // packed postings: the group's exception list (header, directory, escape words)
  uint32_t* d_cells = nullptr;   // concatenated per-term cell tables
  DTermAux* d_aux = nullptr;     // MaxScore route: one record per term of the group (impact frontier, lookup structure), seal
  char* d_look = nullptr;        // the lookup structures of the group's terms (plan.h: kLookMap / kLookCells), one buffer
  uint64_t look_bytes = 0;
  uint32_t n_look[3] = {0, 0, 0};   // terms per lookup kind (plan.h: kLook*; diagnostics)
  uint32_t n_terms = 0;
  std::vector<uint64_t> h_start;  // per term of the group (add order): first posting, postings -- kept until the seal
  std::vector<uint32_t> h_count;
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
  int32_t dim = 0, n_vec = 0;   // dim: the rows' dimension AS RESIDENT -- the field's own (dim_user) rounded up to a multiple of 16,
                                // the extra elements zero (they add nothing to any of the four similarities' sums)
  int32_t dim_user = 0;
  float vnorm2_max = 0.f;                // largest |v|^2 of the rows
  void* d_sketch = nullptr;              // the rows in fp16, matrix-core operand order (knn.hip); nullptr: none (yet)
  int sketch_state = -1;                 // 0: to be built by the first exact search over the field (segment.cpp: ensure_vector_sketch), 1: built, -1: never
  float sketch_scale = 1.f;              // the power of two the rows were multiplied by before rounding
  float absmax = 0.f, vnorm2_min = 0.f;  // largest |element|, smallest non-zero |v|^2 (the sketch's error bound)
  // rows whose doc is live under the segment's current liveDocs (what an exact vector query matches), counted on first
  // use per liveDocs version
  mutable std::atomic<int64_t> live_vec{-1};
  mutable std::atomic<uint64_t> live_vec_version{0};
  FieldData() = default;
  FieldData(const FieldData& o)
      : d_norms(o.d_norms), max_norm(o.max_norm), dict(o.dict), flat(o.flat), groups(o.groups), d_vectors(o.d_vectors),
        d_vnorm2(o.d_vnorm2), d_ord_to_doc(o.d_ord_to_doc), h_ord_to_doc(o.h_ord_to_doc), dim(o.dim), n_vec(o.n_vec),
        vnorm2_max(o.vnorm2_max), d_sketch(o.d_sketch), sketch_state(o.sketch_state), sketch_scale(o.sketch_scale), absmax(o.absmax),
        vnorm2_min(o.vnorm2_min) {}
};

}  // namespace rt
}  // namespace nrtgpu

using namespace nrtgpu;      // (internal header: every translation unit of the runtime wants both)
using namespace nrtgpu::rt;

struct nrtgpu_ctx;
// The immutable part of a segment -- posting columns, norms, vectors, the seal-time structures -- shared by the handles
// of its reader versions (nrtgpu_segment_fork): freed with the last of them.
struct SegCore {
  int device = 0;
  std::map<int32_t, FieldData> fields;
  // the liveDocs folded into the posting columns (apply_live_kernel), if any: once a core is shared nobody folds again,
  // and every handle's liveDocs must be a subset of these (Lucene's deletes only accumulate)
  std::vector<uint64_t> folded_live;
  bool folded = false;
  std::mutex sketch_mu;   // the lazy build of a vector field's sketch
  std::atomic<int64_t> shared_extra_bytes{0};   // device memory added to the core after the seal (the sketches): counted for every handle
  ~SegCore();
};
struct nrtgpu_seg {
  nrtgpu_ctx* ctx = nullptr;
  uint64_t uid = 0;              // unique over the process (never reused, unlike the handle's address): keys the planner's caches;
                                 // the forks of a segment share it (and with it the resident term tables)
  int32_t max_doc = 0;
  uint32_t n_tiles = 0;
  bool sealed = false;
  std::shared_ptr<SegCore> core;
  std::map<int32_t, FieldData>& fields;   // == core->fields
  nrtgpu_seg() : core(std::make_shared<SegCore>()), fields(core->fields) {}
  explicit nrtgpu_seg(std::shared_ptr<SegCore> c) : core(std::move(c)), fields(core->fields) {}
  uint64_t* d_live = nullptr;
  int32_t n_deleted = 0;         // docs cleared in liveDocs (host pop-count at set_live_docs)
  uint64_t live_version = 1;     // a fresh process-wide number at every set_live_docs / fork (keys per-liveDocs caches in the shared core)
  int64_t device_bytes = 0;
  // FILTER / MUST_NOT clauses as doc-set masks: host copies of the registered masks and of liveDocs,
  // and the combined accept sets (live & filter & ~must_not) the scan reads, built on first use
  std::vector<uint64_t> h_live;                      // empty = all live
  bool live_folded = false;  // the posting columns carry the current liveDocs (apply_live_kernel): the scan needs no mask for them
  // The content lock.  Searches hold it SHARED from planning until their kernels have finished; set_live_docs / set_mask
  // take it exclusively, so a reader-version change never rewrites columns or masks under a running scan.  Not a
  // std::shared_mutex: a search begun on one thread may be waited for on another (nrtgpu_search_bm25_batch_device_begin /
  // nrtgpu_pending_wait -- unlocking a std::shared_mutex from a thread that does not hold it is undefined), and
  // nrtgpu_segment_release under running searches must not free what they read (ShardState.java:506-527 closes readers
  // while SEARCH-pool threads run): the handle is then freed by the LAST search that lets go of it.
  mutable std::mutex content_m;
  mutable std::condition_variable content_cv;
  mutable int content_readers = 0;          // searches in flight over this handle
  mutable int content_writers_waiting = 0;  // pending exclusive owners: new SYNCHRONOUS searches let them go first (no writer
                                            // starvation under overlapping request threads).  A PIPELINED search
                                            // (nrtgpu_search_bm25_batch_device_begin) passes a waiting writer while other searches
                                            // are in flight: its thread may be the one that must still wait for one of them --
                                            // begin i + 1 before wait i -- and parking it behind the writer that waits for search
                                            // i is a deadlock (a bounded number of passes only postpones it:
                                            // tests/test_exchange_gpu.py found that on the CPU).  The price: a pipeline that never
                                            // drains can delay set_mask / set_live_docs on the handles it searches -- reader
                                            // versions are forks (no writer), masks are registered before a handle is searched
  mutable bool content_writing = false;
  mutable bool content_released = false;    // nrtgpu_segment_release came while searches were in flight: the last one frees
  void content_lock_shared(bool pipelined) const;
  void content_unlock_shared() const;       // may free the handle (content_released)
  std::map<int32_t, std::vector<uint64_t>> masks;
  mutable std::mutex accept_mu;
  struct AcceptSet { uint64_t* bits; uint64_t used; };        // used: the cache's clock at the last lookup (least recently used goes first)
  mutable std::map<std::vector<int32_t>, AcceptSet> accept;   // key: the filter mask ids ascending, 0, the must_not mask ids ascending
  mutable uint64_t accept_clock = 0;
  mutable std::vector<uint64_t*> accept_retired;              // evicted, maybe still read by a search in flight: freed by the last reader out
  bool accept_retired_empty() const {
    std::lock_guard<std::mutex> lk(accept_mu);
    return accept_retired.empty();
  }
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
  // pipelined: the caller returns to its own caller holding the locks (begin / wait); see nrtgpu_seg::content_writers_waiting.
  // A thread must not start a synchronous search over handles it holds un-waited pipelined searches on.
  SegReadLocks(const nrtgpu_seg* const* segs, int32_t n, bool pipelined = false) {
    for (int32_t i = 0; i < n; ++i)
      if (segs && segs[i]) held.push_back(segs[i]);
    std::sort(held.begin(), held.end());
    held.erase(std::unique(held.begin(), held.end()), held.end());
    for (const nrtgpu_seg* s : held) s->content_lock_shared(pipelined);
  }
  ~SegReadLocks() {   // (any thread: nrtgpu_pending_wait may run on another one than the begin)
    for (const nrtgpu_seg* s : held) s->content_unlock_shared();
  }
  SegReadLocks(const SegReadLocks&) = delete;
  SegReadLocks& operator=(const SegReadLocks&) = delete;
};

namespace nrtgpu {
namespace rt {
// ------------------------------------------------------------------------------------------------
// per-call workspace
// ------------------------------------------------------------------------------------------------
struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  hipEvent_t ev_turn = nullptr;   // recorded behind this call's whole-GPU kernels (nrtgpu_ctx::last_turn)
  hipEvent_t ev_wait = nullptr;   // hipEventBlockingSync: what a caller sleeps on under NRTGPU_FLAG_BLOCKING_WAIT
  PinBuf h_plan;     // host staging of the plan blob
  DevBuf d_plan;     // device copy
  DevBuf d_work;     // theta + item outputs + merge outputs
  DevBuf d_aux;      // hybrid tail: leaf table, query vectors, rescored windows
  PinBuf h_aux;
  PinBuf h_out;      // merged results on the host
  std::vector<hipEvent_t> round_ev;   // collect_timing: start / stop of every knn_score launch of a panel
  bool busy = false;
};

// The caller's wait for its stream: a spin inside hipStreamSynchronize by default (lowest latency; it keeps a CPU busy), a
// sleep on a blocking event under NRTGPU_FLAG_BLOCKING_WAIT (several ranks sharing the host's CPUs: one process per GPU
// with a few calls in flight each would otherwise spin on more CPUs than the box has).
// hipGetLastError() is the CALLING THREAD's: a runtime call of somebody else on this thread (the caller's own GPU code, another
// library) that failed and was never asked about leaves its error there, and the first of OUR functions that checks its launches
// with hipGetLastError() would report it as its own.  Entry points that launch kernels forget it first.
inline void forget_foreign_hip_error() { (void)hipGetLastError(); }

inline hipError_t wait_for_stream(bool blocking, hipStream_t st, hipEvent_t blocking_event) {
  if (!blocking) return hipStreamSynchronize(st);
  const hipError_t e = hipEventRecord(blocking_event, st);
  return e != hipSuccess ? e : hipEventSynchronize(blocking_event);
}

}  // namespace rt
}  // namespace nrtgpu

namespace nrtgpu {
namespace rt {
struct LeafSetCache;

// The context's helper threads (cfg.host_threads - 1 of them, started at nrtgpu_create): run(n, fn) executes
// fn(0) .. fn(n - 1) on the caller and whichever helpers are free, and returns when all are done.  Several
// callers may run jobs at the same time (two batches in flight).  Replaces a std::thread per planner chunk per
// batch: thread creation alone cost more than the work once term lookups came from the cache.
class WorkPool {
 public:
  explicit WorkPool(int helpers);
  ~WorkPool();
  void run(int n, const std::function<void(int)>& fn);
  int helpers() const { return (int)threads_.size(); }

 private:
  struct Job {
    const std::function<void(int)>* fn;
    int n;
    std::atomic<int> next{0}, done{0};
  };
  void loop();
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<std::shared_ptr<Job>> jobs_;
  bool stop_ = false;
};
}  // namespace rt
}  // namespace nrtgpu

struct nrtgpu_dist;   // dist.cpp: RCCL communicator of this context (nrtgpu_dist_init)

// The context's LAUNCHER: one thread that enqueues the searches begun with nrtgpu_search_bm25_*_begin, in the order they were begun
// (search.cpp: device_begin_impl).  A submitting thread then only PLANS -- the plan of batch i + 1 is being built while batch i's
// plan is packed, copied and its seven launches are issued (0.1 ms per 1024-query batch: a quarter of a step of one rank of an
// 8-GPU job, whose scorer runs for 0.3 ms).  Created by the first _begin, joined by nrtgpu_destroy.
struct Launcher {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false;
  explicit Launcher(int device);
  ~Launcher();
  void push(std::function<void()> f);
};

struct nrtgpu_ctx {
  nrtgpu_config cfg{};
  nrtgpu_dist* dist = nullptr;
  int device = 0;
  int n_cus = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::unique_ptr<Slot>> slots;
  hipEvent_t last_turn = nullptr;   // (under gpu_mu) recorded behind the whole-GPU kernels enqueued last: the next ones wait for it ON THE DEVICE
  hipEvent_t last_knn_turn = nullptr;   // (under gpu_mu) the same behind the exact vector search's stage enqueued last (vectors.cpp)
  std::mutex gpu_mu;    // device execution of one batch at a time: a scan kernel wants the whole GPU,
                        // overlapping two only stretches both (host-side planning/unpacking still overlap)
  std::mutex stats_mu;
  nrtgpu_stats stats{};
  std::atomic<int> knn_sketch_skip[4] = {};   // per similarity: panels that go straight to the fp32 rows (the sketch did not certify lately)
  double prof[16] = {0};
  std::vector<uint64_t> last_walls;   // NRTGPU_FLAG_PROFILE: {start, end, item, windows} per output slot of the last MaxScore launch
  int64_t last_walls_items = 0;       // ... whose first this-many slots are the items' owners (helpers behind the scan's slots)
  double ms_prof[16] = {0};   // the same for the items of the MaxScore route (nrtgpu_get_maxscore_profile)
  // request coalescing (nrtgpu_search_bm25_coalesced)
  std::mutex co_mu;
  std::condition_variable co_cv;
  std::vector<struct CoRequest*> co_pending;  // waiting for a leader
  struct CoRequest* co_leader = nullptr;      // the caller lingering for / about to run the next batch
  int co_inflight = 0;                        // coalesced batches executing right now
  int co_inflight_queries = 0;                // ... and how many queries they hold
  int co_last_batch = 0;                      // size of the batch formed last (a lone caller does not linger)
  int32_t co_linger_us = 150;
  // speculative thresholds (plan.h: kHitsSpecInvalid; nrtgpu_debug_spec_counters): queries run under speculation, queries whose guess
  // failed and were run again, and the switch the library throws itself when too many fail (docs not spread like a sample)
  std::atomic<int64_t> spec_queries{0}, spec_reruns{0};   // (sums over the context's leaf sets since nrtgpu_set_speculation: nrtgpu_stats)
  std::atomic<int> spec_off{0};                            // ... 1 once SOME leaf set has had its speculation switched off
  std::atomic<int> spec_scattered{0};                      // ... 1 once SOME leaf set has been moved to the scattered window order
  std::atomic<uint64_t> spec_epoch{1};                     // nrtgpu_set_speculation calls (the leaf sets' verdicts start over)
  std::atomic<int> spec_z16{5 * 16};   // the guess's margin x 16 (nrtgpu_set_speculation)
  std::atomic<int64_t> shard_docs{0}, index_docs{0};   // nrtgpu_set_shard_share: this context's share of a sharded index (0: equal shards)
  std::atomic<int64_t> live_segments{0};   // segment handles (uploads and forks) that have not been freed yet (nrtgpu_debug_live_segments)
  std::atomic<int> co_hold{0};                          // nrtgpu_debug_hold_coalescers: no leader (of either coalescer) leaves with less than a full batch / panel
  // the same for exact vector searches (nrtgpu_knn_exact_coalesced, vectors.cpp)
  std::mutex kco_mu;
  std::vector<struct KnnCoRequest*> kco_pending;
  struct KnnCoRequest* kco_leader = nullptr;   // the caller that will run the next panel
  int kco_inflight = 0;                        // panels executing right now (at most two)
  int kco_last_panel = 0;                      // members of the panel formed last (a leader that finds the device free waits for that cohort)
  // MyIndexSearcher.SlicingParams of the searcher this context serves (nrtgpu_set_slicing)
  std::atomic<int32_t> slice_max_docs{250000}, slice_max_segments{5}, virtual_shards{1};
  std::unique_ptr<nrtgpu::rt::WorkPool> pool;   // helper threads of the host side (planning, unpacking results)
  std::mutex launcher_mu;
  std::unique_ptr<Launcher> launcher;           // (search.cpp: device_begin_impl)
  // planner caches, one per leaf set seen lately (most recent first)
  std::mutex lsc_mu;
  std::vector<std::shared_ptr<nrtgpu::rt::LeafSetCache>> leaf_sets;
  // cross-GPU bound exchange (nrtgpu_exchange_open)
  void* xch_host = nullptr;                 // mmap of the shared table
  unsigned long long* xch_dev = nullptr;    // the same memory as the GPU sees it
  size_t xch_bytes = 0;
  int32_t xch_world = 0, xch_rank = 0;
};

namespace nrtgpu {
namespace rt {

// layout helper: carve 256-byte aligned regions out of one blob
struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
};

// ---- planner caches ----------------------------------------------------------------------------
// One term over one leaf set, resolved once: the per-leaf dictionary lookups and the static half of every DTerm.
// A batch of 1024 five-clause queries over 10 leaves otherwise pays 51 k dictionary probes; query terms repeat
// across batches (and within them), the leaf set changes only when the searcher is refreshed.
struct TermLeaves {
  const DTerm* d_table = nullptr;  // resident copy of the per-leaf records (columns, cell table, aux record, posting count in
                                   // `weight`; docids == nullptr: the leaf lacks the term) -- what DQTerm.table points at
  std::vector<uint32_t> count;     // postings per leaf
  int64_t total = 0;
  uint32_t max_norm = 0;           // largest norm byte of the field over the leaves that hold the term
};
struct LeafSetCache {
  std::vector<uint64_t> uids;    // the leaf set this cache belongs to (segment uids, in call order)
  struct Key {
    int32_t field;
    int64_t hash;
    bool operator==(const Key& o) const { return field == o.field && hash == o.hash; }
  };
  struct KeyHash {
    size_t operator()(const Key& k) const { return (size_t)(((uint64_t)k.hash ^ ((uint64_t)(uint32_t)k.field << 40)) * 0x9E3779B97F4A7C15ull >> 7); }
  };
  static const int kStripes = 64;
  static const size_t kMaxEntries = 1 << 20;  // then the context starts a fresh cache for the leaf set (bounds the memory)
  struct Stripe {
    std::shared_mutex mu;
    std::unordered_map<Key, std::shared_ptr<const TermLeaves>, KeyHash> map;
  };
  Stripe stripes[kStripes];
  // the terms' resident tables: carved out of 1 MiB device chunks that live as long as the cache (a plan in flight
  // holds a reference to the cache)
  int device = 0;
  std::mutex arena_mu;
  std::vector<void*> chunks;
  size_t chunk_used = 0;
  static const size_t kChunkBytes = 1 << 20;
  static const size_t kMaxChunks = 256;       // then the context starts a fresh cache for the leaf set
  bool full() {
    {
      std::lock_guard<std::mutex> lk(arena_mu);
      if (chunks.size() > kMaxChunks) return true;
    }
    return entries() > kMaxEntries;
  }
  void* alloc_tables(size_t stride, size_t want, size_t* got);
  ~LeafSetCache();
  // Entries are never freed while the cache lives (a full cache is replaced as a whole, leaf_set_cache()), so a raw
  // pointer stays valid for as long as the caller holds the cache -- which lets every planner thread keep a private
  // front cache and take no lock and no reference count on a hit (four threads bouncing the stripes' lock words cost
  // more than the lookups themselves).
  uint64_t id = 0;   // unique per cache object: tags the threads' front-cache entries
  // Speculative thresholds (plan.h: kHitsSpecInvalid) are judged PER LEAF SET: the queries run under them over these leaves, the
  // ones whose guess failed and were run again, and the switch the library throws for this leaf set when too many fail (an
  // index whose docid order defeats the estimate must not cost the context's other indexes their speculation; a refresh brings
  // a new leaf set and a fresh verdict).  spec_epoch: the context's nrtgpu_set_speculation count these numbers belong to.
  std::atomic<int64_t> spec_queries{0}, spec_reruns{0};
  std::atomic<int64_t> spec_calls{0}, spec_calls_rerun{0};   // calls judged, calls that needed a second pass (a second pass costs per CALL)
  std::atomic<int> spec_scattered{0};   // the leaf set's windows are walked in the scattered order (maxscore.hip): the second chance
  std::atomic<int> spec_off{0};
  std::atomic<uint64_t> spec_epoch{0};
  const TermLeaves* get(const nrtgpu_seg* const* segs, int32_t n_segs, int32_t field, int64_t hash);
  void get_many(const nrtgpu_seg* const* segs, int32_t n_segs, const Key* keys, size_t n, const TermLeaves** out);
  size_t entries();
};
std::shared_ptr<LeafSetCache> leaf_set_cache(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, int32_t n_segs);

// ---- workspaces (runtime.cpp) ------------------------------------------------------------------
int acquire_slot(nrtgpu_ctx* ctx, Slot** out);
void release_slot(nrtgpu_ctx* ctx, Slot* s);

// ---- segment store (segment.cpp) ---------------------------------------------------------------
// The doc set a query's hits must lie in: liveDocs & FILTER mask & ~MUST_NOT mask, resident in HBM
// ((0, 0): liveDocs itself, or nullptr once they are folded into the posting columns).
int accept_set_of(const nrtgpu_seg* seg, int32_t filter_mask, int32_t must_not_mask, const uint64_t** out);
int accept_set_of(const nrtgpu_seg* seg, const nrtgpu_bm25_query& q, const uint64_t** out);   // all of the query's FILTER / MUST_NOT masks
// vectors of the field whose doc is live (the hits of an exact vector query over the segment)
int64_t live_vector_count(const nrtgpu_seg* seg, const FieldData& f);
int ensure_vector_sketch(const nrtgpu_seg* seg, int32_t field_id);

// ---- vectors (vectors.cpp) ---------------------------------------------------------------------
// nrtgpu_knn_exact with device-resident results (per query k_stride sorted keys, count, live-vector total): the multi-GPU path
int knn_exact_device(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs, int32_t field_id,
                     int32_t sim, const float* queries, int32_t n_queries, int32_t dim, int32_t k, float boost, int32_t k_stride,
                     void* d_keys, void* d_counts, void* d_hits);

// ---- search (search.cpp): pieces the multi-GPU entry (dist.cpp) reuses ------------------------------
// one shard's BM25 search with device-resident results and speculative thresholds guessed against the WHOLE search (search.cpp)
int search_bm25_shard_device(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                             const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t k_stride, void* d_keys, void* d_counts, void* d_hits,
                             int32_t spec_world, void* d_guess, bool* speculated);
// nrtgpu_merge_topk_device + per query the packed key of rank k of the merged list (0: shorter), read from the merged keys
int merge_topk_device_kth(nrtgpu_ctx* ctx, int32_t n_lists, int32_t n_queries, int32_t k_stride, const void* d_keys_in, const void* d_counts_in,
                          const void* d_hits_in, const int32_t* ks, const int32_t* total_hits_thresholds, nrtgpu_topdocs* out, uint64_t* kth);
// a call that ran under speculation has come back: count it for its leaf set (the verdict: search.cpp)
void note_shard_speculation(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, int32_t n_segs, int64_t n_queries, int64_t n_failed);
// TopDocs.merge of n_lists gathered lists per query (layout [list][query]) into DEVICE arrays (keys n_queries x k_stride, counts,
// hit totals), enqueued on `slot`'s stream; the caller synchronises.
int merge_lists_on_device(nrtgpu_ctx* ctx, Slot* slot, int32_t n_lists, int32_t n_queries, int32_t k_stride, const void* g_keys,
                          const void* g_counts, const void* g_hits, const int32_t* ks, void* d_keys, void* d_counts, void* d_hits);
// The vector rescorer over device-resident first-pass lists (hybrid_rescore_kernel), enqueued on `slot`'s stream: uploads the leaf
// table and the query vectors into the slot's aux buffers; windows -> d_win_keys (n_queries x w_stride), d_win_counts.
// Query vectors of `dim_user` elements as the resident rows want them: padded with zeros to the field's resident dimension
// (FieldData.dim).  p == the caller's array when nothing had to be padded.
struct PaddedQueries {
  std::vector<float> buf;
  const float* p = nullptr;
  int32_t dim = 0;
};
int pad_query_vectors(const nrtgpu_seg* const* segs, int32_t n_segs, int32_t field_id, const float* queries, int32_t n, int32_t dim_user,
                      PaddedQueries* out);
int hybrid_tail_on_device(nrtgpu_ctx* ctx, Slot* slot, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                          int32_t field_id, int32_t sim, const float* query_vectors, int32_t dim, float boost, double qw, double rw,
                          int32_t window, int32_t n_queries, const void* d_first_keys, const void* d_first_counts, int32_t k_stride,
                          int32_t drop_foreign, void* d_win_keys, void* d_win_counts, uint32_t w_stride);

// ---- planner (planner.cpp) ---------------------------------------------------------------------
struct HostPlan {
  std::vector<DQuery> queries;
  std::vector<DItem> items;
  std::vector<DPart> parts;
  std::vector<DQTerm> qterms;       // compact plan: per query clause (expand_terms_kernel writes the DTerms on the device)
  std::vector<DQExpand> qexpand;    // per query
  std::vector<uint32_t> qs_begin;   // per (query, leaf): first DTerm of the pair, ~0 = none
  uint32_t n_dterms = 0;            // DTerm records the expansion writes
  uint32_t n_leaves = 0;
  std::shared_ptr<LeafSetCache> lsc;  // keeps the terms' resident tables alive until the call is over
  std::vector<float> caches;
  std::vector<uint32_t> list_idx;   // per query: item indices (merge input lists)
  std::vector<uint32_t> q_base, q_nlists, q_k;
  std::vector<uint64_t> theta_init;  // per query: key below which nothing is collected (min_competitive_score)
  uint32_t k_stride = 0;
  int64_t postings = 0;             // postings in the scanned term ranges (algorithmic work)
  bool fixed_point = false;         // every query of the batch passed the fixed-point range analysis
  bool clause_counting = false;     // exhaustive scan: some query has minimumNumberShouldMatch > 1 / is a DisjunctionMaxQuery: count-carrying variant
  bool masked = false;              // exhaustive scan: some part reads a doc-set mask (liveDocs / FILTER / MUST_NOT)
  bool ms_shapes = false;           // MaxScore route: some query has a FILTER / MUST_NOT mask, minimumNumberShouldMatch > 1 or is a DisjunctionMaxQuery
  bool ms_two = false;              // MaxScore route: some query's score needs the second accumulator (plan.h: kMsSec*)
  uint32_t n_slices = 1;            // searcher slices over the call's leaves (MyIndexSearcher.slices): per query that many hit sums
  // MaxScore route (maxscore.hip): items [0, n_ms_items) run it, the others the exhaustive scan
  uint32_t n_ms_items = 0;
  std::vector<uint32_t> q_wins;     // per query: doc windows of its MaxScore items (plan.h: MsArgs.q_wins)
  std::vector<int64_t> q_lower;     // per query on that route in kMsModePrune: live docs certain to match (> totalHitsThreshold), else 0
  std::vector<uint8_t> q_route;     // per query: kRouteScan (exhaustive scan) or kRouteMs + its kMsMode* (plan.h)
  int64_t ms_postings = 0;          // postings of the queries on that route (algorithmic work, as `postings`)
};

const uint8_t kRouteScan = 0, kRouteMs = 1;   // HostPlan.q_route: kRouteMs + kMsMode*
inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }
int validate_query(const nrtgpu_bm25_query& q, int qi);
// fixed-point eligibility of one query term: the scale 2^E at which all its scores are integers < 2^32
bool fixed_scale_of_term(float weight, const float* cache256, uint32_t max_norm, int32_t* scale);
// prune: 0 = every query is scanned exhaustively; 1 = queries that qualify take the MaxScore route (total_hits then a
// lower bound, relation GTE); 2 = the same for device-resident results: no searchAfter there (a page may hold fewer
// than numHits hits, and then the relation needs the exact count)
int build_plan(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
               const nrtgpu_bm25_query* queries, int32_t n_queries, HostPlan& hp, int prune = 0);

}  // namespace rt
}  // namespace nrtgpu

// test hook halves that live in vectors.cpp (nrtgpu_debug_hold_coalescers / nrtgpu_debug_coalescer_pending, search.cpp)
void nrtgpu_debug_knn_coalescer_wake(nrtgpu_ctx* ctx);
int nrtgpu_debug_knn_coalescer_pending(nrtgpu_ctx* ctx);
