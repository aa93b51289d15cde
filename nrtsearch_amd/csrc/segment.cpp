// segment.cpp -- the segment store: a read-only columnar replica of one Lucene segment's scoring data in HBM
// (upload, seal, liveDocs folded into the posting columns, doc-set masks and their combined accept sets).
#include "runtime_internal.h"

static void destroy_segment(nrtgpu_seg* seg);
SegWriteLock::SegWriteLock(nrtgpu_seg* s) : seg(s) {
  std::unique_lock<std::mutex> lk(seg->content_m);
  seg->content_writers_waiting++;
  seg->content_cv.wait(lk, [&] { return seg->content_readers == 0 && !seg->content_writing; });
  seg->content_writers_waiting--;
  seg->content_writing = true;
}
SegWriteLock::~SegWriteLock() {
  bool free_now = false;
  {
    std::lock_guard<std::mutex> lk(seg->content_m);
    seg->content_writing = false;
    // released while this writer held it: the LAST user frees -- not while readers run or another writer still waits on the
    // condition variable (it is a user of the handle too: it wakes up, writes to a handle nobody will search, and frees here)
    free_now = seg->content_released && seg->content_readers == 0 && seg->content_writers_waiting == 0;
    // notify while the mutex is held: the moment it is dropped another thread may reach "last user" and delete the handle,
    // condition variable included
    if (!free_now) seg->content_cv.notify_all();
  }
  if (free_now) destroy_segment(seg);
}
void nrtgpu_seg::content_lock_shared(bool pipelined) const {
  std::unique_lock<std::mutex> lk(content_m);
  // behind a writer that HOLDS the content: everybody.  Behind a writer that WAITS for the searches in flight: synchronous
  // searches only -- a pipelined one may come from the thread that must still wait for one of those searches
  content_cv.wait(lk, [&] { return !content_writing && (content_writers_waiting == 0 || (pipelined && content_readers > 0)); });
  content_readers++;
}
static void free_retired_accept_sets(const nrtgpu_seg* seg);
void nrtgpu_seg::content_unlock_shared() const {
  bool last_of_released = false;
  {
    std::lock_guard<std::mutex> lk(content_m);
    // the last reader out frees what the accept-set cache evicted while searches were in flight (under content_m: a search that
    // begins now cannot have been handed one of those sets -- they left the cache before they were retired)
    if (content_readers == 1 && !accept_retired_empty()) free_retired_accept_sets(this);
    content_readers--;
    // nrtgpu_segment_release came while this search ran: the last user frees (a writer that still waits counts as one: it is
    // woken below and frees in ~SegWriteLock)
    last_of_released = content_readers == 0 && content_released && !content_writing && content_writers_waiting == 0;
    if (!last_of_released) content_cv.notify_all();   // (under the mutex: after it is dropped `this` may be gone)
  }
  if (last_of_released) destroy_segment(const_cast<nrtgpu_seg*>(this));
}

static const size_t kMaxAcceptSets = 64;   // combined accept sets (liveDocs & filter & ~must_not) resident per segment
static const size_t kMaskPadBytes = 256;  // doc-set masks are readable one sub-tile (128 bytes) past max_doc

static const size_t kMaxRetiredAcceptSets = 64;   // evicted sets that wait for the searches in flight to end (then: refuse instead of evict)

// (no search is in flight over the handle: the caller holds the content exclusively, or is its last reader, or destroys it)
static void drop_accept_sets(nrtgpu_seg* seg) {
  std::lock_guard<std::mutex> lk(seg->accept_mu);
  const int64_t set_bytes = (int64_t)((seg->max_doc + 63) / 64) * 8;
  for (auto& kv : seg->accept) {
    (void)hipFree(kv.second.bits);
    seg->device_bytes -= set_bytes;
  }
  seg->accept.clear();
  for (uint64_t* p : seg->accept_retired) {
    (void)hipFree(p);
    seg->device_bytes -= set_bytes;
  }
  seg->accept_retired.clear();
}
// The last search in flight over the handle has ended: what the cache evicted while searches ran can go.
static void free_retired_accept_sets(const nrtgpu_seg* seg) {
  std::vector<uint64_t*> gone;
  {
    std::lock_guard<std::mutex> lk(seg->accept_mu);
    gone.swap(seg->accept_retired);
    const_cast<nrtgpu_seg*>(seg)->device_bytes -= (int64_t)gone.size() * (int64_t)((seg->max_doc + 63) / 64) * 8;
  }
  if (gone.empty()) return;
  (void)hipSetDevice(seg->ctx->device);
  for (uint64_t* p : gone) (void)hipFree(p);
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
  static std::atomic<uint64_t> next_uid{1};
  auto* seg = new nrtgpu_seg();
  seg->ctx = ctx;
  seg->core->device = ctx->device;
  seg->uid = next_uid.fetch_add(1, std::memory_order_relaxed);
  seg->max_doc = max_doc;
  seg->n_tiles = (uint32_t)(((int64_t)max_doc + kTileDocs - 1) / kTileDocs);
  ctx->live_segments.fetch_add(1, std::memory_order_relaxed);
  *out = seg;
  return NRTGPU_OK;
}

#ifdef NRTGPU_DEV
// Test hook (include/nrtgpu_dev.h): segment handles of the context (uploads and forks) not freed yet -- a handle released under
// running searches is freed by the last of them (nrtgpu_segment_release), and this is how a test sees that it was.
extern "C" int64_t nrtgpu_debug_live_segments(nrtgpu_ctx* ctx) { return ctx ? ctx->live_segments.load(std::memory_order_relaxed) : -1; }
#endif

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

// A sparse term's cell table holds one posting offset per 2^shift sub-tiles; a lookup (maxscore.hip) binary-searches the docids
// of ONE cell, so the table is sized for about kCellPostings postings per cell (it then takes 4 / kCellPostings bytes per posting:
// "table <= 1/4 of the term's postings").
static const int kCellPostings = 4;   // (8 -> 4: -2 % kernel time for +1.4 % segment bytes at C3; 2: -2.6 % for +3.6 %, profiles/r03_kernel_shapes.log)

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
    static const int64_t budget_div = [] {   // (development build: postings per cell aimed for, NRTGPU_CELL_POSTINGS)
      const int64_t v = dev_env_int("NRTGPU_CELL_POSTINGS", kCellPostings);
      return v >= 1 && v <= 64 ? v : (int64_t)kCellPostings;
    }();
    const uint64_t budget = std::max<int64_t>(1, cnt / budget_div);
    // (packed postings: a cell must not span two 2^20-doc super-windows -- a posting's doc offset is relative to its cell's)
    const uint32_t max_shift = (seg->ctx->cfg.flags & NRTGPU_FLAG_PACKED_POSTINGS) ? kPackMaxCellShift : 31u;
    while (((uint64_t)(seg->n_tiles - 1) >> shift) + 1 > budget && shift < max_shift) ++shift;
    const uint32_t n_cells = (uint32_t)(((uint64_t)(seg->n_tiles - 1) >> shift) + 1);
    TermEntry& e = entries[(size_t)t];
    e.group = (uint32_t)f.groups.size();
    e.start = (uint64_t)lo;
    e.count = (uint32_t)cnt;
    e.shift = shift;
    e.cell_start = cells.size();
    e.aux_idx = (uint32_t)t;
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
  {
    // A term id names ONE posting list of the field: two terms under one id -- in this call or across calls (e.g. a 64-bit hash
    // of the term bytes that collides) -- would silently alias.  Refused, so that the caller keeps the leaf on its own path.
    std::unordered_map<int64_t, int64_t> seen_here;
    seen_here.reserve((size_t)n_terms * 2);
    for (int64_t t = 0; t < n_terms; ++t) {
      if (f.dict.count(term_hash[t]) || !seen_here.emplace(term_hash[t], t).second)
        return fail(NRTGPU_ERR_INVALID_ARG, "term id %lld added twice to field %d (term ids must be unique per field)", (long long)term_hash[t], field_id);
    }
  }

  TermGroup g;
  g.n_postings = (uint64_t)total;
  g.n_terms = (uint32_t)n_terms;
  g.h_start.resize((size_t)n_terms);
  g.h_count.resize((size_t)n_terms);
  for (int64_t t = 0; t < n_terms; ++t) {
    g.h_start[(size_t)t] = (uint64_t)offsets[t];
    g.h_count[(size_t)t] = (uint32_t)(offsets[t + 1] - offsets[t]);
  }
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
  // Resident rows are a multiple of 16 elements long (the matrix-core kernels' pieces): a field of another dimension is padded
  // with zeros, which change none of the sums (x + 0 * 0 = x, (q - v)^2 = 0 for a zero pair) -- results are the field's own.
  const int32_t dim_user = dim;
  dim = (dim + 15) & ~15;
  if (dim > 2048) return fail(NRTGPU_ERR_UNSUPPORTED, "vector dimension %d (device path takes <= 2048)", dim_user);
  f.dim = dim;
  f.dim_user = dim_user;
  f.n_vec = n;
  if (n == 0) return NRTGPU_OK;
  void* p = nullptr;
  if (int rc = dev_alloc(seg, &p, (size_t)n * dim * 4 + 256)) return rc;
  f.d_vectors = (float*)p;
  if (dim == dim_user) {
    HIP_TRY(hipMemcpy(f.d_vectors, row_major, (size_t)n * dim * 4, hipMemcpyHostToDevice));
  } else {
    HIP_TRY(hipMemset(f.d_vectors, 0, (size_t)n * dim * 4));
    HIP_TRY(hipMemcpy2D(f.d_vectors, (size_t)dim * 4, row_major, (size_t)dim_user * 4, (size_t)dim_user * 4, (size_t)n, hipMemcpyHostToDevice));
  }
  if (int rc = dev_alloc(seg, &p, (size_t)n * 4 + 64)) return rc;
  f.d_vnorm2 = (float*)p;
  launch_knn_row_norms(nullptr, f.d_vectors, dim, n, f.d_vnorm2);
  // the largest |v|^2 of the field in this segment: the rounding bound the exact search certifies its answer with (vectors.cpp)
  uint32_t* d_max = (uint32_t*)(f.d_vnorm2 + n);   // (the allocation's spare tail)
  HIP_TRY(hipMemsetAsync(d_max, 0, 4, nullptr));
  launch_knn_norm_max(nullptr, f.d_vnorm2, n, d_max);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&f.vnorm2_max, d_max, 4, hipMemcpyDeviceToHost));
  {
    // what the exact search's error bounds need besides (vectors.cpp): max |element|, min non-zero |v|^2 (float bits).  The
    // fp16 sketch itself is built by the first exact search over the field (ensure_vector_sketch below): a field that is only
    // ever used to RESCORE hits never pays its +50 % of HBM
    uint32_t stats[2] = {0u, 0xFFFFFFFFu};
    HIP_TRY(hipMemcpy(d_max + 1, stats, 8, hipMemcpyHostToDevice));
    launch_knn_absmax(nullptr, f.d_vectors, (int64_t)n * dim, d_max + 1);
    launch_knn_norm_min(nullptr, f.d_vnorm2, n, d_max + 2);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(stats, d_max + 1, 8, hipMemcpyDeviceToHost));
    memcpy(&f.absmax, &stats[0], 4);
    if (stats[1] != 0xFFFFFFFFu) memcpy(&f.vnorm2_min, &stats[1], 4);
    // (rows with inf / NaN, or a context that declines the sketch: the fp32 pass only)
    f.sketch_state = ((seg->ctx->cfg.flags & NRTGPU_FLAG_NO_VECTOR_SKETCH) || !std::isfinite(f.absmax) || !std::isfinite(f.vnorm2_max)) ? -1 : 0;
  }
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

// What the MaxScore route needs per term besides the columns (plan.h: DTermAux), built on the device from the sealed
// columns: the impact frontier of every term and, for the terms the segment's LOOKUP BUDGET pays for, a doc -> posting lookup
// structure for the walk's later clauses (plan.h: kLook*).  Terms are served in the order of their posting counts, largest
// first -- lookups go to a query's densest clauses, so that is the order of benefit per byte -- by the first rule of the POLICY
// that applies to the term and whose structure still fits the budget:
//   kind:N  the N largest terms of the upload group        kind@D  terms with a posting per D docs or more        kind  every term
// kinds: bits (records, 0.25 B per doc), cells (4 - 8 B per posting).  Terms under kLookMinPostings postings, and whatever no
// rule or no budget covers, are searched in their cell of the tile-granular table.
// The budget (nrtgpu_config.lookup_budget_pct, default kLookBudgetPct): lookup bytes <= that share of the upload group's
// resident posting bytes (8 B per posting, 4 B under NRTGPU_FLAG_PACKED_POSTINGS).
// Rounds 2-4 kept records for "the 2048 largest terms", no budget, and the coarse tile cells behind them: 5.1 GB of records
// next to 0.6 GB of postings at C3's 10 M docs.  Measured in round 5 (profiles/r05_look_policy_budget_curve.log, same box,
// kernel ms per 1024 C3 queries / resident GB): records for 2048 terms 2.11 / 5.73; records for the terms with a posting per
// 128 docs + lookup cells 2.13 / 1.15; per 256 docs 2.12 / 1.43; per 1024 docs (budget 300 %) 2.14 / 2.49 -- with fine cells
// behind them, records beyond the ~250 most frequent terms buy nothing.
static const uint32_t kLookMinPostings = 64;
static const int kLookBudgetPct = 150;
static const char* const kLookPolicy = "bits@256,cells";
struct LookRule {
  uint32_t kind;
  int64_t rank_limit;   // < 0: none
  int64_t density;      // docs per posting, < 0: none
};
static std::vector<LookRule> parse_look_policy(const char* text) {
  std::vector<LookRule> out;
  std::string t(text ? text : "");
  size_t i = 0;
  while (i < t.size()) {
    size_t j = t.find(',', i);
    if (j == std::string::npos) j = t.size();
    const std::string item = t.substr(i, j - i);
    i = j + 1;
    const size_t sep = item.find_first_of(":@");
    const std::string name = item.substr(0, sep);
    LookRule r{kLookNone, -1, -1};
    if (name == "bits") r.kind = kLookBits;
    else if (name == "cells") r.kind = kLookCells;
    else continue;
    if (sep != std::string::npos) {
      const int64_t v = atoll(item.c_str() + sep + 1);
      if (item[sep] == ':') r.rank_limit = v;
      else r.density = std::max<int64_t>(v, 1);
    }
    out.push_back(r);
  }
  return out;
}
static int build_term_aux(nrtgpu_seg* seg, TermGroup& g) {
  if (g.d_aux || g.n_terms == 0) return NRTGPU_OK;
  const size_t nt = g.n_terms;
  const uint32_t max_doc = (uint32_t)seg->max_doc;
  const int pct = seg->ctx->cfg.lookup_budget_pct == 0 ? kLookBudgetPct : std::max(seg->ctx->cfg.lookup_budget_pct, 0);
  const uint64_t posting_bytes = g.n_postings * ((seg->ctx->cfg.flags & NRTGPU_FLAG_PACKED_POSTINGS) ? 4ull : 8ull);
  uint64_t budget = posting_bytes / 100ull * (uint64_t)pct + posting_bytes % 100ull * (uint64_t)pct / 100ull;
  const std::vector<LookRule> rules = parse_look_policy(dev_env_str("NRTGPU_LOOK_POLICY", kLookPolicy));
  std::vector<uint64_t> look((size_t)nt, ~0ull);   // byte offset of the term's structure inside the group's buffer
  std::vector<uint32_t> meta((size_t)nt, 0u);      // kind | log2 docs per cell << 8
  std::vector<uint32_t> which[3];                  // per kind the terms that got it
  uint64_t look_bytes = 0;
  uint32_t max_count = 0, max_cells = 0;
  {
    std::vector<uint32_t> order;
    for (size_t t = 0; t < nt; ++t)
      if (g.h_count[t] >= kLookMinPostings) order.push_back((uint32_t)t);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return g.h_count[a] != g.h_count[b] ? g.h_count[a] > g.h_count[b] : a < b; });
    for (size_t oi = 0; oi < order.size(); ++oi) {
      const uint32_t t = order[oi];
      const uint64_t cnt = g.h_count[t];
      for (const LookRule& r : rules) {
        if (r.rank_limit >= 0 && (int64_t)oi >= r.rank_limit) continue;
        if (r.density >= 0 && cnt * (uint64_t)r.density < (uint64_t)max_doc) continue;
        uint64_t cost = 0;
        uint32_t shift = 0;
        // (every structure: 16-byte aligned, entry 0 readable for idle slots, one entry of slack behind the last doc's)
        if (r.kind == kLookBits) cost = (((((uint64_t)max_doc + 31ull) / 32ull + 1ull) * 8ull + 15ull) & ~15ull);
        else {
          // cells of 2^shift docs, the largest power of two with at most one posting per cell on average
          while (shift < 31u && (cnt << (shift + 1u)) <= (uint64_t)max_doc) ++shift;
          const uint64_t n_cells = (((uint64_t)max_doc - 1ull) >> shift) + 1ull;
          cost = (((n_cells + 2ull) * 4ull + 15ull) & ~15ull);
        }
        if (cost > budget) continue;   // (a later rule's structure may fit)
        budget -= cost;
        look[t] = look_bytes;
        look_bytes += cost;
        meta[t] = r.kind | (shift << 8);
        which[r.kind].push_back(t);
        max_count = std::max(max_count, (uint32_t)cnt);
        if (r.kind == kLookCells) max_cells = std::max<uint32_t>(max_cells, (uint32_t)((((uint64_t)max_doc - 1ull) >> shift) + 2ull));
        break;
      }
    }
  }
  void* p = nullptr;
  if (int rc = dev_alloc(seg, &p, nt * sizeof(DTermAux))) return rc;
  g.d_aux = (DTermAux*)p;
  if (look_bytes) {
    if (int rc = dev_alloc(seg, &p, (size_t)look_bytes + 64)) return rc;
    g.d_look = (char*)p;
    g.look_bytes = look_bytes;
    HIP_TRY(hipMemset(g.d_look, 0, (size_t)look_bytes + 64));
  }
  std::vector<uint32_t> flat;
  size_t first_of[3] = {0, 0, 0};
  for (uint32_t k = 1; k < 3; ++k) {
    first_of[k] = flat.size();
    flat.insert(flat.end(), which[k].begin(), which[k].end());
  }
  uint64_t* d_start = nullptr;
  uint64_t* d_look = nullptr;
  uint32_t* d_count = nullptr;
  uint32_t* d_meta = nullptr;
  uint32_t* d_which = nullptr;
  auto free_tmp = [&] {
    if (d_start) (void)hipFree(d_start);
    if (d_look) (void)hipFree(d_look);
    if (d_count) (void)hipFree(d_count);
    if (d_meta) (void)hipFree(d_meta);
    if (d_which) (void)hipFree(d_which);
  };
  hipError_t e = hipMalloc((void**)&d_start, nt * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&d_look, nt * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&d_count, nt * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_meta, nt * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_which, std::max<size_t>(flat.size(), 1) * 4);
  if (e == hipSuccess) e = hipMemcpy(d_start, g.h_start.data(), nt * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_look, look.data(), nt * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_count, g.h_count.data(), nt * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_meta, meta.data(), nt * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && !flat.empty()) e = hipMemcpy(d_which, flat.data(), flat.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_term_frontier(nullptr, g.d_fnorm, d_start, d_count, d_look, d_meta, g.d_look, (uint32_t)nt, g.d_aux);
    launch_term_bits(nullptr, g.d_docids, d_start, d_count, d_look, d_which + first_of[kLookBits], (uint32_t)which[kLookBits].size(), max_count, g.d_look);
    launch_term_cells(nullptr, g.d_docids, d_start, d_count, d_look, d_meta, d_which + first_of[kLookCells], (uint32_t)which[kLookCells].size(), max_cells,
                      max_doc, g.d_look);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  free_tmp();
  if (e != hipSuccess) return fail(NRTGPU_ERR_HIP, "building the MaxScore lookup structures failed: %s", hipGetErrorString(e));
  for (uint32_t k = 1; k < 3; ++k) g.n_look[k] = (uint32_t)which[k].size();
  g.h_start.clear();
  g.h_start.shrink_to_fit();
  g.h_count.clear();
  g.h_count.shrink_to_fit();
  return NRTGPU_OK;
}

// NRTGPU_FLAG_PACKED_POSTINGS: the group's docid and code columns become ONE column of 32-bit words plus the group's
// exception list (plan.h: kPack*); the two columns are freed.  From here on g.d_docids is the packed column and
// g.d_fnorm the list (what DTerm.docids / DTerm.fnorm then mean to the kernels).
static int pack_group(nrtgpu_seg* seg, TermGroup& g) {
  if (g.packed) return NRTGPU_OK;
  const uint64_t n = g.n_postings;
  if (n >= (1ull << 32)) return fail(NRTGPU_ERR_UNSUPPORTED, "packed postings: an upload group of 2^32 postings or more");
  const uint32_t n_blocks = (uint32_t)((n + (1ull << kPackEscBlockShift) - 1) >> kPackEscBlockShift);
  // exceptions per block of 2048 postings -> directory (exclusive prefix)
  std::vector<uint32_t> dir((size_t)n_blocks + 1, 0u);
  if (n_blocks) {
    uint32_t* d_counts = nullptr;
    HIP_TRY(hipMalloc((void**)&d_counts, (size_t)n_blocks * 4));
    launch_pack_count(nullptr, g.d_fnorm, n, n_blocks, d_counts);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(dir.data() + 1, d_counts, (size_t)n_blocks * 4, hipMemcpyDeviceToHost);   // (syncs the null stream)
    (void)hipFree(d_counts);
    if (e != hipSuccess) return fail(NRTGPU_ERR_HIP, "packing the postings failed: %s", hipGetErrorString(e));
  }
  uint64_t run = 0;
  for (uint32_t b = 0; b < n_blocks; ++b) {
    const uint32_t c = dir[(size_t)b + 1];
    dir[(size_t)b + 1] = 0;
    dir[(size_t)b] = (uint32_t)run;
    run += c;
  }
  dir[(size_t)n_blocks] = (uint32_t)run;
  const uint32_t n_exc = (uint32_t)run;
  const size_t list_words = 4 + (size_t)n_blocks + 1 + (size_t)n_exc;
  const size_t col_bytes = (((size_t)n * 4 + 15) & ~(size_t)15) + 64;
  void* p = nullptr;
  if (int rc = dev_alloc(seg, &p, list_words * 4 + 64)) return rc;
  uint32_t* d_list = (uint32_t*)p;
  if (int rc = dev_alloc(seg, &p, col_bytes)) {
    (void)hipFree(d_list);
    seg->device_bytes -= (int64_t)(list_words * 4 + 64);
    return rc;
  }
  uint32_t* d_packed = (uint32_t*)p;
  const uint32_t header[4] = {n_blocks, n_exc, 0u, 0u};
  hipError_t e = hipMemset(d_packed, 0, col_bytes);
  if (e == hipSuccess) e = hipMemset(d_list, 0, list_words * 4 + 64);
  if (e == hipSuccess) e = hipMemcpy(d_list, header, sizeof(header), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_list + 4, dir.data(), dir.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_pack_write(nullptr, g.d_docids, g.d_fnorm, n, n_blocks, d_list + 4, d_list + 4 + n_blocks + 1, d_packed);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {  // the two-column form stays in place (and is freed with the segment)
    (void)hipFree(d_packed);
    (void)hipFree(d_list);
    seg->device_bytes -= (int64_t)(col_bytes + list_words * 4 + 64);
    return fail(NRTGPU_ERR_HIP, "packing the postings failed: %s", hipGetErrorString(e));
  }
  (void)hipFree(g.d_docids);   // [docid column | code column]: one allocation
  seg->device_bytes -= (int64_t)(2 * col_bytes);
  g.d_docids = d_packed;
  g.d_fnorm = d_list;
  g.d_dict = d_list;
  g.packed = true;
  return NRTGPU_OK;
}

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
  for (auto& kv : seg->fields)
    for (auto& g : kv.second.groups)
      if (int rc2 = build_term_aux(seg, g)) return rc2;
  if (seg->ctx->cfg.flags & NRTGPU_FLAG_PACKED_POSTINGS)   // (the seal-time builders above read the two-column form)
    for (auto& kv : seg->fields)
      for (auto& g : kv.second.groups)
        if (int rc2 = pack_group(seg, g)) return rc2;
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
  if (seg->ctx->cfg.flags & (NRTGPU_FLAG_NO_LIVE_FOLD | NRTGPU_FLAG_PACKED_POSTINGS)) return NRTGPU_OK;   // (a packed word has no spare bit)
  if (!seg->sealed) return NRTGPU_OK;  // seal folds
  if (seg->core.use_count() > 1) return NRTGPU_OK;  // the columns belong to several reader versions: this one's deletes stay a mask
  for (auto& kv : seg->fields)
    for (auto& g : kv.second.groups)
      launch_apply_live(nullptr, g.d_docids, g.d_fnorm, g.n_postings, seg->d_live);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  seg->live_folded = seg->d_live != nullptr;
  seg->core->folded = seg->live_folded;
  seg->core->folded_live = seg->live_folded ? seg->h_live : std::vector<uint64_t>();
  return NRTGPU_OK;
}

extern "C" int nrtgpu_segment_set_live_docs(nrtgpu_seg* seg, const uint64_t* bits, int32_t n_words) {
  if (!seg) return fail(NRTGPU_ERR_INVALID_ARG, "seg is NULL");
  SegWriteLock content(seg);  // waits for the searches running over this segment; later ones wait for it
  HIP_TRY(hipSetDevice(seg->ctx->device));
  const int32_t need = (seg->max_doc + 63) / 64;
  if (bits && n_words < need) return fail(NRTGPU_ERR_INVALID_ARG, "live bits: %d words given, %d needed", n_words, need);
  if (seg->core.use_count() > 1 && seg->core->folded) {
    // the shared posting columns carry another reader version's deletes: this version may only have MORE of them
    for (int32_t i = 0; i < need; ++i) {
      const uint64_t want = bits ? bits[i] : ~0ull;
      uint64_t extra = want & ~seg->core->folded_live[(size_t)i];
      if (i == need - 1 && (seg->max_doc & 63)) extra &= (1ull << (seg->max_doc & 63)) - 1ull;
      if (extra)
        return fail(NRTGPU_ERR_UNSUPPORTED, "liveDocs of a forked segment may only lose docs (doc %d is deleted in the shared columns)",
                    i * 64 + __builtin_ctzll(extra));
    }
  }
  static std::atomic<uint64_t> next_live_version{2};
  drop_accept_sets(seg);
  seg->live_version = next_live_version.fetch_add(1, std::memory_order_relaxed);
  if (!bits) {
    if (seg->d_live) (void)hipFree(seg->d_live);
    seg->d_live = nullptr;
    seg->h_live.clear();
    seg->n_deleted = 0;
    return fold_live_docs(seg);
  }
  if (n_words < need) return fail(NRTGPU_ERR_INVALID_ARG, "live bits: %d words given, %d needed", n_words, need);
  seg->h_live.assign(bits, bits + need);
  {
    int64_t live = 0;
    for (int32_t i = 0; i < need; ++i) {
      uint64_t w = bits[i];
      if (i == need - 1 && (seg->max_doc & 63)) w &= (1ull << (seg->max_doc & 63)) - 1ull;
      live += __builtin_popcountll(w);
    }
    seg->n_deleted = (int32_t)((int64_t)seg->max_doc - live);
  }
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
// The doc set a query's hits must lie in on this leaf: liveDocs AND every FILTER mask AND NOT any MUST_NOT mask (nullptr: every
// doc, or liveDocs folded into the postings).  Combined on the host -- one pass over 64-bit words -- the first time a combination
// is asked for, resident from then on (until liveDocs or a mask change).
static int accept_set_of_ids(const nrtgpu_seg* seg, std::vector<int32_t> filters, std::vector<int32_t> must_nots, const uint64_t** out) {
  auto canon = [](std::vector<int32_t>& v) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
  };
  canon(filters);
  canon(must_nots);
  if (filters.empty() && must_nots.empty()) {
    *out = seg->live_folded ? nullptr : seg->d_live;
    return 0;
  }
  std::vector<int32_t> key(filters);
  key.push_back(0);
  key.insert(key.end(), must_nots.begin(), must_nots.end());
  std::lock_guard<std::mutex> lk(seg->accept_mu);
  auto it = seg->accept.find(key);
  if (it != seg->accept.end()) {
    it->second.used = ++seg->accept_clock;
    *out = it->second.bits;
    return 0;
  }
  // Bounded like the reference's LRUQueryCache, and like it the least recently used set makes room -- but a set
  // may be in use by a search in flight (the caller's own plan included: it holds the content like every search), so an evicted
  // set is RETIRED: out of the cache at once, freed by the last search to leave the handle (content_unlock_shared).  Only when
  // as many retired sets wait as the cache holds -- a handle that is never idle, under more combinations than it can keep -- the
  // bound refuses, and that combination runs on the caller's own path.
  if (seg->accept.size() >= kMaxAcceptSets) {
    if (seg->accept_retired.size() >= kMaxRetiredAcceptSets)
      return fail(NRTGPU_ERR_UNSUPPORTED, "%zu combined doc sets are resident on a segment and %zu evicted ones wait for searches in flight",
                  seg->accept.size(), seg->accept_retired.size());
    auto lru = seg->accept.begin();
    for (auto a = seg->accept.begin(); a != seg->accept.end(); ++a)
      if (a->second.used < lru->second.used) lru = a;
    seg->accept_retired.push_back(lru->second.bits);
    seg->accept.erase(lru);
  }
  std::vector<const std::vector<uint64_t>*> f, mn;
  for (int32_t id : filters) {
    auto m = seg->masks.find(id);
    if (m == seg->masks.end()) return fail(NRTGPU_ERR_UNSUPPORTED, "filter mask %d is not resident on a segment", id);
    f.push_back(&m->second);
  }
  for (int32_t id : must_nots) {
    auto m = seg->masks.find(id);
    if (m == seg->masks.end()) return fail(NRTGPU_ERR_UNSUPPORTED, "must_not mask %d is not resident on a segment", id);
    mn.push_back(&m->second);
  }
  const size_t need = (size_t)(seg->max_doc + 63) / 64;
  std::vector<uint64_t> w(need + kMaskPadBytes / 8, 0ull);  // padded like liveDocs
  for (size_t i = 0; i < need; ++i) {
    uint64_t v = seg->h_live.empty() ? ~0ull : seg->h_live[i];
    for (const std::vector<uint64_t>* m : f) v &= (*m)[i];
    for (const std::vector<uint64_t>* m : mn) v &= ~(*m)[i];
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
  seg->accept[key] = nrtgpu_seg::AcceptSet{(uint64_t*)p, ++seg->accept_clock};
  *out = (uint64_t*)p;
  return 0;
}

int nrtgpu::rt::accept_set_of(const nrtgpu_seg* seg, int32_t filter_mask, int32_t must_not_mask, const uint64_t** out) {
  std::vector<int32_t> f, mn;
  if (filter_mask) f.push_back(filter_mask);
  if (must_not_mask) mn.push_back(must_not_mask);
  return accept_set_of_ids(seg, std::move(f), std::move(mn), out);
}

int nrtgpu::rt::accept_set_of(const nrtgpu_seg* seg, const nrtgpu_bm25_query& q, const uint64_t** out) {
  std::vector<int32_t> f, mn;
  if (q.filter_mask) f.push_back(q.filter_mask);
  if (q.must_not_mask) mn.push_back(q.must_not_mask);
  for (int i = 0; i < q.n_more_filters; ++i) f.push_back(q.more_filters[i]);
  for (int i = 0; i < q.n_more_must_not; ++i) mn.push_back(q.more_must_not[i]);
  return accept_set_of_ids(seg, std::move(f), std::move(mn), out);
}

int64_t nrtgpu::rt::live_vector_count(const nrtgpu_seg* seg, const FieldData& f) {
  if (seg->h_live.empty()) return f.n_vec;
  if (f.live_vec_version.load(std::memory_order_acquire) == seg->live_version) {
    const int64_t c = f.live_vec.load(std::memory_order_acquire);
    if (c >= 0) return c;
  }
  int64_t n = 0;
  const uint64_t* live = seg->h_live.data();
  if (f.h_ord_to_doc.empty()) {  // row == docid
    const int32_t full = f.n_vec / 64;
    for (int32_t i = 0; i < full; ++i) n += __builtin_popcountll(live[i]);
    if (f.n_vec & 63) n += __builtin_popcountll(live[full] & ((1ull << (f.n_vec & 63)) - 1ull));
  } else {
    for (int32_t d : f.h_ord_to_doc) n += (live[d >> 6] >> (d & 63)) & 1ull;
  }
  f.live_vec.store(n, std::memory_order_release);
  f.live_vec_version.store(seg->live_version, std::memory_order_release);
  return n;
}

// The fp16 sketch of a field's rows (knn.hip): the rows scaled by a power of two that puts the largest |element| at 2^14 at
// most, rounded to fp16, in matrix-core operand order.  Built on first use, once per segment core (its forks share it); a
// segment that cannot have one (declined, inf / NaN rows, no HBM left) is searched from its fp32 rows.
int nrtgpu::rt::ensure_vector_sketch(const nrtgpu_seg* seg, int32_t field_id) {
  auto fit = seg->fields.find(field_id);
  if (fit == seg->fields.end() || !fit->second.d_vectors || fit->second.n_vec == 0) return NRTGPU_OK;
  FieldData& f = fit->second;
  std::lock_guard<std::mutex> lk(seg->core->sketch_mu);
  if (f.sketch_state != 0) return NRTGPU_OK;   // built already, or never
  HIP_TRY(hipSetDevice(seg->core->device));
  void* p = nullptr;
  const size_t bytes = knn_sketch_bytes(f.dim, f.n_vec) + 256;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    f.sketch_state = -1;
    return NRTGPU_OK;
  }
  int e = 0;
  (void)std::frexp(f.absmax, &e);   // absmax < 2^e
  f.sketch_scale = f.absmax > 0.f ? std::ldexp(1.0f, 14 - e) : 1.0f;
  launch_knn_sketch_build(nullptr, f.d_vectors, f.dim, f.n_vec, f.sketch_scale, p);
  hipError_t he = hipGetLastError();
  if (he == hipSuccess) he = hipStreamSynchronize(nullptr);
  if (he != hipSuccess) {   // no sketch for this field, ever (the fp32 pass serves it): nothing leaks, nobody retries
    (void)hipFree(p);
    f.sketch_state = -1;
    return fail(NRTGPU_ERR_HIP, "vector sketch build: %s", hipGetErrorString(he));
  }
  f.d_sketch = p;
  f.sketch_state = 1;
  seg->core->shared_extra_bytes.fetch_add((int64_t)bytes, std::memory_order_relaxed);   // (the core's, not the handle's that happened to trigger it)
  return NRTGPU_OK;
}

SegCore::~SegCore() {
  (void)hipSetDevice(device);
  for (auto& kv : fields) {
    FieldData& f = kv.second;
    if (f.d_norms) (void)hipFree(f.d_norms);
    if (f.d_vectors) (void)hipFree(f.d_vectors);
    if (f.d_vnorm2) (void)hipFree(f.d_vnorm2);
    if (f.d_sketch) (void)hipFree(f.d_sketch);
    if (f.d_ord_to_doc) (void)hipFree(f.d_ord_to_doc);
    for (auto& g : f.groups) {
      if (g.d_docids) (void)hipFree(g.d_docids);
      if (g.d_dict) (void)hipFree(g.d_dict);
      if (g.d_freqs) (void)hipFree(g.d_freqs);
      if (g.d_cells) (void)hipFree(g.d_cells);
      if (g.d_aux) (void)hipFree(g.d_aux);
      if (g.d_look) (void)hipFree(g.d_look);
    }
  }
}

static void destroy_segment(nrtgpu_seg* seg) {
  (void)hipSetDevice(seg->ctx->device);
  seg->ctx->live_segments.fetch_sub(1, std::memory_order_relaxed);
  drop_accept_sets(seg);
  if (seg->d_live) (void)hipFree(seg->d_live);
  delete seg;   // (the shared core -- columns, norms, vectors -- goes with its last handle)
}

// Safe under running searches (the reference closes readers while SEARCH-pool threads run: ShardState.java:506-527; close
// listeners: TextBaseFieldDef.java:335-371): the searches in flight over the handle keep what they read, and the last of
// them frees it.  The caller must not START a search with the handle after this call.
extern "C" void nrtgpu_segment_release(nrtgpu_seg* seg) {
  if (!seg) return;
  {
    std::lock_guard<std::mutex> lk(seg->content_m);
    if (seg->content_readers > 0 || seg->content_writing || seg->content_writers_waiting > 0) {   // (a parked set_mask / set_live_docs is a user too)
      seg->content_released = true;
      return;
    }
  }
  destroy_segment(seg);
}

// A new reader version of a sealed segment: same immutable data (shared, not copied), its own liveDocs / masks.  Searches
// over the previous handle keep seeing the previous liveDocs -- the point-in-time view an IndexSearcher has in Lucene.
extern "C" int nrtgpu_segment_fork(nrtgpu_seg* seg, const uint64_t* live_bits, int32_t n_words, nrtgpu_seg** out) {
  if (!seg || !out) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (!seg->sealed) return fail(NRTGPU_ERR_STATE, "segment is not sealed");
  auto* f = new nrtgpu_seg(seg->core);
  f->ctx = seg->ctx;
  f->uid = seg->uid;
  f->max_doc = seg->max_doc;
  f->n_tiles = seg->n_tiles;
  f->sealed = true;
  f->ctx->live_segments.fetch_add(1, std::memory_order_relaxed);
  if (int rc = nrtgpu_segment_set_live_docs(f, live_bits, n_words)) {
    nrtgpu_segment_release(f);
    return rc;
  }
  *out = f;
  return NRTGPU_OK;
}

extern "C" int64_t nrtgpu_segment_device_bytes(const nrtgpu_seg* seg) {
  return seg ? seg->device_bytes + seg->core->shared_extra_bytes.load(std::memory_order_relaxed) : 0;
}
