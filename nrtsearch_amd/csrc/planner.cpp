// planner.cpp -- one batch of queries -> the launch plan of the scan and merge kernels: term resolution per
// (clause, leaf), score-table and fixed-point analysis, work items of equal cost.
#include "runtime_internal.h"

// ------------------------------------------------------------------------------------------------
// plan building
// ------------------------------------------------------------------------------------------------


int nrtgpu::rt::validate_query(const nrtgpu_bm25_query& q, int qi) {
  // LazyQueueTopScoreDocCollectorManager.java:90-98
  if (q.k <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: numHits must be > 0; got %d", qi, q.k);
  if (q.total_hits_threshold < 0) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: totalHitsThreshold must be >= 0, got %d", qi, q.total_hits_threshold);
  if (q.k > NRTGPU_MAX_K) return fail(NRTGPU_ERR_UNSUPPORTED, "query %d: numHits %d > %d", qi, q.k, NRTGPU_MAX_K);
  if (q.n_terms <= 0 || !q.terms) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: no terms", qi);
  if (q.n_terms > NRTGPU_MAX_TERMS) return fail(NRTGPU_ERR_UNSUPPORTED, "query %d: %d clauses > %d", qi, q.n_terms, NRTGPU_MAX_TERMS);
  if (q.min_should_match < 0) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: minimumNumberShouldMatch %d", qi, q.min_should_match);
  if (q.disjunction_max != 0 && q.disjunction_max != 1) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: disjunction_max must be 0 or 1", qi);
  if (q.disjunction_max == 1 && q.min_should_match > 1)
    return fail(NRTGPU_ERR_UNSUPPORTED, "query %d: a DisjunctionMaxQuery has no minimumNumberShouldMatch", qi);
  if (!(q.tie_breaker >= 0.0f && q.tie_breaker <= 1.0f)) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: tie_breaker must be in [0, 1]", qi);
  if (q.tie_breaker != 0.0f && q.disjunction_max != 1) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: tie_breaker without disjunction_max", qi);
  {
    int n_must = 0;
    for (int t = 0; t < q.n_terms; ++t) {
      if (q.terms[t].occur != 0 && q.terms[t].occur != 1) return fail(NRTGPU_ERR_INVALID_ARG, "query %d term %d: occur must be 0 (SHOULD) or 1 (MUST)", qi, t);
      n_must += q.terms[t].occur;
    }
    if (n_must != 0 && q.disjunction_max == 1) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: the disjuncts of a DisjunctionMaxQuery have no occur", qi);
    if (n_must != 0 && q.min_should_match > 0)   // (Lucene then scores the SHOULD part as one more required scorer: another sum structure)
      return fail(NRTGPU_ERR_UNSUPPORTED, "query %d: MUST clauses next to minimumNumberShouldMatch > 0", qi);
  }
  if (q.n_caches <= 0 || !q.norm_cache) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: norm_cache missing", qi);
  if (q.n_caches > kLdsCaches) return fail(NRTGPU_ERR_UNSUPPORTED, "query %d: %d scored fields > %d", qi, q.n_caches, kLdsCaches);
  for (int t = 0; t < q.n_terms; ++t) {
    if (q.terms[t].cache_slot < 0 || q.terms[t].cache_slot >= q.n_caches)
      return fail(NRTGPU_ERR_INVALID_ARG, "query %d term %d: cache_slot out of range", qi, t);
    if (!(q.terms[t].weight >= 0.0f)) return fail(NRTGPU_ERR_INVALID_ARG, "query %d term %d: weight must be >= 0", qi, t);
  }
  if (!(q.min_competitive_score >= 0.0f)) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: min_competitive_score must be >= 0", qi);
  if (q.filter_mask < 0 || q.must_not_mask < 0) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: mask ids must be >= 0", qi);
  if (q.n_more_filters < 0 || q.n_more_must_not < 0 || (q.n_more_filters > 0 && !q.more_filters) || (q.n_more_must_not > 0 && !q.more_must_not))
    return fail(NRTGPU_ERR_INVALID_ARG, "query %d: more_filters / more_must_not", qi);
  if (q.n_more_filters + (q.filter_mask != 0) > NRTGPU_MAX_MASKS || q.n_more_must_not + (q.must_not_mask != 0) > NRTGPU_MAX_MASKS)
    return fail(NRTGPU_ERR_UNSUPPORTED, "query %d: more than %d FILTER or MUST_NOT clauses", qi, NRTGPU_MAX_MASKS);
  for (int i = 0; i < q.n_more_filters; ++i)
    if (q.more_filters[i] <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: more_filters[%d] must be a mask id > 0", qi, i);
  for (int i = 0; i < q.n_more_must_not; ++i)
    if (q.more_must_not[i] <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "query %d: more_must_not[%d] must be a mask id > 0", qi, i);
  return 0;
}

// What the clauses' occur flags mean for the scorers (nrtgpu.h: nrtgpu_term.occur): every clause MUST = the conjunction = the
// disjunction with minimumNumberShouldMatch = n; MUST next to SHOULD = the second-accumulator shape kMsSecReqOpt.
static int n_must_of(const nrtgpu_bm25_query& q) {
  int n = 0;
  for (int t = 0; t < q.n_terms; ++t) n += q.terms[t].occur;
  return n;
}
static int effective_msm(const nrtgpu_bm25_query& q) { return n_must_of(q) == q.n_terms ? q.n_terms : q.min_should_match; }
static uint32_t sec_mode_of(const nrtgpu_bm25_query& q) {
  if (q.disjunction_max == 1 && q.tie_breaker > 0.0f) return kMsSecTieBreaker;
  const int nm = n_must_of(q);
  return nm > 0 && nm < q.n_terms ? kMsSecReqOpt : kMsSecNone;
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
bool nrtgpu::rt::fixed_scale_of_term(float weight, const float* cache256, uint32_t max_norm, int32_t* scale) {
  const float s_min = nrtgpu::hostmath::bm25_score(weight, 1.0f, cache256[max_norm & 255u]);
  // (the binary exponents straight from the bits -- what ilogb returns for a normal number; this runs once per clause of every
  //  query planned: two libm calls per clause were a fifth of the planner's first pass)
  uint32_t sb, wb;
  memcpy(&sb, &s_min, 4);
  memcpy(&wb, &weight, 4);
  const uint32_t se = (sb >> 23) & 0xFFu, we = (wb >> 23) & 0xFFu;
  if ((sb >> 31) != 0u || (wb >> 31) != 0u || se == 0u || se == 0xFFu || we == 0u || we == 0xFFu) return false;   // not positive and normal
  const int e_min = (int)se - 127, e_w = (int)we - 127;
  if (e_w - e_min > 7) return false;  // 24 mantissa bits + 8 binades of range fill the 32-bit table entry
  *scale = 23 - e_min;
  return *scale > -64 && *scale < 64;
}
struct PlanPiece {
  std::vector<DQTerm> qterms;
  std::vector<float> caches;
  std::vector<QS> qs;          // per (query, leaf) with at least one clause, in query order
  uint32_t n_dterms = 0;
  int64_t postings = 0, cost = 0;
  bool oom = false;
};

// ------------------------------------------------------------------------------------------------
// planner caches
// ------------------------------------------------------------------------------------------------
std::shared_ptr<LeafSetCache> nrtgpu::rt::leaf_set_cache(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, int32_t n_segs) {
  std::lock_guard<std::mutex> lk(ctx->lsc_mu);
  for (size_t i = 0; i < ctx->leaf_sets.size(); ++i) {
    const std::vector<uint64_t>& u = ctx->leaf_sets[i]->uids;
    bool same = u.size() == (size_t)n_segs;
    for (int32_t s = 0; same && s < n_segs; ++s) same = u[(size_t)s] == segs[s]->uid;
    if (same && ctx->leaf_sets[i]->full()) {  // its arena has grown past the cap: start over (plans in flight keep the old one)
      ctx->leaf_sets.erase(ctx->leaf_sets.begin() + (ptrdiff_t)i);
      break;
    }
    if (same) {
      if (i != 0) std::rotate(ctx->leaf_sets.begin(), ctx->leaf_sets.begin() + (ptrdiff_t)i, ctx->leaf_sets.begin() + (ptrdiff_t)i + 1);
      return ctx->leaf_sets[0];
    }
  }
  static std::atomic<uint64_t> next_id{1};
  auto c = std::make_shared<LeafSetCache>();
  c->id = next_id.fetch_add(1, std::memory_order_relaxed);
  c->device = ctx->device;
  c->uids.resize((size_t)n_segs);
  for (int32_t s = 0; s < n_segs; ++s) c->uids[(size_t)s] = segs[s]->uid;
  ctx->leaf_sets.insert(ctx->leaf_sets.begin(), c);
  if (ctx->leaf_sets.size() > 8) ctx->leaf_sets.pop_back();  // (a search still planning with it keeps its reference)
  return c;
}

size_t LeafSetCache::entries() {
  size_t n = 0;
  for (Stripe& st : stripes) {
    std::shared_lock<std::shared_mutex> rd(st.mu);
    n += st.map.size();
  }
  return n;
}

namespace {
// the calling thread's front cache: direct-mapped, tagged with the cache object's id
struct Front { uint64_t owner; int64_t hash; int32_t field; const TermLeaves* tl; };
const size_t kFront = 8192;
Front& front_slot(size_t kh) {
  thread_local std::vector<Front> front(kFront, Front{0, 0, 0, nullptr});
  return front[(kh >> 6) & (kFront - 1)];
}
}  // namespace

const TermLeaves* LeafSetCache::get(const nrtgpu_seg* const* segs, int32_t n_segs, int32_t field, int64_t hash) {
  const Key key{field, hash};
  const TermLeaves* out = nullptr;
  get_many(segs, n_segs, &key, 1, &out);
  return out;
}

// Looks up n terms.  The tables of the terms the cache lacks are built together and reach the device in one copy per
// arena chunk rather than one per term: a cold batch of 1024 five-term queries was 18 ms of 5120 small synchronous
// copies.  The copies complete before the entries become visible, so whoever finds an entry may use its table.
void LeafSetCache::get_many(const nrtgpu_seg* const* segs, int32_t n_segs, const Key* keys, size_t n, const TermLeaves** out) {
  struct Miss { size_t first; std::shared_ptr<TermLeaves> tl; size_t table; };   // first: index of the key's first use
  std::vector<Miss> miss;
  std::unordered_map<Key, size_t, KeyHash> pending;   // missing key -> its Miss (a batch names a term many times)
  std::vector<std::pair<size_t, size_t>> dup;         // (index in keys, Miss) of the repeats
  for (size_t i = 0; i < n; ++i) {
    const size_t kh = KeyHash()(keys[i]);
    Front& f = front_slot(kh);
    if (f.owner == id && f.hash == keys[i].hash && f.field == keys[i].field) {
      out[i] = f.tl;
      continue;
    }
    out[i] = nullptr;
    Stripe& st = stripes[kh % (size_t)kStripes];
    {
      std::shared_lock<std::shared_mutex> rd(st.mu);
      auto it = st.map.find(keys[i]);
      if (it != st.map.end()) {
        f = Front{id, keys[i].hash, keys[i].field, it->second.get()};
        out[i] = f.tl;
        continue;
      }
    }
    auto pit = pending.find(keys[i]);
    if (pit != pending.end()) {
      dup.emplace_back(i, pit->second);
      continue;
    }
    pending.emplace(keys[i], miss.size());
    miss.push_back(Miss{i, nullptr, (size_t)-1});
  }
  if (miss.empty()) return;
  // the missing terms' per-leaf records, one 256-byte aligned table each, back to back as they will lie in the arena
  const size_t stride = ((size_t)n_segs * sizeof(DTerm) + 255) & ~(size_t)255;
  std::vector<char> stage;
  size_t n_tables = 0;
  for (Miss& m : miss) {
    m.tl = std::make_shared<TermLeaves>();
    TermLeaves& tl = *m.tl;
    tl.count.assign((size_t)n_segs, 0u);
    const Key& key = keys[m.first];
    DTerm* leaf = nullptr;
    for (int32_t si = 0; si < n_segs; ++si) {
      auto fit = segs[si]->fields.find(key.field);
      if (fit == segs[si]->fields.end()) continue;
      const FieldData& fd = fit->second;
      const TermEntry* e = fd.flat.find(key.hash);
      if (!e || e->count == 0) continue;
      if (!leaf) {  // (a term no leaf holds needs no table: it never reaches the device)
        m.table = n_tables++;
        stage.resize(n_tables * stride, 0);
        leaf = (DTerm*)(stage.data() + m.table * stride);
      }
      const TermGroup& g = fd.groups[e->group];
      DTerm& d = leaf[si];
      d.docids = g.d_docids;
      d.fnorm = g.d_fnorm;
      d.cell_off = g.d_cells + e->cell_start;
      d.start = e->start;
      d.aux = g.d_aux + e->aux_idx;
      d.shift = e->shift;
      memcpy(&d.weight, &e->count, 4);  // the posting count rides in the weight slot (expand_terms_kernel orders by it)
      tl.count[(size_t)si] = e->count;
      tl.total += e->count;
      tl.max_norm = std::max(tl.max_norm, fd.max_norm);
    }
  }
  // arena space: runs of tables, a run never crosses a chunk; one copy per run
  std::vector<char*> d_of((size_t)n_tables, nullptr);
  bool ok = stride <= kChunkBytes;
  for (size_t t0 = 0; ok && t0 < n_tables;) {
    size_t cnt = 0;
    char* dst = (char*)alloc_tables(stride, n_tables - t0, &cnt);
    if (!dst || cnt == 0 || hipMemcpy(dst, stage.data() + t0 * stride, cnt * stride, hipMemcpyHostToDevice) != hipSuccess) {
      ok = false;
      break;
    }
    for (size_t j = 0; j < cnt; ++j) d_of[t0 + j] = dst + j * stride;
    t0 += cnt;
  }
  for (Miss& m : miss) {
    if (m.table != (size_t)-1) {
      if (ok && d_of[m.table]) m.tl->d_table = (const DTerm*)d_of[m.table];
      else m.tl->total = -1;  // out of device memory: the caller fails the batch (the entry is kept: later batches fail
    }                         // alike until the cache is replaced)
    const Key& key = keys[m.first];
    const size_t kh = KeyHash()(key);
    Stripe& st = stripes[kh % (size_t)kStripes];
    const TermLeaves* got;
    {
      std::unique_lock<std::shared_mutex> wr(st.mu);
      got = st.map.emplace(key, m.tl).first->second.get();   // (a racing thread may have inserted the term meanwhile: its entry stays)
    }
    front_slot(kh) = Front{id, key.hash, key.field, got};
    out[m.first] = got;
  }
  for (const auto& d : dup) out[d.first] = out[miss[d.second].first];
}

// Room for up to `want` tables of `stride` bytes in the current arena chunk (a new one when it is full): returns the
// first and how many fit.
void* LeafSetCache::alloc_tables(size_t stride, size_t want, size_t* got) {
  std::lock_guard<std::mutex> lk(arena_mu);
  *got = 0;
  if (stride > kChunkBytes || want == 0) return nullptr;
  if (chunks.empty() || chunk_used + stride > kChunkBytes) {
    void* p = nullptr;
    (void)hipSetDevice(device);
    if (hipMalloc(&p, kChunkBytes) != hipSuccess) return nullptr;
    chunks.push_back(p);
    chunk_used = 0;
  }
  const size_t fit = std::min(want, (kChunkBytes - chunk_used) / stride);
  void* r = (char*)chunks.back() + chunk_used;
  chunk_used += fit * stride;
  *got = fit;
  return r;
}

LeafSetCache::~LeafSetCache() {
  (void)hipSetDevice(device);
  for (void* p : chunks) (void)hipFree(p);
}

// Pass 1 of the planner for queries [q_begin, q_end): one cache lookup per clause (LeafSetCache); score
// tables go to the clauses with the most postings; terms of a (query, leaf) sorted densest first.
// Offsets (term_begin, cache offsets) are relative to the piece.
static void resolve_queries(LeafSetCache& lsc, const nrtgpu_seg* const* segs, int32_t n_segs, const int32_t* n_deleted,
                            const int32_t* slice_of_leaf, int32_t n_slices, const nrtgpu_bm25_query* queries, int q_begin, int q_end, PlanPiece& pc, uint32_t* q_qs_begin,
                            uint32_t* q_qs_cnt, uint32_t* qs_begin, DQExpand* qexpand,
                            std::vector<uint32_t>& cache_base, std::vector<QTabs>& qtabs, int prune,
                            std::vector<int64_t>& q_lower, std::vector<uint8_t>& q_route, std::vector<int64_t>& q_ms_key) {
  // (per query, on the stack: validate_query bounds n_terms by NRTGPU_MAX_TERMS)
  int64_t term_total[NRTGPU_MAX_TERMS];
  int32_t tab_of_term[NRTGPU_MAX_TERMS], term_scale[NRTGPU_MAX_TERMS];
  const TermLeaves* ents[NRTGPU_MAX_TERMS];
  std::vector<int64_t> slice_sum((size_t)std::max(n_slices, 1));
  bool any_deleted = false;
  for (int si = 0; si < n_segs; ++si) any_deleted = any_deleted || n_deleted[si] != 0;
  // (liveDocs that are not folded into the postings -- packed layout, forked reader versions, NRTGPU_FLAG_NO_LIVE_FOLD --
  //  are a mask the MaxScore kernel tests when a doc's score is complete: no obstacle to the route)
  size_t prev_cache_off = 0, prev_cache_len = 0;
  pc.qterms.reserve((size_t)(q_end - q_begin) * 6);
  pc.qs.reserve((size_t)(q_end - q_begin) * (size_t)std::max(n_segs, 1));
  // every clause of the range in one lookup (the cache fetches what it lacks in one go)
  std::vector<LeafSetCache::Key> all_keys;
  std::vector<const TermLeaves*> all_ents;
  all_keys.reserve((size_t)(q_end - q_begin) * 6);
  for (int qi = q_begin; qi < q_end; ++qi)
    for (int t = 0; t < queries[qi].n_terms; ++t) all_keys.push_back(LeafSetCache::Key{queries[qi].terms[t].field_id, queries[qi].terms[t].term_hash});
  all_ents.assign(all_keys.size(), nullptr);
  lsc.get_many(segs, n_segs, all_keys.data(), all_keys.size(), all_ents.data());
  size_t ent_at = 0;
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
    for (int t = 0; t < q.n_terms; ++t) {
      ents[(size_t)t] = all_ents[ent_at++];
      if (ents[(size_t)t]->total < 0) pc.oom = true;
      term_total[(size_t)t] = std::max<int64_t>(ents[(size_t)t]->total, 0);
    }
    // fixed-point analysis: per clause the scale of its scores, per query the common scale
    for (int t = 0; t < q.n_terms; ++t) term_scale[t] = 0;
    bool fx_ok = true;
    int32_t fx_E = kNoFixed;
    for (int t = 0; t < q.n_terms && fx_ok; ++t) {
      if (term_total[(size_t)t] == 0) continue;  // matches nothing here
      const uint32_t max_norm = ents[(size_t)t]->max_norm;
      int32_t sc = 0;
      fx_ok = fixed_scale_of_term(q.terms[t].weight, q.norm_cache + (size_t)q.terms[t].cache_slot * 256, max_norm, &sc);
      term_scale[(size_t)t] = sc;
      if (fx_ok) fx_E = std::max(fx_E, sc);
    }
    for (int t = 0; t < q.n_terms && fx_ok; ++t)  // 32-bit entries shifted into the common scale, summed over
      if (term_total[(size_t)t] != 0 && fx_E - term_scale[(size_t)t] > 15) fx_ok = false;  // <= 32 clauses: < 2^53
    if (!fx_ok) fx_E = kNoFixed;
    // MaxScore route (maxscore.hip)?  Any disjunction of up to kMsMaxTerms clauses with the exact fixed-point sums -- plain,
    // with a FILTER / MUST_NOT doc set, with minimumNumberShouldMatch, or a DisjunctionMaxQuery -- unless a bound from
    // outside (min_competitive_score) makes the count's meaning the caller's business.  The mode says when bounds may skip
    // (plan.h: kMsMode*): like Lucene, which starts skipping only once a collector's totalHits has passed the threshold
    // (LazyQueueTopScoreDocCollector.java:176-199).
    //   * the planner KNOWS that some slice passes max(totalHitsThreshold, numHits) -- some clause has that many postings
    //     left in it after discounting every deleted doc of the slice's leaves, and nothing else narrows the hits: skipping
    //     from the start, the certain lower bound is what is reported;
    //   * it cannot know (few postings, a mask, a clause count): the kernel counts exactly until a slice passes;
    //   * ScoreMode.COMPLETE: nothing may ever be skipped -- worth it only for a small query, for which the exhaustive scan
    //     would walk every sub-tile of the shard for a handful of postings.
    int64_t lower = 0;
    uint8_t route = kRouteScan;
    if (prune != 0 && fx_ok && q.n_terms <= kMsMaxTerms && !(q.min_competitive_score > 0.0f)) {
      const bool shaped = effective_msm(q) > 1 || q.filter_mask != 0 || q.must_not_mask != 0 || q.n_more_filters != 0 || q.n_more_must_not != 0 ||
                          sec_mode_of(q) == kMsSecReqOpt;   // (how many docs hold every MUST clause: nothing the posting counts say)
      if (q.total_hits_threshold == INT32_MAX) {
        int64_t all = 0;
        for (int t = 0; t < q.n_terms; ++t) all += term_total[(size_t)t];
        if (all > 0 && all <= kMsExactMaxPostings) route = kRouteMs + (uint8_t)kMsModeExact;
      } else {
        if (!shaped) {
          // the reference counts per slice (one collector each): some slice must certainly pass the threshold
          const int64_t floor_ = std::max<int64_t>(q.total_hits_threshold, q.k);
          int64_t best_slice = 0;
          for (int t = 0; t < q.n_terms; ++t) {
            const uint32_t* cnt = ents[(size_t)t]->count.data();
            int64_t certain = 0;
            if (n_slices <= 1) {
              certain = term_total[(size_t)t];
              if (any_deleted) {
                certain = 0;
                for (int si = 0; si < n_segs; ++si) certain += std::max<int64_t>(0, (int64_t)cnt[si] - n_deleted[si]);
              }
              best_slice = std::max(best_slice, certain);
            } else {
              std::fill(slice_sum.begin(), slice_sum.end(), 0);
              for (int si = 0; si < n_segs; ++si) {
                const int64_t c = std::max<int64_t>(0, (int64_t)cnt[si] - n_deleted[si]);
                slice_sum[(size_t)slice_of_leaf[si]] += c;
                certain += c;
              }
              for (int sl = 0; sl < n_slices; ++sl) best_slice = std::max(best_slice, slice_sum[(size_t)sl]);
            }
            lower = std::max(lower, certain);   // what is reported: certain matches of the whole search
          }
          if (best_slice <= floor_) lower = 0;
        }
        route = kRouteMs + (uint8_t)(lower > 0 ? kMsModePrune : kMsModeCount);
      }
    }
    q_lower[(size_t)qi] = lower;
    q_route[(size_t)qi] = route;
    {
      // What a MaxScore item costs follows the docs it EVALUATES, not the postings of its clauses: the dense clauses of a query
      // are non-essential almost from the start and never streamed.  Over 156 C3 queries the walk model's cost correlates +0.26
      // with all postings and +0.84 with the postings of the two heaviest clauses (scripts/cpu_launch_order_sim.py): the key
      // build_plan orders that route's items by.
      int64_t p0 = 0, p1 = 0;
      float w0 = -1.0f, w1 = -1.0f;
      for (int t = 0; t < q.n_terms; ++t) {
        if (term_total[(size_t)t] == 0) continue;
        const float wt = q.terms[t].weight;
        if (wt > w0) { w1 = w0; p1 = p0; w0 = wt; p0 = term_total[(size_t)t]; }
        else if (wt > w1) { w1 = wt; p1 = term_total[(size_t)t]; }
      }
      q_ms_key[(size_t)qi] = p0 + p1;
    }
    for (int t = 0; t < q.n_terms; ++t) tab_of_term[t] = -1;
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
    // the compact plan of the query: one DQTerm per clause that matches anything, and per leaf how many of them it holds
    DQExpand& qx = qexpand[(size_t)qi];
    qx.term_begin = (uint32_t)pc.qterms.size();
    qx.by_weight = route != kRouteScan ? 1u : 0u;
    qx.pad = 0;
    const uint32_t* cnt_of[kMaxTerms];
    bool req_of[kMaxTerms];
    uint32_t n_live = 0;
    const bool req_opt = sec_mode_of(q) == kMsSecReqOpt;
    bool no_hits = false;   // a MUST clause that matches nothing anywhere
    for (int t = 0; t < q.n_terms; ++t) {
      const TermLeaves& tl = *ents[(size_t)t];
      if (tl.total <= 0) {
        no_hits = no_hits || (req_opt && q.terms[t].occur == 1);
        continue;
      }
      req_of[n_live] = req_opt && q.terms[t].occur == 1;
      DQTerm d{};
      d.table = tl.d_table;
      d.weight = q.terms[t].weight;
      d.cache_slot = (uint32_t)q.terms[t].cache_slot;
      d.tab_slot = (tab_of_term[(size_t)t] >= 0 ? (uint32_t)tab_of_term[(size_t)t] : kTabSlotNone) |
                   (sec_mode_of(q) == kMsSecReqOpt && q.terms[t].occur == 1 ? kTabSlotRequired : 0u);
      d.fx_scale = term_scale[(size_t)t];
      d.fx_shift = fx_ok ? (uint32_t)(fx_E - term_scale[(size_t)t]) : 0u;
      pc.qterms.push_back(d);
      cnt_of[n_live++] = tl.count.data();
    }
    qx.n_terms = n_live;
    q_qs_begin[qi] = (uint32_t)pc.qs.size();
    for (int si = 0; si < n_segs; ++si) {
      uint32_t n = 0;
      int64_t postings = 0;
      bool lacks_must = no_hits;   // (a leaf without one of the MUST terms has no hits: it is not planned at all)
      for (uint32_t t = 0; t < n_live; ++t) {
        const uint32_t c = cnt_of[t][si];
        n += c != 0u ? 1u : 0u;
        postings += c;
        lacks_must = lacks_must || (req_of[t] && c == 0u);
      }
      if (lacks_must) n = 0;
      qs_begin[(size_t)qi * (size_t)n_segs + (size_t)si] = n ? pc.n_dterms : 0xFFFFFFFFu;
      if (n == 0) continue;
      pc.qs.push_back(QS{pc.n_dterms, n, si, postings});
      pc.n_dterms += n;
      pc.postings += postings;
      pc.cost += postings + (int64_t)segs[si]->n_tiles * kTileCostPostings;
    }
    q_qs_cnt[qi] = (uint32_t)pc.qs.size() - q_qs_begin[qi];
  }
}

int nrtgpu::rt::build_plan(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                      const nrtgpu_bm25_query* queries, int32_t n_queries, HostPlan& hp, int prune) {
  const double t_entry = now_ms();
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

  static const bool plan_trace = dev_env_int("NRTGPU_PLAN_TRACE", 0) != 0;  // debug aid: phase times on stderr
  const double tp0 = plan_trace ? now_ms() : 0.0;
  // pass 1: resolve terms per (query, segment), densest term first; remember posting counts.
  // Queries are independent here, so the batch is cut into contiguous chunks resolved by
  // cfg.host_threads planner threads and concatenated (offsets rebased) afterwards.
  std::vector<uint32_t> q_qs_begin((size_t)n_queries), q_qs_cnt((size_t)n_queries);
  std::vector<uint32_t> cache_base((size_t)n_queries);
  std::vector<QTabs> qtabs((size_t)n_queries);
  int n_thr = ctx->cfg.host_threads > 0 ? ctx->cfg.host_threads : 4;
  n_thr = std::max(1, std::min(n_thr, n_queries / 64));
  // A helper thread has to be woken (tens of microseconds, twice per plan) and the caller then waits for the slowest of them: over
  // few (query, leaf) pairs that costs what it brings.  Measured on the GPU boxes' hosts without the GPU
  // (profiles/r05_planner_host_ab.log, scripts/cpu_plan_bench.py): 1024 C3 queries over ONE leaf (the shard of one rank of eight)
  // 1 thread 0.25-0.30 ms per plan, 2 threads 0.20-0.33: below 2048 pairs the caller plans alone.  Over the whole index's 10
  // leaves the plan alone is no faster with helpers either (1 thread 0.60-0.74 ms, 2 / 4 threads 0.80-0.87) -- but a pipeline
  // that starts empty waits for its FIRST plan, and the driver's 20-step form shows it: planned by the caller alone 487-493 k
  // queries/s (profiles/r05_bench_steps20_*.json of r05p), with the configured helpers 498-505 k (r05g, r05j).  The helpers stay.
  static const long alone_below = dev_env_int("NRTGPU_PLAN_ALONE_PAIRS", 2048);   // (development build: A/B)
  if ((int64_t)n_queries * std::max(n_segs, 1) < alone_below) n_thr = 1;
  // More chunks than threads, handed out by a counter (WorkPool::run; the caller works too): a helper thread that wakes late --
  // tens of microseconds on a busy or virtualised host, of a phase that takes a few hundred -- then costs the batch one small
  // chunk, not its whole share.  Measured without a GPU (scripts/cpu_plan_bench.py, 1024 C3 queries over one leaf): one share
  // per thread 0.65 ms per plan with 2 threads against 0.45 ms with 1; the pieces are concatenated in chunk order either way,
  // so the plan does not depend on who resolved what.
  const int n_chunks = n_thr == 1 ? 1 : std::max(n_thr, std::min(n_queries / 64, 4 * n_thr));
  std::vector<PlanPiece> pieces((size_t)n_chunks);
  auto chunk_begin = [&](int t) { return (int)((int64_t)n_queries * t / n_chunks); };
  {
    const int variant = (ctx->cfg.flags >> 8) & 15;  // (the timing ablations of the scan stay exhaustive; 7 = instrumented kernels)
    // (the MaxScore route adds exact fixed-point integers: a context that asks for fp64 sums gets the exhaustive scan)
    if ((ctx->cfg.flags & (NRTGPU_FLAG_NO_PRUNE | NRTGPU_FLAG_NO_FIXED_POINT)) != 0 || !(variant == 0 || variant == 7)) prune = 0;
  }
  hp.q_lower.assign((size_t)n_queries, 0);
  hp.q_route.assign((size_t)n_queries, kRouteScan);
  std::vector<int64_t> q_ms_key((size_t)n_queries, 0);   // (resolve_queries: launch-order key of the query's MaxScore items)
  hp.lsc = leaf_set_cache(ctx, segs, n_segs);
  hp.n_leaves = (uint32_t)n_segs;
  hp.qexpand.resize((size_t)n_queries);
  hp.qs_begin.assign((size_t)n_queries * (size_t)std::max(n_segs, 1), 0xFFFFFFFFu);
  std::vector<int32_t> n_deleted((size_t)std::max(n_segs, 1), 0);
  for (int si = 0; si < n_segs; ++si) n_deleted[(size_t)si] = segs[si]->n_deleted;
  // the searcher's slices over these leaves (MyIndexSearcher.slices / slicesForShards): relation and route depend on them
  std::vector<int32_t> slice_of_leaf((size_t)std::max(n_segs, 1), 0);
  int32_t n_slices = 1;
  if (n_segs > 0 && (int32_t)g_thread_slices.size() == n_segs) {
    // the caller's slices (nrtgpu_set_thread_slices: a call over a subset of the searcher's leaves counts by the WHOLE searcher's
    // slices), renumbered densely in order of first appearance
    std::unordered_map<int32_t, int32_t> dense;
    for (int32_t i = 0; i < n_segs; ++i) {
      auto it = dense.emplace(g_thread_slices[(size_t)i], (int32_t)dense.size()).first;
      slice_of_leaf[(size_t)i] = it->second;
    }
    n_slices = (int32_t)std::max<size_t>(dense.size(), 1);
  } else if (ctx->slice_max_docs.load() > 0 && n_segs > 0) {
    std::vector<hostmath::LeafInfo> all((size_t)n_segs);
    int32_t base = 0;
    for (int32_t i = 0; i < n_segs; ++i) {
      all[(size_t)i] = {i, segs[i]->max_doc, segs[i]->max_doc - segs[i]->n_deleted, doc_bases ? doc_bases[i] : base};
      base += segs[i]->max_doc;
    }
    const int32_t vs = ctx->virtual_shards.load();
    const std::vector<std::vector<int32_t>> sl = vs > 1
        ? hostmath::slices_for_shards(all, vs, ctx->slice_max_docs.load(), ctx->slice_max_segments.load(), nullptr)
        : hostmath::slices(all, ctx->slice_max_docs.load(), ctx->slice_max_segments.load(), all);
    n_slices = (int32_t)std::max<size_t>(sl.size(), 1);
    for (size_t s_ = 0; s_ < sl.size(); ++s_)
      for (int32_t li : sl[s_]) slice_of_leaf[(size_t)li] = (int32_t)s_;
  }
  auto work = [&](int t) {
    resolve_queries(*hp.lsc, segs, n_segs, n_deleted.data(), slice_of_leaf.data(), n_slices, queries, chunk_begin(t), chunk_begin(t + 1), pieces[(size_t)t],
                    q_qs_begin.data(), q_qs_cnt.data(), hp.qs_begin.data(), hp.qexpand.data(), cache_base, qtabs, prune, hp.q_lower, hp.q_route, q_ms_key);
  };
  ctx->pool->run(n_chunks, work);
  const double tp1 = plan_trace ? now_ms() : 0.0;
  // concatenate the pieces: offsets were relative to the piece
  int64_t total_postings = 0;
  {
    size_t nt = 0, nc = 0;
    for (const PlanPiece& pc : pieces) { nt += pc.qterms.size(); nc += pc.caches.size(); }
    hp.qterms.reserve(nt);
    hp.caches.reserve(nc);
  }
  std::vector<int> piece_of((size_t)n_queries, 0);
  for (int t = 0; t < n_chunks; ++t) {
    PlanPiece& pc = pieces[(size_t)t];
    if (pc.oom) return fail(NRTGPU_ERR_OOM, "out of device memory for the resident term tables");
    const uint32_t qterm_base = (uint32_t)hp.qterms.size(), c_base = (uint32_t)hp.caches.size(), dterm_base = hp.n_dterms;
    hp.qterms.insert(hp.qterms.end(), pc.qterms.begin(), pc.qterms.end());
    hp.caches.insert(hp.caches.end(), pc.caches.begin(), pc.caches.end());
    for (QS& qs : pc.qs) qs.term_begin += dterm_base;
    for (int qi = chunk_begin(t); qi < chunk_begin(t + 1); ++qi) {
      cache_base[(size_t)qi] += c_base;
      hp.qexpand[(size_t)qi].term_begin += qterm_base;
      piece_of[(size_t)qi] = t;
      if (dterm_base)
        for (int si = 0; si < n_segs; ++si) {
          uint32_t& b = hp.qs_begin[(size_t)qi * (size_t)n_segs + (size_t)si];
          if (b != 0xFFFFFFFFu) b += dterm_base;
        }
    }
    hp.n_dterms += pc.n_dterms;
    total_postings += pc.postings;
  }
  // the (query, leaf) pairs of query qi
  auto qs_of = [&](int qi) { return pieces[(size_t)piece_of[(size_t)qi]].qs.data() + q_qs_begin[(size_t)qi]; };
  hp.postings = total_postings;
  auto on_ms_kernel = [&](uint32_t q_) { return hp.q_route[q_] != kRouteScan; };
  // (the accumulator form is a property of the exhaustive scan's launch; the MaxScore route only takes fixed-point queries)
  hp.fixed_point = (ctx->cfg.flags & NRTGPU_FLAG_NO_FIXED_POINT) == 0;
  for (int qi = 0; qi < n_queries && hp.fixed_point; ++qi)
    if (!on_ms_kernel((uint32_t)qi) && q_qs_cnt[(size_t)qi] != 0 && qtabs[(size_t)qi].fx_E == kNoFixed) hp.fixed_point = false;
  for (int qi = 0; qi < n_queries; ++qi)
    if (on_ms_kernel((uint32_t)qi))
      for (uint32_t j = 0; j < q_qs_cnt[(size_t)qi]; ++j) hp.ms_postings += qs_of(qi)[j].postings;
  // Exhaustive scan, minimumNumberShouldMatch > 1 (QueryNodeMapper.java:259-261) / DisjunctionMaxQuery: the clause count rides
  // in the fixed-point accumulator, so the scan's whole launch must be in fixed-point mode; otherwise the caller runs Lucene
  hp.clause_counting = false;
  hp.ms_shapes = false;
  hp.ms_two = false;
  for (int qi = 0; qi < n_queries; ++qi) {
    const nrtgpu_bm25_query& q = queries[qi];
    if (effective_msm(q) > 1 || q.disjunction_max == 1) (on_ms_kernel((uint32_t)qi) ? hp.ms_shapes : hp.clause_counting) = true;
    if (sec_mode_of(q) != kMsSecNone) {
      // a second accumulator per doc: the MaxScore kernel's SHAPES == 2 instantiation carries one, the exhaustive scan's LDS
      // accumulators do not
      if (!on_ms_kernel((uint32_t)qi) && q_qs_cnt[(size_t)qi] != 0)
        return fail(NRTGPU_ERR_UNSUPPORTED, "query %d: a tie breaker > 0 / MUST next to SHOULD clauses run on the MaxScore route only (<= %d clauses, "
                                            "fixed-point sums, no min_competitive_score, not ScoreMode.COMPLETE over more than %lld postings)",
                    qi, kMsMaxTerms, (long long)kMsExactMaxPostings);
      hp.ms_two = true;
    }
    if ((q.filter_mask != 0 || q.must_not_mask != 0 || q.n_more_filters != 0 || q.n_more_must_not != 0) && on_ms_kernel((uint32_t)qi)) hp.ms_shapes = true;
  }
  hp.n_slices = (uint32_t)n_slices;
  if (hp.clause_counting && !hp.fixed_point)
    return fail(NRTGPU_ERR_UNSUPPORTED, "minimumNumberShouldMatch > 1 / DisjunctionMaxQuery need the fixed-point accumulators (weights of "
                                        "a query in this batch span too many binades, or NRTGPU_FLAG_NO_FIXED_POINT is set)");

  const double tp2 = plan_trace ? now_ms() : 0.0;
  // pass 2: cut every query's leaves (in docBase order) into items of roughly equal cost.  An item
  // may span several segments (like a LeafSlice) and a large segment may be cut by tile range.
  // Measured on MI355X (one workgroup per CU): every extra item of a query costs a cold
  // top-k start, so a query is cut only when it alone would take longer than its fair share of the
  // batch on one CU.  target_items == 0 => one share per CU.
  const int64_t target_items = ctx->cfg.target_items > 0 ? ctx->cfg.target_items : (int64_t)std::max(ctx->n_cus, 1);
  const int64_t min_item_cost = 1 << 17;
  // An item counts its hits per searcher slice (plan.h: kSliceSlots): at most that many distinct slices per item; the
  // (query, leaf) pairs of a query are visited slice by slice so that a slice's parts are neighbours.
  struct Pending { int64_t cost; uint32_t query; uint32_t part_begin, n_parts; uint32_t tiles; int32_t first_slice, last_slice; uint32_t n_slices; };
  std::vector<Pending> pend;
  std::vector<int64_t> q_costs((size_t)n_queries, 0), q_items((size_t)n_queries, 0);
  for (int qi = 0; qi < n_queries; ++qi)
    for (uint32_t j = 0; j < q_qs_cnt[(size_t)qi]; ++j) {
      const QS& qs = qs_of(qi)[j];
      q_costs[(size_t)qi] += qs.postings + (int64_t)segs[qs.seg]->n_tiles * kTileCostPostings;
    }
  hostmath::plan_item_counts(q_costs.data(), n_queries, target_items, min_item_cost, q_items.data());  // host_math.h
  // One contiguous range of queries per worker, each into its own part / item vectors; concatenated in range order
  // below, so the plan is the one a single thread would have produced.
  struct CutPiece { std::vector<DPart> parts; std::vector<Pending> pend; bool masked = false; int rc = 0; };
  std::vector<CutPiece> cuts((size_t)n_chunks);
  auto cut_work = [&](int t) {
    const int q0 = chunk_begin(t), q1 = chunk_begin(t + 1);
    std::vector<DPart>& parts = cuts[(size_t)t].parts;
    std::vector<Pending>& pend = cuts[(size_t)t].pend;
    bool& masked = cuts[(size_t)t].masked;
    int& cut_rc = cuts[(size_t)t].rc;
    std::vector<QS> by_slice;
    parts.reserve((size_t)(q1 - q0) * (size_t)std::max(n_segs, 1));
    pend.reserve((size_t)(q1 - q0) * 2);
    for (int qi = q0; qi < q1; ++qi) {
      const int64_t q_cost = q_costs[(size_t)qi];
      if (q_cost == 0) continue;
      const int64_t n_it = q_items[(size_t)qi];
      const double budget = (double)q_cost / (double)n_it;
      const QS* qsv = qs_of(qi);
      const uint32_t n_qs = q_qs_cnt[(size_t)qi];
      if (n_slices > 1) {
        by_slice.assign(qsv, qsv + n_qs);
        std::stable_sort(by_slice.begin(), by_slice.end(), [&](const QS& a, const QS& b) { return slice_of_leaf[(size_t)a.seg] < slice_of_leaf[(size_t)b.seg]; });
        qsv = by_slice.data();
      }
      const bool scan_route = hp.q_route[(size_t)qi] == kRouteScan;
      Pending cur{0, (uint32_t)qi, (uint32_t)parts.size(), 0, 0, -1, -1, 0};
      double filled = 0.0;
      auto close_item = [&]() {
        pend.push_back(cur);
        cur = Pending{0, (uint32_t)qi, (uint32_t)parts.size(), 0, 0, -1, -1, 0};
        filled = 0.0;
      };
      for (uint32_t j = 0; j < n_qs; ++j) {
        const QS& qs = qsv[j];
        const nrtgpu_seg* seg = segs[qs.seg];
        const int32_t sl = slice_of_leaf[(size_t)qs.seg];
        if (cur.n_parts > 0 && sl != cur.last_slice && cur.n_slices == (uint32_t)kSliceSlots) close_item();  // no slot left for another slice
        const double tile_cost = (double)qs.postings / (double)seg->n_tiles + (double)kTileCostPostings;
        const uint64_t* accept = nullptr;  // liveDocs, narrowed by the query's FILTER / MUST_NOT masks
        if (int rc = accept_set_of(seg, queries[qi], &accept)) { cut_rc = rc; return; }
        uint32_t tb = 0;
        while (tb < seg->n_tiles) {
          double room = budget - filled;
          uint32_t take = (uint32_t)std::max(1.0, std::floor(room / tile_cost + 0.5));
          take = std::min<uint32_t>(take, seg->n_tiles - tb);
          if (cur.n_parts == 0) cur.first_slice = sl;
          if (sl != cur.last_slice) { cur.n_slices++; cur.last_slice = sl; }
          DPart p{};
          p.live_bits = accept;
          if (accept && scan_route) masked = true;
          p.term_begin = qs.term_begin;
          p.n_terms = qs.n_terms;
          p.tile_begin = tb;
          p.tile_end = tb + take;
          p.max_doc = (uint32_t)seg->max_doc;
          p.doc_base = doc_bases ? doc_bases[qs.seg] : 0;
          p.tile_offset = cur.tiles;
          p.slice = (uint32_t)sl;   // (the slot is assigned once the items are final, below)
          parts.push_back(p);
          cur.n_parts++;
          cur.tiles += take;
          cur.cost += (int64_t)(take * tile_cost);
          filled += take * tile_cost;
          tb += take;
          if (filled >= budget * 0.999) close_item();  // item full
        }
      }
      if (cur.n_parts > 0) {
        // A short remainder (the tile rounding of the items before it) does not become an item of its own: it would
        // finish without a single compaction, never publish its quantile, and with one peer silent the bound
        // exchange between the query's items never forms (kernels.hip: peers_bound).  It joins the item before it.
        if (!pend.empty() && pend.back().query == (uint32_t)qi && (double)cur.cost < 0.5 * budget &&
            pend.back().part_begin + pend.back().n_parts == cur.part_begin &&
            pend.back().n_slices + cur.n_slices - (pend.back().last_slice == cur.first_slice ? 1u : 0u) <= (uint32_t)kSliceSlots) {
          Pending& prev = pend.back();
          for (uint32_t pi2 = 0; pi2 < cur.n_parts; ++pi2) parts[cur.part_begin + pi2].tile_offset += prev.tiles;
          prev.n_slices += cur.n_slices - (prev.last_slice == cur.first_slice ? 1u : 0u);
          prev.last_slice = cur.last_slice;
          prev.n_parts += cur.n_parts;
          prev.tiles += cur.tiles;
          prev.cost += cur.cost;
        } else {
          pend.push_back(cur);
        }
      }
    }
    // the slot of every part: the running number of its slice among the item's slices (a slice's parts are neighbours)
    for (const Pending& a : pend) {
      uint32_t slot = 0;
      for (uint32_t pi2 = 0; pi2 < a.n_parts; ++pi2) {
        DPart& p = parts[a.part_begin + pi2];
        if (pi2 > 0 && p.slice != (parts[a.part_begin + pi2 - 1].slice & 0xFFFFFFu)) ++slot;
        p.slice |= slot << 24;
      }
    }
  };
  ctx->pool->run(n_chunks, cut_work);
  {
    size_t np = 0, ni = 0;
    for (const CutPiece& c : cuts) { np += c.parts.size(); ni += c.pend.size(); }
    hp.parts.reserve(hp.parts.size() + np);
    pend.reserve(ni);
    for (CutPiece& c : cuts) {
      if (c.rc) return c.rc;
      const uint32_t base = (uint32_t)hp.parts.size();
      hp.parts.insert(hp.parts.end(), c.parts.begin(), c.parts.end());
      for (Pending& a : c.pend) { a.part_begin += base; pend.push_back(a); }
      if (c.masked) hp.masked = true;
    }
  }
  // longest-processing-time-first launch order: the hardware dispatcher hands out workgroups in
  // index order, so big items start first and small ones fill the tail
  // (cf. slices ordered largest first, MyIndexSearcher.java:154-158)
  {
    // The MaxScore items' longest-first key.  Round 3 closed with "the postings of the query's two heaviest clauses" as the
    // default on a CPU model's word (makespan 1.56 -> 1.33 x the balanced load) and no measurement.  Measured in round 4, same box,
    // interleaved (profiles/r04_helpers_v1_ab.log): the two-clause key is SLOWER than all postings -- 3.00 vs 2.85 ms per 1024 C3
    // queries without helper workgroups, 2.62 vs 2.51 with them.  The default is all postings again; NRTGPU_MS_LPT=1 keeps the
    // two-clause key for A/B.  Results do not depend on the launch order (an item's output slot is assigned after the sort).
    static const bool ms_lpt = dev_env_int("NRTGPU_MS_LPT", 0) != 0;
    if (ms_lpt)
      for (Pending& a : pend)
        if (on_ms_kernel(a.query) && q_costs[(size_t)a.query] > 0)   // an item's share of its query's key
          a.cost = 1 + (int64_t)((double)q_ms_key[(size_t)a.query] * ((double)a.cost / (double)q_costs[(size_t)a.query]));
  }
  std::stable_sort(pend.begin(), pend.end(), [](const Pending& a, const Pending& b) { return a.cost > b.cost; });
  // the items of the queries on the MaxScore route first: they run in a launch of their own
  std::stable_partition(pend.begin(), pend.end(), [&](const Pending& a) { return on_ms_kernel(a.query); });
  hp.n_ms_items = 0;
  for (const Pending& a : pend) hp.n_ms_items += on_ms_kernel(a.query) ? 1u : 0u;
  // Finer doc windows for the slowest queries of a launch (maxscore.hip: a window is what ONE wave walks alone -- the grain at
  // which an item's owner and its helpers share work; a wave holds its current window and has its next one reserved).  Measured
  // under speculative thresholds (scripts/gpu_tail_items.py, profiles/r04_tail_items.log): the launch's slowest item -- five
  // frequent terms, nothing to prune by -- needs 154 us per 64-tile window and wave against a 2.0 ms launch; 24 of its 159
  // windows are in its own waves' hands at any time, so helpers never find the 16 unassigned ones they ask for, and it ends
  // 170 us after 99 % of the items: the launch's whole tail.  Such queries get windows of a quarter the size (plan.h:
  // DItem.flags bits 2-3); the others keep theirs -- a window's head (cell lookups, bitset clear, clause ranges) is 7 % of an
  // average item's time.  Which queries: by the postings of their two heaviest clauses -- what the walk must evaluate, +0.84
  // with an item's cost where all postings have +0.26 (scripts/cpu_launch_order_sim.py) -- one MaxScore query in
  // NRTGPU_MS_FINE_ITEMS (default 32; 0: none; profiles/r04_fine_windows_ab.log).
  std::vector<uint8_t> q_fine((size_t)n_queries, 0);
  {
    static const int env_fine = (int)dev_env_int("NRTGPU_MS_FINE_ITEMS", 32);
    static const int env_shift = std::min(std::max((int)dev_env_int("NRTGPU_MS_FINE_SHIFT", 2), 1), 3);
    if (env_fine > 0 && hp.n_ms_items >= 64) {
      std::vector<std::pair<int64_t, int32_t>> by_key;
      for (int qi = 0; qi < n_queries; ++qi)
        if (on_ms_kernel((uint32_t)qi) && q_qs_cnt[(size_t)qi] != 0) by_key.emplace_back(-q_ms_key[(size_t)qi], qi);
      const size_t n_fine = std::min(by_key.size(), std::max<size_t>(1, by_key.size() / (size_t)env_fine));
      std::partial_sort(by_key.begin(), by_key.begin() + (ptrdiff_t)n_fine, by_key.end());
      for (size_t i = 0; i < n_fine; ++i) q_fine[(size_t)by_key[i].second] = (uint8_t)env_shift;
    }
  }
  hp.items.resize(pend.size());
  // the items of a query, in launch order: counting sort by query (q_base = first slot of the query's list)
  hp.q_base.assign((size_t)n_queries, 0);
  hp.q_nlists.assign((size_t)n_queries, 0);
  for (const Pending& a : pend) hp.q_nlists[a.query]++;
  {
    uint32_t run = 0;
    for (int qi = 0; qi < n_queries; ++qi) { hp.q_base[(size_t)qi] = run; run += hp.q_nlists[(size_t)qi]; }
  }
  hp.list_idx.assign(pend.size(), 0);
  hp.q_wins.assign((size_t)n_queries, 0u);
  std::vector<uint32_t> q_fill((size_t)n_queries, 0);
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
    it.peer_slot = hp.q_base[pend[i].query] + q_fill[pend[i].query]++;
    it.flags = on_ms_kernel(pend[i].query) ? (uint32_t)(hp.q_route[pend[i].query] - kRouteMs) : 0u;
    if (on_ms_kernel(pend[i].query)) {
      // its doc windows (maxscore.hip: windows start on kMsWinTiles boundaries of the segment, a part's first one may be short):
      // what the owner's waves and the item's helpers hand out from one counter (plan.h: DHelp)
      uint32_t wins = 0;
      const uint32_t fine = q_fine[pend[i].query], wt = (uint32_t)kMsWinTiles >> fine;   // (every item of a query: one window size)
      for (uint32_t pi2 = 0; pi2 < it.n_parts; ++pi2) {
        const DPart& p = hp.parts[it.part_begin + pi2];
        wins += (p.tile_end - (p.tile_begin & ~(wt - 1u)) + wt - 1u) / wt;
      }
      it.flags |= (fine << 2) | (std::min(wins, 0xFFFFFFu) << 8);
      hp.q_wins[pend[i].query] += wins;
    }
    hp.items[i] = it;
    hp.list_idx[it.peer_slot] = (uint32_t)i;
  }
  for (int qi = 0; qi < n_queries; ++qi) {
    const nrtgpu_bm25_query& q = queries[qi];
    hp.q_k[(size_t)qi] = (uint32_t)q.k;
    DQuery& dq = hp.queries[(size_t)qi];
    dq.k = (uint32_t)q.k;
    dq.has_after = q.has_after ? 1u : 0u;
    dq.after_doc = q.after_doc;
    dq.after_score = q.after_score;
    dq.item_begin = hp.q_base[(size_t)qi];
    dq.n_items = hp.q_nlists[(size_t)qi];
    dq.min_should_match = (uint32_t)std::max(effective_msm(q), 0);
    dq.sec_mode = sec_mode_of(q);
    dq.tie_breaker = q.tie_breaker;
    dq.combine_max = q.disjunction_max == 1 && dq.sec_mode == kMsSecNone ? 1u : 0u;   // (tie breaker 0: the best clause IS the score)
    // what a slice's hits must exceed for GREATER_THAN_OR_EQUAL_TO; COMPLETE mode: nothing does
    dq.gte_floor = q.total_hits_threshold == INT32_MAX ? 0xFFFFFFFFu : (uint32_t)std::max(q.total_hits_threshold, q.k);
    dq.slice_base = (uint32_t)qi * hp.n_slices;
  }
  if (plan_trace) {   // the items as a SET (order-free) and in launch order: two plans of one batch under different launch orders agree on the first
    uint64_t h_set = 0, h_order = 0;
    for (size_t i = 0; i < hp.items.size(); ++i) {
      const DItem& it = hp.items[i];
      uint64_t h = ((uint64_t)it.query << 40) ^ ((uint64_t)it.part_begin << 12) ^ (uint64_t)it.n_parts ^ ((uint64_t)it.flags << 60);
      h *= 0x9E3779B97F4A7C15ull;
      h ^= h >> 29;
      h_set += h;
      h_order = (h_order ^ h) * 0x100000001B3ull;
    }
    fprintf(stderr, "[nrtgpu plan] items as a set %016llx, in launch order %016llx; first items' queries:", (unsigned long long)h_set, (unsigned long long)h_order);
    for (size_t i = 0; i < hp.items.size() && i < 8; ++i) fprintf(stderr, " %u", hp.items[i].query);
    fprintf(stderr, "\n");
  }
  if (plan_trace)
    fprintf(stderr, "[nrtgpu plan] %d queries: head %.3f ms, resolve %.3f ms (%d threads), concat %.3f, cut+items %.3f; %zu terms %zu parts %zu items\n",
            n_queries, tp0 - t_entry, tp1 - tp0, n_thr, tp2 - tp1, now_ms() - tp2, (size_t)hp.n_dterms, hp.parts.size(), hp.items.size());
  return 0;
}
