// search.cpp -- BM25 entry points of the C ABI: plan upload + scan + merge, the fused hybrid tail, request
// coalescing, device-resident results for the multi-GPU exchange.
#include "runtime_internal.h"


// Speculative thresholds of the MaxScore route (plan.h: kHitsSpecInvalid): the guess's safety margin in standard deviations x 16
// (nrtgpu_set_speculation; a context starts with 5; 0: no speculation).  Measured on C3
// (profiles/r04_speculation_ab.log): margin 6 / 4 / 3 -> kernel 1.94 / 1.92 / (1.9) ms against 2.29 without, 0 / 0 / 108 of
// 122 880 queries run again.
static uint32_t spec_margin16(const nrtgpu_ctx* ctx) { return (uint32_t)std::max(ctx->spec_z16.load(std::memory_order_relaxed), 0); }
// The verdict on speculation belongs to the LEAF SET (runtime_internal.h: LeafSetCache.spec_*), in two steps.  A leaf set starts
// with its windows walked in docid order (the cheapest: a wave stays in a part for several windows).  When more than 2 % of
// >= 2048 queries had to be run again -- docids that are no sample of the index: an index sorted by something the score follows,
// time-ordered vocabulary -- it is given a second chance in the SCATTERED window order (maxscore.hip: any prefix of the windows
// taken is spread over the item's docs; +8 % kernel time on independently drawn docids, measured), counters reset; when that
// fails too, speculation is switched off for this leaf set alone.  Measured at C3's size (profiles/r05_scatter_*.log): an
// index numbered by doc length -- docid order 2043 of 2048 queries run again, scattered 0.5 % and the kernel at 1.10 ms
// against 2.34 without speculation; terms in docid bursts -- 17 % run again in docid order, 5 % scattered (the bursts'
// variance is not a sample's): off.
static void spec_sync_epoch(const nrtgpu_ctx* ctx, LeafSetCache* lsc) {
  const uint64_t e = ctx->spec_epoch.load(std::memory_order_relaxed);
  if (lsc->spec_epoch.exchange(e, std::memory_order_relaxed) != e) {   // nrtgpu_set_speculation since: start over
    lsc->spec_queries.store(0, std::memory_order_relaxed);
    lsc->spec_reruns.store(0, std::memory_order_relaxed);
    lsc->spec_calls.store(0, std::memory_order_relaxed);
    lsc->spec_calls_rerun.store(0, std::memory_order_relaxed);
    lsc->spec_scattered.store(0, std::memory_order_relaxed);
    lsc->spec_off.store(0, std::memory_order_relaxed);
  }
}
static bool speculating(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, int32_t n_segs) {
  if (!ctx || spec_margin16(ctx) == 0u || !segs || n_segs <= 0) return false;
  for (int32_t i = 0; i < n_segs; ++i)
    if (!segs[i]) return false;   // (the call fails with its own message)
  std::shared_ptr<LeafSetCache> lsc = leaf_set_cache(ctx, segs, n_segs);
  spec_sync_epoch(ctx, lsc.get());
  return lsc->spec_off.load(std::memory_order_relaxed) == 0;
}
// A call that ran under speculation has come back: count it, and throw the leaf set's switch when too many guesses fail --
// every failure is a second pass.
static void note_speculation_of(nrtgpu_ctx* ctx, LeafSetCache* lsc, int64_t n_queries, int64_t n_rerun) {
  spec_sync_epoch(ctx, lsc);
  const int64_t seen = lsc->spec_queries.fetch_add(n_queries, std::memory_order_relaxed) + n_queries;
  const int64_t failed = lsc->spec_reruns.fetch_add(n_rerun, std::memory_order_relaxed) + n_rerun;
  ctx->spec_queries.fetch_add(n_queries, std::memory_order_relaxed);
  ctx->spec_reruns.fetch_add(n_rerun, std::memory_order_relaxed);
  // Two ways to fail the verdict.  By QUERIES: more than 2 % of >= 2048 run again.  By CALLS (round 6): a second pass costs per
  // call -- plan, launch, a wait of its own -- whether it re-runs one query or fifty, so what decides is how many CALLS need
  // one: more than a quarter of >= 32 calls.  Measured at C3's size with 1024-query batches (profiles/r06_followup_seeded_reruns.log):
  // the sorted corpus in the scattered order fails 0.45 % of its queries -- under the first rule "cured" -- which is a failed
  // query in 99 % of its batches: 2.32 - 2.62 ms per step under speculation against 2.08 with it off, although the scorer
  // itself runs 1.05 ms against 2.04.  A closed loop's small batches keep their speculation under the same failure rate (8-query
  // batches: one call in 28 needs a second pass).
  const int64_t calls = lsc->spec_calls.fetch_add(1, std::memory_order_relaxed) + 1;
  const int64_t calls_bad = lsc->spec_calls_rerun.fetch_add(n_rerun > 0 ? 1 : 0, std::memory_order_relaxed) + (n_rerun > 0 ? 1 : 0);
  const bool no_verdict = dev_env_int("NRTGPU_SPEC_NO_VERDICT", 0) != 0;   // (development build: a fixed setting; read per call: tests set it)
  // (A third step was tried in round 6 -- the guess's margin from the MEASURED dispersion of the candidates over the doc windows
  //  instead of a sample's sqrt(m): re-runs fell to 0.2 - 0.5 % on the clustered corpus and to none on the sorted one, and the
  //  deeper guesses gave the gain back: clustered 2.92 ms per step against 2.46 with speculation off, sorted 2.17 against 2.09.
  //  profiles/r06_dispersion_experiment.patch, r06_dispersion_second_pass.log.)
  if (((seen >= 2048 && failed * 50 > seen) || (calls >= 32 && calls_bad * 4 > calls)) && !no_verdict) {
    if (lsc->spec_scattered.exchange(1, std::memory_order_relaxed) == 0) {   // first: the scattered window order, a fresh count
      lsc->spec_queries.store(0, std::memory_order_relaxed);
      lsc->spec_reruns.store(0, std::memory_order_relaxed);
      lsc->spec_calls.store(0, std::memory_order_relaxed);
      lsc->spec_calls_rerun.store(0, std::memory_order_relaxed);
      ctx->spec_scattered.store(1, std::memory_order_relaxed);
    } else {
      lsc->spec_off.store(1, std::memory_order_relaxed);
      ctx->spec_off.store(1, std::memory_order_relaxed);
    }
  }
}
static void note_speculation(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, int32_t n_segs, int64_t n_queries, int64_t n_rerun) {
  std::shared_ptr<LeafSetCache> lsc = leaf_set_cache(ctx, segs, n_segs);
  note_speculation_of(ctx, lsc.get(), n_queries, n_rerun);
}

void nrtgpu::rt::note_shard_speculation(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, int32_t n_segs, int64_t n_queries, int64_t n_failed) {
  note_speculation(ctx, segs, n_segs, n_queries, n_failed);
}

// NRTGPU_MS_PERSISTENT=0 (development build): one workgroup per item + helper workgroups behind them (A/B)
static bool ms_persistent() {
  static const bool v = dev_env_int("NRTGPU_MS_PERSISTENT", 1) != 0;
  return v;
}

struct DeviceRun {
  // device pointers valid until the slot is reused
  uint64_t* out_keys = nullptr;
  uint32_t* out_counts = nullptr;
  uint64_t* out_hits = nullptr;
  uint64_t* prof = nullptr;   // instrumented variant: 8 counters per item
  size_t n_items = 0;
  size_t n_slots = 0;      // items + the MaxScore route's helper slots behind them
  uint64_t* walls = nullptr;   // instrumented variant: per slot {start, end, item, windows} (plan.h: DHelp.walls)
  size_t n_ms_items = 0;
};

// Enqueue plan upload + scan + merge on the slot's stream.  Merge output goes to (ext_keys,
// ext_counts, ext_hits) when given (device-resident variant), else into the slot's scratch.
// Everything is enqueued at once on the slot's stream and nothing waits on the host.  The kernels that want the whole
// GPU (the two scorers) take turns ON THE DEVICE: under `gpu` (unlocked on entry and on return) the stream is made to
// wait for the event recorded behind the previous batch's scorers, then this batch's are enqueued and their event
// recorded behind their merge.  The plan upload and its expansion run ahead of that wait: they overlap the scorers of the
// batch before instead of sitting between the two (before: lock -> launch -> host sync -> unlock, a host round trip plus
// upload and expansion between any two scorer launches).
static int enqueue_search(nrtgpu_ctx* ctx, Slot* slot, const HostPlan& hp, int32_t n_queries, uint32_t k_stride_out,
                          uint64_t* ext_keys, uint32_t* ext_counts, uint64_t* ext_hits, DeviceRun* run,
                          std::unique_lock<std::mutex>& gpu, int64_t epoch = -1, bool allow_spec = false, int32_t spec_world = 1,
                          uint64_t* ext_guess = nullptr, bool follow_up = false) {
  forget_foreign_hip_error();
  // follow_up (round 6): the second pass of a speculative call -- the handful of queries whose guess failed the merge's check, run
  // again (seeded: search_batch_spec).  As a launch of the usual kind it took a TURN of its own, and BEHIND whatever batch another
  // thread had enqueued meanwhile: its caller waited a whole batch for five queries (sorted-by-length corpus at C3's size, two
  // submitting threads: 2.6 ms per 1024-query step around a 1.05 ms kernel -- slower than with speculation off).  A follow-up takes
  // no turn: its few persistent workgroups run on the CUs every big launch leaves alone, beside whatever batch is running.
  // Only MaxScore items (the exhaustive scan launches a workgroup per item and wants the whole device).
  static const bool follow_up_on = dev_env_int("NRTGPU_FOLLOW_UP", 1) != 0;   // (development build: 0 = a turn of its own, A/B)
  const bool small = follow_up && follow_up_on && hp.n_ms_items != 0 && hp.n_ms_items == hp.items.size() && ms_persistent();
  // spec_world > 1 (the library's multi-GPU search, dist.cpp): this call is ONE SHARD of a spec_world-way search over equal docid
  // ranges, and its speculative thresholds are guesses at the k-th score of the WHOLE search -- a shard's docs are a 1 / world
  // sample of the index, so the guess rule holds with the windows of all shards in its denominator.  Such a guess cannot be
  // checked against this shard's list: the largest one per query goes to ext_guess and the caller checks it against the list
  // merged over all shards (and runs a failed query again on every shard).
  const size_t n_items = hp.items.size();
  Carver pc;
  const size_t o_queries = pc.take(hp.queries.size() * sizeof(DQuery));
  const size_t o_items = pc.take(n_items * sizeof(DItem));
  const size_t o_parts = pc.take(hp.parts.size() * sizeof(DPart));
  const size_t o_qterms = pc.take(hp.qterms.size() * sizeof(DQTerm));
  const size_t o_qexp = pc.take(hp.qexpand.size() * sizeof(DQExpand));
  const size_t o_qsb = pc.take(hp.qs_begin.size() * 4);
  const size_t o_caches = pc.take(hp.caches.size() * sizeof(float));
  const size_t o_lidx = pc.take(hp.list_idx.size() * 4);
  const size_t o_qbase = pc.take(hp.q_base.size() * 4);
  const size_t o_qnl = pc.take(hp.q_nlists.size() * 4);
  const size_t o_qk = pc.take(hp.q_k.size() * 4);
  const size_t o_theta = pc.take(hp.theta_init.size() * 8);  // uploaded with the plan, then updated by the kernel
  const size_t o_quant = pc.take(hp.list_idx.size() * 8);    // per item: published quantile bound (zeros)
  const size_t o_lower = pc.take(ext_hits ? hp.q_lower.size() * 8 : 0);  // device-resident results: certain lower bounds
  const bool use_xch = epoch >= 0 && ctx->xch_dev != nullptr;
  const size_t o_xch = pc.take(use_xch ? sizeof(DExchange) : 0);
  const size_t o_qwins = pc.take(hp.q_wins.size() * 4);
  const size_t o_help = pc.take(sizeof(MsArgs));   // the MaxScore launch's record (plan.h); filled in below, once the workspace is carved
  const size_t plan_bytes = pc.off;
  if (int rc = slot->h_plan.reserve(plan_bytes)) return rc;
  if (int rc = slot->d_plan.reserve(plan_bytes)) return rc;
  char* hb = (char*)slot->h_plan.p;
  memcpy(hb + o_queries, hp.queries.data(), hp.queries.size() * sizeof(DQuery));
  if (n_items) memcpy(hb + o_items, hp.items.data(), n_items * sizeof(DItem));
  if (!hp.parts.empty()) memcpy(hb + o_parts, hp.parts.data(), hp.parts.size() * sizeof(DPart));
  if (!hp.qterms.empty()) memcpy(hb + o_qterms, hp.qterms.data(), hp.qterms.size() * sizeof(DQTerm));
  memcpy(hb + o_qexp, hp.qexpand.data(), hp.qexpand.size() * sizeof(DQExpand));
  if (!hp.qs_begin.empty()) memcpy(hb + o_qsb, hp.qs_begin.data(), hp.qs_begin.size() * 4);
  memcpy(hb + o_caches, hp.caches.data(), hp.caches.size() * sizeof(float));
  if (!hp.list_idx.empty()) memcpy(hb + o_lidx, hp.list_idx.data(), hp.list_idx.size() * 4);
  memcpy(hb + o_qbase, hp.q_base.data(), hp.q_base.size() * 4);
  memcpy(hb + o_qnl, hp.q_nlists.data(), hp.q_nlists.size() * 4);
  memcpy(hb + o_qk, hp.q_k.data(), hp.q_k.size() * 4);
  memcpy(hb + o_theta, hp.theta_init.data(), hp.theta_init.size() * 8);
  if (!hp.q_wins.empty()) memcpy(hb + o_qwins, hp.q_wins.data(), hp.q_wins.size() * 4);
  if (spec_world > 1) {   // the denominator of "how much of the query's docs have I seen": the windows of ALL shards
    // (this shard's REAL share of the index where the caller has stated it -- nrtgpu_set_shard_share: virtual shards balance
    //  live docs, not docid ranges, and a shard that holds 40 % of the index is no "one of two" -- else spec_world equal shards)
    uint64_t num = (uint64_t)spec_world, den = 1;
    const int64_t sd = ctx->shard_docs.load(std::memory_order_relaxed), id = ctx->index_docs.load(std::memory_order_relaxed);
    if (sd > 0 && id >= sd) {
      num = (uint64_t)id;
      den = (uint64_t)sd;
    }
    uint32_t* qw = (uint32_t*)(hb + o_qwins);
    for (size_t i = 0; i < hp.q_wins.size(); ++i) qw[i] = (uint32_t)std::min<uint64_t>(((uint64_t)qw[i] * num + den - 1) / den, 0xFFFFFFFFull);
  }
  memset(hb + o_quant, 0, hp.list_idx.size() * 8);
  if (ext_hits) memcpy(hb + o_lower, hp.q_lower.data(), hp.q_lower.size() * 8);
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

  // MaxScore route: helper workgroups behind the items (plan.h: DHelp; maxscore.hip) -- each with an output slot of its own
  // behind the items' slots.  NRTGPU_MS_HELPERS: how many (default 4 per CU; 0: none), NRTGPU_MS_HELP_MIN: an item with fewer
  // unassigned windows is not joined.
  static const int env_helpers = (int)dev_env_int("NRTGPU_MS_HELPERS", -1);
  static const int env_help_min = (int)dev_env_int("NRTGPU_MS_HELP_MIN", 16);
  const size_t n_help = hp.n_ms_items == 0 ? 0 : (size_t)(env_helpers >= 0 ? env_helpers : 4 * std::max(ctx->n_cus, 1));
  const size_t n_slots = n_items + n_help;
  Carver wc;
  const size_t o_ikeys = wc.take(n_slots * (size_t)hp.k_stride * 8);
  const size_t o_icnt = wc.take(n_slots * 4);
  const size_t o_ihits = wc.take(n_slots * 8);
  const size_t o_okeys = wc.take((size_t)n_queries * k_stride_out * 8);
  const size_t o_ocnt = wc.take((size_t)n_queries * 4);
  const size_t o_ohits = wc.take((size_t)n_queries * 8);
  const size_t o_terms = wc.take((size_t)hp.n_dterms * sizeof(DTerm));  // written by expand_terms_kernel
  // per (query, searcher slice) the hits its items counted, per query "some item's slice has passed the floor": zeroed per call
  const size_t o_ssum = wc.take((size_t)n_queries * hp.n_slices * 4), o_qprune = wc.take((size_t)n_queries * 4);
  // the helpers' state (zeroed per call as well): per MaxScore item the window counter, the helpers that joined, the owner's
  // start time; per query the head of its helper-slot list; the "nothing left to help" flag
  const size_t o_hwin = wc.take(hp.n_ms_items * 4), o_hcnt = wc.take(hp.n_ms_items * 4), o_ht0 = wc.take(hp.n_ms_items * 8);
  const size_t o_hhead = wc.take((size_t)n_queries * 4), o_hnext = wc.take(n_help * 4), o_hoff = wc.take(4);
  const size_t o_hqueue = wc.take(4), o_hused = wc.take(4), o_hstart = wc.take(8);
  // speculative thresholds (plan.h: kHitsSpecInvalid): only where the caller can run a query again (allow_spec: the batch and the
  // hybrid entry), never next to the cross-GPU bound exchange (its quantile uses the selection's second rank)
  const uint32_t spec_z16 = spec_margin16(ctx);
  const bool spec = allow_spec && spec_z16 != 0u && hp.n_ms_items != 0 && !use_xch && hp.lsc && hp.lsc->spec_off.load(std::memory_order_relaxed) == 0;
  const size_t o_spec = wc.take(spec ? (size_t)n_queries * 8 : 0);
  const size_t zero_bytes = wc.off - o_ssum;
  // kernel variant: clause counting (8), doc-set masks somewhere in the batch (9), else what the flags ask for
  const int flag_variant = (ctx->cfg.flags >> 8) & 15;
  const int ablation = hp.clause_counting ? 8 : ((hp.masked && flag_variant == 0 && !(ctx->cfg.flags & NRTGPU_FLAG_NO_MASK_VARIANT)) ? 9 : flag_variant);
  const bool profile = flag_variant == 7;
  const size_t o_prof = wc.take((ablation == 7 || profile) ? n_slots * 128 : 0);
  const size_t o_walls = wc.take(profile ? n_slots * 64 : 0);
  if (int rc = slot->d_work.reserve(wc.off)) return rc;
  char* db = (char*)slot->d_plan.p;
  char* wb = (char*)slot->d_work.p;
  DHelp help{};
  help.win_next = (uint32_t*)(wb + o_hwin);
  help.help_cnt = (uint32_t*)(wb + o_hcnt);
  help.item_t0 = (unsigned long long*)(wb + o_ht0);
  help.help_head = (uint32_t*)(wb + o_hhead);
  help.help_query = (uint32_t*)(wb + o_hnext);
  help.help_off = (uint32_t*)(wb + o_hoff);
  help.item_next = (uint32_t*)(wb + o_hqueue);
  help.help_used = (uint32_t*)(wb + o_hused);
  help.t_start = (unsigned long long*)(wb + o_hstart);
  {
    // NRTGPU_MS_HELP_ALPHA (x 16; 0: helpers only once the queue is empty): while items are queued a workgroup helps an item
    // whose expected time left exceeds alpha x what is left of the launch.  Measured at 8 spare CUs (same log): alpha 0 2.48 ms
    // per step, 1.0 2.50, 1.5 2.42, 2.0 2.42, 3.0 2.46 -- the estimate of what is left runs high early in the launch (the
    // heavy items lead), so the bar sits above 1: 1.5 is the default.
    static const int env_help_alpha = (int)dev_env_int("NRTGPU_MS_HELP_ALPHA", 24);
    uint64_t wins = 0;
    for (size_t i = 0; i < hp.n_ms_items; ++i) wins += hp.items[i].flags >> 8;
    help.total_wins = (uint32_t)std::min<uint64_t>(wins, 0xFFFFFFFFull);
    help.alpha16 = (uint32_t)std::max(env_help_alpha, 0);
    // NRTGPU_MS_SPARE_CUS: CUs a persistent launch leaves alone.  A persistent workgroup holds its CU (all of the LDS, 504 of
    // the 512 vector registers of every SIMD) until the launch ends, so nothing else runs there -- and the NEXT batch's plan
    // expansion and the memsets in front of it, which used to slip in between two workgroups of this launch, would queue
    // behind it instead of overlapping it.
    // Measured (profiles/r04_persistent_spare_ab.log, 1024 C3 queries per step): 0 spare CUs 2.85 ms per step (kernel 2.34), 4: 2.72,
    // 8: 2.50 (kernel 2.39), 16: 2.56, 32: 2.70 -- eight CUs of 256 is the default.
    static const int env_spare = (int)dev_env_int("NRTGPU_MS_SPARE_CUS", 8);
    help.n_cus = (uint32_t)std::max(ctx->n_cus - std::max(env_spare, 0), 1);
    if (small) help.n_cus = (uint32_t)std::max(env_spare, 1);   // (a follow-up: the spare CUs are its whole device)
    // NRTGPU_MS_PERSISTENT=0: one workgroup per item + helper workgroups behind them (A/B)
    help.persistent = ms_persistent() ? 1u : 0u;
  }
  help.n_own = (uint32_t)hp.n_ms_items;
  help.n_help = (uint32_t)n_help;
  help.slot_base = (uint32_t)n_items;
  static const bool env_help_greedy = dev_env_int("NRTGPU_MS_HELP_GREEDY", 0) != 0;
  help.min_rem = (uint32_t)std::min(std::max(env_help_min, 1), 0xFFFF) | (env_help_greedy ? 1u << 16 : 0u);
  help.walls = profile ? (unsigned long long*)(wb + o_walls) : nullptr;
  help.spec_g = spec ? (unsigned long long*)(wb + o_spec) : nullptr;
  help.spec_z16 = spec ? spec_z16 : 0u;
  {   // NRTGPU_MS_SPEC_FIRST / NRTGPU_MS_SPEC_GROW (x 16): when a workgroup's estimates are due (plan.h: DHelp.spec_sched)
    static const int env_first = (int)dev_env_int("NRTGPU_MS_SPEC_FIRST", 2 * kMsWaves);
    static const int env_grow = (int)dev_env_int("NRTGPU_MS_SPEC_GROW", 32);
    help.spec_sched = (uint32_t)std::min(std::max(env_first, 1), 255) | ((uint32_t)std::min(std::max(env_grow, 17), 255) << 8);
  }
  MsArgs ms_args{};   // (the kernel reads the record from the plan: maxscore.hip)
  ms_args.items = (const DItem*)(db + o_items);
  ms_args.parts = (const DPart*)(db + o_parts);
  ms_args.terms = (const DTerm*)(wb + o_terms);
  ms_args.queries = (const DQuery*)(db + o_queries);
  ms_args.caches = (const float*)(db + o_caches);
  ms_args.theta_g = (unsigned long long*)(db + o_theta);
  ms_args.slice_sum = (uint32_t*)(wb + o_ssum);
  ms_args.q_prune = (uint32_t*)(wb + o_qprune);
  ms_args.xch = use_xch ? (const DExchange*)(db + o_xch) : nullptr;
  ms_args.item_keys = (uint64_t*)(wb + o_ikeys);
  ms_args.item_counts = (uint32_t*)(wb + o_icnt);
  ms_args.item_hits = (uint64_t*)(wb + o_ihits);
  ms_args.item_prof = profile ? (uint64_t*)(wb + o_prof) : nullptr;
  ms_args.q_wins = (const uint32_t*)(db + o_qwins);
  {   // the leaf set's window order (note_speculation); NRTGPU_MS_SCATTER = 0 / 1 (development build): forced, A/B
    const long forced = dev_env_int("NRTGPU_MS_SCATTER", -1);
    // (the leaf set's step: search.cpp: note_speculation_of; forced: the bits as given)
    ms_args.scatter = forced >= 0 ? (forced != 0 ? 1u : 0u) : ((spec && hp.lsc->spec_scattered.load(std::memory_order_relaxed) != 0) ? 1u : 0u);
  }
  ms_args.k_stride = hp.k_stride;
  ms_args.help = help;
  memcpy(hb + o_help, &ms_args, sizeof(ms_args));

  hipStream_t st = slot->stream;
  HIP_TRY(hipMemcpyAsync(db, hb, plan_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(wb + o_ssum, 0, zero_bytes, st));
  const bool timing = ctx->cfg.collect_timing != 0;
  // the queries on the MaxScore route (items [0, n_ms)), then the exhaustive scan of the others
  const size_t n_ms = hp.n_ms_items;
  // the compact plan -> the DTerm records of every (query, leaf), on the device
  launch_expand_terms(st, (const DQExpand*)(db + o_qexp), (const DQTerm*)(db + o_qterms), (const uint32_t*)(db + o_qsb),
                      (uint32_t)n_queries, hp.n_leaves, (DTerm*)(wb + o_terms));
  gpu.lock();
  // (experiment, off unless NRTGPU_OVERLAP_SCORERS=1: no turn between consecutive calls' scorers -- nothing but the turn itself
  //  orders them: workspaces are per slot, term tables reach the device before they become visible -- so that the next batch's
  //  items fill the tail of this one's launch.  Measured, round 5 (profiles/r05_overlap_scorers_ab.log): SLOWER, 2.50 against 1.95 ms
  //  per step -- the batch's merge queues behind the next batch's items, 0.09 -> 0.98 ms)
  static const bool overlap_scorers = dev_env_int("NRTGPU_OVERLAP_SCORERS", 0) != 0;
  if (ctx->last_turn && !overlap_scorers && !small) HIP_TRY(hipStreamWaitEvent(st, ctx->last_turn, 0));
  if (ctx->last_knn_turn) HIP_TRY(hipStreamWaitEvent(st, ctx->last_knn_turn, 0));   // (vector searches do not queue behind each other, the scorers queue behind them)
  if (timing) HIP_TRY(hipEventRecord(slot->ev3, st));
  if (profile && n_help) HIP_TRY(hipMemsetAsync(wb + o_prof + n_items * 128, 0, n_help * 128, st));   // (a helper that leaves at once writes nothing)
  if (profile) HIP_TRY(hipMemsetAsync(wb + o_walls, 0, n_slots * 64, st));
  launch_bm25_maxscore(st, profile, (ctx->cfg.flags & NRTGPU_FLAG_PACKED_POSTINGS) != 0, hp.ms_two ? 2 : (hp.ms_shapes ? 1 : 0), ms_args, (const MsArgs*)(db + o_help));
  if (timing) HIP_TRY(hipEventRecord(slot->ev0, st));
  launch_bm25_scan(st, hp.fixed_point, (ctx->cfg.flags & NRTGPU_FLAG_NO_PREFETCH) == 0, (ctx->cfg.flags & NRTGPU_FLAG_PACKED_POSTINGS) != 0, ablation, (uint32_t)(n_items - n_ms),
                   (const DItem*)(db + o_items) + n_ms, (const DPart*)(db + o_parts), (const DTerm*)(wb + o_terms),
                   (const DQuery*)(db + o_queries), (const float*)(db + o_caches),
                   (unsigned long long*)(db + o_theta), (unsigned long long*)(db + o_quant),
                   use_xch ? (const DExchange*)(db + o_xch) : nullptr, (uint32_t*)(wb + o_ssum), (uint64_t*)(wb + o_ikeys) + n_ms * (size_t)hp.k_stride,
                   (uint32_t*)(wb + o_icnt) + n_ms, (uint64_t*)(wb + o_ihits) + n_ms, hp.k_stride,
                   ablation == 7 ? (uint64_t*)(wb + o_prof) + n_ms * 16 : nullptr);
  if (timing) HIP_TRY(hipEventRecord(slot->ev1, st));
  // The turn ends behind the SCORERS: the next batch's scorers start while this batch's merge (one workgroup per query, 0.05 ms
  // of device time per 1024 queries) runs on the CUs the persistent MaxScore launch leaves alone.  Through round 3 the merge was
  // part of the turn -- behind the next batch's one-workgroup-per-item launch it waited for a free CU until that launch drained
  // (closed loop at 64 callers: p99 1.4 -> 4.0 ms) -- which the spare CUs have changed: measured, same box
  // (profiles/r04_turn_before_merge_ab.log), 2.413 -> 2.357 ms per 1024-query step, batch p50 4.80 -> 4.69 ms, closed loop at
  // 64 / 512 callers p99 1.20 / 2.61 -> 1.20 / 2.55 ms.  NRTGPU_TURN_BEFORE_MERGE=0: the old turn (A/B).
  static const bool turn_before_merge = dev_env_int("NRTGPU_TURN_BEFORE_MERGE", 1) != 0;
  if (turn_before_merge && !small) {
    HIP_TRY(hipEventRecord(slot->ev_turn, st));
    ctx->last_turn = slot->ev_turn;
  }
  uint64_t* okeys = ext_keys ? ext_keys : (uint64_t*)(wb + o_okeys);
  uint32_t* ocnt = ext_counts ? ext_counts : (uint32_t*)(wb + o_ocnt);
  uint64_t* ohits = ext_hits ? ext_hits : (uint64_t*)(wb + o_ohits);
  launch_merge_topk(st, (uint32_t)n_queries, (const uint64_t*)(wb + o_ikeys), (const uint32_t*)(wb + o_icnt),
                    (const uint64_t*)(wb + o_ihits), (const uint32_t*)(db + o_lidx), (const uint32_t*)(db + o_qbase),
                    (const uint32_t*)(db + o_qnl), hp.k_stride, (const uint32_t*)(db + o_qk), okeys, ocnt, ohits,
                    k_stride_out, n_help ? help.help_query : nullptr, (uint32_t)n_help, help.slot_base, spec_world > 1 ? nullptr : help.spec_g);
  if (ext_guess) {   // (spec_world > 1: the guesses are checked by the caller, against the list merged over all shards)
    if (spec) HIP_TRY(hipMemcpyAsync(ext_guess, wb + o_spec, (size_t)n_queries * 8, hipMemcpyDeviceToDevice, st));
    else HIP_TRY(hipMemsetAsync(ext_guess, 0, (size_t)n_queries * 8, st));
  }
  // TotalHits.relation by the reference's per-slice rule, tagged into the merged counts
  launch_slice_relation(st, (const uint32_t*)(wb + o_ssum), (const DQuery*)(db + o_queries), hp.n_slices, ohits, (uint32_t)n_queries);
  if (ext_hits && hp.n_ms_items) launch_patch_hits(st, (const uint64_t*)(db + o_lower), ohits, (uint32_t)n_queries);
  if (timing) HIP_TRY(hipEventRecord(slot->ev2, st));
  if (!turn_before_merge && !small) {
    HIP_TRY(hipEventRecord(slot->ev_turn, st));
    ctx->last_turn = slot->ev_turn;
  }
  gpu.unlock();
  HIP_TRY(hipGetLastError());
  run->out_keys = okeys;
  run->out_counts = ocnt;
  run->out_hits = ohits;
  run->prof = (ablation == 7 || profile) ? (uint64_t*)(wb + o_prof) : nullptr;
  run->n_items = n_items;
  run->n_slots = n_slots;
  run->walls = profile ? (uint64_t*)(wb + o_walls) : nullptr;
  run->n_ms_items = n_ms;
  return 0;
}

static void account(nrtgpu_ctx* ctx, Slot* slot, const HostPlan& hp, int32_t n_queries, double plan_ms, double t_entry_ms = 0.0,
                    double queue_ms = 0.0) {
  float scan_ms = 0.f, merge_ms = 0.f, ms_ms = 0.f;
  if (ctx->cfg.collect_timing) {
    (void)hipEventElapsedTime(&ms_ms, slot->ev3, slot->ev0);
    (void)hipEventElapsedTime(&scan_ms, slot->ev0, slot->ev1);
    (void)hipEventElapsedTime(&merge_ms, slot->ev1, slot->ev2);
  }
  const size_t n_scan = hp.items.size() - hp.n_ms_items;
  {   // what this call cost, for the calling thread (nrtgpu_last_diagnostics)
    nrtgpu_diagnostics d{};
    d.total_ms = t_entry_ms > 0.0 ? now_ms() - t_entry_ms : 0.0;
    d.plan_ms = plan_ms;
    d.queue_ms = queue_ms;
    d.device_ms = (double)(hp.n_ms_items ? ms_ms : 0.f) + (double)(n_scan ? scan_ms : 0.f) + (double)merge_ms;
    d.postings = hp.postings;
    d.queries = n_queries;
    d.items_maxscore = (int32_t)hp.n_ms_items;
    d.items_scan = (int32_t)n_scan;
    g_diag = d;
  }
  std::lock_guard<std::mutex> lk(ctx->stats_mu);
  ctx->stats.batches += 1;
  ctx->stats.queries += n_queries;
  ctx->stats.scan_launches += n_scan ? 1 : 0;
  ctx->stats.fixed_point_launches += (n_scan && hp.fixed_point) ? 1 : 0;
  ctx->stats.scan_ms += n_scan ? scan_ms : 0.f;
  ctx->stats.merge_ms += merge_ms;
  ctx->stats.scan_postings += hp.postings - hp.ms_postings;
  ctx->stats.scan_items += (int64_t)n_scan;
  ctx->stats.maxscore_launches += hp.n_ms_items ? 1 : 0;
  ctx->stats.maxscore_ms += hp.n_ms_items ? ms_ms : 0.f;
  ctx->stats.maxscore_postings += hp.ms_postings;
  ctx->stats.maxscore_items += (int64_t)hp.n_ms_items;
  ctx->stats.host_plan_ms += plan_ms;
}

// `hits` as the device leaves it: low 48 bits = a count, high 16 != 0 = the relation is GREATER_THAN_OR_EQUAL_TO
// (plan.h: kHitsPrunedUnit) -- some slice of the searcher collected more than max(totalHitsThreshold, numHits) hits,
// where one reference collector starts publishing a min competitive score (LazyQueueTopScoreDocCollector.java:176-199;
// slices: MyIndexSearcher.java:163-208; reduce: LazyQueueTopScoreDocCollectorManager.java:137-144).  The tag comes from
// slice_relation_kernel (exhaustive scan: the count is exact) or from the MaxScore route's items (the count is a lower
// bound; what is reported then is `lower` -- the live docs the planner knew to match, the same on every run: the
// reference's own value there is an artefact of its traversal, SURVEY 7 hard part 3).  Either way the queue must be
// full: a page of a searchAfter walk that returns fewer than numHits hits is EQUAL_TO.
static void unpack_topdocs(const uint64_t* keys, uint32_t n, uint64_t hits, const int32_t k, int64_t lower, uint32_t n_first,
                           nrtgpu_topdocs* out) {
  const int32_t cap = out->capacity > 0 ? out->capacity : k;
  const int32_t m = std::min<int32_t>((int32_t)n, cap);
  // two plain loops (vectorisable): doc = ~low word, score = high word reinterpreted
  if (int32_t* __restrict__ docs = out->docs)
    for (int32_t i = 0; i < m; ++i) docs[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)keys[i]);
  if (uint32_t* __restrict__ sc = (uint32_t*)out->scores)
    for (int32_t i = 0; i < m; ++i) sc[i] = (uint32_t)(keys[i] >> 32);
  out->n_hits = m;
  const int64_t counted = (int64_t)(hits & (kHitsPrunedUnit - 1));
  // (lower > 0: the query took the MaxScore route, i.e. the planner KNEW some slice passes the threshold, whether
  // or not the kernel then skipped anything.  Merged per-GPU results carry their shards' bounds in `counted`:
  // nrtgpu_search_bm25_batch_device)
  const bool gte = ((hits >> 48) != 0 || lower > 0) && n_first == (uint32_t)k;
  out->total_hits = (gte && lower > 0) ? lower : counted;
  out->total_hits_is_lower_bound = gte ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// ABI: search
// ------------------------------------------------------------------------------------------------
// rerun: where the indices of the queries go whose speculative threshold failed the merge's check (plan.h: kHitsSpecInvalid);
// nullptr: no speculation in this call.
static int search_batch_impl(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                             int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                             nrtgpu_topdocs* out, std::vector<int32_t>* rerun, bool content_held = false, const uint64_t* seeds = nullptr,
                             bool follow_up = false) {
  if (!ctx || !queries || !out || (n_segs > 0 && !segs)) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_queries <= 0 || n_segs < 0) return fail(NRTGPU_ERR_INVALID_ARG, "n_queries must be > 0");
  if (n_queries > ctx->cfg.max_batch) return fail(NRTGPU_ERR_INVALID_ARG, "batch of %d exceeds max_batch %d", n_queries, ctx->cfg.max_batch);
  NRT_CHECK_DEADLINE("before the search was planned");
  HIP_TRY(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  HostPlan hp;
  for (int si = 0; si < n_segs; ++si)
    if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
  std::optional<SegReadLocks> content;   // until this call's kernels have finished (content_held: the caller holds them over both passes)
  if (!content_held) content.emplace(segs, n_segs);
  if (int rc = build_plan(ctx, segs, doc_bases, n_segs, queries, n_queries, hp, 1)) return rc;
  // seeds (search_batch_spec's second pass): per query a key that k docs of the search are KNOWN to exceed -- the walk starts from
  // it as from a threshold another item of the query had published (theta_g), and collects nothing at or below it
  if (seeds)
    for (int qi = 0; qi < n_queries; ++qi) hp.theta_init[(size_t)qi] = std::max(hp.theta_init[(size_t)qi], seeds[qi]);
  const double plan_ms = now_ms() - t0;

  Slot* slot = nullptr;
  acquire_slot(ctx, &slot);
  struct Guard { nrtgpu_ctx* c; Slot* s; ~Guard() { release_slot(c, s); } } guard{ctx, slot};
  const double queue_ms = now_ms() - t0 - plan_ms;
  NRT_CHECK_DEADLINE("while the search waited for a workspace");   // (nothing has been launched)
  DeviceRun run;
  const size_t kb = (size_t)n_queries * hp.k_stride * 8, cb = (size_t)n_queries * 4, hb = (size_t)n_queries * 8;
  Carver oc;
  const size_t o_k = oc.take(kb), o_c = oc.take(cb), o_h = oc.take(hb);
  if (int rc = slot->h_out.reserve(oc.off)) return rc;
  char* ho = (char*)slot->h_out.p;
  static const bool call_trace = dev_env_int("NRTGPU_PLAN_TRACE", 0) != 0;  // debug aid: phase times on stderr
  const double tc0 = call_trace ? now_ms() : 0.0;
  {
    std::unique_lock<std::mutex> gpu(ctx->gpu_mu, std::defer_lock);
    if (int rc = enqueue_search(ctx, slot, hp, n_queries, hp.k_stride, nullptr, nullptr, nullptr, &run, gpu, -1, rerun != nullptr, 1, nullptr, follow_up)) return rc;
  }
  // The answers come back behind the kernels on the same stream: ONE copy where the workspace lays keys, counts and hits out as the
  // host buffer does (both are carved by the same rule), and one wait for everything -- through round 5 the call waited for the
  // kernels, then issued three copies and waited again: a second wake-up and two more copy launches, a fifth of a one-query call
  // (profiles/r06_single_query_timeline.txt).
  const double tc1 = call_trace ? now_ms() : 0.0;
  if ((const char*)run.out_counts - (const char*)run.out_keys == (ptrdiff_t)(o_c - o_k) && (const char*)run.out_hits - (const char*)run.out_keys == (ptrdiff_t)(o_h - o_k)) {
    HIP_TRY(hipMemcpyAsync(ho + o_k, run.out_keys, (o_h - o_k) + hb, hipMemcpyDeviceToHost, slot->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(ho + o_k, run.out_keys, kb, hipMemcpyDeviceToHost, slot->stream));
    HIP_TRY(hipMemcpyAsync(ho + o_c, run.out_counts, cb, hipMemcpyDeviceToHost, slot->stream));
    HIP_TRY(hipMemcpyAsync(ho + o_h, run.out_hits, hb, hipMemcpyDeviceToHost, slot->stream));
  }
  HIP_TRY(wait_for_stream((ctx->cfg.flags & NRTGPU_FLAG_BLOCKING_WAIT) != 0, slot->stream, slot->ev_wait));
  const double tc2 = call_trace ? now_ms() : 0.0;
  const uint64_t* keys = (const uint64_t*)(ho + o_k);
  const uint32_t* cnts = (const uint32_t*)(ho + o_c);
  const uint64_t* hits = (const uint64_t*)(ho + o_h);
  {
    const int n_chunks = n_queries >= 256 ? std::min(8, ctx->pool->helpers() + 1) : 1;   // ~8 KB of docs + scores per query
    ctx->pool->run(n_chunks, [&](int c) {
      const int q0 = (int)((int64_t)n_queries * c / n_chunks), q1 = (int)((int64_t)n_queries * (c + 1) / n_chunks);
      for (int qi = q0; qi < q1; ++qi)
        unpack_topdocs(keys + (size_t)qi * hp.k_stride, cnts[qi], hits[qi] & ~kHitsSpecInvalid, queries[qi].k, hp.q_lower[(size_t)qi], cnts[qi], &out[qi]);
    });
  }
  if (rerun)
    for (int qi = 0; qi < n_queries; ++qi)
      if (hits[qi] & kHitsSpecInvalid) rerun->push_back(qi);
  if (call_trace)
    fprintf(stderr, "[nrtgpu call] %d queries: plan %.3f ms, upload + kernels %.3f, results to host %.3f, unpack %.3f\n", n_queries, plan_ms,
            tc1 - tc0, tc2 - tc1, now_ms() - tc2);
  if (run.prof && run.n_items) {
    std::vector<uint64_t> hp_prof(run.n_slots * 16);
    HIP_TRY(hipMemcpy(hp_prof.data(), run.prof, hp_prof.size() * 8, hipMemcpyDeviceToHost));
    std::lock_guard<std::mutex> lk(ctx->stats_mu);
    for (size_t i = 0; i < run.n_slots; ++i)   // (slots behind the items: helpers of the MaxScore route)
      for (int j = 0; j < 16; ++j) ((i < run.n_ms_items || i >= run.n_items) ? ctx->ms_prof : ctx->prof)[j] += (double)hp_prof[i * 16 + j];
    if (run.walls) {   // the last instrumented launch's workgroups in time (nrtgpu_get_maxscore_item_walls)
      std::vector<uint64_t> w(run.n_slots * 8);
      (void)hipMemcpy(w.data(), run.walls, w.size() * 8, hipMemcpyDeviceToHost);
      ctx->last_walls.swap(w);
      ctx->last_walls_items = (int64_t)run.n_ms_items;
    }
  }
  account(ctx, slot, hp, n_queries, plan_ms, t0, queue_ms);
  return NRTGPU_OK;
}

// after_first (the coalescer's): called between the two passes with the indices of the queries that are run again -- every OTHER
// query of the batch holds its final answer in `out` at that moment, and its caller need not wait for the second pass.
static int search_batch_spec(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                             const nrtgpu_bm25_query* queries, int32_t n_queries, nrtgpu_topdocs* out,
                             const std::function<void(const std::vector<int32_t>&)>* after_first) {
  // The MaxScore route may run under SPECULATIVE thresholds here (plan.h: kHitsSpecInvalid; nrtgpu_set_speculation): a query whose
  // guess the merge could not confirm comes back tagged and is run again without speculation.  Both passes run under ONE set of
  // content locks (a set_mask / set_live_docs between them would show the re-run queries other content than their batch mates,
  // or evict a mask they name), and the second pass ignores the thread's deadline: its work is the tail of a search that was
  // launched in time, and every untagged query of the call already holds its answer.
  std::vector<int32_t> rerun;
  const bool spec = speculating(ctx, segs, n_segs);
  if (!spec) return search_batch_impl(ctx, segs, doc_bases, n_segs, queries, n_queries, out, nullptr);
  if (ctx && segs)
    for (int si = 0; si < n_segs; ++si)
      if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
  SegReadLocks content(segs, n_segs);
  const int rc = search_batch_impl(ctx, segs, doc_bases, n_segs, queries, n_queries, out, &rerun, true);
  if (rc != 0) return rc;
  const nrtgpu_diagnostics first = g_diag;
  note_speculation(ctx, segs, n_segs, n_queries, (int64_t)rerun.size());
  if (rerun.empty()) return rc;
  if (after_first) (*after_first)(rerun);
  std::vector<nrtgpu_bm25_query> rq(rerun.size());
  std::vector<nrtgpu_topdocs> ro(rerun.size());
  // The second pass starts where the first one ended (round 6).  A guess that failed was too HIGH: the first pass skipped docs it
  // should have kept -- but every hit it returned is a real doc with its real score, so a query that came back with k hits has k
  // docs at or above its k-th key, and the final k-th key cannot lie below that one.  The re-run takes "the first pass's k-th key
  // minus one" as its initial threshold (strictly below the final k-th key: keys are distinct integers, and a collector keeps only
  // what EXCEEDS its threshold) and prunes from its first posting the way a converged walk does, where an unseeded re-run --
  // no speculation, theta from zero -- was the slowest kind of walk there is: on the sorted-by-length corpus the five queries of
  // a batch that are run again cost the step 1.6 ms around a 1.1 ms kernel (profiles/r05_bench_c3_sorted.json).
  // (Only from the caller's own arrays where they hold the k-th hit: a NULL or short array seeds nothing.)
  std::vector<uint64_t> seeds(rerun.size(), 0ull);
  static const bool seed_reruns = dev_env_int("NRTGPU_SEED_RERUNS", 1) != 0;   // (development build: 0 = from zero, A/B)
  for (size_t i = 0; i < rerun.size(); ++i) {
    rq[i] = queries[rerun[i]];
    ro[i] = out[rerun[i]];
    const nrtgpu_topdocs& o = out[rerun[i]];
    const int32_t k = rq[i].k;
    if (seed_reruns && k > 0 && o.n_hits >= k && o.docs && o.scores && (o.capacity <= 0 || o.capacity >= k)) {
      const uint64_t kth = pack_key(o.scores[k - 1], (uint32_t)o.docs[k - 1]);
      seeds[i] = kth > 0 ? kth - 1 : 0ull;
    }
  }
  const int64_t deadline = g_deadline_ns;
  g_deadline_ns = 0;
  // (a few queries: a follow-up launch beside the next batch -- enqueue_search; more than that and it is a batch of its own)
  const int rc2 = search_batch_impl(ctx, segs, doc_bases, n_segs, rq.data(), (int32_t)rq.size(), ro.data(), nullptr, true, seeds.data(), rq.size() <= 32);
  g_deadline_ns = deadline;
  if (rc2 != 0) return rc2;
  for (size_t i = 0; i < rerun.size(); ++i) out[rerun[i]] = ro[i];
  nrtgpu_diagnostics d = g_diag;   // both passes
  d.total_ms += first.total_ms;
  d.plan_ms += first.plan_ms;
  d.queue_ms += first.queue_ms;
  d.device_ms += first.device_ms;
  d.postings += first.postings;
  d.queries = first.queries;
  d.items_maxscore += first.items_maxscore;
  d.items_scan += first.items_scan;
  g_diag = d;
  return NRTGPU_OK;
}

extern "C" int nrtgpu_search_bm25_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                        int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                        nrtgpu_topdocs* out) {
  return search_batch_spec(ctx, segs, doc_bases, n_segs, queries, n_queries, out, nullptr);
}

extern "C" int nrtgpu_set_shard_share(nrtgpu_ctx* ctx, int64_t shard_docs, int64_t index_docs) {
  if (!ctx || shard_docs < 0 || index_docs < 0 || (shard_docs > 0 && index_docs < shard_docs))
    return fail(NRTGPU_ERR_INVALID_ARG, "shard share: 0 <= shard_docs <= index_docs expected");
  ctx->shard_docs.store(shard_docs, std::memory_order_relaxed);
  ctx->index_docs.store(index_docs, std::memory_order_relaxed);
  return NRTGPU_OK;
}

extern "C" int nrtgpu_set_speculation(nrtgpu_ctx* ctx, float margin) {
  if (!ctx || !(margin >= 0.0f)) return fail(NRTGPU_ERR_INVALID_ARG, "speculation: a margin >= 0 expected");
  ctx->spec_z16.store((int)std::min(255.0f * 16.0f, margin * 16.0f + 0.5f), std::memory_order_relaxed);
  ctx->spec_queries.store(0, std::memory_order_relaxed);
  ctx->spec_reruns.store(0, std::memory_order_relaxed);
  ctx->spec_off.store(0, std::memory_order_relaxed);
  ctx->spec_scattered.store(0, std::memory_order_relaxed);
  ctx->spec_epoch.fetch_add(1, std::memory_order_relaxed);   // (every leaf set's verdict starts over: LeafSetCache.spec_epoch)
  return NRTGPU_OK;
}

#ifdef NRTGPU_DEV   // include/nrtgpu_dev.h (the product reports the same three values in nrtgpu_stats)
extern "C" int nrtgpu_debug_spec_counters(nrtgpu_ctx* ctx, int64_t* out3) {
  if (!ctx || !out3) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  out3[0] = ctx->spec_queries.load(std::memory_order_relaxed);
  out3[1] = ctx->spec_reruns.load(std::memory_order_relaxed);
  out3[2] = ctx->spec_off.load(std::memory_order_relaxed);
  return NRTGPU_OK;
}
#endif

// Hybrid tail: BM25 recall -> exact-vector rescore -> window, one stream, no host round trip between
// the stages (RescoreTask.java:47-50 -> QueryRescore.java:39-57 applied to the hits of
// SearchHandler.java:1412-1413).  Same results as nrtgpu_search_bm25_batch followed per query by
// nrtgpu_rescore_vectors.
static int search_hybrid_impl(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                              int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                              int32_t field_id, int32_t sim, const float* query_vectors, int32_t dim, float boost,
                              double query_weight, double rescore_weight, int32_t window, nrtgpu_topdocs* out, std::vector<int32_t>* rerun,
                              bool content_held = false) {
  forget_foreign_hip_error();
  if (!ctx || !queries || !out || !query_vectors || (n_segs > 0 && (!segs || !doc_bases)))
    return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_queries <= 0 || n_segs < 0) return fail(NRTGPU_ERR_INVALID_ARG, "n_queries must be > 0");
  if (n_queries > ctx->cfg.max_batch) return fail(NRTGPU_ERR_INVALID_ARG, "batch of %d exceeds max_batch %d", n_queries, ctx->cfg.max_batch);
  if (dim <= 0 || sim < 0 || sim > 3 || window <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad rescore arguments");
  if (!(query_weight >= 0.0) || !(rescore_weight >= 0.0) || !(boost >= 0.0f))
    return fail(NRTGPU_ERR_UNSUPPORTED, "hybrid tail: negative weights (combined scores must stay >= 0)");
  NRT_CHECK_DEADLINE("before the search was planned");
  HIP_TRY(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  HostPlan hp;
  for (int si = 0; si < n_segs; ++si)
    if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
  std::optional<SegReadLocks> content;   // until this call's kernels have finished (content_held: the caller holds them over both passes)
  if (!content_held) content.emplace(segs, n_segs);
  if (int rc = build_plan(ctx, segs, doc_bases, n_segs, queries, n_queries, hp, 1)) return rc;
  const double plan_ms = now_ms() - t0;
  for (int si = 0; si < n_segs; ++si) {
    auto fit = segs[si]->fields.find(field_id);
    if (fit != segs[si]->fields.end() && fit->second.d_vectors && fit->second.dim_user != dim)
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
    if (int rc = enqueue_search(ctx, slot, hp, n_queries, hp.k_stride, nullptr, nullptr, nullptr, &run, gpu, -1, rerun != nullptr)) {
      (void)hipStreamSynchronize(st);   // the upload of the query vectors reads the slot's pinned buffer: not in flight when the slot is released
      return rc;
    }
    launch_hybrid_rescore(st, (uint32_t)n_queries, run.out_keys, run.out_counts, hp.k_stride, (const DVecSeg*)(da + o_segs), n_segs,
                          dim, (const float*)(da + o_qv), (const float*)(da + o_qn), sim, boost, query_weight, rescore_weight,
                          (uint32_t)window, (uint64_t*)(da + o_wk), (uint32_t*)(da + o_wc), w_stride);
    HIP_TRY(hipGetLastError());
  }
  // (the answers behind the kernels on the same stream, ONE wait: see search_batch_impl)
  HIP_TRY(hipMemcpyAsync(ha + oh_k, da + o_wk, kb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ha + oh_c, da + o_wc, cb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ha + oh_fc, run.out_counts, cb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ha + oh_h, run.out_hits, hb, hipMemcpyDeviceToHost, st));
  HIP_TRY(wait_for_stream((ctx->cfg.flags & NRTGPU_FLAG_BLOCKING_WAIT) != 0, st, slot->ev_wait));
  const uint64_t* keys = (const uint64_t*)(ha + oh_k);
  const uint32_t* cnts = (const uint32_t*)(ha + oh_c);
  const uint32_t* first_cnts = (const uint32_t*)(ha + oh_fc);
  const uint64_t* hits = (const uint64_t*)(ha + oh_h);
  for (int qi = 0; qi < n_queries; ++qi) {
    // QueryRescorer keeps the first pass's TotalHits; the window only trims the hits
    unpack_topdocs(keys + (size_t)qi * w_stride, cnts[qi], hits[qi] & ~kHitsSpecInvalid, std::min<int32_t>(window, NRTGPU_MAX_K), hp.q_lower[(size_t)qi],
                   first_cnts[qi] == (uint32_t)queries[qi].k ? (uint32_t)std::min<int32_t>(window, NRTGPU_MAX_K) : 0xFFFFFFFFu, &out[qi]);
    // (a first pass whose speculative threshold failed the merge's check fed the tail a wrong recall set: the query is run again)
    if (rerun && (hits[qi] & kHitsSpecInvalid)) rerun->push_back(qi);
  }
  account(ctx, slot, hp, n_queries, plan_ms);
  return NRTGPU_OK;
}

extern "C" int nrtgpu_search_hybrid_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                          int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                          int32_t field_id, int32_t sim, const float* query_vectors, int32_t dim, float boost,
                                          double query_weight, double rescore_weight, int32_t window, nrtgpu_topdocs* out) {
  // The first pass may run under speculative thresholds (plan.h: kHitsSpecInvalid), as in nrtgpu_search_bm25_batch: recall, tail
  // and the copy back stay one stream with no host round trip; the merge's tags arrive with the results, and a tagged query --
  // its recall set may lack docs -- is run again, first pass and tail, without speculation.
  // Both passes under one set of content locks, the second one without the thread's deadline: as nrtgpu_search_bm25_batch.
  std::vector<int32_t> rerun;
  const bool spec = speculating(ctx, segs, n_segs);
  if (!spec)
    return search_hybrid_impl(ctx, segs, doc_bases, n_segs, queries, n_queries, field_id, sim, query_vectors, dim, boost, query_weight,
                              rescore_weight, window, out, nullptr);
  if (ctx && segs)
    for (int si = 0; si < n_segs; ++si)
      if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
  SegReadLocks content(segs, n_segs);
  const int rc = search_hybrid_impl(ctx, segs, doc_bases, n_segs, queries, n_queries, field_id, sim, query_vectors, dim, boost, query_weight,
                                    rescore_weight, window, out, &rerun, true);
  if (rc != 0) return rc;
  note_speculation(ctx, segs, n_segs, n_queries, (int64_t)rerun.size());
  if (rerun.empty()) return rc;
  std::vector<nrtgpu_bm25_query> rq(rerun.size());
  std::vector<nrtgpu_topdocs> ro(rerun.size());
  std::vector<float> rv(rerun.size() * (size_t)dim);
  for (size_t i = 0; i < rerun.size(); ++i) {
    rq[i] = queries[rerun[i]];
    ro[i] = out[rerun[i]];
    memcpy(rv.data() + i * (size_t)dim, query_vectors + (size_t)rerun[i] * (size_t)dim, (size_t)dim * sizeof(float));
  }
  const int64_t deadline = g_deadline_ns;
  g_deadline_ns = 0;
  const int rc2 = search_hybrid_impl(ctx, segs, doc_bases, n_segs, rq.data(), (int32_t)rq.size(), field_id, sim, rv.data(), dim, boost, query_weight,
                                     rescore_weight, window, ro.data(), nullptr, true);
  g_deadline_ns = deadline;
  if (rc2 != 0) return rc2;
  for (size_t i = 0; i < rerun.size(); ++i) out[rerun[i]] = ro[i];
  return NRTGPU_OK;
}

int nrtgpu::rt::hybrid_tail_on_device(nrtgpu_ctx* ctx, Slot* slot, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                      int32_t field_id, int32_t sim, const float* query_vectors, int32_t dim, float boost, double qw, double rw,
                                      int32_t window, int32_t n_queries, const void* d_first_keys, const void* d_first_counts,
                                      int32_t k_stride, int32_t drop_foreign, void* d_win_keys, void* d_win_counts, uint32_t w_stride) {
  forget_foreign_hip_error();
  const size_t nq = (size_t)n_queries;
  PaddedQueries padded;   // (rows are resident padded to a multiple of 16 elements: the queries likewise)
  if (int rc = pad_query_vectors(segs, n_segs, field_id, query_vectors, n_queries, dim, &padded)) return rc;
  query_vectors = padded.p;
  dim = padded.dim;
  Carver ac;
  const size_t o_segs = ac.take((size_t)std::max(n_segs, 1) * sizeof(DVecSeg)), o_qv = ac.take(nq * (size_t)dim * 4), o_qn = ac.take(nq * 4);
  if (int rc = slot->d_aux.reserve(ac.off)) return rc;
  if (int rc = slot->h_aux.reserve(ac.off)) return rc;
  char* ha = (char*)slot->h_aux.p;
  char* da = (char*)slot->d_aux.p;
  DVecSeg* hs = (DVecSeg*)(ha + o_segs);
  for (int si = 0; si < n_segs; ++si) {
    DVecSeg v{};
    auto fit = segs[si]->fields.find(field_id);
    if (fit != segs[si]->fields.end() && fit->second.d_vectors) {
      if (fit->second.dim != dim) return fail(NRTGPU_ERR_INVALID_ARG, "vector dimension mismatch");
      v.vecs = fit->second.d_vectors;
      v.vnorm2 = fit->second.d_vnorm2;
      v.ord_to_doc = fit->second.d_ord_to_doc;
      v.n_vec = fit->second.n_vec;
    }
    v.doc_base = doc_bases[si];
    v.max_doc = segs[si]->max_doc;
    hs[si] = v;
  }
  memcpy(ha + o_qv, query_vectors, nq * (size_t)dim * 4);
  float* hqn = (float*)(ha + o_qn);
  for (size_t q = 0; q < nq; ++q) {  // |q|^2 in the order nrtgpu_rescore_vectors uses
    const float* qv = query_vectors + q * (size_t)dim;
    float qn = 0.f;
    for (int d = 0; d < dim; ++d) {
      volatile float p2 = qv[d] * qv[d];
      qn = qn + p2;
    }
    hqn[q] = qn;
  }
  HIP_TRY(hipMemcpyAsync(da, ha, ac.off, hipMemcpyHostToDevice, slot->stream));
  launch_hybrid_rescore(slot->stream, (uint32_t)n_queries, (const uint64_t*)d_first_keys, (const uint32_t*)d_first_counts, (uint32_t)k_stride,
                        (const DVecSeg*)(da + o_segs), n_segs, dim, (const float*)(da + o_qv), (const float*)(da + o_qn), sim, boost, qw, rw,
                        (uint32_t)window, (uint64_t*)d_win_keys, (uint32_t*)d_win_counts, w_stride, drop_foreign);
  HIP_TRY(hipGetLastError());
  return NRTGPU_OK;
}

int nrtgpu::rt::merge_lists_on_device(nrtgpu_ctx* ctx, Slot* slot, int32_t n_lists, int32_t n_queries, int32_t k_stride, const void* g_keys,
                                      const void* g_counts, const void* g_hits, const int32_t* ks, void* d_keys, void* d_counts, void* d_hits) {
  forget_foreign_hip_error();
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
  char* db = (char*)slot->d_plan.p;
  HIP_TRY(hipMemcpyAsync(db, hb, pc.off, hipMemcpyHostToDevice, slot->stream));
  launch_merge_topk(slot->stream, (uint32_t)n_queries, (const uint64_t*)g_keys, (const uint32_t*)g_counts, (const uint64_t*)g_hits,
                    (const uint32_t*)(db + o_lidx), (const uint32_t*)(db + o_qbase), (const uint32_t*)(db + o_qnl), (uint32_t)k_stride,
                    (const uint32_t*)(db + o_qk), (uint64_t*)d_keys, (uint32_t*)d_counts, (uint64_t*)d_hits, (uint32_t)k_stride);
  HIP_TRY(hipGetLastError());
  return NRTGPU_OK;
}

extern "C" int nrtgpu_query_supported(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, int32_t n_segs, const nrtgpu_bm25_query* q) {
  if (!ctx || !q || (n_segs > 0 && !segs)) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_segs < 0) return fail(NRTGPU_ERR_INVALID_ARG, "n_segs must be >= 0");
  for (int si = 0; si < n_segs; ++si)
    if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
  HIP_TRY(hipSetDevice(ctx->device));
  SegReadLocks content(segs, n_segs);
  HostPlan hp;   // the planner is the predicate: whatever it accepts, the kernels run
  return build_plan(ctx, segs, nullptr, n_segs, q, 1, hp, 1);
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
  int64_t deadline_ns = 0;        // the calling thread's (nrtgpu_set_thread_deadline_ns)
  std::vector<int32_t> slices;    // the calling thread's slice of every leaf (nrtgpu_set_thread_slices; empty: the library slices the leaves itself)
  nrtgpu_diagnostics diag{};      // of the batch the request travelled in
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
  if (a->doc_bases && memcmp(a->doc_bases, b->doc_bases, (size_t)a->n_segs * 4) != 0) return false;
  // The batch is planned on its LEADER's thread, under the leader's slices (planner.cpp reads the planning thread's
  // nrtgpu_set_thread_slices): a follower travels with a leader only if its own slices are the same -- two searcher versions that
  // share a resident subset, or virtual shards dealt over different leaf lists, slice the same leaves differently, and the
  // per-slice hit counts (the totalHits relation, the route) follow the slices.
  return a->slices == b->slices;
}

extern "C" int nrtgpu_set_coalescing(nrtgpu_ctx* ctx, int32_t linger_us) {
  if (!ctx || linger_us < 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad coalescing arguments");
  std::lock_guard<std::mutex> lk(ctx->co_mu);
  ctx->co_linger_us = linger_us;
  return NRTGPU_OK;
}

#ifdef NRTGPU_DEV
// Test hook (include/nrtgpu_dev.h): while held, a coalescer's leader leaves only with a full batch / panel, so a test can queue
// a known set of callers and assert what batches they form by construction instead of by the host's speed.
extern "C" int nrtgpu_debug_hold_coalescers(nrtgpu_ctx* ctx, int32_t hold) {
  if (!ctx) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  {
    std::lock_guard<std::mutex> lk(ctx->co_mu);
    ctx->co_hold = hold != 0 ? 1 : 0;   // (atomic: the kNN coalescer reads it under its own lock)
    if (!hold && ctx->co_leader) ctx->co_leader->cv.notify_one();
  }
  nrtgpu_debug_knn_coalescer_wake(ctx);
  return NRTGPU_OK;
}
extern "C" int nrtgpu_debug_coalescer_pending(nrtgpu_ctx* ctx, int32_t which) {
  if (!ctx) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (which == 0) {
    std::lock_guard<std::mutex> lk(ctx->co_mu);
    return (int)ctx->co_pending.size();
  }
  return nrtgpu_debug_knn_coalescer_pending(ctx);
}
#endif

extern "C" int nrtgpu_search_bm25_coalesced(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                            int32_t n_segs, const nrtgpu_bm25_query* q, nrtgpu_topdocs* out) {
  if (!ctx || !q || !out || (n_segs > 0 && !segs)) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_segs < 0) return fail(NRTGPU_ERR_INVALID_ARG, "n_segs must be >= 0");
  if (int rc = validate_query(*q, 0)) return rc;  // a bad request must not fail its batch mates
  // (minimumNumberShouldMatch > 1 / DisjunctionMaxQuery: on the MaxScore route they run next to anything; where one of them needs
  //  the exhaustive scan's count-carrying variant and a batch mate forces fp64 sums, the batch fails as a whole and is re-run
  //  member by member below: only the offender sees NRTGPU_ERR_UNSUPPORTED)
  for (int si = 0; si < n_segs; ++si) {
    if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
    if (!segs[si]->sealed) return fail(NRTGPU_ERR_STATE, "segment %d is not sealed", si);
    if (segs[si]->ctx != ctx) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d belongs to another context", si);
  }
  NRT_CHECK_DEADLINE("before the request was queued");
  CoRequest me{segs, doc_bases, n_segs, q, out};
  me.deadline_ns = g_deadline_ns;
  if ((int32_t)g_thread_slices.size() == n_segs) me.slices = g_thread_slices;   // (a list of another length is ignored by the planner as well)
  std::vector<CoRequest*> batch;
  {
    std::unique_lock<std::mutex> lk(ctx->co_mu);
    ctx->co_pending.push_back(&me);
    if (ctx->co_leader) {  // follower: the lingering leader takes this request (or a later one does)
      // (wake it when the queue is full -- or when, with the device idle, as many callers wait as the last batch held: the cohort
      //  of a closed loop has come back, lingering longer only adds latency)
      if ((int32_t)ctx->co_pending.size() >= ctx->cfg.max_batch ||
          (ctx->co_inflight == 0 && ctx->co_last_batch > 1 && ctx->co_last_batch < 192 && (int32_t)ctx->co_pending.size() >= ctx->co_last_batch))
        ctx->co_leader->cv.notify_one();
      lk.unlock();
      {
        std::unique_lock<std::mutex> mine(me.m);
        me.cv.wait(mine, [&] { return me.done || me.lead; });
      }
      if (me.done) {
        if (me.rc != 0) g_last_error = me.err;
        else g_diag = me.diag;
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
      // The cohort is back: with nothing in flight, as many callers wait as the last batch held -- callers in a closed loop return
      // together, within tens of microseconds of their batch's end -- so the batch leaves now instead of at the linger's end (the
      // linger stays the bound for arrivals that are no cohort).  NRTGPU_CO_COHORT=0 (development build): the linger alone, A/B.
      static const bool cohort_rule = dev_env_int("NRTGPU_CO_COHORT", 1) != 0;
      // Measured (profiles/r05_coalescer_cohort_ab.log, C3): 8 callers 13.7 k -> 20.8 k queries/s, p50 0.58 -> 0.38 ms; 64 callers
      // 82 k -> 98 k, 0.77 -> 0.64 ms; C2: 8 callers 19.5 k -> 38.2 k, 64 callers 128 k -> 174 k.  Cohorts of kCoOverlapMin and more
      // are the two-batches-in-flight regime below, which the rule leaves alone (512 callers: 306 k before, 298 k with the rule
      // applied to them too).
      if (cohort_rule && !ctx->co_hold && ctx->co_inflight == 0 && ctx->co_last_batch > 1 && ctx->co_last_batch < kCoOverlapMin &&
          waiting >= ctx->co_last_batch)
        break;
      const bool late = std::chrono::steady_clock::now() >= deadline;
      if (late && !ctx->co_hold && (ctx->co_inflight == 0 ||
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
    std::vector<CoRequest*> rest, expired;
    batch.push_back(&me);
    for (CoRequest* r : ctx->co_pending) {
      if (r == &me) continue;
      if (deadline_passed(r->deadline_ns)) expired.push_back(r);   // waited too long: leaves without being searched
      else if ((int32_t)batch.size() < cap && same_leaves(&me, r)) batch.push_back(r);
      else rest.push_back(r);
    }
    for (CoRequest* r : expired) {
      std::lock_guard<std::mutex> theirs(r->m);
      r->rc = NRTGPU_ERR_TIMEOUT;
      r->err = "deadline passed while the request waited for a batch";
      r->done = true;
      r->cv.notify_one();
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
  // (a batch of several requests is not failed for the leader's deadline: its mates have not expired -- they were checked when the
  //  batch was formed -- and a batch is a few milliseconds.  A lone request keeps its deadline.)
  struct DeadlineScope {
    int64_t saved;
    explicit DeadlineScope(bool clear) : saved(g_deadline_ns) { if (clear) g_deadline_ns = 0; }
    ~DeadlineScope() { g_deadline_ns = saved; }
  } deadline_scope(batch.size() > 1);
  // A member whose query is run again (a speculative threshold failed the merge's check) must not hold up its batch mates: they
  // are woken between the two passes with the answer they already have.
  std::vector<char> released(batch.size(), 0);
  const std::function<void(const std::vector<int32_t>&)> after_first = [&](const std::vector<int32_t>& rerun) {
    std::vector<char> again(batch.size(), 0);
    for (int32_t qi : rerun) again[(size_t)qi] = 1;
    const nrtgpu_diagnostics first_diag = g_diag;
    for (size_t i = 1; i < batch.size(); ++i) {   // (batch[0] is this caller: it runs the second pass)
      if (again[i]) continue;
      CoRequest* r = batch[i];
      std::lock_guard<std::mutex> theirs(r->m);
      *r->out = outs[i];
      r->diag = first_diag;
      r->rc = 0;
      r->done = true;
      r->cv.notify_one();
      released[i] = 1;
    }
  };
  int rc = search_batch_spec(ctx, segs, doc_bases, n_segs, qs.data(), (int32_t)qs.size(), outs.data(), &after_first);
  std::string err = rc ? g_last_error : std::string();
  const nrtgpu_diagnostics batch_diag = g_diag;
  // A request the planner rejects (a mask that is not resident on a leaf, a clause shape outside the fixed-point range ...)
  // must not fail its batch mates: the batch is re-run member by member and only the offender gets the error.
  std::vector<int> rcs(batch.size(), rc);
  std::vector<std::string> errs(batch.size(), err);
  if (rc != 0 && batch.size() > 1) {
    for (size_t i = 0; i < batch.size(); ++i) {
      if (released[i]) continue;   // (left with its answer between the passes)
      rcs[i] = nrtgpu_search_bm25_batch(ctx, segs, doc_bases, n_segs, &qs[i], 1, &outs[i]);
      errs[i] = rcs[i] ? g_last_error : std::string();
    }
    rc = rcs[0];   // batch[0] is this caller
    err = errs[0];
  }
  for (size_t i = 0; i < batch.size(); ++i) {
    CoRequest* r = batch[i];
    if (r == &me) {
      if (rcs[i] == 0) *out = outs[i];
      continue;
    }
    if (released[i]) continue;   // (gone already: its request object may no longer exist)
    // (notified under the request's own lock: the woken caller cannot return -- and free its request -- before
    // we are done with it, and it contends with nobody but us)
    std::lock_guard<std::mutex> theirs(r->m);
    if (rcs[i] == 0) *r->out = outs[i];
    r->diag = batch_diag;
    r->rc = rcs[i];
    if (rcs[i] != 0) r->err = errs[i];
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

extern "C" int nrtgpu_search_bm25_batch_device(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                               int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                               int32_t k_stride, void* d_keys, void* d_counts, void* d_hits) {
  return nrtgpu_search_bm25_batch_device_epoch(ctx, segs, doc_bases, n_segs, queries, n_queries, k_stride, d_keys, d_counts,
                                               d_hits, -1);
}

// A device-resident search in flight (nrtgpu_search_bm25_batch_device_begin): what must outlive the enqueue -- the plan (its
// resident term tables), the workspace, the shared locks on the segments' content -- until nrtgpu_pending_wait.
struct nrtgpu_pending {
  nrtgpu_ctx* ctx = nullptr;
  Slot* slot = nullptr;
  HostPlan hp;
  std::unique_ptr<SegReadLocks> content;
  int32_t n_queries = 0;
  double plan_ms = 0.0;
  // speculative thresholds (plan.h: kHitsSpecInvalid) on this path: the launch ran under them; nrtgpu_pending_wait reads the merged
  // counts' tags and, if a guess failed, runs the batch again without speculation into the same buffers before it returns
  bool speculated = false;
  bool shard_spec = false;      // one shard of the library's multi-GPU search: guesses against the global k-th score, checked by dist.cpp
  uint32_t k_stride = 0;
  uint64_t* d_keys = nullptr;
  uint32_t* d_counts = nullptr;
  uint64_t* d_hits = nullptr;
  // the enqueue of a search begun through the context's launcher thread (runtime_internal.h: Launcher): done, how it went (shared
  // with the launcher's job: the handle may be freed the moment the job has said "done")
  struct Enqueue {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int rc = 0;
    std::string err;
  };
  std::shared_ptr<Enqueue> enq;   // null: enqueued by the thread that began it
};

static int device_begin_impl(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                             const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t k_stride, void* d_keys, void* d_counts, void* d_hits,
                             int64_t epoch, int32_t spec_world, uint64_t* d_guess, nrtgpu_pending** out, bool through_launcher = false) {
  if (!ctx || !queries || !d_keys || !d_counts || !d_hits || !out || (n_segs > 0 && !segs)) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  *out = nullptr;
  NRT_CHECK_DEADLINE("before the search was planned");
  if (n_queries <= 0 || n_segs < 0) return fail(NRTGPU_ERR_INVALID_ARG, "n_queries must be > 0");
  if (n_queries > ctx->cfg.max_batch) return fail(NRTGPU_ERR_INVALID_ARG, "batch of %d exceeds max_batch %d", n_queries, ctx->cfg.max_batch);
  HIP_TRY(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  for (int si = 0; si < n_segs; ++si)
    if (!segs[si]) return fail(NRTGPU_ERR_INVALID_ARG, "segment %d is NULL", si);
  auto p = std::make_unique<nrtgpu_pending>();
  p->ctx = ctx;
  p->n_queries = n_queries;
  p->content = std::make_unique<SegReadLocks>(segs, n_segs, true);  // until this call's kernels have finished (nrtgpu_pending_wait)
  // (both scorers take part in a bound exchange that is open: nrtgpu_exchange_open)
  if (int rc = build_plan(ctx, segs, doc_bases, n_segs, queries, n_queries, p->hp, 2)) return rc;
  if (k_stride < (int32_t)p->hp.k_stride && k_stride < NRTGPU_MAX_K) {
    for (int qi = 0; qi < n_queries; ++qi)
      if (queries[qi].k > k_stride) return fail(NRTGPU_ERR_INVALID_ARG, "k_stride %d smaller than numHits %d", k_stride, queries[qi].k);
  }
  p->plan_ms = now_ms() - t0;
  acquire_slot(ctx, &p->slot);
  // Speculative thresholds here too (round 5): every shard of a multi-GPU search computes ITS top-k under them, the tags are read
  // in nrtgpu_pending_wait.  Not next to the cross-GPU bound exchange (epoch >= 0 with an exchange open: its quantile uses the
  // selection's second rank).
  // (spec_world: 1 = this call alone, checked and re-run in nrtgpu_pending_wait; >= 2 = one shard of that many, checked by the
  //  caller; 0 = no speculation)
  p->speculated = spec_world != 0 && epoch < 0 && p->hp.lsc && spec_margin16(ctx) != 0u && (spec_sync_epoch(ctx, p->hp.lsc.get()), p->hp.lsc->spec_off.load(std::memory_order_relaxed) == 0) &&
                  p->hp.n_ms_items != 0;
  p->shard_spec = spec_world > 1;   // (then nrtgpu_pending_wait neither reads tags nor re-runs: the caller checks the guesses)
  p->k_stride = (uint32_t)k_stride;
  p->d_keys = (uint64_t*)d_keys;
  p->d_counts = (uint32_t*)d_counts;
  p->d_hits = (uint64_t*)d_hits;
  static const bool use_launcher = dev_env_int("NRTGPU_LAUNCHER", 1) != 0;   // (development build: 0 = the caller enqueues, A/B)
  if (through_launcher && use_launcher) {
    // the caller goes on planning its next batch; the launcher packs and enqueues this one (in the order of the _begin calls).
    // What the enqueue says is read by nrtgpu_pending_wait.
    Launcher* l = nullptr;
    {
      std::lock_guard<std::mutex> lk(ctx->launcher_mu);
      if (!ctx->launcher) ctx->launcher = std::make_unique<Launcher>(ctx->device);
      l = ctx->launcher.get();
    }
    nrtgpu_pending* pp = p.release();
    std::shared_ptr<nrtgpu_pending::Enqueue> enq = pp->enq = std::make_shared<nrtgpu_pending::Enqueue>();
    const int32_t sw = std::max(spec_world, 1);
    l->push([pp, enq, n_queries, k_stride, epoch, sw, d_guess] {
      DeviceRun run;
      int rc;
      {
        std::unique_lock<std::mutex> gpu(pp->ctx->gpu_mu, std::defer_lock);
        rc = enqueue_search(pp->ctx, pp->slot, pp->hp, n_queries, (uint32_t)k_stride, pp->d_keys, pp->d_counts, pp->d_hits, &run, gpu, epoch,
                            pp->speculated, sw, d_guess);
      }
      std::lock_guard<std::mutex> lk(enq->mu);   // (from here on `pp` may be gone)
      enq->rc = rc;
      if (rc) enq->err = g_last_error;
      enq->done = true;
      enq->cv.notify_all();
    });
    *out = pp;
    return NRTGPU_OK;
  }
  DeviceRun run;
  {
    std::unique_lock<std::mutex> gpu(ctx->gpu_mu, std::defer_lock);
    if (int rc = enqueue_search(ctx, p->slot, p->hp, n_queries, (uint32_t)k_stride, (uint64_t*)d_keys, (uint32_t*)d_counts,
                                (uint64_t*)d_hits, &run, gpu, epoch, p->speculated, std::max(spec_world, 1), d_guess)) {
      (void)hipStreamSynchronize(p->slot->stream);   // whatever was enqueued reads the slot's buffers
      release_slot(ctx, p->slot);
      return rc;
    }
  }
  *out = p.release();
  return NRTGPU_OK;
}

extern "C" int nrtgpu_search_bm25_batch_device_begin(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                                     int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                                     int32_t k_stride, void* d_keys, void* d_counts, void* d_hits, int64_t epoch,
                                                     nrtgpu_pending** out) {
  return device_begin_impl(ctx, segs, doc_bases, n_segs, queries, n_queries, k_stride, d_keys, d_counts, d_hits, epoch, 1, nullptr, out, true);
}

extern "C" int nrtgpu_search_bm25_shard_device_begin(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                                     const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t k_stride, void* d_keys,
                                                     void* d_counts, void* d_hits, int32_t spec_world, void* d_guess, nrtgpu_pending** out) {
  if (spec_world >= 2 && !d_guess) return fail(NRTGPU_ERR_INVALID_ARG, "a shard of %d needs a place for its guesses", spec_world);
  if (spec_world < 0 || spec_world > 4096) return fail(NRTGPU_ERR_INVALID_ARG, "spec_world %d", spec_world);
  return device_begin_impl(ctx, segs, doc_bases, n_segs, queries, n_queries, k_stride, d_keys, d_counts, d_hits, -1, spec_world >= 2 ? spec_world : 0,
                           (uint64_t*)d_guess, out, true);
}

extern "C" int nrtgpu_note_shard_speculation(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, int32_t n_segs, int32_t n_queries, int32_t n_failed) {
  if (!ctx || (n_segs > 0 && !segs) || n_queries < 0 || n_failed < 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad arguments");
  nrtgpu::rt::note_shard_speculation(ctx, segs, n_segs, n_queries, n_failed);
  return NRTGPU_OK;
}

extern "C" int nrtgpu_pending_wait(nrtgpu_pending* pending);
// One shard of the library's multi-GPU search (dist.cpp): device-resident results as nrtgpu_search_bm25_batch_device, with the
// speculative thresholds guessed against the WHOLE search's k-th score (spec_world shards of equal docid ranges) and the largest
// guess per query left in d_guess (0: none) for the caller to check against the merged list.  spec_world <= 1: no speculation
// at all (the re-run of the queries whose guess failed).  *speculated: whether guesses were made (the leaf set's verdict allows).
int nrtgpu::rt::search_bm25_shard_device(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                         const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t k_stride, void* d_keys, void* d_counts,
                                         void* d_hits, int32_t spec_world, void* d_guess, bool* speculated) {
  nrtgpu_pending* p = nullptr;
  if (int rc = device_begin_impl(ctx, segs, doc_bases, n_segs, queries, n_queries, k_stride, d_keys, d_counts, d_hits, -1, spec_world >= 2 ? spec_world : 0,
                                 (uint64_t*)d_guess, &p))
    return rc;
  if (speculated) *speculated = p->speculated;
  return nrtgpu_pending_wait(p);
}

extern "C" int nrtgpu_pending_wait(nrtgpu_pending* pending) {
  if (!pending) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  std::unique_ptr<nrtgpu_pending> p(pending);
  nrtgpu_ctx* ctx = p->ctx;
  (void)hipSetDevice(ctx->device);
  if (p->enq) {   // begun through the launcher: its enqueue first
    std::unique_lock<std::mutex> lk(p->enq->mu);
    p->enq->cv.wait(lk, [&] { return p->enq->done; });
    if (p->enq->rc) {
      lk.unlock();
      (void)hipStreamSynchronize(p->slot->stream);   // whatever was enqueued reads the slot's buffers
      release_slot(ctx, p->slot);
      return fail(p->enq->rc, "%s", p->enq->err.c_str());
    }
  }
  hipError_t e = wait_for_stream((ctx->cfg.flags & NRTGPU_FLAG_BLOCKING_WAIT) != 0, p->slot->stream, p->slot->ev_wait);
  if (e == hipSuccess && p->speculated && !p->shard_spec) {
    // the merge tagged the queries whose guess it could not confirm (kHitsSpecInvalid in the hit totals): read the totals, clear the
    // tags for the caller, and -- if any -- run the batch again without speculation into the same buffers (the plan and the
    // workspace are still this call's).  A re-run is a second pass of the batch; the leaf set's verdict bounds how often.
    // (A batch with a failed guess is run again WHOLE: the handle keeps the batch's plan, not the caller's queries -- those were
    //  borrowed for the _begin call only -- so there is nothing to plan a sub-batch from.  The synchronous entries, which still hold
    //  the queries, re-run the tagged ones alone: search_batch_spec.)
    const size_t nq = (size_t)p->n_queries;
    int64_t n_bad = 0;
    if (int rc = p->slot->h_out.reserve(nq * 8)) {   // (the tags could not be read: the results are unverified, not an answer)
      release_slot(ctx, p->slot);
      return rc;
    }
    uint64_t* hh = (uint64_t*)p->slot->h_out.p;
    e = hipMemcpyAsync(hh, p->d_hits, nq * 8, hipMemcpyDeviceToHost, p->slot->stream);
    if (e == hipSuccess) e = wait_for_stream((ctx->cfg.flags & NRTGPU_FLAG_BLOCKING_WAIT) != 0, p->slot->stream, p->slot->ev_wait);
    if (e == hipSuccess) {
      for (size_t q = 0; q < nq; ++q)
        if (hh[q] & kHitsSpecInvalid) {
          hh[q] &= ~kHitsSpecInvalid;
          ++n_bad;
        }
      note_speculation_of(ctx, p->hp.lsc.get(), (int64_t)nq, n_bad);
      if (n_bad) {
        DeviceRun run;
        std::unique_lock<std::mutex> gpu(ctx->gpu_mu, std::defer_lock);
        if (int rc = enqueue_search(ctx, p->slot, p->hp, p->n_queries, p->k_stride, p->d_keys, p->d_counts, p->d_hits, &run, gpu, -1, false)) {
          gpu = std::unique_lock<std::mutex>();   // (its message is this thread's g_last_error: kept)
          (void)hipStreamSynchronize(p->slot->stream);
          release_slot(ctx, p->slot);
          return rc;
        }
        e = wait_for_stream((ctx->cfg.flags & NRTGPU_FLAG_BLOCKING_WAIT) != 0, p->slot->stream, p->slot->ev_wait);
      }
    }
  }
  if (e == hipSuccess) account(ctx, p->slot, p->hp, p->n_queries, p->plan_ms);   // (fills the WAITING thread's diagnostics)
  release_slot(ctx, p->slot);
  if (e != hipSuccess) return fail(NRTGPU_ERR_HIP, "device-resident search failed: %s", hipGetErrorString(e));
  return NRTGPU_OK;
}

extern "C" int nrtgpu_search_bm25_batch_device_epoch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                                     int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                                     int32_t k_stride, void* d_keys, void* d_counts, void* d_hits,
                                                     int64_t epoch) {
  nrtgpu_pending* p = nullptr;   // (the synchronous form enqueues itself: a hand-off to the launcher thread would only add its latency)
  if (int rc = device_begin_impl(ctx, segs, doc_bases, n_segs, queries, n_queries, k_stride, d_keys, d_counts, d_hits, epoch, 1, nullptr, &p)) return rc;
  return nrtgpu_pending_wait(p);
}

extern "C" int nrtgpu_merge_topk_device(nrtgpu_ctx* ctx, int32_t n_lists, int32_t n_queries, int32_t k_stride,
                                        const void* d_keys_in, const void* d_counts_in, const void* d_hits_in,
                                        const int32_t* ks, const int32_t* total_hits_thresholds, nrtgpu_topdocs* out) {
  return nrtgpu::rt::merge_topk_device_kth(ctx, n_lists, n_queries, k_stride, d_keys_in, d_counts_in, d_hits_in, ks, total_hits_thresholds, out, nullptr);
}

// nrtgpu_merge_topk_device, and per query the packed key of rank k of the MERGED list (kth[q]; 0: the list is shorter than k) --
// read from the merged keys themselves, not from the caller's output arrays: those may be absent or shorter than k
// (unpack_topdocs copies what fits), and a verdict taken from them would differ between ranks whose callers sized them
// differently (dist.cpp: the check of the shards' speculative thresholds).
int nrtgpu::rt::merge_topk_device_kth(nrtgpu_ctx* ctx, int32_t n_lists, int32_t n_queries, int32_t k_stride, const void* d_keys_in,
                                      const void* d_counts_in, const void* d_hits_in, const int32_t* ks, const int32_t* total_hits_thresholds,
                                      nrtgpu_topdocs* out, uint64_t* kth) {
  forget_foreign_hip_error();
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
  HIP_TRY(wait_for_stream((ctx->cfg.flags & NRTGPU_FLAG_BLOCKING_WAIT) != 0, st, slot->ev_wait));
  const uint64_t* keys = (const uint64_t*)(ho + o_okeys);
  const uint32_t* cnts = (const uint32_t*)(ho + o_ocnt);
  const uint64_t* hits = (const uint64_t*)(ho + o_ohits);
  {
    const int n_chunks = n_queries >= 64 ? std::min(std::min(8, n_queries / 32), ctx->pool->helpers() + 1) : 1;
    ctx->pool->run(n_chunks, [&](int c) {
      const int q0 = (int)((int64_t)n_queries * c / n_chunks), q1 = (int)((int64_t)n_queries * (c + 1) / n_chunks);
      for (int qi = q0; qi < q1; ++qi) {
        unpack_topdocs(keys + (size_t)qi * k_stride, cnts[qi], hits[qi], ks[qi], 0, cnts[qi], &out[qi]);
        if (kth) kth[qi] = cnts[qi] >= (uint32_t)ks[qi] ? keys[(size_t)qi * k_stride + (size_t)ks[qi] - 1u] : 0ull;
      }
    });
  }
  return NRTGPU_OK;
}

