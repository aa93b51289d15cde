// vectors.cpp -- exact vector search (the knn request path and ExactVectorQuery) and the vector rescorer.
#include "runtime_internal.h"

// ------------------------------------------------------------------------------------------------
// ABI: exact vector search / vector rescore
// ------------------------------------------------------------------------------------------------
static const uint32_t kKnnCap = 1u << 18;   // candidate keys per query and round (2 MiB)
static const int kKnnMaxQ = 64;   // queries per pass over the rows: two panels of 32 on paired workgroups (knn.hip)

// Shared by the two vector entry points.  knn_request = false: ExactVectorQuery (every doc with a vector
// matches, boost inside the score).  knn_request = true: the `knn` request path -- pre-filter mask, score
// threshold on the unboosted score (MinThresholdQuery's MinScoreWrapper), boost applied afterwards,
// totalHits = the docs returned.
int nrtgpu::rt::pad_query_vectors(const nrtgpu_seg* const* segs, int32_t n_segs, int32_t field_id, const float* queries, int32_t n,
                                  int32_t dim_user, PaddedQueries* out) {
  int32_t dim_dev = (dim_user + 15) & ~15;
  for (int si = 0; si < n_segs; ++si) {
    if (!segs[si]) continue;
    auto fit = segs[si]->fields.find(field_id);
    if (fit == segs[si]->fields.end() || !fit->second.d_vectors) continue;
    if (fit->second.dim_user != dim_user)
      return fail(NRTGPU_ERR_INVALID_ARG, "segment %d: field %d has dimension %d, query has %d", si, field_id, fit->second.dim_user, dim_user);
    dim_dev = fit->second.dim;
  }
  out->dim = dim_dev;
  if (dim_dev == dim_user) {
    out->p = queries;
    return NRTGPU_OK;
  }
  out->buf.assign((size_t)n * (size_t)dim_dev, 0.0f);
  for (int32_t q = 0; q < n; ++q) memcpy(out->buf.data() + (size_t)q * dim_dev, queries + (size_t)q * dim_user, (size_t)dim_user * 4);
  out->p = out->buf.data();
  return NRTGPU_OK;
}

static int knn_impl(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                    int32_t field_id, int32_t sim, const float* queries, int32_t n_queries, int32_t dim,
                    int32_t k, float boost, bool knn_request, int32_t filter_mask, float min_score, nrtgpu_topdocs* out,
                    char* ext_keys = nullptr, char* ext_cnts = nullptr, char* ext_hits = nullptr) {
  forget_foreign_hip_error();
  // ext_*: device-resident results instead of `out` (the multi-GPU path): per query k_stride sorted keys, its count, and the live
  // vectors of these leaves as the hit total -- the layout the exchange stage takes
  if (!ctx || !queries || (!out && !ext_keys) || (n_segs > 0 && !segs)) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n_queries <= 0 || k <= 0 || dim <= 0 || sim < 0 || sim > 3) return fail(NRTGPU_ERR_INVALID_ARG, "bad knn arguments");
  if (k > NRTGPU_MAX_K) return fail(NRTGPU_ERR_UNSUPPORTED, "k %d > %d", k, NRTGPU_MAX_K);
  if (dim > 2048) return fail(NRTGPU_ERR_UNSUPPORTED, "vector dimension %d (device path takes <= 2048)", dim);
  NRT_CHECK_DEADLINE("before the vector search started");
  HIP_TRY(hipSetDevice(ctx->device));
  for (int si = 0; si < n_segs; ++si)
    if (!segs[si]) return fail(NRTGPU_ERR_STATE, "segment %d missing or not sealed", si);
  SegReadLocks content(segs, n_segs);  // liveDocs / masks stay as they are until the kernels have finished
  for (int si = 0; si < n_segs; ++si)
    if (!segs[si] || !segs[si]->sealed) return fail(NRTGPU_ERR_STATE, "segment %d missing or not sealed", si);
  // from here on `dim` is the RESIDENT dimension (a multiple of 16) and the queries are padded like the rows
  PaddedQueries padded;
  if (int rc = pad_query_vectors(segs, n_segs, field_id, queries, n_queries, dim, &padded)) return rc;
  queries = padded.p;
  dim = padded.dim;
  for (int si = 0; si < n_segs; ++si)   // the fp16 sketches this search nominates from: built on a field's first exact search
    if (int rc = ensure_vector_sketch(segs[si], field_id)) return rc;
  const uint32_t k_stride = round_up((uint32_t)k, 16);
  // The matrix-core pass NOMINATES: it keeps the k_int best rows by its estimate of the score (an fp32 fma chain in the
  // MFMA's order, |q|^2 + |v|^2 - 2 q.v for EUCLIDEAN).  The answer is the top k of the nominations rescored in the
  // oracle's order (knn_score_seq), certified against the rows left outside (knn.hip: knn_select_kernel<true>).
  const uint32_t k_int = std::min<uint32_t>((uint32_t)NRTGPU_MAX_K, (uint32_t)k + std::max<uint32_t>(32u, (uint32_t)k / 2u));
  const uint32_t ki_stride = round_up(k_int, 16);
  Slot* slot = nullptr;
  acquire_slot(ctx, &slot);
  struct Guard { nrtgpu_ctx* c; Slot* s; ~Guard() { release_slot(c, s); } } guard{ctx, slot};
  // Ordering on the device (search.cpp: enqueue_search has the BM25 side): a stage of this call -- nomination launches,
  // selections, rescoring -- starts behind the BM25 scorers enqueued last (they want every CU's LDS) and they start behind it;
  // but two vector searches do NOT queue behind each other: the nomination kernel streams, its workgroups are handed out as
  // CUs come free, so a second call's launches fill the tail of this call's and run under its small selection / rescoring
  // kernels (34 KB of LDS beside a nomination workgroup's 112).  Measured with two callers at 10 M x 768: 32 queries per call
  // 3.21 -> 2.90 ms per step, 64: 3.61 -> 3.20.  The host lock is held for the two event operations only.
  hipStream_t st = slot->stream;
  auto take_turn = [&]() -> int {
    std::lock_guard<std::mutex> gpu(ctx->gpu_mu);
    if (ctx->last_turn) HIP_TRY(hipStreamWaitEvent(st, ctx->last_turn, 0));
    return NRTGPU_OK;
  };
  auto end_turn = [&]() -> int {
    std::lock_guard<std::mutex> gpu(ctx->gpu_mu);
    HIP_TRY(hipEventRecord(slot->ev_turn, st));
    ctx->last_knn_turn = slot->ev_turn;
    return NRTGPU_OK;
  };
  const bool timing = ctx->cfg.collect_timing != 0;
  Carver wc;
  const size_t o_q = wc.take((size_t)kKnnMaxQ * dim * 4), o_qn = wc.take(kKnnMaxQ * 4), o_eb = wc.take(kKnnMaxQ * 4);
  const size_t o_eb16 = wc.take(kKnnMaxQ * 4), o_qs = wc.take(kKnnMaxQ * 4);
  const size_t o_segs = wc.take((size_t)std::max(n_segs, 1) * sizeof(DVecSeg));
  const size_t o_leaves = wc.take((size_t)std::max(n_segs, 1) * sizeof(DKnnLeaf));   // the sketch kernel's leaf table (plan.h)
  const size_t o_th = wc.take(kKnnMaxQ * 8);
  const size_t o_tk = wc.take((size_t)kKnnMaxQ * ki_stride * 8), o_tc = wc.take(kKnnMaxQ * 4);
  const size_t o_cc = wc.take(kKnnMaxQ * 4), o_xk = wc.take((size_t)kKnnMaxQ * k_stride * 8);
  const size_t o_xc = wc.take(kKnnMaxQ * 4), o_cert = wc.take(kKnnMaxQ * 4), o_ov = wc.take(64);   // (fetched in one copy)
  const size_t o_p16 = wc.take(160 * 1024);   // the panel in fp16, the sketch kernel's operand order (knn_panel_fp16_kernel)
  const size_t o_cd = wc.take((size_t)kKnnMaxQ * kKnnCap * 8);
  if (int rc = slot->d_work.reserve(wc.off)) return rc;
  const size_t oh_cnt = (size_t)kKnnMaxQ * k_stride * 8, oh_cert = oh_cnt + (o_cert - o_xc), oh_ov = oh_cnt + (o_ov - o_xc);
  if (int rc = slot->h_out.reserve(oh_ov + 64)) return rc;
  // the panel's inputs are staged in pinned memory laid out like the workspace's head [o_q, o_th): two copies per panel, no sync
  if (int rc = slot->h_aux.reserve(o_th)) return rc;
  char* wb = (char*)slot->d_work.p;
  char* ho = (char*)slot->h_out.p;
  char* hs = (char*)slot->h_aux.p;
  float *qn = (float*)(hs + o_qn), *eb = (float*)(hs + o_eb), *eb16 = (float*)(hs + o_eb16), *qsc = (float*)(hs + o_qs);
  // the leaves' vector matrices, for docid -> row on the device (the rescoring reads rows by docid)
  DVecSeg* hsegs = (DVecSeg*)(hs + o_segs);
  double nv_max = 0.0, nv_min = INFINITY, rows_unit = 0.0;   // rows_unit: the largest 1 / scale of the leaves' sketches
  bool all_sketched = true, any_vectors = false;
  for (int si = 0; si < n_segs; ++si) {
    DVecSeg v{};
    v.doc_base = doc_bases ? doc_bases[si] : 0;
    v.max_doc = segs[si]->max_doc;
    auto fit = segs[si]->fields.find(field_id);
    if (fit != segs[si]->fields.end() && fit->second.d_vectors) {
      v.vecs = fit->second.d_vectors;
      v.vnorm2 = fit->second.d_vnorm2;
      v.ord_to_doc = fit->second.d_ord_to_doc;
      v.n_vec = fit->second.n_vec;
      nv_max = std::max(nv_max, (double)fit->second.vnorm2_max);
      if (fit->second.n_vec > 0) {
        if (!fit->second.d_sketch) all_sketched = false;
        any_vectors = true;
        if (fit->second.vnorm2_min > 0.f) nv_min = std::min(nv_min, (double)fit->second.vnorm2_min);
        rows_unit = std::max(rows_unit, 1.0 / (double)fit->second.sketch_scale);
      }
    }
    hsegs[(size_t)si] = v;
  }
  // the sketch kernel's view of the same leaves: ONE launch walks them all (tiles of 16 rows numbered through the leaves)
  DKnnLeaf* hleaves = (DKnnLeaf*)(hs + o_leaves);
  int32_t n_kleaves = 0;
  int64_t total_tiles = 0, total_rows = 0, live_vectors = 0;
  for (int si = 0; si < n_segs; ++si) {
    auto fit = segs[si]->fields.find(field_id);
    if (fit == segs[si]->fields.end() || !fit->second.d_vectors || fit->second.n_vec == 0) continue;
    const FieldData& f = fit->second;
    live_vectors += live_vector_count(segs[si], f);   // (deleted docs are masked inside the kernel and are no hits)
    const uint64_t* accept = segs[si]->d_live;        // (vectors are not re-coded for liveDocs: always the mask)
    if (knn_request && filter_mask != 0)
      if (int rc = accept_set_of(segs[si], filter_mask, 0, &accept)) return rc;
    DKnnLeaf l{};
    l.sketch = f.d_sketch;
    l.vnorm2 = f.d_vnorm2;
    l.ord_to_doc = f.d_ord_to_doc;
    l.accept = accept;
    l.tile_begin = total_tiles;
    l.n_rows = f.n_vec;
    l.doc_base = doc_bases ? doc_bases[si] : 0;
    l.inv_rows_scale = 1.0f / f.sketch_scale;
    hleaves[n_kleaves++] = l;
    total_tiles += ((int64_t)f.n_vec + 15) >> 4;
    total_rows += f.n_vec;
  }
  // |estimate - result| <= E: both are fp32 evaluations of the same length-dim sums, each within gamma = dim * 2^-24 (relative
  // to the sum of the terms' magnitudes) of the real value whatever the order; the maps to a score have slope <= 1 and add a
  // few roundings (erel).  DESIGN §4.5 derives the constants.
  const double u_fp32 = std::ldexp(1.0, -24), gam = (double)(dim + 4) * u_fp32;
  const float score_boost = knn_request ? 1.0f : boost;
  const float erel = (float)(32.0 * u_fp32);
  auto bound_of = [&](double nq) {   // e_abs of plan.h's knn_result_upper / knn_estimate_lower
    double e = 0.0;
    if (sim == 0) e = 2.0 * gam * (double)score_boost;
    else if (sim == 1) e = 1.1 * gam * std::sqrt(nq * nv_max) * (double)score_boost;
    else if (sim == 2) e = 4.5 * gam * (nq + nv_max);   // squared-distance units, no boost
    else e = 2.2 * gam * std::sqrt(nq * nv_max) * (double)score_boost;
    return std::nextafter((float)((e + 4.0 * u_fp32) * (1.0 + 1e-6)), INFINITY);
  };
  // The fp16 sketch (knn.hip): rows and queries rounded to 11 significant bits (relative 2^-11 each), products exact in fp32,
  // fp32 accumulation; elements that fall under fp16's normal range (2^-14 after scaling: 2^-28 of the largest) may be flushed.
  //   |dot16 - q.v| <= (2^-10 + 2^-22 + 4 gamma) sum |q_i v_i|  +  |q|_1 * 2^-14 / rows' scale  +  |v|_1 * 2^-14 / query's scale
  // with sum |q_i v_i| <= |q||v| and |v|_1 <= sqrt(dim) |v|.  Cosine divides by the row's own |v|: the first term's |v| cancels,
  // the flush terms need the smallest non-zero |v| of the leaves.  On top: the fp32 bound above (the estimate's norms are fp32).
  // (the query panel in fp16 and at least a small nomination queue behind it must fit the CU's 160 KB of LDS)
  const bool sketch_ok = all_sketched && any_vectors && std::isfinite(nv_max) &&
                         knn_sketch_fits(dim, std::min(n_queries, dim > 1280 ? 16 : kKnnMaxQ));
  auto bound16_of = [&](double nq, double q_l1, double q_unit, double e32) {
    // (4 gamma for the accumulation: the matrix cores' internal summation tree is not specified to round to nearest at every node)
    const double e16 = std::ldexp(1.0, -10) + std::ldexp(1.0, -22) + 4.0 * gam;
    const double flush = q_l1 * std::ldexp(1.0, -14) * rows_unit + std::sqrt((double)dim * nv_max) * std::ldexp(1.0, -14) * q_unit;
    const double e_dot = 1.01 * (e16 * std::sqrt(nq * nv_max) + flush);
    double e = 0.0;
    if (sim == 0) e = 0.5 * 1.01 * (e16 + (nq > 0.0 && std::isfinite(nv_min) ? flush / std::sqrt(nq * nv_min) : 0.0)) * (double)score_boost;
    else if (sim == 1) e = 0.5 * e_dot * (double)score_boost;
    else if (sim == 2) e = 2.0 * e_dot;
    else e = e_dot * (double)score_boost;
    // (+ 32 u: the kernel maps the dot product to a score with hardware rsq / rcp and a handful of fp32 roundings, scores <= 1;
    //  MAXIMUM_INNER_PRODUCT's unbounded scores take theirs from e_rel)
    return std::nextafter((float)((e + e32 + 32.0 * u_fp32 * (sim == 2 ? 4.0 : (double)score_boost)) * (1.0 + 1e-6)), INFINITY);
  };
  // Rows are scored in rounds with a selection in between (theta tightens from round to round).  The first round of
  // a panel gives every row a slot of the candidate list; later rounds only append rows that beat theta, so they can
  // be long: with rows in no particular order a round that multiplies the rows seen by 16 appends about
  // k * ln 16 candidates.  Rows ordered by rising similarity could overflow the list (every row beats theta): the
  // select kernel flags that and the panel is redone with rounds no longer than the list (`safe`).
  // 160 KB of LDS hold two 16-query panels up to 1280 dimensions; beyond that one panel (16 queries per pass) up to 2048
  const int pass_q = dim > 1280 ? 16 : kKnnMaxQ;
  for (int q0 = 0; q0 < n_queries; q0 += pass_q) {
    const int nq = std::min(pass_q, n_queries - q0);
    if (deadline_passed(g_deadline_ns)) {   // between two passes over the rows: nothing of the next one has been launched
      (void)hipStreamSynchronize(st);
      return fail(NRTGPU_ERR_TIMEOUT, "deadline passed between two passes over the rows (%d of %d queries answered)", q0, n_queries);
    }
    // nominate from the sketch unless it failed to certify lately for this similarity (then: a few panels straight from fp32)
    bool panel_sketch = sketch_ok;
    if (panel_sketch && ctx->knn_sketch_skip[sim].load(std::memory_order_relaxed) > 0) {
      ctx->knn_sketch_skip[sim].fetch_sub(1, std::memory_order_relaxed);
      panel_sketch = false;
    }
    // squareMagnitude (fp32, in element order), the largest |element| and the 1-norm of every query: each sum is a chain of
    // dependent additions, so eight queries are walked side by side -- eight independent chains, every query's own order untouched
    // (0.11 ms -> 0.035 ms per 64-query panel of 768 dimensions: a pass's staging is host time no kernel runs under for one caller)
    float q_max_of[kKnnMaxQ];
    double q_l1_of[kKnnMaxQ];
    for (int qb = 0; qb < nq; qb += 8) {
      const int nb = std::min(8, nq - qb);
      float s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, mx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      double l1[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const float* qv0 = queries + (size_t)(q0 + qb) * dim;
      {
#pragma clang fp contract(off)   // (x * x rounded, then added: never one fused operation, whatever the build's flags)
        auto walk = [&](const int n) {
          for (int d = 0; d < dim; ++d)
            for (int j = 0; j < n; ++j) {
              const float x = qv0[(size_t)j * dim + d];
              const float p2 = x * x;
              s2[j] = s2[j] + p2;
              mx[j] = std::max(mx[j], std::fabs(x));
              l1[j] += std::fabs((double)x);
            }
        };
        if (nb == 8) walk(8);   // (a constant trip count: unrolled, the eight sums in registers)
        else walk(nb);
      }
      for (int j = 0; j < nb; ++j) {
        qn[(size_t)(qb + j)] = s2[j];
        q_max_of[qb + j] = mx[j];
        q_l1_of[qb + j] = l1[j];
      }
    }
    for (int q = 0; q < nq; ++q) {
      const float s2 = qn[(size_t)q];
      eb[(size_t)q] = bound_of((double)s2);
      const float q_max = q_max_of[q];
      const double q_l1 = q_l1_of[q];
      int e2 = 0;
      (void)std::frexp(q_max, &e2);   // q_max < 2^e2: the scaled query's largest |element| is below 2^14
      qsc[(size_t)q] = (q_max > 0.f && std::isfinite(q_max)) ? std::ldexp(1.0f, 14 - e2) : 1.0f;
      eb16[(size_t)q] = bound16_of((double)s2, q_l1, 1.0 / (double)qsc[(size_t)q], (double)eb[(size_t)q]);
      if (!std::isfinite(q_max) || !std::isfinite(eb16[(size_t)q])) panel_sketch = false;
    }
    memcpy(hs + o_q, queries + (size_t)q0 * dim, (size_t)nq * dim * 4);
    HIP_TRY(hipMemcpyAsync(wb + o_q, hs + o_q, (size_t)nq * dim * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(wb + o_qn, hs + o_qn, o_th - o_qn, hipMemcpyHostToDevice, st));   // |q|^2, bounds, scales, leaf table
    if (int rc = take_turn()) return rc;   // (the staging above and its copies are not part of the turn)
    int64_t total_vec = 0, rows_scored = 0;
    size_t n_ev = 0;
    // One pass over the rows of every leaf.  nominate: the estimates' running top-k_int, theta tightening (knn_select_kernel
    // <false>); else theta stays what the certification left and every nomination is rescored into the answer (<true>).
    // (rows of the first round: every row takes a slot of the list and the first selection scans them all)
    static const int64_t kFirstRound = []() { const long v = dev_env_int("NRTGPU_KNN_FIRST_ROUND", 0); return (int64_t)(v >= 1024 && v <= (1 << 18) ? (v & ~15L) : (1 << 16)); }();
    int64_t sketch_launches = 0;
    auto rows_pass = [&](bool nominate, int safe) -> int {   // (the fp32 rows: a launch per leaf and round)
      int64_t seen = 0, round = kFirstRound;
      // Nominating, theta tightens fast: after two selections (>= 1M rows seen) a later launch appends about k ln(rows / rows
      // seen) keys per query, so the remaining launches run back to back (append_only) and ONE selection closes the pass.  A
      // theta still unknown then (hardly any live row) makes those launches append every live row: the list overflows, the
      // flag is raised and the panel is redone in bounded rounds with a selection after each.
      int selections = 0;
      bool pending = false;
      total_vec = 0;
      for (int si = 0; si < n_segs; ++si) {
        const nrtgpu_seg* seg = segs[si];
        auto fit = seg->fields.find(field_id);
        if (fit == seg->fields.end() || !fit->second.d_vectors) continue;
        const FieldData& f = fit->second;
        total_vec += live_vector_count(seg, f);   // (deleted docs are masked inside the kernel and are no hits)
        const uint64_t* accept = seg->d_live;  // (vectors are not re-coded for liveDocs: always the mask)
        if (knn_request && filter_mask != 0)
          if (int rc = accept_set_of(seg, filter_mask, 0, &accept)) return rc;
        // rounds never exceed the candidate capacity, so a list cannot overflow; theta tightens between rounds
        int64_t r = 0;
        if (seen == 0) round = kFirstRound;
        while (r < f.n_vec) {
          const int64_t rb = r;
          int64_t len = (safe || (nominate && seen == 0)) ? std::min<int64_t>(round, kKnnCap) : round;
          if (!nominate && !safe) len = f.n_vec;   // theta is fixed and tight: the whole leaf at once
          const int64_t re = std::min<int64_t>(f.n_vec, (r + len + 15) & ~(int64_t)15);   // rounds begin on tile boundaries (16 rows)
          uint32_t blocks = (uint32_t)std::min<int64_t>((re - r + 255) / 256, (int64_t)std::max(ctx->n_cus, 1));  // 256 rows per workgroup step
          if (nq > 32) blocks = std::max(16u, std::min((uint32_t)std::max(ctx->n_cus, 16), 2u * blocks) / 16u * 16u);   // paired workgroups
          if (timing) {
            while (slot->round_ev.size() < n_ev + 2) {
              hipEvent_t ev = nullptr;
              HIP_TRY(hipEventCreate(&ev));
              slot->round_ev.push_back(ev);
            }
            HIP_TRY(hipEventRecord(slot->round_ev[n_ev], st));
          }
          const bool defer = nominate && !safe && selections >= 2;
          const int e = launch_knn_score(st, blocks, f.d_vectors, f.d_vnorm2, f.d_ord_to_doc, accept, dim, r, re,
                                         doc_bases ? doc_bases[si] : 0, (const float*)(wb + o_q), (const float*)(wb + o_qn), nq,
                                         sim, score_boost, (const unsigned long long*)(wb + o_th), (uint64_t*)(wb + o_cd),
                                         (uint32_t*)(wb + o_cc), kKnnCap, defer ? 1 : 0);
          if (e) return fail(NRTGPU_ERR_HIP, "knn_score launch: %s", hipGetErrorString((hipError_t)e));
          if (timing) {
            HIP_TRY(hipEventRecord(slot->round_ev[n_ev + 1], st));
            n_ev += 2;
          }
          if (defer) {
            pending = true;
          } else if (nominate) {
            launch_knn_select(st, (uint32_t)nq, (uint64_t*)(wb + o_tk), (uint32_t*)(wb + o_tc), ki_stride, k_int,
                              (const uint64_t*)(wb + o_cd), (uint32_t*)(wb + o_cc), kKnnCap,
                              (unsigned long long*)(wb + o_th), (uint32_t*)(wb + o_ov));
            ++selections;
          } else
            launch_knn_refine_select(st, (uint32_t)nq, (uint64_t*)(wb + o_xk), (uint32_t*)(wb + o_xc), k_stride, (uint32_t)k,
                                     (const uint64_t*)(wb + o_cd), (uint32_t*)(wb + o_cc), kKnnCap,
                                     (unsigned long long*)(wb + o_th), (uint32_t*)(wb + o_ov), (const DVecSeg*)(wb + o_segs), n_segs, dim,
                                     sim, (const float*)(wb + o_q), (const float*)(wb + o_qn), score_boost, (const float*)(wb + o_eb), erel,
                                     knn_request ? min_score : 0.0f, k_int, 0, (uint32_t*)(wb + o_cert));
          r = re;
          seen += re - rb;
          round = safe ? std::min<int64_t>(round * 4, kKnnCap) : std::min<int64_t>(seen * 15, (int64_t)1 << 40);
        }
      }
      if (pending)
        launch_knn_select(st, (uint32_t)nq, (uint64_t*)(wb + o_tk), (uint32_t*)(wb + o_tc), ki_stride, k_int,
                          (const uint64_t*)(wb + o_cd), (uint32_t*)(wb + o_cc), kKnnCap,
                          (unsigned long long*)(wb + o_th), (uint32_t*)(wb + o_ov));
      rows_scored += seen;
      return NRTGPU_OK;
    };
    // The same from the fp16 sketch: a round is a range of the leaves' tiles, ONE launch whatever the number of leaves it crosses.
    auto sketch_pass = [&](int safe, bool nominate) -> int {   // nominate = false: theta is fixed, every nomination is rescored into the answer
      int64_t seen = 0, round = kFirstRound >> 4;   // in tiles of 16 rows
      int selections = 0;
      bool pending = false;
      total_vec = live_vectors;
      for (int64_t t = 0; t < total_tiles;) {
        int64_t len = (safe || (nominate && seen == 0)) ? std::min<int64_t>(round, kKnnCap >> 4) : round;
        if (!nominate && !safe) len = total_tiles;   // theta is fixed and tight: everything at once
        len = std::min<int64_t>(len, (int64_t)1 << 22);   // (a queue entry carries the padded row inside the launch in 26 bits)
        const int64_t te = std::min<int64_t>(total_tiles, t + len);
        const uint32_t blocks = (uint32_t)std::min<int64_t>(((te - t) * 16 + 255) / 256, (int64_t)std::max(ctx->n_cus, 1));
        if (timing) {
          while (slot->round_ev.size() < n_ev + 2) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            slot->round_ev.push_back(ev);
          }
          HIP_TRY(hipEventRecord(slot->round_ev[n_ev], st));
        }
        const bool defer = nominate && !safe && selections >= 2;
        const int e = launch_knn_sketch(st, blocks, (const DKnnLeaf*)(wb + o_leaves), n_kleaves, dim, t, te, (const void*)(wb + o_p16),
                                        (const float*)(wb + o_qn), (const float*)(wb + o_qs), nq, sim, score_boost,
                                        (const unsigned long long*)(wb + o_th), (uint64_t*)(wb + o_cd), (uint32_t*)(wb + o_cc), kKnnCap,
                                        defer ? 1 : 0);
        if (e) return fail(NRTGPU_ERR_HIP, "knn_sketch launch: %s", hipGetErrorString((hipError_t)e));
        ++sketch_launches;
        if (timing) {
          HIP_TRY(hipEventRecord(slot->round_ev[n_ev + 1], st));
          n_ev += 2;
        }
        if (defer) {
          pending = true;
        } else if (nominate) {
          launch_knn_select(st, (uint32_t)nq, (uint64_t*)(wb + o_tk), (uint32_t*)(wb + o_tc), ki_stride, k_int,
                            (const uint64_t*)(wb + o_cd), (uint32_t*)(wb + o_cc), kKnnCap, (unsigned long long*)(wb + o_th),
                            (uint32_t*)(wb + o_ov));
          ++selections;
        } else {
          launch_knn_refine_select(st, (uint32_t)nq, (uint64_t*)(wb + o_xk), (uint32_t*)(wb + o_xc), k_stride, (uint32_t)k,
                                   (const uint64_t*)(wb + o_cd), (uint32_t*)(wb + o_cc), kKnnCap, (unsigned long long*)(wb + o_th),
                                   (uint32_t*)(wb + o_ov), (const DVecSeg*)(wb + o_segs), n_segs, dim, sim, (const float*)(wb + o_q),
                                   (const float*)(wb + o_qn), score_boost, (const float*)(wb + o_eb16), erel,
                                   knn_request ? min_score : 0.0f, k_int, 0, (uint32_t*)(wb + o_cert));
        }
        seen += te - t;
        t = te;
        round = safe ? std::min<int64_t>(round * 4, kKnnCap >> 4) : std::min<int64_t>(seen * 15, (int64_t)1 << 36);
      }
      if (pending)
        launch_knn_select(st, (uint32_t)nq, (uint64_t*)(wb + o_tk), (uint32_t*)(wb + o_tc), ki_stride, k_int,
                          (const uint64_t*)(wb + o_cd), (uint32_t*)(wb + o_cc), kKnnCap, (unsigned long long*)(wb + o_th),
                          (uint32_t*)(wb + o_ov));
      rows_scored += total_rows;
      return NRTGPU_OK;
    };
    auto fetch = [&]() -> int {   // the answer so far + the flags
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(ho, wb + o_xk, (size_t)nq * k_stride * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(ho + oh_cnt, wb + o_xc, o_ov + 4 - o_xc, hipMemcpyDeviceToHost, st));   // counts, flags, overflow: one copy
      HIP_TRY(hipStreamSynchronize(st));
      return NRTGPU_OK;
    };
    // 1. nominate, rescore the nominations, certify
    for (int safe = 0;; ++safe) {
      if (int rc = take_turn()) return rc;
      HIP_TRY(hipMemsetAsync(wb + o_th, 0, o_cd - o_th, st));  // theta, lists, counters
      if (knn_request && min_score > 0.0f) {  // start theta below the lowest key whose RESULT can still reach min_score
        std::vector<uint64_t> th0((size_t)nq);
        for (int q = 0; q < nq; ++q) {
          const float lo = (float)knn_estimate_lower(sim, (double)min_score, (double)(panel_sketch ? eb16 : eb)[(size_t)q], (double)erel, 1.0);
          th0[(size_t)q] = lo > 0.0f ? pack_key(lo, 0xFFFFFFFFu) - 1ull : 0ull;
        }
        HIP_TRY(hipMemcpyAsync(wb + o_th, th0.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));  // th0 is a stack vector
      }
      if (panel_sketch) launch_knn_panel_fp16(st, (const float*)(wb + o_q), (const float*)(wb + o_qs), dim, nq, wb + o_p16);
      if (int rc = panel_sketch ? sketch_pass(safe, true) : rows_pass(true, safe)) return rc;
      launch_knn_refine_select(st, (uint32_t)nq, (uint64_t*)(wb + o_xk), (uint32_t*)(wb + o_xc), k_stride, (uint32_t)k,
                               (const uint64_t*)(wb + o_tk), (uint32_t*)(wb + o_tc), ki_stride, (unsigned long long*)(wb + o_th),
                               (uint32_t*)(wb + o_ov), (const DVecSeg*)(wb + o_segs), n_segs, dim, sim, (const float*)(wb + o_q),
                               (const float*)(wb + o_qn), score_boost, (const float*)(wb + (panel_sketch ? o_eb16 : o_eb)), erel,
                               knn_request ? min_score : 0.0f, k_int, 1, (uint32_t*)(wb + o_cert));
      if (int rc = end_turn()) return rc;
      if (int rc = fetch()) return rc;
      if (*(const uint32_t*)(ho + oh_ov) == 0u) break;
      if (safe) return fail(NRTGPU_ERR_HIP, "knn: candidate list overflow in a bounded round");
    }
    // 2. queries whose answer the nominations do not certify (rows outside the list within the rounding bound of the k-th
    //    result: near-duplicates, or k at the list's capacity): a second pass nominates every row whose estimate is within
    //    the bound of the k-th rescored score, and rescores all of them
    int uncertified = 0;
    for (int q = 0; q < nq; ++q) uncertified += ((const uint32_t*)(ho + oh_cert))[q] == 0u;
    // Where: first the sketch again (half the bytes; its bound is wider, so more rows are nominated -- all of them are rescored,
    // which costs nothing much while they are thousands); if that overflows a list -- the bound reaches down to rows by the
    // hundred thousand: large norms against small distances -- the fp32 rows with their tight bound, and the next panels of this
    // similarity go there directly.
    auto second_theta = [&](const float* bound, std::vector<uint64_t>& th2) {
      // a row of the answer scores >= the k-th rescored score known so far (or >= min_score while fewer than k are known): the
      // lowest key its estimate can carry; certified queries nominate nothing (theta = ~0)
      th2.assign((size_t)nq, ~0ull);
      const uint64_t* xk = (const uint64_t*)ho;
      const uint32_t* xc = (const uint32_t*)(ho + oh_cnt);
      for (int q = 0; q < nq; ++q) {
        if (((const uint32_t*)(ho + oh_cert))[q] != 0u) continue;
        const double base = xc[q] >= (uint32_t)k ? (double)key_score(xk[(size_t)q * k_stride + (size_t)k - 1])
                                                 : (knn_request ? (double)min_score : 0.0);
        const float lo = (float)knn_estimate_lower(sim, base, (double)bound[(size_t)q], (double)erel, (double)score_boost);
        th2[(size_t)q] = lo > 0.0f ? pack_key(lo, 0xFFFFFFFFu) - 1ull : 0ull;
      }
    };
    auto second_pass = [&](bool from_sketch, int safe, const std::vector<uint64_t>& th2) -> int {
      if (int rc = take_turn()) return rc;
      HIP_TRY(hipMemcpyAsync(wb + o_th, th2.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
      for (int q = 0; q < nq; ++q)   // their answers start over (a nomination found again must not be counted twice)
        if (((const uint32_t*)(ho + oh_cert))[q] == 0u) HIP_TRY(hipMemsetAsync(wb + o_xc + (size_t)q * 4, 0, 4, st));
      HIP_TRY(hipMemsetAsync(wb + o_cc, 0, kKnnMaxQ * 4, st));
      HIP_TRY(hipMemsetAsync(wb + o_ov, 0, 4, st));
      if (int rc = from_sketch ? sketch_pass(safe, false) : rows_pass(false, safe)) return rc;
      if (int rc = end_turn()) return rc;
      return fetch();   // (the second pass leaves the flags as they are)
    };
    if (uncertified) {
      std::vector<uint64_t> th2;
      bool done = false;
      if (panel_sketch) {
        const std::vector<uint32_t> cert0((const uint32_t*)(ho + oh_cert), (const uint32_t*)(ho + oh_cert) + nq);
        const std::vector<uint64_t> xk0((const uint64_t*)ho, (const uint64_t*)ho + (size_t)nq * k_stride);
        const std::vector<uint32_t> xc0((const uint32_t*)(ho + oh_cnt), (const uint32_t*)(ho + oh_cnt) + nq);
        second_theta(eb16, th2);
        if (int rc = second_pass(true, 0, th2)) return rc;
        done = *(const uint32_t*)(ho + oh_ov) == 0u;
        if (!done) {   // back to what the first stage left (the flags are untouched on the device; the host's copies are restored)
          memcpy(ho, xk0.data(), xk0.size() * 8);
          memcpy(ho + oh_cnt, xc0.data(), xc0.size() * 4);
          memcpy(ho + oh_cert, cert0.data(), cert0.size() * 4);
          ctx->knn_sketch_skip[sim].store(16, std::memory_order_relaxed);
        }
      }
      if (!done) {
        second_theta(eb, th2);
        for (int safe = 0;; ++safe) {
          if (int rc = second_pass(false, safe, th2)) return rc;
          if (*(const uint32_t*)(ho + oh_ov) == 0u) break;
          if (safe) return fail(NRTGPU_ERR_HIP, "knn: candidate list overflow in a bounded round");
        }
      }
    }
    {
      double ms = 0.0;
      for (size_t i = 0; i + 1 < n_ev; i += 2) {
        float one = 0.f;
        (void)hipEventElapsedTime(&one, slot->round_ev[i], slot->round_ev[i + 1]);
        ms += (double)one;
      }
      std::lock_guard<std::mutex> lk(ctx->stats_mu);
      ctx->stats.knn_panels += 1;
      ctx->stats.knn_score_launches += (int64_t)(n_ev / 2);
      ctx->stats.knn_score_ms += ms;
      ctx->stats.knn_rows += rows_scored;
      ctx->stats.knn_second_passes += uncertified ? 1 : 0;
      ctx->stats.knn_sketch_launches += sketch_launches;
    }
    if (ext_keys) {   // stays in HBM: what nrtgpu_dist_knn_exact exchanges
      std::vector<uint64_t> tv((size_t)nq, (uint64_t)total_vec);
      HIP_TRY(hipMemcpyAsync(ext_keys + (size_t)q0 * k_stride * 8, wb + o_xk, (size_t)nq * k_stride * 8, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipMemcpyAsync(ext_cnts + (size_t)q0 * 4, wb + o_xc, (size_t)nq * 4, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipMemcpyAsync(ext_hits + (size_t)q0 * 8, tv.data(), (size_t)nq * 8, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));   // (tv is a stack vector; the workspace is reused by the next panel)
      continue;
    }
    const uint64_t* keys = (const uint64_t*)ho;
    const uint32_t* cnts = (const uint32_t*)(ho + oh_cnt);
    for (int q = 0; q < nq; ++q) {
      nrtgpu_topdocs* o = &out[q0 + q];
      const int32_t cap = o->capacity > 0 ? o->capacity : k;
      const int32_t m = std::min<int32_t>((int32_t)cnts[q], cap);
      for (int32_t i = 0; i < m; ++i) {
        if (o->docs) o->docs[i] = (int32_t)key_doc(keys[(size_t)q * k_stride + i]);
        if (o->scores) o->scores[i] = key_score(keys[(size_t)q * k_stride + i]);
      }
      o->n_hits = m;
      o->total_hits = total_vec;   // every live doc with a vector matches an exact vector query
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

int nrtgpu::rt::knn_exact_device(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs, int32_t field_id,
                                 int32_t sim, const float* queries, int32_t n_queries, int32_t dim, int32_t k, float boost, int32_t k_stride,
                                 void* d_keys, void* d_counts, void* d_hits) {
  if (!d_keys || !d_counts || !d_hits) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (k_stride != (int32_t)round_up((uint32_t)k, 16)) return fail(NRTGPU_ERR_INVALID_ARG, "k_stride must be numHits rounded up to 16");
  return knn_impl(ctx, segs, doc_bases, n_segs, field_id, sim, queries, n_queries, dim, k, boost, false, 0, 0.0f, nullptr, (char*)d_keys,
                  (char*)d_counts, (char*)d_hits);
}

extern "C" int nrtgpu_knn_exact(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                int32_t field_id, int32_t sim, const float* queries, int32_t n_queries, int32_t dim,
                                int32_t k, float boost, nrtgpu_topdocs* out) {
  return knn_impl(ctx, segs, doc_bases, n_segs, field_id, sim, queries, n_queries, dim, k, boost, false, 0, 0.0f, out);
}

// ------------------------------------------------------------------------------------------------
// Request coalescing for exact vector searches (cf. search.cpp: nrtgpu_search_bm25_coalesced)
// ------------------------------------------------------------------------------------------------
struct KnnCoRequest {
  const nrtgpu_seg* const* segs;
  const int32_t* doc_bases;
  int32_t n_segs, field_id, sim, dim, k;
  float boost;
  const float* query;
  nrtgpu_topdocs* out;
  int64_t deadline_ns = 0;
  int rc = 0;
  bool done = false, lead = false;
  std::string err;
  std::mutex m;   // (every caller sleeps on its own mutex + condition variable: no convoy on the coalescer's lock)
  std::condition_variable cv;
};

static bool knn_co_compatible(const KnnCoRequest* a, const KnnCoRequest* b) {
  if (a->n_segs != b->n_segs || a->field_id != b->field_id || a->sim != b->sim || a->dim != b->dim || a->boost != b->boost) return false;
  if (a->n_segs && memcmp(a->segs, b->segs, (size_t)a->n_segs * sizeof(void*)) != 0) return false;
  if ((a->doc_bases == nullptr) != (b->doc_bases == nullptr)) return false;
  return !a->doc_bases || memcmp(a->doc_bases, b->doc_bases, (size_t)a->n_segs * 4) == 0;
}

// TotalHits.relation of an exact vector query by the reference's rule (include/nrtgpu.h): one collector per searcher slice
// (MyIndexSearcher.java:163-208), each flips to GREATER_THAN_OR_EQUAL_TO once it has collected more than
// max(totalHitsThreshold, numHits) hits with its queue full (LazyQueueTopScoreDocCollector.java:176-199).  A slice collects the
// live docs of its leaves that carry a vector.  Host only.
extern "C" int nrtgpu_knn_exact_relation(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                         int32_t field_id, int32_t k, int32_t total_hits_threshold) {
  if (!ctx || (n_segs > 0 && !segs) || n_segs < 0 || k <= 0 || total_hits_threshold < 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad knn_exact_relation arguments");
  if (total_hits_threshold == INT32_MAX) return 0;   // ScoreMode.COMPLETE: the collector never publishes a min competitive score
  for (int32_t i = 0; i < n_segs; ++i)
    if (!segs[i]) return fail(NRTGPU_ERR_STATE, "segment %d missing or not sealed", i);
  SegReadLocks content(segs, n_segs);   // liveDocs of the handles are read: not under a set_live_docs on one of them
  std::vector<int64_t> live((size_t)std::max(n_segs, 1), 0);
  std::vector<hostmath::LeafInfo> all((size_t)n_segs);
  int32_t base = 0;
  for (int32_t i = 0; i < n_segs; ++i) {
    if (!segs[i] || !segs[i]->sealed) return fail(NRTGPU_ERR_STATE, "segment %d missing or not sealed", i);
    auto fit = segs[i]->fields.find(field_id);
    if (fit != segs[i]->fields.end() && fit->second.d_vectors) live[(size_t)i] = live_vector_count(segs[i], fit->second);
    all[(size_t)i] = {i, segs[i]->max_doc, segs[i]->max_doc - segs[i]->n_deleted, doc_bases ? doc_bases[i] : base};
    base += segs[i]->max_doc;
  }
  const int64_t floor_ = (int64_t)std::max(total_hits_threshold, k);
  if (ctx->slice_max_docs.load() <= 0 || n_segs == 0) {   // the whole search counts as one slice
    int64_t sum = 0;
    for (int64_t v : live) sum += v;
    return sum > floor_ ? 1 : 0;
  }
  const int32_t vs = ctx->virtual_shards.load();
  const std::vector<std::vector<int32_t>> sl = vs > 1
      ? hostmath::slices_for_shards(all, vs, ctx->slice_max_docs.load(), ctx->slice_max_segments.load(), nullptr)
      : hostmath::slices(all, ctx->slice_max_docs.load(), ctx->slice_max_segments.load(), all);
  for (const std::vector<int32_t>& s_ : sl) {
    int64_t sum = 0;
    for (int32_t li : s_) sum += live[(size_t)li];
    if (sum > floor_) return 1;
  }
  return 0;
}

void nrtgpu_debug_knn_coalescer_wake(nrtgpu_ctx* ctx) {
  std::lock_guard<std::mutex> lk(ctx->kco_mu);
  if (ctx->kco_leader) ctx->kco_leader->cv.notify_one();
}
int nrtgpu_debug_knn_coalescer_pending(nrtgpu_ctx* ctx) {
  std::lock_guard<std::mutex> lk(ctx->kco_mu);
  return (int)ctx->kco_pending.size();
}

extern "C" int nrtgpu_knn_exact_coalesced(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                          int32_t field_id, int32_t sim, const float* query, int32_t dim, int32_t k, float boost,
                                          nrtgpu_topdocs* out) {
  if (!ctx || !query || !out || (n_segs > 0 && !segs)) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  // (what would fail the panel must fail this request alone)
  if (k <= 0 || dim <= 0 || sim < 0 || sim > 3 || n_segs < 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad knn arguments");
  if (k > NRTGPU_MAX_K) return fail(NRTGPU_ERR_UNSUPPORTED, "k %d > %d", k, NRTGPU_MAX_K);
  if (dim > 2048) return fail(NRTGPU_ERR_UNSUPPORTED, "vector dimension %d (device path takes <= 2048)", dim);
  for (int si = 0; si < n_segs; ++si) {
    if (!segs[si] || !segs[si]->sealed) return fail(NRTGPU_ERR_STATE, "segment %d missing or not sealed", si);
    auto fit = segs[si]->fields.find(field_id);
    if (fit != segs[si]->fields.end() && fit->second.d_vectors && fit->second.dim_user != dim)
      return fail(NRTGPU_ERR_INVALID_ARG, "segment %d: field %d has dimension %d, query has %d", si, field_id, fit->second.dim_user, dim);
  }
  NRT_CHECK_DEADLINE("before the request was queued");
  KnnCoRequest me{segs, doc_bases, n_segs, field_id, sim, dim, k, boost, query, out};
  me.deadline_ns = g_deadline_ns;
  std::vector<KnnCoRequest*> batch;
  {
    std::unique_lock<std::mutex> lk(ctx->kco_mu);
    ctx->kco_pending.push_back(&me);
    if (ctx->kco_leader) {   // follower: the leader (or a later one) takes this request
      if (ctx->kco_pending.size() >= (size_t)kKnnMaxQ ||
          (ctx->kco_inflight == 0 && ctx->kco_last_panel > 1 && ctx->kco_pending.size() >= (size_t)ctx->kco_last_panel))
        ctx->kco_leader->cv.notify_one();   // a whole panel waits, or (device free) the last panel's cohort is back
      lk.unlock();
      {
        std::unique_lock<std::mutex> mine(me.m);
        me.cv.wait(mine, [&] { return me.done || me.lead; });
      }
      if (me.done) {
        if (me.rc != 0) g_last_error = me.err;
        return me.rc;
      }
      lk.lock();   // promoted: continue as the leader
    } else {
      ctx->kco_leader = &me;
    }
    // leader: leave at once when the device is free; else when the running panel finishes, or -- as a second panel in flight,
    // whose launches fill the tails of the first one's -- as soon as a whole panel is waiting
    // (NO linger: a caller that finds the device free runs alone -- company comes from the callers that arrive while a panel is
    //  running.  co_hold: the test hook of nrtgpu_debug_hold_coalescers.)
    // The cohort of a closed loop: the members of a panel come back within tens of microseconds of each other, and the first one
    // back used to find the device free and leave alone -- the others then waited a whole pass (2.7 ms at C4) for theirs: 8
    // callers ran at 1.8 k queries/s with p50 5.5 ms where one panel of 8 per pass gives 2.9 k at 2.8 ms.  So a leader that finds
    // the device free after a panel of N > 1 waits until N callers are pending again, for at most kKnnCohortLingerUs; a lone
    // stream of callers (last panel 1) never waits.  NRTGPU_KCO_COHORT=0 (development build): leave at once, A/B.
    static const bool cohort_rule = dev_env_int("NRTGPU_KCO_COHORT", 1) != 0;
    constexpr int kKnnCohortLingerUs = 150;
    const auto cohort_deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(kKnnCohortLingerUs);
    for (;;) {
      const size_t waiting = ctx->kco_pending.size();
      if ((ctx->kco_inflight == 1 || ctx->co_hold) && waiting >= (size_t)kKnnMaxQ) break;
      if (ctx->kco_inflight == 0 && !ctx->co_hold) {
        const bool cohort_due = cohort_rule && ctx->kco_last_panel > 1 && waiting < (size_t)std::min(ctx->kco_last_panel, kKnnMaxQ) &&
                                std::chrono::steady_clock::now() < cohort_deadline;
        if (!cohort_due) break;
        me.cv.wait_until(lk, cohort_deadline);
        continue;
      }
      me.cv.wait(lk);
    }
    std::vector<KnnCoRequest*> rest, expired;
    batch.push_back(&me);
    for (KnnCoRequest* r : ctx->kco_pending) {
      if (r == &me) continue;
      if (deadline_passed(r->deadline_ns)) expired.push_back(r);
      else if (batch.size() < (size_t)kKnnMaxQ && knn_co_compatible(&me, r)) batch.push_back(r);
      else rest.push_back(r);
    }
    for (KnnCoRequest* r : expired) {
      std::lock_guard<std::mutex> theirs(r->m);
      r->rc = NRTGPU_ERR_TIMEOUT;
      r->err = "deadline passed while the request waited for a panel";
      r->done = true;
      r->cv.notify_one();
    }
    ctx->kco_pending.swap(rest);
    ctx->kco_leader = nullptr;
    if (!ctx->kco_pending.empty()) {   // hand the lead to the oldest request left behind
      KnnCoRequest* next = ctx->kco_pending.front();
      ctx->kco_leader = next;
      std::lock_guard<std::mutex> theirs(next->m);
      next->lead = true;
      next->cv.notify_one();
    }
    ctx->kco_inflight++;
    ctx->kco_last_panel = (int)batch.size();
  }
  // the panel, outside the lock: the members' queries side by side, the largest k; every member's buffers take its own k
  int32_t kmax = 0;
  for (KnnCoRequest* r : batch) kmax = std::max(kmax, r->k);
  std::vector<float> qs(batch.size() * (size_t)dim);
  std::vector<nrtgpu_topdocs> outs(batch.size());
  for (size_t i = 0; i < batch.size(); ++i) {
    memcpy(qs.data() + i * (size_t)dim, batch[i]->query, (size_t)dim * 4);
    outs[i] = *batch[i]->out;
    const int32_t cap = outs[i].capacity > 0 ? outs[i].capacity : batch[i]->k;
    outs[i].capacity = std::min(cap, batch[i]->k);
  }
  int rc;
  {
    // (a panel of several requests does not run under its leader's deadline: its mates have not expired)
    struct DeadlineScope {
      int64_t saved;
      explicit DeadlineScope(bool clear) : saved(g_deadline_ns) { if (clear) g_deadline_ns = 0; }
      ~DeadlineScope() { g_deadline_ns = saved; }
    } deadline_scope(batch.size() > 1);
    rc = knn_impl(ctx, segs, doc_bases, n_segs, field_id, sim, qs.data(), (int32_t)batch.size(), dim, kmax, boost, false, 0, 0.0f, outs.data());
  }
  const std::string err = rc ? g_last_error : std::string();
  for (size_t i = 0; i < batch.size(); ++i) {
    KnnCoRequest* r = batch[i];
    if (rc == 0) {
      const int32_t cap = r->out->capacity;
      *r->out = outs[i];
      r->out->capacity = cap;
    }
    if (r == &me) continue;
    std::lock_guard<std::mutex> theirs(r->m);   // (the woken caller cannot return -- and free its request -- before we are done with it)
    r->rc = rc;
    if (rc != 0) r->err = err;
    r->done = true;
    r->cv.notify_one();
  }
  {
    std::lock_guard<std::mutex> lk(ctx->kco_mu);
    ctx->kco_inflight--;
    if (ctx->kco_leader) ctx->kco_leader->cv.notify_one();   // a leader may be waiting for the device
  }
  if (rc != 0) g_last_error = err;
  return rc;
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
  forget_foreign_hip_error();
  if (!ctx || !query || !out || (n > 0 && (!docs || !first_scores)) || (n_segs > 0 && (!segs || !doc_bases)))
    return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  if (n < 0 || dim <= 0 || sim < 0 || sim > 3 || window <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad rescore arguments");
  HIP_TRY(hipSetDevice(ctx->device));
  PaddedQueries padded;   // (rows are resident padded to a multiple of 16 elements: the query likewise)
  if (int rc = pad_query_vectors(segs, n_segs, field_id, query, 1, dim, &padded)) return rc;
  query = padded.p;
  dim = padded.dim;
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
  if (ctx->last_turn) HIP_TRY(hipStreamWaitEvent(slot->stream, ctx->last_turn, 0));
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
