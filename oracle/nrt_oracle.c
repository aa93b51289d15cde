/*
 * nrt_oracle.c -- CPU restatement (plain C) of the reference's query-execution hot path.
 * TEST INFRASTRUCTURE ONLY: see nrt_oracle.h for who may use it and for the parity status.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no -ffast-math: float op order is the spec)
 */
#include "nrt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * SmallFloat.intToByte4 / byte4ToInt  (org.apache.lucene.util.SmallFloat, Lucene 10.4.0;
 * used by BM25Similarity.computeNorm and LENGTH_TABLE).  SURVEY.md Appendix A.1.
 * ------------------------------------------------------------------------------------------ */
static int32_t long_to_int4(int64_t i) {
  /* numBits = 64 - Long.numberOfLeadingZeros(i) */
  int num_bits = (i == 0) ? 0 : 64 - __builtin_clzll((unsigned long long)i);
  if (num_bits < 4) {
    return (int32_t)i; /* subnormal */
  }
  int shift = num_bits - 4;
  int32_t encoded = (int32_t)((uint64_t)i >> shift); /* keep the 4 most significant bits */
  encoded &= 0x07;                                   /* the top one is implicit */
  encoded |= (shift + 1) << 3;                       /* 0 is reserved for subnormals */
  return encoded;
}

static int64_t int4_to_long(int32_t i) {
  int64_t bits = i & 0x07;
  int shift = (int)((uint32_t)i >> 3) - 1;
  if (shift == -1) return bits;
  return (bits | 0x08) << shift;
}

/* MAX_INT4 = longToInt4(Integer.MAX_VALUE) = 231; NUM_FREE_VALUES = 255 - MAX_INT4 = 24 */
static int32_t num_free_values(void) { return 255 - long_to_int4(2147483647LL); }

int32_t nrt_oracle_int_to_byte4(int32_t i) {
  int32_t nfv = num_free_values();
  if (i < 0) return -1;
  if (i < nfv) return i;
  return (nfv + long_to_int4((int64_t)i - nfv)) & 0xFF;
}

int32_t nrt_oracle_byte4_to_int(int32_t b) {
  int32_t nfv = num_free_values();
  b &= 0xFF;
  if (b < nfv) return b;
  return (int32_t)(nfv + int4_to_long(b - nfv));
}

/* ------------------------------------------------------------------------------------------
 * BM25Similarity (Lucene 10.4.0), default instance per SimilarityCreator.java:33,41.
 *   idf   = (float) Math.log(1 + (docCount - docFreq + 0.5D) / (docFreq + 0.5D))
 *   avgdl = (float) (sumTotalTermFreq / (double) docCount)
 *   cache[i] = 1f / (k1 * ((1 - b) + b * LENGTH_TABLE[i] / avgdl))        (all float ops)
 *   score = weight - weight / (1f + freq * normInverse)                  (all float ops)
 * Explain text that pins the shape: src/test/java/com/yelp/nrtsearch/server/grpc/QueryTest.java
 * :1003-1019.
 * ------------------------------------------------------------------------------------------ */
float nrt_oracle_bm25_idf(int64_t doc_count, int64_t doc_freq) {
  double v = log(1.0 + ((double)(doc_count - doc_freq) + 0.5) / ((double)doc_freq + 0.5));
  return (float)v;
}

float nrt_oracle_bm25_avgdl(int64_t sum_total_term_freq, int64_t doc_count) {
  return (float)((double)sum_total_term_freq / (double)doc_count);
}

void nrt_oracle_bm25_norm_cache(float avgdl, float k1, float b, float out256[256]) {
  for (int i = 0; i < 256; ++i) {
    volatile float len = (float)nrt_oracle_byte4_to_int(i); /* LENGTH_TABLE[i] */
    volatile float one_minus_b = 1.0f - b;
    volatile float t = b * len;
    t = t / avgdl;
    t = one_minus_b + t;
    t = k1 * t;
    out256[i] = 1.0f / t;
  }
}

float nrt_oracle_bm25_score(float weight, float freq, float norm_inverse) {
  volatile float t = freq * norm_inverse; /* volatile: forbid contraction / reassociation */
  t = 1.0f + t;
  t = weight / t;
  return weight - t;
}

/* ------------------------------------------------------------------------------------------
 * Collector: LazyQueueTopScoreDocCollector.java:37-203 over Lucene's HitQueue
 * (lessThan(a,b): a.score == b.score ? a.doc > b.doc : a.score < b.score  -- SURVEY A.5).
 * totalHits / relation: this restatement drives the collector from an EXHAUSTIVE scorer, so
 * totalHits is the exact count; the relation flips to GREATER_THAN_OR_EQUAL_TO exactly where
 * the reference collector would start publishing a min competitive score (…Collector.java
 * :176-199: totalHits > totalHitsThreshold and the queue is full).
 * ------------------------------------------------------------------------------------------ */
typedef struct { int32_t doc; float score; } hit_t;

struct nrt_oracle_collector {
  int32_t num_hits;
  int32_t has_after;
  int32_t after_doc_global;
  float   after_score;
  int32_t total_hits_threshold; /* already max(threshold, numHits), …Manager.java:102 */
  int32_t doc_base;
  int32_t after_doc_leaf;
  int64_t total_hits;
  int32_t relation_gte;
  float   min_competitive;
  int32_t size;
  hit_t*  heap; /* 1-based binary min-heap under less_than */
};

static int less_than(const hit_t* a, const hit_t* b) {
  if (a->score == b->score) return a->doc > b->doc;
  return a->score < b->score;
}

static void heap_up(hit_t* h, int i) {
  hit_t node = h[i];
  int j = i >> 1;
  while (j > 0 && less_than(&node, &h[j])) {
    h[i] = h[j];
    i = j;
    j >>= 1;
  }
  h[i] = node;
}

static void heap_down(hit_t* h, int size, int i) {
  hit_t node = h[i];
  int j = i << 1, k = j + 1;
  if (k <= size && less_than(&h[k], &h[j])) j = k;
  while (j <= size && less_than(&h[j], &node)) {
    h[i] = h[j];
    i = j;
    j = i << 1;
    k = j + 1;
    if (k <= size && less_than(&h[k], &h[j])) j = k;
  }
  h[i] = node;
}

nrt_oracle_collector* nrt_oracle_collector_new(int32_t num_hits, int32_t has_after, int32_t after_doc,
                                               float after_score, int32_t total_hits_threshold) {
  if (num_hits <= 0 || total_hits_threshold < 0) return NULL; /* …Manager.java:90-98 */
  nrt_oracle_collector* c = (nrt_oracle_collector*)calloc(1, sizeof(*c));
  c->num_hits = num_hits;
  c->has_after = has_after;
  c->after_doc_global = after_doc;
  c->after_score = after_score;
  c->total_hits_threshold = total_hits_threshold > num_hits ? total_hits_threshold : num_hits;
  c->heap = (hit_t*)malloc(sizeof(hit_t) * ((size_t)num_hits + 1));
  c->min_competitive = 0.0f;
  nrt_oracle_collector_set_leaf(c, 0);
  return c;
}

void nrt_oracle_collector_free(nrt_oracle_collector* c) {
  if (!c) return;
  free(c->heap);
  free(c);
}

void nrt_oracle_collector_set_leaf(nrt_oracle_collector* c, int32_t doc_base) {
  c->doc_base = doc_base;
  c->after_doc_leaf = c->has_after ? c->after_doc_global - doc_base : 2147483647;
}

/* updateMinCompetitiveScore, …Collector.java:176-199 */
static void update_min_competitive(nrt_oracle_collector* c) {
  if (c->total_hits > c->total_hits_threshold) {
    float top = (c->size == c->num_hits) ? c->heap[1].score : -INFINITY;
    float local_min = nextafterf(top, INFINITY); /* Math.nextUp */
    if (local_min > c->min_competitive) {
      c->relation_gte = 1;
      c->min_competitive = local_min;
    }
  }
}

void nrt_oracle_collector_collect(nrt_oracle_collector* c, int32_t doc, float score) {
  int64_t hit_count_so_far = ++c->total_hits; /* :106 */
  if (c->has_after &&
      (score > c->after_score || (score == c->after_score && doc <= c->after_doc_leaf))) {
    /* hit was collected on a previous page, :112-120 */
    if (!c->relation_gte) update_min_competitive(c);
    return;
  }
  float top_score = (c->size == c->num_hits) ? c->heap[1].score : -INFINITY; /* :122-127 */
  if (score <= top_score) { /* :129-140: ties lose, docs arrive in increasing docid */
    if (hit_count_so_far == (int64_t)c->total_hits_threshold + 1) update_min_competitive(c);
    return;
  }
  /* collectCompetitiveHit, :146-156 */
  if (c->size < c->num_hits) {
    c->size++;
    c->heap[c->size].doc = doc + c->doc_base;
    c->heap[c->size].score = score;
    heap_up(c->heap, c->size);
  } else {
    c->heap[1].doc = doc + c->doc_base;
    c->heap[1].score = score;
    heap_down(c->heap, c->size, 1);
  }
  update_min_competitive(c);
}

int32_t nrt_oracle_collector_topdocs(nrt_oracle_collector* c, int32_t* docs, float* scores,
                                     int64_t* total_hits, int32_t* total_hits_is_lower_bound) {
  /* TopDocsCollector.topDocs(): pop least-first into the tail => (score desc, doc asc) */
  int32_t n = c->size;
  hit_t* h = (hit_t*)malloc(sizeof(hit_t) * ((size_t)n + 1));
  memcpy(h, c->heap, sizeof(hit_t) * ((size_t)n + 1));
  int size = n;
  for (int i = n - 1; i >= 0; --i) {
    docs[i] = h[1].doc;
    scores[i] = h[1].score;
    h[1] = h[size];
    size--;
    if (size > 0) heap_down(h, size, 1);
  }
  free(h);
  if (total_hits) *total_hits = c->total_hits;
  if (total_hits_is_lower_bound) *total_hits_is_lower_bound = c->relation_gte;
  return n;
}

/* ------------------------------------------------------------------------------------------
 * Exhaustive disjunction over one segment.  What is restated: the observable contract of
 * IndexSearcher.search -> Weight.bulkScorer().score(leafCollector, liveDocs, 0, maxDoc) for a
 * rewritten pure-SHOULD BooleanQuery of TermQuery clauses / a single TermQuery, reached from
 * src/main/java/com/yelp/nrtsearch/server/handler/SearchHandler.java:1412-1413.
 * ------------------------------------------------------------------------------------------ */
#define ORACLE_WINDOW 4096

void nrt_oracle_search_segment(int32_t max_doc, int32_t doc_base, const uint64_t* live_bits,
                               int32_t n_terms, const nrt_oracle_term* terms,
                               nrt_oracle_collector* collector) {
  nrt_oracle_search_segment_msm(max_doc, doc_base, live_bits, n_terms, terms, 1, collector);
}

/* BooleanQuery.Builder.setMinimumNumberShouldMatch(n) (src/main/java/com/yelp/nrtsearch/server/query/
 * QueryNodeMapper.java:259-261): a doc is a hit iff at least n SHOULD clauses match it (every clause
 * counts, also a repeated term); its score is still the sum over ALL its matching clauses (Lucene's
 * WANDScorer / MinShouldMatchSumScorer contract).  n <= 1 is the plain disjunction. */
void nrt_oracle_search_segment_msm(int32_t max_doc, int32_t doc_base, const uint64_t* live_bits,
                                   int32_t n_terms, const nrt_oracle_term* terms, int32_t min_should_match,
                                   nrt_oracle_collector* collector) {
  double acc[ORACLE_WINDOW];
  uint8_t matched[ORACLE_WINDOW];
  int64_t* cursor = (int64_t*)calloc((size_t)(n_terms > 0 ? n_terms : 1), sizeof(int64_t));
  nrt_oracle_collector_set_leaf(collector, doc_base);

  for (int32_t base = 0; base < max_doc; base += ORACLE_WINDOW) {
    int32_t end = base + ORACLE_WINDOW;
    if (end > max_doc) end = max_doc;
    int any = 0;
    for (int t = 0; t < n_terms; ++t) {
      const nrt_oracle_term* tm = &terms[t];
      int64_t p = cursor[t];
      if (p < tm->n && tm->docids[p] < end) {
        if (!any) {
          memset(acc, 0, sizeof(double) * (size_t)(end - base));
          memset(matched, 0, (size_t)(end - base));
          any = 1;
        }
        for (; p < tm->n && tm->docids[p] < end; ++p) {
          int32_t d = tm->docids[p];
          float freq = tm->freqs ? (float)tm->freqs[p] : 1.0f;
          uint8_t norm = tm->norms ? tm->norms[d] : (uint8_t)1;
          float s = nrt_oracle_bm25_score(tm->weight, freq, tm->cache[norm]);
          acc[d - base] += (double)s;
          matched[d - base]++;
        }
        cursor[t] = p;
      }
    }
    if (!any) continue;
    const int need = min_should_match > 1 ? min_should_match : 1;
    for (int32_t d = base; d < end; ++d) {
      if (matched[d - base] < need) continue;
      if (live_bits && !((live_bits[d >> 6] >> (d & 63)) & 1ULL)) continue;
      nrt_oracle_collector_collect(collector, d, (float)acc[d - base]);
    }
  }
  free(cursor);
}

/* BooleanQuery with MUST and SHOULD term clauses, minimumNumberShouldMatch 0 (QueryNodeMapper.java:257-283 adds clauses of any
 * occur).  lucene-core 10.4.0 [Lucene-recall, SURVEY A]: BooleanScorerSupplier scores it with ReqOptSumScorer(req, opt):
 *   req = the MUST clauses: one -> its TermScorer (the float itself); several -> ConjunctionScorer, score() = (float) of the
 *         double sum of the sub scores;
 *   opt = the SHOULD clauses: one -> its TermScorer; several -> a disjunction whose score() is (float) of the double sum of
 *         the MATCHING sub scores;
 *   ReqOptSumScorer.score(): float score = req.score(); if (opt is on the doc) score += opt.score();  -- a FLOAT addition of
 *         two separately rounded sums, not one sum.
 * A hit matches every MUST clause; SHOULD clauses only add.  required[t] != 0: clause t is MUST.  A segment that lacks a MUST
 * term has no hits.  Parity of this shape is pinned by recall only: the reference's tests hold no score of such a query. */
void nrt_oracle_search_segment_reqopt(int32_t max_doc, int32_t doc_base, const uint64_t* live_bits,
                                      int32_t n_terms, const nrt_oracle_term* terms, const uint8_t* required,
                                      nrt_oracle_collector* collector) {
  double req[ORACLE_WINDOW], opt[ORACLE_WINDOW];
  uint8_t n_req[ORACLE_WINDOW], n_opt[ORACLE_WINDOW];
  int64_t* cursor = (int64_t*)calloc((size_t)(n_terms > 0 ? n_terms : 1), sizeof(int64_t));
  int need = 0;
  for (int t = 0; t < n_terms; ++t) need += required[t] ? 1 : 0;
  nrt_oracle_collector_set_leaf(collector, doc_base);
  for (int32_t base = 0; base < max_doc; base += ORACLE_WINDOW) {
    int32_t end = base + ORACLE_WINDOW;
    if (end > max_doc) end = max_doc;
    int any = 0;
    for (int t = 0; t < n_terms; ++t) {
      const nrt_oracle_term* tm = &terms[t];
      int64_t p = cursor[t];
      if (p < tm->n && tm->docids[p] < end) {
        if (!any) {
          memset(req, 0, sizeof(double) * (size_t)(end - base));
          memset(opt, 0, sizeof(double) * (size_t)(end - base));
          memset(n_req, 0, (size_t)(end - base));
          memset(n_opt, 0, (size_t)(end - base));
          any = 1;
        }
        for (; p < tm->n && tm->docids[p] < end; ++p) {
          int32_t d = tm->docids[p];
          float freq = tm->freqs ? (float)tm->freqs[p] : 1.0f;
          uint8_t norm = tm->norms ? tm->norms[d] : (uint8_t)1;
          float s = nrt_oracle_bm25_score(tm->weight, freq, tm->cache[norm]);
          if (required[t]) {
            req[d - base] += (double)s;
            n_req[d - base]++;
          } else {
            opt[d - base] += (double)s;
            n_opt[d - base]++;
          }
        }
        cursor[t] = p;
      }
    }
    if (!any) continue;
    for (int32_t d = base; d < end; ++d) {
      if (n_req[d - base] != need || need == 0) continue;
      if (live_bits && !((live_bits[d >> 6] >> (d & 63)) & 1ULL)) continue;
      float score = (float)req[d - base];
      if (n_opt[d - base]) score += (float)opt[d - base];
      nrt_oracle_collector_collect(collector, d, score);
    }
  }
  free(cursor);
}

/* DisjunctionMaxQuery over term clauses (src/main/java/com/yelp/nrtsearch/server/query/QueryNodeMapper.java:350-358
 * builds org.apache.lucene.search.DisjunctionMaxQuery(disjuncts, tieBreakerMultiplier)).  lucene-core 10.4.0's
 * DisjunctionMaxScorer.score() [Lucene-recall, SURVEY A]: over the matching sub-scorers keep scoreMax (float) and
 * otherScoreSum (double: every sub score that is not the max), return (float)(scoreMax + otherScoreSum * tieBreaker).
 * With tieBreaker == 0 -- the only value the device route takes, and the one the reference's own test uses
 * (src/test/java/com/yelp/nrtsearch/server/grpc/QueryTest.java:541-583) -- the score is the best clause's score and no
 * order of the sub-scorers matters; with a tie breaker the sub-scorers are visited here in clause order (Lucene visits
 * its DisiPriorityQueue's order: the double sum may differ in the last bit). */
void nrt_oracle_search_segment_dismax(int32_t max_doc, int32_t doc_base, const uint64_t* live_bits,
                                      int32_t n_terms, const nrt_oracle_term* terms, float tie_breaker,
                                      nrt_oracle_collector* collector) {
  float smax[ORACLE_WINDOW];
  double other[ORACLE_WINDOW];
  uint8_t matched[ORACLE_WINDOW];
  int64_t* cursor = (int64_t*)calloc((size_t)(n_terms > 0 ? n_terms : 1), sizeof(int64_t));
  nrt_oracle_collector_set_leaf(collector, doc_base);
  for (int32_t base = 0; base < max_doc; base += ORACLE_WINDOW) {
    int32_t end = base + ORACLE_WINDOW;
    if (end > max_doc) end = max_doc;
    int any = 0;
    for (int t = 0; t < n_terms; ++t) {
      const nrt_oracle_term* tm = &terms[t];
      int64_t p = cursor[t];
      if (p < tm->n && tm->docids[p] < end) {
        if (!any) {
          memset(smax, 0, sizeof(float) * (size_t)(end - base));
          memset(other, 0, sizeof(double) * (size_t)(end - base));
          memset(matched, 0, (size_t)(end - base));
          any = 1;
        }
        for (; p < tm->n && tm->docids[p] < end; ++p) {
          int32_t d = tm->docids[p];
          float freq = tm->freqs ? (float)tm->freqs[p] : 1.0f;
          uint8_t norm = tm->norms ? tm->norms[d] : (uint8_t)1;
          float s = nrt_oracle_bm25_score(tm->weight, freq, tm->cache[norm]);
          if (s >= smax[d - base]) {
            other[d - base] += (double)smax[d - base];
            smax[d - base] = s;
          } else {
            other[d - base] += (double)s;
          }
          matched[d - base] = 1;
        }
        cursor[t] = p;
      }
    }
    if (!any) continue;
    for (int32_t d = base; d < end; ++d) {
      if (!matched[d - base]) continue;
      if (live_bits && !((live_bits[d >> 6] >> (d & 63)) & 1ULL)) continue;
      nrt_oracle_collector_collect(collector, d, (float)((double)smax[d - base] + other[d - base] * (double)tie_breaker));
    }
  }
  free(cursor);
}

/* ------------------------------------------------------------------------------------------
 * The same disjunction with dynamic pruning: the algorithm family of Lucene's
 * MaxScoreBulkScorer (lucene-core 10.4.0; chosen for a top-level pure-SHOULD BooleanQuery under
 * ScoreMode.TOP_SCORES), driven by the min competitive score the
 * collector publishes (src/main/java/org/apache/lucene/search/LazyQueueTopScoreDocCollector.java
 * :168-171, :188-192 -> Scorable.setMinCompetitiveScore).
 *
 * Per 4096-doc window (Lucene's INNER_WINDOW_SIZE): every clause gets an upper bound for the
 * window from per-block maxima (128 postings per block, the granularity of Lucene's level-0
 * impacts); clauses are ordered by that bound; the longest prefix whose bounds sum below the min
 * competitive score is "non-essential".  Essential clauses are scored term-at-a-time into the
 * window; for each doc they matched, non-essential clauses are advanced to the doc from the
 * largest bound down, and the doc is dropped as soon as partial + remaining bounds cannot reach
 * the min competitive score.  Docs matched only by non-essential clauses are never visited.
 *
 * Safe: the top-k (docs and scores) equals nrt_oracle_search_segment's; totalHits becomes a
 * lower bound (the collector has flagged GREATER_THAN_OR_EQUAL_TO before any pruning can start,
 * since pruning needs min_competitive > 0).  This is NOT a line-level restatement of Lucene's
 * outer-window selection or of its totalHits value (unpinned); it is the pruned CPU
 * baseline SURVEY 8d asks to have timed beside the GPU.
 * ------------------------------------------------------------------------------------------ */
#define ORACLE_BLOCK 128

void nrt_oracle_block_max(const nrt_oracle_term* tm, float* out) {
  int64_t nb = (tm->n + ORACLE_BLOCK - 1) / ORACLE_BLOCK;
  for (int64_t b = 0; b < nb; ++b) {
    int64_t lo = b * ORACLE_BLOCK, hi = lo + ORACLE_BLOCK;
    if (hi > tm->n) hi = tm->n;
    float m = 0.0f;
    for (int64_t p = lo; p < hi; ++p) {
      float freq = tm->freqs ? (float)tm->freqs[p] : 1.0f;
      uint8_t norm = tm->norms ? tm->norms[tm->docids[p]] : (uint8_t)1;
      float s = nrt_oracle_bm25_score(tm->weight, freq, tm->cache[norm]);
      if (s > m) m = s;
    }
    out[b] = m;
  }
}

/* first index >= p whose docid is >= target (exponential then binary search) */
static int64_t advance_to(const int32_t* d, int64_t p, int64_t n, int32_t target) {
  if (p >= n || d[p] >= target) return p;
  int64_t lo = p, step = 1;
  while (lo + step < n && d[lo + step] < target) {
    lo += step;
    step <<= 1;
  }
  int64_t hi = lo + step < n ? lo + step : n;
  while (hi - lo > 1) {
    int64_t mid = lo + ((hi - lo) >> 1);
    if (d[mid] < target) lo = mid; else hi = mid;
  }
  return hi;
}

void nrt_oracle_search_segment_maxscore(int32_t max_doc, int32_t doc_base, const uint64_t* live_bits,
                                        int32_t n_terms, const nrt_oracle_term* terms,
                                        const float* const* block_max,
                                        nrt_oracle_collector* collector, int64_t* postings_scored) {
  enum { MAXT = 64 };
  double acc[ORACLE_WINDOW];
  uint8_t matched[ORACLE_WINDOW];
  int64_t cursor[MAXT], blk[MAXT];
  float win_max[MAXT];
  int order[MAXT];
  double bound[MAXT + 1]; /* bound[j] = sum of win_max over order[0..j-1] */
  int64_t scored = 0;
  if (n_terms > MAXT) n_terms = MAXT;
  for (int t = 0; t < n_terms; ++t) cursor[t] = blk[t] = 0;
  nrt_oracle_collector_set_leaf(collector, doc_base);

  for (int32_t base = 0; base < max_doc; base += ORACLE_WINDOW) {
    int32_t end = base + ORACLE_WINDOW;
    if (end > max_doc) end = max_doc;
    /* upper bound of every clause inside [base, end) */
    for (int t = 0; t < n_terms; ++t) {
      const nrt_oracle_term* tm = &terms[t];
      int64_t nb = (tm->n + ORACLE_BLOCK - 1) / ORACLE_BLOCK;
      int64_t b = blk[t];
      for (; b < nb; ++b) { /* skip blocks that end before the window */
        int64_t last = (b + 1) * ORACLE_BLOCK - 1;
        if (last >= tm->n) last = tm->n - 1;
        if (tm->docids[last] >= base) break;
      }
      blk[t] = b;
      float m = 0.0f;
      for (; b < nb && tm->docids[b * ORACLE_BLOCK] < end; ++b)
        if (block_max[t][b] > m) m = block_max[t][b];
      win_max[t] = m;
      order[t] = t;
    }
    for (int i = 1; i < n_terms; ++i) { /* ascending by bound */
      int o = order[i], j = i - 1;
      for (; j >= 0 && win_max[order[j]] > win_max[o]; --j) order[j + 1] = order[j];
      order[j + 1] = o;
    }
    bound[0] = 0.0;
    for (int j = 0; j < n_terms; ++j) bound[j + 1] = bound[j] + (double)win_max[order[j]];
    float theta = collector->min_competitive;
    int ne = 0; /* clauses order[0..ne-1] are non-essential */
    while (ne < n_terms && (float)bound[ne + 1] < theta) ++ne;
    if (ne == n_terms) continue; /* nothing in this window can be competitive */

    int any = 0;
    for (int j = ne; j < n_terms; ++j) {
      int t = order[j];
      const nrt_oracle_term* tm = &terms[t];
      int64_t p = advance_to(tm->docids, cursor[t], tm->n, base);
      if (p < tm->n && tm->docids[p] < end) {
        if (!any) {
          memset(acc, 0, sizeof(double) * (size_t)(end - base));
          memset(matched, 0, (size_t)(end - base));
          any = 1;
        }
        for (; p < tm->n && tm->docids[p] < end; ++p) {
          int32_t d = tm->docids[p];
          float freq = tm->freqs ? (float)tm->freqs[p] : 1.0f;
          uint8_t norm = tm->norms ? tm->norms[d] : (uint8_t)1;
          acc[d - base] += (double)nrt_oracle_bm25_score(tm->weight, freq, tm->cache[norm]);
          matched[d - base] = 1;
          ++scored;
        }
      }
      cursor[t] = p;
    }
    if (!any) continue;
    for (int32_t d = base; d < end; ++d) {
      if (!matched[d - base]) continue;
      if (live_bits && !((live_bits[d >> 6] >> (d & 63)) & 1ULL)) continue;
      double score = acc[d - base];
      int competitive = 1;
      for (int j = ne - 1; j >= 0; --j) {
        if ((float)(score + bound[j + 1]) < collector->min_competitive) {
          competitive = 0;
          break;
        }
        int t = order[j];
        const nrt_oracle_term* tm = &terms[t];
        int64_t p = advance_to(tm->docids, cursor[t], tm->n, d);
        cursor[t] = p;
        if (p < tm->n && tm->docids[p] == d) {
          float freq = tm->freqs ? (float)tm->freqs[p] : 1.0f;
          uint8_t norm = tm->norms ? tm->norms[d] : (uint8_t)1;
          score += (double)nrt_oracle_bm25_score(tm->weight, freq, tm->cache[norm]);
          ++scored;
        }
      }
      if (competitive) nrt_oracle_collector_collect(collector, d, (float)score);
    }
  }
  if (postings_scored) *postings_scored += scored;
}

/* ------------------------------------------------------------------------------------------
 * Many independent searches on the host cores: one collector per query visiting its leaves in
 * docBase order (one Lucene slice per query), queries spread over OpenMP threads.  This is the
 * timed region of bench.py's cpu_baseline leg -- no Python inside it.
 * ------------------------------------------------------------------------------------------ */
void nrt_oracle_search_batch(int32_t n_queries, const int64_t* leaf_offsets,
                             const nrt_oracle_leaf* leaves, int32_t k, int32_t total_hits_threshold,
                             int32_t maxscore, int32_t n_threads, int32_t* out_docs, float* out_scores,
                             int32_t* out_n, int64_t* out_total_hits, int32_t* out_gte,
                             int64_t* postings_scored) {
  int64_t scored_all = 0;
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : scored_all)
  for (int32_t q = 0; q < n_queries; ++q) {
    nrt_oracle_collector* c = nrt_oracle_collector_new(k, 0, 0, 0.0f, total_hits_threshold);
    int64_t scored = 0;
    for (int64_t j = leaf_offsets[q]; j < leaf_offsets[q + 1]; ++j) {
      const nrt_oracle_leaf* lf = &leaves[j];
      if (maxscore)
        nrt_oracle_search_segment_maxscore(lf->max_doc, lf->doc_base, lf->live_bits, lf->n_terms, lf->terms,
                                           lf->block_max, c, &scored);
      else
        nrt_oracle_search_segment(lf->max_doc, lf->doc_base, lf->live_bits, lf->n_terms, lf->terms, c);
    }
    out_n[q] = nrt_oracle_collector_topdocs(c, out_docs + (int64_t)q * k, out_scores + (int64_t)q * k,
                                            &out_total_hits[q], &out_gte[q]);
    nrt_oracle_collector_free(c);
    scored_all += scored;
  }
  if (postings_scored) *postings_scored = scored_all;
}

/* ------------------------------------------------------------------------------------------
 * TopDocs.merge(0, topN, shardHits) with every shardIndex == -1 (…Manager.java:137-144).
 * Lucene's merge pops a priority queue ordered by (score desc, shardIndex, doc asc); with equal
 * shardIndex that is a stable global (score desc, doc asc) order, produced here by sorting.
 * ------------------------------------------------------------------------------------------ */
static int cmp_hit_desc(const void* pa, const void* pb) {
  const hit_t* a = (const hit_t*)pa;
  const hit_t* b = (const hit_t*)pb;
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  return (a->doc > b->doc) - (a->doc < b->doc);
}

int32_t nrt_oracle_topdocs_merge(int32_t top_n, int32_t n_lists, const int32_t* lens,
                                 const int32_t* docs, const float* scores,
                                 int32_t* out_docs, float* out_scores) {
  int64_t total = 0;
  for (int i = 0; i < n_lists; ++i) total += lens[i];
  hit_t* all = (hit_t*)malloc(sizeof(hit_t) * (size_t)(total > 0 ? total : 1));
  for (int64_t i = 0; i < total; ++i) {
    all[i].doc = docs[i];
    all[i].score = scores[i];
  }
  qsort(all, (size_t)total, sizeof(hit_t), cmp_hit_desc);
  int32_t n = (int32_t)(total < top_n ? total : top_n);
  for (int32_t i = 0; i < n; ++i) {
    out_docs[i] = all[i].doc;
    out_scores[i] = all[i].score;
  }
  free(all);
  return n;
}

/* ------------------------------------------------------------------------------------------
 * Vector similarity -> score.  Mapping table: src/main/java/com/yelp/nrtsearch/server/field/
 * VectorFieldDef.java:77-88; formulas documented by the reference at docs/field_types/
 * vector.rst:26-35 and mirrored by similarityToScore (VectorFieldDef.java:664-673).
 * Lucene's VectorUtil accumulates in float in a JVM-dependent (scalar / Panama) order, so this
 * scalar left-to-right float sum is one member of the tolerance class, not a
 * bit-exact spec.
 * ------------------------------------------------------------------------------------------ */
float nrt_oracle_vector_score(int32_t sim, const float* q, const float* v, int32_t dim) {
  if (sim == 2) { /* EUCLIDEAN: 1 / (1 + squareDistance) */
    float d2 = 0.0f;
    for (int i = 0; i < dim; ++i) {
      volatile float diff = q[i] - v[i];
      volatile float sq = diff * diff;
      d2 = d2 + sq;
    }
    return 1.0f / (1.0f + d2);
  }
  float dot = 0.0f, nq = 0.0f, nv = 0.0f;
  for (int i = 0; i < dim; ++i) {
    volatile float p = q[i] * v[i];
    dot = dot + p;
    if (sim == 0) {
      volatile float a = q[i] * q[i];
      volatile float b = v[i] * v[i];
      nq = nq + a;
      nv = nv + b;
    }
  }
  if (sim == 0) { /* COSINE: max((1 + cos) / 2, 0), cos = (float)(sum / sqrt((double)n1 * n2)) */
    float c = (float)((double)dot / sqrt((double)nq * (double)nv));
    float s = (1.0f + c) / 2.0f;
    return s > 0.0f ? s : 0.0f;
  }
  if (sim == 1) { /* DOT_PRODUCT: max((1 + dot) / 2, 0) */
    float s = (1.0f + dot) / 2.0f;
    return s > 0.0f ? s : 0.0f;
  }
  /* MAXIMUM_INNER_PRODUCT: dot < 0 ? 1 / (1 - dot) : dot + 1 */
  if (dot < 0.0f) return 1.0f / (1.0f - dot);
  return dot + 1.0f;
}

/* Summation orders of the three sums (dot / squareMagnitudes / squareDistance):
 *   order 0  scalar, left to right, one accumulator, multiply and add rounded separately.  THE PINNED ORDER: what the device
 *            returns bit for bit (tests/test_vectors_gpu.py) and what the d = 3 goldens of VectorFieldDefTest.java check (order
 *            1 gives the same bits at d <= 32 -- Lucene's scalar code does not unroll there; order 2 may not: fused rounding).
 *   order 1  [Lucene-recall] lucene-core 10.4.0's DefaultVectorUtilSupport (the non-Panama code a stock JVM runs without
 *            --add-modules jdk.incubator.vector): for d > 32 four accumulators striding the dimension (dotProduct,
 *            squareDistance: acc_j += a[i + j] * b[i + j], i += 4; res = acc1 + acc2 + acc3 + acc4, then the tail left to
 *            right), two for cosine (sum, norm1, norm2 each as a pair); multiply-add as two roundings
 *            (Constants.HAS_FAST_SCALAR_FMA false).
 *   order 2  the same with every multiply-add fused (Math.fma: HAS_FAST_SCALAR_FMA true, what an x86-64 JVM with FMA3 picks).
 * Orders 1 and 2 are restated from memory of the Lucene sources -- not in /root/reference, not runnable here (no JVM) -- and
 * exist to BOUND what "the oracle's bits" can differ from Lucene's by: scripts/cpu_vector_order_study.py counts the rank and
 * score differences between the three at the C4 shape.  Parity claims name order 0. */
static inline float mul_add(int fused, float a, float b, float c) {
  if (fused) return fmaf(a, b, c);
  volatile float p = a * b;
  return p + c;
}

float nrt_oracle_vector_score_order(int32_t order, int32_t sim, const float* q, const float* v, int32_t dim) {
  if (order == 0) return nrt_oracle_vector_score(sim, q, v, dim);
  const int fused = order == 2;
  int i = 0;
  if (sim == 2) { /* squareDistance: 4 accumulators */
    float res = 0.0f;
    if (dim > 32) {
      float a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
      const int ub = dim & ~3;
      for (; i < ub; i += 4) {
        volatile float d1 = q[i] - v[i], d2 = q[i + 1] - v[i + 1], d3 = q[i + 2] - v[i + 2], d4 = q[i + 3] - v[i + 3];
        a1 = mul_add(fused, d1, d1, a1);
        a2 = mul_add(fused, d2, d2, a2);
        a3 = mul_add(fused, d3, d3, a3);
        a4 = mul_add(fused, d4, d4, a4);
      }
      volatile float t = a1 + a2;
      t = t + a3;
      t = t + a4;
      res = res + t;
    }
    for (; i < dim; ++i) {
      volatile float d = q[i] - v[i];
      res = mul_add(fused, d, d, res);
    }
    return 1.0f / (1.0f + res);
  }
  float dot = 0.0f, nq = 0.0f, nv = 0.0f;
  if (sim == 0) { /* cosine: pairs of accumulators */
    if (dim > 32) {
      float s1 = 0.f, s2 = 0.f, q1 = 0.f, q2 = 0.f, v1 = 0.f, v2 = 0.f;
      const int ub = dim & ~1;
      for (; i < ub; i += 2) {
        s1 = mul_add(fused, q[i], v[i], s1);
        q1 = mul_add(fused, q[i], q[i], q1);
        v1 = mul_add(fused, v[i], v[i], v1);
        s2 = mul_add(fused, q[i + 1], v[i + 1], s2);
        q2 = mul_add(fused, q[i + 1], q[i + 1], q2);
        v2 = mul_add(fused, v[i + 1], v[i + 1], v2);
      }
      volatile float ts = s1 + s2, tq = q1 + q2, tv = v1 + v2;
      dot = dot + ts;
      nq = nq + tq;
      nv = nv + tv;
    }
    for (; i < dim; ++i) {
      dot = mul_add(fused, q[i], v[i], dot);
      nq = mul_add(fused, q[i], q[i], nq);
      nv = mul_add(fused, v[i], v[i], nv);
    }
    const float c = (float)((double)dot / sqrt((double)nq * (double)nv));
    const float s_ = (1.0f + c) / 2.0f;
    return s_ > 0.0f ? s_ : 0.0f;
  }
  if (dim > 32) { /* dotProduct: 4 accumulators */
    float a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
    const int ub = dim & ~3;
    for (; i < ub; i += 4) {
      a1 = mul_add(fused, q[i], v[i], a1);
      a2 = mul_add(fused, q[i + 1], v[i + 1], a2);
      a3 = mul_add(fused, q[i + 2], v[i + 2], a3);
      a4 = mul_add(fused, q[i + 3], v[i + 3], a4);
    }
    volatile float t = a1 + a2;
    t = t + a3;
    t = t + a4;
    dot = dot + t;
  }
  for (; i < dim; ++i) dot = mul_add(fused, q[i], v[i], dot);
  if (sim == 1) {
    const float s_ = (1.0f + dot) / 2.0f;
    return s_ > 0.0f ? s_ : 0.0f;
  }
  if (dot < 0.0f) return 1.0f / (1.0f - dot);
  return dot + 1.0f;
}

/* ------------------------------------------------------------------------------------------
 * Exact vector search: ExactVectorQuery scores EVERY doc that has a vector
 * (src/main/java/com/yelp/nrtsearch/server/query/vector/ExactVectorQuery.java:137-173, VectorValuesScorer.score() =
 * similarity.compare(query, vector) * boost) into a top-k collector (score descending, ties by docid ascending: the
 * order TopScoreDocCollector + TopDocs.merge produce).  n_q queries x n rows (row r is doc doc_base + r; live == NULL
 * or bit r of live set = the row's doc is live).  Queries spread over OpenMP threads, rows of one query scored in
 * order (the order does not matter to the result: the comparison is total).
 * out_docs / out_scores: n_q x k, out_n: hits per query.
 * ------------------------------------------------------------------------------------------ */
void nrt_oracle_knn_exact(int32_t sim, const float* queries, int32_t n_q, const float* vecs, int64_t n, int32_t dim,
                          const uint64_t* live, int32_t doc_base, float boost, int32_t k, int32_t n_threads,
                          int32_t* out_docs, float* out_scores, int32_t* out_n) {
  nrt_oracle_knn_exact_order(0, sim, queries, n_q, vecs, n, dim, live, doc_base, boost, k, n_threads, out_docs, out_scores, out_n);
}

/* (order: the summation order of nrt_oracle_vector_score_order) */
void nrt_oracle_knn_exact_order(int32_t order, int32_t sim, const float* queries, int32_t n_q, const float* vecs, int64_t n, int32_t dim,
                                const uint64_t* live, int32_t doc_base, float boost, int32_t k, int32_t n_threads,
                                int32_t* out_docs, float* out_scores, int32_t* out_n) {
  if (n_threads < 1) n_threads = 1;
  /* rows are cut into blocks so that one query also scales over the threads; per (query, block) a sorted top-k, merged after */
  const int64_t block = 1 << 16;
  const int64_t n_blocks = (n + block - 1) / block;
  const int64_t n_tasks = (int64_t)n_q * (n_blocks > 0 ? n_blocks : 1);
  float* bs = (float*)malloc((size_t)n_tasks * (size_t)k * sizeof(float));
  int32_t* bd = (int32_t*)malloc((size_t)n_tasks * (size_t)k * sizeof(int32_t));
  int32_t* bn = (int32_t*)calloc((size_t)n_tasks, sizeof(int32_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int64_t t = 0; t < n_tasks; ++t) {
    const int32_t q = (int32_t)(t / (n_blocks > 0 ? n_blocks : 1));
    const int64_t b = t % (n_blocks > 0 ? n_blocks : 1);
    float* s = bs + (size_t)t * (size_t)k;
    int32_t* d = bd + (size_t)t * (size_t)k;
    int32_t m = 0;
    const int64_t r1 = (b + 1) * block < n ? (b + 1) * block : n;
    for (int64_t r = b * block; r < r1; ++r) {
      if (live && !((live[r >> 6] >> (r & 63)) & 1ull)) continue;
      const float sc = nrt_oracle_vector_score_order(order, sim, queries + (size_t)q * (size_t)dim, vecs + (size_t)r * (size_t)dim, dim) * boost;
      if (m == k && !(sc > s[k - 1])) continue; /* rows come in docid order: an equal score loses to the earlier doc */
      int32_t i = m < k ? m : k - 1;
      while (i > 0 && s[i - 1] < sc) {
        s[i] = s[i - 1];
        d[i] = d[i - 1];
        --i;
      }
      s[i] = sc;
      d[i] = doc_base + (int32_t)r;
      if (m < k) ++m;
    }
    bn[t] = m;
  }
  for (int32_t q = 0; q < n_q; ++q) { /* k-way merge of the blocks' lists (blocks in docid order: ties keep it) */
    int32_t* pos = (int32_t*)calloc((size_t)(n_blocks > 0 ? n_blocks : 1), sizeof(int32_t));
    int32_t m = 0;
    while (m < k) {
      int64_t best = -1;
      for (int64_t b = 0; b < n_blocks; ++b) {
        const int64_t t = (int64_t)q * n_blocks + b;
        if (pos[b] >= bn[t]) continue;
        if (best < 0 || bs[(size_t)t * (size_t)k + (size_t)pos[b]] > bs[(size_t)((int64_t)q * n_blocks + best) * (size_t)k + (size_t)pos[best]]) best = b;
      }
      if (best < 0) break;
      const int64_t t = (int64_t)q * n_blocks + best;
      out_scores[(size_t)q * (size_t)k + (size_t)m] = bs[(size_t)t * (size_t)k + (size_t)pos[best]];
      out_docs[(size_t)q * (size_t)k + (size_t)m] = bd[(size_t)t * (size_t)k + (size_t)pos[best]];
      ++pos[best];
      ++m;
    }
    out_n[q] = m;
    free(pos);
  }
  free(bs);
  free(bd);
  free(bn);
}

/* QueryRescore.combine, src/main/java/com/yelp/nrtsearch/server/rescore/QueryRescore.java:40-45 */
float nrt_oracle_rescore_combine(float first_pass, int32_t matched, float second_pass,
                                 double query_weight, double rescore_weight) {
  if (!matched) return (float)(query_weight * (double)first_pass);
  return (float)(query_weight * (double)first_pass + rescore_weight * (double)second_pass);
}
