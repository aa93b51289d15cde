/*
 * nrt_oracle.h -- CPU restatement of the nrtsearch / Lucene 10.4 query-execution hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nrtsearch_amd/ may include, link or dlopen this.
 * Allowed users: tests/, __graft_entry__.smoke(), and bench.py's cpu_baseline leg (as the
 * checker / reported baseline, never as the thing shipped).
 *
 * Parity status: the scalar arithmetic (SmallFloat, idf, avgdl, norm cache, BM25 score) is
 * PINNED against the reference's own golden values (tests/test_oracle_golden.py; SURVEY.md
 * section 8c).  Multi-term double-accumulated sums, norm quantisation for long fields and kNN at
 * d=768 are "parity unpinned": the reference holds no golden vector for them and Lucene 10.4.0
 * itself (un-vendored Maven dependency, gradle/libs.versions.toml:7) cannot be run here.
 *
 * Every function cites the reference file:line (relative to /root/reference) or, for logic that
 * lives only inside lucene-core 10.4.0, the published Lucene class it restates.
 */
#ifndef NRT_ORACLE_H
#define NRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- SmallFloat (org.apache.lucene.util.SmallFloat, Lucene 10.4.0) ---- */
int32_t nrt_oracle_int_to_byte4(int32_t i);   /* returns 0..255 */
int32_t nrt_oracle_byte4_to_int(int32_t b);   /* b in 0..255 */

/* ---- BM25Similarity (default instance: src/main/java/com/yelp/nrtsearch/server/similarity/
 *      SimilarityCreator.java:33,41; k1 = 1.2f, b = 0.75f, discountOverlaps = true) ---- */
float nrt_oracle_bm25_idf(int64_t doc_count, int64_t doc_freq);
float nrt_oracle_bm25_avgdl(int64_t sum_total_term_freq, int64_t doc_count);
void  nrt_oracle_bm25_norm_cache(float avgdl, float k1, float b, float out256[256]);
float nrt_oracle_bm25_score(float weight, float freq, float norm_inverse);

/* ---- top-k collector: restates src/main/java/org/apache/lucene/search/
 *      LazyQueueTopScoreDocCollector.java:37-203 (== Lucene TopScoreDocCollector) ---- */
typedef struct nrt_oracle_collector nrt_oracle_collector;

nrt_oracle_collector* nrt_oracle_collector_new(int32_t num_hits, int32_t has_after, int32_t after_doc,
                                               float after_score, int32_t total_hits_threshold);
void nrt_oracle_collector_free(nrt_oracle_collector*);
/* getLeafCollector(ctx): rebases `after` to the leaf (…Collector.java:73-84) */
void nrt_oracle_collector_set_leaf(nrt_oracle_collector*, int32_t doc_base);
/* LeafCollector.collect(doc) with scorer.score() == score (…Collector.java:103-144) */
void nrt_oracle_collector_collect(nrt_oracle_collector*, int32_t leaf_doc, float score);
/* TopDocsCollector.topDocs(): pops the queue into (score desc, doc asc). returns n_hits. */
int32_t nrt_oracle_collector_topdocs(nrt_oracle_collector*, int32_t* docs, float* scores,
                                     int64_t* total_hits, int32_t* total_hits_is_lower_bound);

/* ---- one term's postings inside one segment, as the reference sees them through
 *      PostingsEnum (docid ascending, freq) + NumericDocValues norms ---- */
typedef struct {
  const int32_t* docids;   /* ascending */
  const int32_t* freqs;    /* NULL => freq == 1 (DOCS-only field, AtomFieldDef.java:123-126) */
  int64_t        n;
  float          weight;   /* boost * idf  (BM25Similarity.scorer) */
  const uint8_t* norms;    /* max_doc bytes, NULL => norm value 1 (norms omitted) */
  const float*   cache;    /* 256 floats: normInverse table for this field */
} nrt_oracle_term;

/*
 * Exhaustive disjunction (pure SHOULD, minShouldMatch <= 1) over one segment, scored exactly as
 * Lucene does: per (term, doc) float BM25 score, per doc the sum over matching terms accumulated
 * in double then cast to float (MaxScoreBulkScorer / BooleanScorer / WANDScorer all do this; a
 * single TermQuery returns the float itself, which the same cast reproduces), docs delivered to
 * the collector in ascending docid order, deleted docs (live_bits) never delivered.
 * Implemented as windowed term-at-a-time over 4096-doc windows (the shape of Lucene's
 * MaxScoreBulkScorer inner window), so it is also a reasonable CPU baseline.
 */
void nrt_oracle_search_segment(int32_t max_doc, int32_t doc_base, const uint64_t* live_bits,
                               int32_t n_terms, const nrt_oracle_term* terms,
                               nrt_oracle_collector* collector);

/* The same with minimumNumberShouldMatch (QueryNodeMapper.java:259-261): hits need that many matching
 * clauses, scores still sum every matching clause. */
void nrt_oracle_search_segment_msm(int32_t max_doc, int32_t doc_base, const uint64_t* live_bits,
                                   int32_t n_terms, const nrt_oracle_term* terms, int32_t min_should_match,
                                   nrt_oracle_collector* collector);
/* MUST (required[t] != 0) and SHOULD term clauses, minimumNumberShouldMatch 0: hits match every MUST clause, score =
 * (float) sum of the MUST scores + (float) sum of the matching SHOULD scores, added in float (ReqOptSumScorer) */
void nrt_oracle_search_segment_reqopt(int32_t max_doc, int32_t doc_base, const uint64_t* live_bits,
                                      int32_t n_terms, const nrt_oracle_term* terms, const uint8_t* required,
                                      nrt_oracle_collector* collector);
/* DisjunctionMaxQuery over the same term clauses (QueryNodeMapper.java:350-358): best clause + tie_breaker * the others */
void nrt_oracle_search_segment_dismax(int32_t max_doc, int32_t doc_base, const uint64_t* live_bits,
                                      int32_t n_terms, const nrt_oracle_term* terms, float tie_breaker,
                                      nrt_oracle_collector* collector);

/*
 * The same search with MaxScore dynamic pruning (the algorithm family of Lucene's
 * MaxScoreBulkScorer): identical top-k, totalHits a lower bound.
 * block_max[t] = per-128-posting-block maximum score of clause t (nrt_oracle_block_max; the role
 * of Lucene's level-0 impacts, computed once per (term, weight) outside any timed region).
 * *postings_scored (may be NULL) is incremented by the postings actually scored.
 */
void nrt_oracle_block_max(const nrt_oracle_term* term, float* out /* ceil(n / 128) */);
void nrt_oracle_search_segment_maxscore(int32_t max_doc, int32_t doc_base, const uint64_t* live_bits,
                                        int32_t n_terms, const nrt_oracle_term* terms,
                                        const float* const* block_max,
                                        nrt_oracle_collector* collector, int64_t* postings_scored);

/* One leaf of one query, prepared by the caller (block_max may be NULL when maxscore == 0). */
typedef struct {
  int32_t max_doc;
  int32_t doc_base;
  const uint64_t* live_bits;
  int32_t n_terms;
  const nrt_oracle_term* terms;
  const float* const* block_max;
} nrt_oracle_leaf;

/* n_queries independent searches (leaves[leaf_offsets[q] .. leaf_offsets[q+1]) in docBase order,
 * one collector each) over n_threads OpenMP threads; outputs are [n_queries][k]. */
void nrt_oracle_search_batch(int32_t n_queries, const int64_t* leaf_offsets,
                             const nrt_oracle_leaf* leaves, int32_t k, int32_t total_hits_threshold,
                             int32_t maxscore, int32_t n_threads, int32_t* out_docs, float* out_scores,
                             int32_t* out_n, int64_t* out_total_hits, int32_t* out_gte,
                             int64_t* postings_scored);

/* TopDocs.merge(0, topN, shardHits[]) with all shardIndex == -1: order (score desc, doc asc)
 * (LazyQueueTopScoreDocCollectorManager.java:137-144).  Lists are concatenated in `docs/scores`
 * with lengths `lens[n_lists]`.  Returns number of merged hits written. */
int32_t nrt_oracle_topdocs_merge(int32_t top_n, int32_t n_lists, const int32_t* lens,
                                 const int32_t* docs, const float* scores,
                                 int32_t* out_docs, float* out_scores);

/* ---- exact vector scoring (ExactVectorQuery.java:137-173 + VectorSimilarityFunction) ----
 * sim: 0 cosine, 1 dot_product, 2 l2_norm (euclidean), 3 max_inner_product. */
float nrt_oracle_vector_score(int32_t sim, const float* q, const float* v, int32_t dim);
/* ExactVectorQuery over one matrix of rows + the top-k collector behind it (ExactVectorQuery.java:137-173) */
void nrt_oracle_knn_exact(int32_t sim, const float* queries, int32_t n_q, const float* vecs, int64_t n, int32_t dim,
                          const uint64_t* live, int32_t doc_base, float boost, int32_t k, int32_t n_threads,
                          int32_t* out_docs, float* out_scores, int32_t* out_n);

/* The same under another summation order of the similarity's sums (nrt_oracle.c: 0 = the pinned scalar left-to-right order,
 * 1 / 2 = [Lucene-recall] DefaultVectorUtilSupport's unrolled accumulators without / with fused multiply-add): how far
 * "the oracle's bits" can be from a given Lucene build's.  Test infrastructure for scripts/cpu_vector_order_study.py. */
float nrt_oracle_vector_score_order(int32_t order, int32_t sim, const float* q, const float* v, int32_t dim);
void nrt_oracle_knn_exact_order(int32_t order, int32_t sim, const float* queries, int32_t n_q, const float* vecs, int64_t n, int32_t dim,
                                const uint64_t* live, int32_t doc_base, float boost, int32_t k, int32_t n_threads,
                                int32_t* out_docs, float* out_scores, int32_t* out_n);

/* QueryRescorer.combine as overridden by QueryRescore.java:40-45:
 * (float)(queryWeight * firstPass + rescoreWeight * secondPass), double arithmetic. */
float nrt_oracle_rescore_combine(float first_pass, int32_t matched, float second_pass,
                                 double query_weight, double rescore_weight);

#ifdef __cplusplus
}
#endif
#endif
