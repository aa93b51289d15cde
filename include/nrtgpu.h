/*
 * nrtgpu.h -- C ABI of the MI355X-native query-execution path for nrtsearch.
 *
 * This is the drop-in boundary: everything nrtsearch does below
 *   searcher.search(query, collectorManager)
 *     src/main/java/com/yelp/nrtsearch/server/handler/SearchHandler.java:1412-1413 (single query)
 *     src/main/java/com/yelp/nrtsearch/server/handler/SearchHandler.java:556        (per retriever)
 * for an eligible query (pure-SHOULD BooleanQuery of TermQuery clauses / single TermQuery under
 * the default BM25Similarity, or an exact float vector query) is replaced by calls into this
 * library.  The reference has no native boundary of its own (pure JVM); the JNI / FFM stub a
 * maintainer adds on the Java side is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns NRTGPU_OK (0) or a negative nrtgpu_status; nrtgpu_last_error()
 *    returns a thread-local UTF-8 message for the last failure on the calling thread.
 *  - thread-safety: any number of host threads may call the search / knn / rescore functions
 *    concurrently on one ctx (SEARCH-pool threads, src/main/java/com/yelp/nrtsearch/server/
 *    concurrent/ExecutorFactory.java:80-117).  Segment lifecycle calls for one nrtgpu_seg must not
 *    race with each other; releasing a segment that a running search uses is undefined.
 *  - ownership: input pointers are borrowed for the duration of the call; outputs are written
 *    into caller-allocated arrays of the stated capacity.  No callbacks into the caller.
 *  - plain C types only: no torch / HIP types cross this boundary.
 *  - there is NO CPU fallback inside the library: with no usable gfx950 device nrtgpu_create
 *    fails with NRTGPU_ERR_HIP; NRTGPU_ERR_UNSUPPORTED tells the caller to run its own
 *    (Lucene) path for that query.
 */
#ifndef NRTGPU_H
#define NRTGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  NRTGPU_OK = 0,
  NRTGPU_ERR_INVALID_ARG = -1, /* IllegalArgumentException in the reference (e.g. numHits <= 0) */
  NRTGPU_ERR_HIP = -2,         /* device / runtime failure -> IOException -> gRPC INTERNAL */
  NRTGPU_ERR_OOM = -3,
  NRTGPU_ERR_UNSUPPORTED = -4, /* shape not handled on device: caller falls back to Lucene */
  NRTGPU_ERR_STATE = -5,       /* lifecycle misuse (e.g. search on an unsealed segment) */
  NRTGPU_ERR_TIMEOUT = -6      /* the calling thread's deadline passed before the work was launched (nrtgpu_set_thread_deadline_ns):
                                * gRPC DEADLINE_EXCEEDED / SearchResponse.hitTimeout */
} nrtgpu_status;

typedef struct nrtgpu_ctx nrtgpu_ctx;
typedef struct nrtgpu_seg nrtgpu_seg;

/* Limits of the device fast path (queries outside them get NRTGPU_ERR_UNSUPPORTED). */
#define NRTGPU_MAX_K 1024          /* numHits handled by the LDS top-k */
#define NRTGPU_MAX_TERMS 32        /* SHOULD clauses per query */
#define NRTGPU_MAX_MASKS 8         /* FILTER (and, separately, MUST_NOT) clauses per query */
#define NRTGPU_TILE_DOCS 1024      /* docs per wave-private LDS accumulator sub-tile */

typedef struct {
  int32_t device_id;        /* HIP device ordinal this ctx owns (one process per GPU) */
  int32_t max_batch;        /* max queries per batch call; 0 => 1024 */
  int32_t target_items;     /* work items (query x doc-range) aimed for per batch; 0 => auto */
  int32_t collect_timing;   /* !=0: bracket the scan kernel with HIP events (nrtgpu_get_stats) */
  int32_t flags;            /* NRTGPU_FLAG_* */
  int32_t host_threads;     /* planner threads used inside one batch call (term-dictionary lookups); 0 => 4 */
  int32_t lookup_budget_pct; /* HBM a segment may spend on doc -> posting lookup structures of the dynamic-pruning route (per term a
                             * 16-bit code map or lookup cells), in percent of its resident posting bytes, granted to the largest
                             * terms first; 0 => 150; < 0 => none (every lookup is a binary search in the postings: slower, same
                             * results) */
  int32_t reserved;
} nrtgpu_config;

#define NRTGPU_FLAG_NO_PREFETCH 1     /* scan kernel without the one-tile-ahead posting prefetch (A/B) */
#define NRTGPU_FLAG_NO_FIXED_POINT 2  /* always accumulate in fp64 (A/B; results are identical either way) */
#define NRTGPU_FLAG_NO_LIVE_FOLD 8     /* A/B: liveDocs stay a mask read by the scan instead of being folded into the posting columns */
#define NRTGPU_FLAG_NO_MASK_VARIANT 4   /* A/B: docs outside liveDocs / a mask are checked one by one (general sweep) */
#define NRTGPU_FLAG_PACKED_POSTINGS 32  /* compressed postings: every segment of this context keeps ONE 32-bit word per
                                        * posting in HBM (20-bit doc offset inside its 2^20-doc super-window | 12-bit score code) instead of
                                        * a docid and a code column: half the posting bytes, same results bit for bit.  liveDocs are then not
                                        * folded into the postings (the scorers test the mask).  Postings the 12-bit code cannot name (freq > 12
                                        * or norm byte >= 128) go through a per-upload-group exception list (4 B each).  A separately
                                        * reported configuration (its own roofline denominator). */
#define NRTGPU_FLAG_BLOCKING_WAIT 64    /* callers sleep until their results are there instead of spinning on the stream: for deployments
                                        * where several processes (one per GPU) with several calls in flight each share the host's CPUs.
                                        * Costs a wake-up (tens of microseconds) per call */
#define NRTGPU_FLAG_NO_VECTOR_SKETCH 128 /* vector fields keep no fp16 copy of their rows (+50 % of the fp32 matrix, built by a field's first exact search): the exact search then
                                         * nominates from the fp32 rows (2x the bytes per pass, 32 queries per pass instead of 64).
                                         * Results are the same bits either way */
#define NRTGPU_FLAG_NO_PRUNE 16        /* never take the MaxScore route: every query is scanned exhaustively and total_hits is
                                        * always the exact count (the relation still follows totalHitsThreshold) */

const char* nrtgpu_version(void);
const char* nrtgpu_last_error(void);

int  nrtgpu_create(const nrtgpu_config* cfg, nrtgpu_ctx** out);
void nrtgpu_destroy(nrtgpu_ctx* ctx);

/* ---------------------------------------------------------------------------------------------
 * Segment store: a read-only columnar replica of one immutable Lucene segment's scoring data,
 * filled through Lucene's public reader APIs at searcher-refresh / warm time
 * (hook: ShardSearcherFactory.newSearcher, src/main/java/com/yelp/nrtsearch/server/index/
 * ShardState.java:506-527; keyed by the segment core CacheKey on the Java side).
 * --------------------------------------------------------------------------------------------- */
int  nrtgpu_segment_begin(nrtgpu_ctx* ctx, int32_t max_doc, int32_t device_hint, nrtgpu_seg** out);
/* norms of one field: leaf.getNormValues(field) as bytes (SmallFloat.intToByte4 of the field
 * length); NULL => norms omitted, norm value 1 for every doc (AtomFieldDef.java:123-126). */
int  nrtgpu_segment_add_field_norms(nrtgpu_seg* seg, int32_t field_id, const uint8_t* norm_bytes);
/* postings of n_terms terms of one field: TermsEnum/PostingsEnum flattened column-major.
 * term_hash identifies a term (any injective 64-bit id chosen by the caller); offsets has
 * n_terms+1 entries into docids/freqs; docids ascending within a term; freqs NULL => all 1
 * (IndexOptions.DOCS).  May be called several times per segment/field. */
int  nrtgpu_segment_add_terms(nrtgpu_seg* seg, int32_t field_id, int64_t n_terms, const int64_t* term_hash,
                              const int64_t* offsets, const int32_t* docids, const int32_t* freqs);
/* float vectors of one field (leaf.getFloatVectorValues): n rows of `dim` fp32, row-major;
 * ord_to_doc NULL => ordinal == docid (dense).  Any dim <= 2048 (more: NRTGPU_ERR_UNSUPPORTED, the field stays on the
 * caller's path): rows are kept zero-padded to a multiple of 16 elements and query vectors are padded alike inside the
 * search calls -- zeros add nothing to a dot product, a squared norm or a squared distance, so scores are the field's own
 * (the reference's vector tests run at dim = 3: VectorFieldDefTest.java:1885-1965).  Queries pass the field's dim. */
int  nrtgpu_segment_add_vectors(nrtgpu_seg* seg, int32_t field_id, int32_t dim, int32_t n,
                                const int32_t* ord_to_doc, const float* row_major);
/* builds the per-term doc-range tables; the segment becomes searchable */
int  nrtgpu_segment_seal(nrtgpu_seg* seg);
/* leaf.getLiveDocs() as 64-bit words, bit d set = doc d live; NULL => all live.  May be called
 * again after seal (only liveDocs change between reader versions of one segment).  Costs one device
 * pass over the segment's postings: the postings of deleted docs are re-coded to score the neutral
 * element, so searches pay nothing per query for deletes.  Thread-safe against searches: the call waits for
 * the searches running over this segment and later ones wait for it (they see the new liveDocs; one
 * liveDocs version per segment handle at a time). */
int  nrtgpu_segment_set_live_docs(nrtgpu_seg* seg, const uint64_t* bits, int32_t n_words);
/* A new reader version of a sealed segment (what a refresh produces when only liveDocs changed): a handle that SHARES
 * the segment's postings, norms and vectors (nothing is copied or re-uploaded) and carries its own liveDocs (bits as
 * above; NULL = all live) and its own doc-set masks.  Searches over `seg` keep seeing `seg`'s liveDocs -- the
 * point-in-time view an IndexSearcher has in Lucene -- and nobody waits for anybody.  The shared data is freed with the
 * last handle (nrtgpu_segment_release on each).  While a segment has several handles its deletes are tested as a mask
 * instead of being folded into the postings, and -- as in Lucene, where a segment's deletes only accumulate -- a handle's
 * liveDocs may not bring back a doc that the shared postings already carry as deleted (NRTGPU_ERR_UNSUPPORTED). */
int  nrtgpu_segment_fork(nrtgpu_seg* seg, const uint64_t* live_bits, int32_t n_words, nrtgpu_seg** out);
/* Non-scoring clauses as doc-set masks (FILTER / MUST_NOT of the BooleanQuery built at
 * src/main/java/com/yelp/nrtsearch/server/query/QueryNodeMapper.java:257-283).  The shim materialises the
 * clause's per-leaf DocIdSet (what LRUQueryCache caches) as 64-bit words, bit d set = doc d matches,
 * and registers it under an id > 0 of its choosing; queries name ids (nrtgpu_bm25_query.filter_mask /
 * must_not_mask).  bits == NULL drops the mask.  Like set_live_docs: excludes itself from the searches
 * running over this segment. */
int  nrtgpu_segment_set_mask(nrtgpu_seg* seg, int32_t mask_id, const uint64_t* bits, int32_t n_words);
void nrtgpu_segment_release(nrtgpu_seg* seg);
/* bytes of HBM held by the segment (diagnostics) */
int64_t nrtgpu_segment_device_bytes(const nrtgpu_seg* seg);

/* ---------------------------------------------------------------------------------------------
 * BM25 disjunction search.  Replaces, for one IndexSearcher.search call: per-leaf
 * Weight.scorerSupplier(ctx).bulkScorer().score(leafCollector, liveDocs, 0, maxDoc) (postings
 * traversal + BM25Similarity SimScorer + double-accumulated disjunction sum), the
 * TopScoreDocCollector / LazyQueueTopScoreDocCollector (src/main/java/org/apache/lucene/search/
 * LazyQueueTopScoreDocCollector.java:103-199) and CollectorManager.reduce == TopDocs.merge
 * (LazyQueueTopScoreDocCollectorManager.java:137-144).
 * Index-global statistics stay on the host: the caller passes weight = boost * idf and the
 * 256-entry normInverse cache of BM25Similarity.scorer() (helpers below compute both).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t field_id;
  int32_t cache_slot;     /* which 256-float table of nrtgpu_bm25_query.norm_cache this term uses */
  int64_t term_hash;
  float   weight;         /* boost * idf, float (BM25Similarity.scorer) */
  int32_t occur;          /* 0: SHOULD; 1: MUST (BooleanClause.Occur, QueryNodeMapper.java:257-283).  Every clause MUST: the
                           * conjunction (all clauses match, score = (float) of the double sum -- ConjunctionScorer), the same as
                           * min_should_match = n_terms over SHOULD clauses.  MUST next to SHOULD clauses (min_should_match 0):
                           * a hit matches every MUST clause and scores (float) sum of the MUST scores + (float) sum of its
                           * matching SHOULD scores, the two added in float (ReqOptSumScorer [Lucene-recall]); MaxScore route
                           * only (<= 8 clauses, fixed-point sums, not ScoreMode.COMPLETE on a large query), else
                           * NRTGPU_ERR_UNSUPPORTED; with min_should_match > 0 or disjunction_max: NRTGPU_ERR_UNSUPPORTED */
} nrtgpu_term;

typedef struct {
  int32_t n_terms;                 /* 1..NRTGPU_MAX_TERMS SHOULD clauses (duplicates allowed) */
  const nrtgpu_term* terms;
  int32_t n_caches;                /* number of 256-float tables in norm_cache (one per field) */
  const float* norm_cache;         /* n_caches * 256 floats */
  int32_t k;                       /* numHits, 1..NRTGPU_MAX_K  (DocCollector.java:79-114) */
  int32_t total_hits_threshold;    /* >= 0; INT32_MAX => exact count (ScoreMode.COMPLETE) */
  int32_t has_after;               /* searchAfter (LazyQueueTopScoreDocCollector.java:112-120) */
  int32_t after_doc;               /* global docid of the last hit of the previous page */
  float   after_score;
  int32_t min_should_match;        /* minimumNumberShouldMatch (QueryNodeMapper.java:259-261): 0 and 1 are the plain
                                    * disjunction; > 1: only docs matched by that many clauses are hits, the score
                                    * is still the sum over all matching clauses (what Lucene's WANDScorer returns).
                                    * > 1 on the exhaustive route needs the fixed-point accumulators for the whole batch,
                                    * else NRTGPU_ERR_UNSUPPORTED (nrtgpu_search_bm25_coalesced: for that request only) */
  float   min_competitive_score;   /* Scorable.setMinCompetitiveScore across shards: a lower bound of the k-th best
                                    * score of the WHOLE search this call is one shard of (other GPUs' results so
                                    * far, LazyMaxScoreAccumulator).  Docs scoring strictly below it are counted in
                                    * total_hits but not collected; 0 = none */
  int32_t filter_mask;             /* 0 = none; else hits must lie in this registered mask on every segment: a
                                    * FILTER clause next to the SHOULD clauses with minimumNumberShouldMatch = 1,
                                    * or "+(should clauses) #filter" -- either way a hit matches >= 1 scoring
                                    * clause and the filter adds nothing to the score */
  int32_t must_not_mask;           /* 0 = none; else hits must NOT lie in this mask (MUST_NOT clause) */
  int32_t disjunction_max;         /* 0: BooleanQuery, a doc scores the sum of its matching clauses.
                                    * 1: DisjunctionMaxQuery over the same term clauses (src/main/java/com/yelp/nrtsearch/
                                    * server/query/QueryNodeMapper.java:350-358): a doc scores its BEST matching clause plus
                                    * tie_breaker x the others.  Fixed-point sums (on the exhaustive route: for the whole batch), else
                                    * NRTGPU_ERR_UNSUPPORTED; min_should_match <= 1, every clause SHOULD; disjuncts that are
                                    * not term queries stay on the caller's path */
  int32_t n_more_filters;          /* further FILTER clauses next to filter_mask (QueryNodeMapper.java:257-283 builds any number): */
  const int32_t* more_filters;     /* ... resident mask ids > 0; a hit lies in ALL of the query's filter masks */
  int32_t n_more_must_not;         /* further MUST_NOT clauses next to must_not_mask: */
  const int32_t* more_must_not;    /* ... resident mask ids > 0; a hit lies in NONE of the query's must_not masks.  The masks are
                                    * combined at plan time (one AND / AND NOT pass over 64-bit words per leaf and combination,
                                    * cached on the segment like a single pair); n_more_* = 0: the arrays are not read */
  float   tie_breaker;             /* DisjunctionMaxQuery.tieBreakerMultiplier, 0..1 (disjunction_max = 1 only, else 0): the score
                                    * is (float)(scoreMax + otherScoreSum * tieBreaker) in double, as DisjunctionMaxScorer
                                    * computes it [Lucene-recall]; 0 is what the reference's own test uses (src/test/java/com/
                                    * yelp/nrtsearch/server/grpc/QueryTest.java:541-583).  > 0: MaxScore route only (see
                                    * nrtgpu_term.occur), else NRTGPU_ERR_UNSUPPORTED */
  int32_t reserved;
} nrtgpu_bm25_query;

typedef struct {
  int32_t  n_hits;                     /* out: hits written, <= k */
  int32_t  capacity;                   /* in : capacity of docs/scores (>= k) */
  int32_t* docs;                       /* out: global docids (doc_base + leaf doc) */
  float*   scores;                     /* out: (score desc, doc asc) */
  int64_t  total_hits;                 /* out: live matching docs: the exact number, or -- when the query ran with dynamic pruning
                                        * (MaxScore route; total_hits_is_lower_bound = 1) -- a lower bound above totalHitsThreshold.
                                        * WHICH lower bound: where the planner knew beforehand that some slice passes the threshold
                                        * (a plain disjunction whose largest term alone does), the live docs it knew to match -- the
                                        * same number on every run; where the query had to count its way to the threshold (masks,
                                        * minimumNumberShouldMatch, DisjunctionMax, an uncertain count), the docs the walk had
                                        * evaluated -- which depends on when its workgroups saw the threshold passed and on what the
                                        * bounds then skipped: NOT the same from run to run (Lucene's own value there is an artefact
                                        * of its traversal too).  The relation and the returned hits are deterministic either way. */
  int32_t  total_hits_is_lower_bound;  /* out: 1 == GREATER_THAN_OR_EQUAL_TO: some slice of the searcher (nrtgpu_set_slicing)
                                        * collected more than max(totalHitsThreshold, numHits) hits and numHits hits were returned */
} nrtgpu_topdocs;

/* Slicing of the searcher this context serves (MyIndexSearcher.SlicingParams: the index live settings sliceMaxDocs,
 * sliceMaxSegments, virtualShards; src/main/java/com/yelp/nrtsearch/server/search/MyIndexSearcher.java:79-208).  The
 * reference runs one collector per slice and merges: TotalHits.relation is GREATER_THAN_OR_EQUAL_TO iff SOME SLICE
 * collected more than max(totalHitsThreshold, numHits) hits (LazyQueueTopScoreDocCollector.java:176-199 per slice,
 * LazyQueueTopScoreDocCollectorManager.java:137-144) -- 1500 hits spread over ten slices are EQUAL_TO 1500.  The
 * library derives the same slices from the leaves of each call (maxDoc, live docs, docBase) and reports the relation
 * by that rule; dynamic pruning is used only where some slice certainly passes the threshold.  Defaults: 250000, 5, 1.
 * slice_max_docs == 0: the whole search counts as one slice. */
int  nrtgpu_set_slicing(nrtgpu_ctx* ctx, int32_t slice_max_docs, int32_t slice_max_segments, int32_t virtual_shards);
/* PARTIAL RESIDENCY ("non-resident segments are searched by Lucene and merged with TopDocs.merge").  Under NRT refresh a
 * searcher usually holds a few young segments that are not resident yet (ShardState.java:506-527).  The caller then splits the
 * searcher's SLICES (IndexSearcher.getSlices(), MyIndexSearcher.java:79-208): the slices whose leaves are all resident go to
 * nrtgpu_search_bm25* in ONE call over their leaves, the other slices run through Lucene's own collectors, and the per-slice
 * results are reduced as the reference reduces them -- TopDocs.merge, totalHits summed, GREATER_THAN_OR_EQUAL_TO if any part's is
 * (LazyQueueTopScoreDocCollectorManager.java:137-144).  Statistics stay index-global (the weights the caller passes).
 * A call over a SUBSET of the searcher's leaves must count its hits by the WHOLE searcher's slices: with the default slicing the
 * library's own slicing of whole slices' leaves reproduces them (leaves are packed in size order; a removed slice removes a whole
 * run), with virtual shards it does not (leaves are dealt to shards over all leaves first).  So the calling thread may state the
 * slice of every leaf of its next calls: slice_of_leaf[i] >= 0 for the call's i-th leaf (any numbering; equal numbers = one slice),
 * n_leaves = the call's leaf count (a call with another count ignores the override and slices by itself); NULL / 0 clears it.
 * Thread-local, like the deadline: set before the search call, cleared after. */
int  nrtgpu_set_thread_slices(const int32_t* slice_of_leaf, int32_t n_leaves);

int  nrtgpu_search_bm25(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                        const nrtgpu_bm25_query* q, nrtgpu_topdocs* out);
/* The eligibility test alone (the predicate a GpuIndexSearcher applies to the rewritten query before it
 * chooses between this library and super.search): NRTGPU_OK if nrtgpu_search_bm25 would run `q` over these leaves,
 * else the status it would return (NRTGPU_ERR_UNSUPPORTED: too many clauses / fields, numHits > NRTGPU_MAX_K, a mask
 * that is not resident on a leaf, minimumNumberShouldMatch > 1 without the fixed-point range, ...) with the reason
 * in nrtgpu_last_error().  No device work. */
int  nrtgpu_query_supported(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, int32_t n_segs, const nrtgpu_bm25_query* q);
/* n_queries independent searches over the same leaves in one device pass */
int  nrtgpu_search_bm25_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                              const nrtgpu_bm25_query* queries, int32_t n_queries, nrtgpu_topdocs* out);

/* The same single search, but concurrent callers (the SEARCH / gRPC pool's threads, each blocked in its own
 * call: SearchHandler.java:1412 runs on the request thread) are coalesced by the library into device batches:
 * a caller that finds no batch forming waits at most `linger_us` (default 150) for company -- less when, with nothing in
 * flight, as many callers wait as the last batch held (the cohort of a closed loop is back) -- then runs everybody's
 * queries over the same leaves as one batch.  No extra thread; results identical to nrtgpu_search_bm25. */
int  nrtgpu_search_bm25_coalesced(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                  const nrtgpu_bm25_query* q, nrtgpu_topdocs* out);
int  nrtgpu_set_coalescing(nrtgpu_ctx* ctx, int32_t linger_us);
/* Speculative thresholds of the MaxScore route (nrtgpu_search_bm25 / _batch / _coalesced / nrtgpu_search_hybrid_batch
 * -- the calls that can run a query again): a workgroup that has walked a fraction of a query's docs guesses the final k-th score from the best of what
 * it has seen, `margin` standard deviations on the safe side, and skips what cannot reach the guess; the merge checks every
 * guess against the merged list and a query whose guess failed is run again without speculation inside the same call.  Results
 * are exact either way.  margin 0 switches it off; a context starts with 5.  The library judges every leaf set (the segments of
 * one searcher version) by its own failures: more than 2 % of >= 2048 queries run again moves the leaf set to a scattered window
 * order, and if the guesses fail there too speculation is switched off for it.  Resets the counters nrtgpu_get_stats reports
 * (spec_queries / spec_reruns / spec_scattered / spec_disabled) and every leaf set's verdict. */
int  nrtgpu_set_speculation(nrtgpu_ctx* ctx, float margin);

/* Device-resident variant for the multi-GPU path (one process per GPU): results stay
 * in HBM as packed keys so the caller can RCCL all-gather them without a host round trip.
 *   d_keys  : n_queries * k_stride uint64 (device), key = (float_bits(score) << 32) | (0xFFFFFFFF - doc),
 *             sorted descending == (score desc, doc asc); unused tail slots are 0
 *   d_counts: n_queries uint32 (device) hits per query
 *   d_hits  : n_queries uint64 (device) total hits per query: the exact count, or -- for a query that ran with dynamic
 *             pruning -- (1 << 48) + a lower bound above totalHitsThreshold; sums of these over shards keep both
 *             parts, and nrtgpu_merge_topk_device reports such a sum as GREATER_THAN_OR_EQUAL_TO
 * Returns after the work is enqueued AND complete on the library's stream (synchronous). */
int  nrtgpu_search_bm25_batch_device(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                     int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                     int32_t k_stride, void* d_keys, void* d_counts, void* d_hits);
/* The same in two halves, for callers that pipeline: _begin plans the batch, enqueues its kernels on one of the library's streams
 * and returns at once (the plan, the workspace and shared locks on the segments' content stay with the pending handle);
 * nrtgpu_pending_wait blocks until the results are complete in d_keys / d_counts / d_hits and releases everything.  ONE
 * submitting thread then keeps the device fed -- the plan of batch i + 1 is built while the kernels of batch i run -- and
 * nobody spins on a stream in between (a rank of a multi-GPU job: ~1 CPU instead of one per call in flight).  At most 4
 * batches in flight per context (a fifth _begin waits for a workspace); every handle must be waited for exactly once.
 * epoch: as nrtgpu_search_bm25_batch_device_epoch below (-1: no bound exchange). */
/* Lifetime contract of a pending search: it holds its segments' content from begin to wait (set_live_docs / set_mask on those
 * handles wait for it; nrtgpu_segment_release is deferred to it).  nrtgpu_pending_wait may be called from any thread.  A thread
 * that holds un-waited pending searches may begin more (they pass a writer that is waiting for the earlier ones) but must not
 * start a SYNCHRONOUS search over the same handles before it has waited: that one queues behind the waiting writer. */
typedef struct nrtgpu_pending nrtgpu_pending;
int  nrtgpu_search_bm25_batch_device_begin(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                           int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                           int32_t k_stride, void* d_keys, void* d_counts, void* d_hits, int64_t epoch,
                                           nrtgpu_pending** out);
int  nrtgpu_pending_wait(nrtgpu_pending* pending);
/* One SHARD of a search that `spec_world` GPUs share by equal docid ranges (the multi-GPU form below, for callers that pipeline
 * it themselves): as nrtgpu_search_bm25_batch_device_begin without an epoch, but the speculative thresholds of this call are
 * guesses at the k-th score of the WHOLE search -- a shard's docs are a 1 / spec_world sample of the index, so after a fraction f
 * of its windows it holds about k f / spec_world of the final top-k and the (that + z sd + 2)-th best key it has seen is the
 * guess (csrc/maxscore.hip: ms_compact).  Every shard then collects about k / spec_world candidates instead of k and no bound
 * travels between the GPUs while they walk.  A guess can only be checked against the MERGED list: the largest guess per query is
 * left in d_guess (n_queries x u64 in HBM, 0: none) -- nrtgpu_dist_exchange_merge_checked takes it along, checks it and says
 * which queries every shard has to run again (spec_world 0: without speculation).  The results are exact either way.
 * spec_world counts shards OF THIS SHARD'S SIZE: total docs / this shard's docs, rounded down.  A shard that holds more of the
 * index than it says guesses too high -- its guesses are caught by the check and cost re-runs, never a wrong answer
 * (nrtgpu_dist_search_bm25_batch_mode passes the communicator's size: docid-range shards are equal to within a segment boundary).
 * nrtgpu_note_shard_speculation tells the leaf set's verdict (nrtgpu_stats.spec_*) how a batch went, for callers that do the
 * check themselves; nrtgpu_dist_search_bm25_batch_mode does all of this itself. */
/* The share of a sharded index this context holds (round 6): shard_docs of index_docs live docs.  Where it is set, the guesses of
 * the shard-level speculation (nrtgpu_search_bm25_shard_device_begin with spec_world >= 2, nrtgpu_dist_search_bm25_batch[_mode])
 * count "the windows of all shards" as this shard's windows x index_docs / shard_docs instead of x spec_world: the reference's
 * virtual shards balance LIVE docs by greedy LPT over whole segments (MyIndexSearcher.java:117-160), which leaves shards of 40 %
 * and 20 % of an index side by side, and a shard that holds more than 1 / spec_world of the docs guesses too high (caught by the
 * check, but every catch is a second pass).  (0, 0): equal shards again.  Changes nothing but how often guesses fail. */
int  nrtgpu_set_shard_share(nrtgpu_ctx* ctx, int64_t shard_docs, int64_t index_docs);
int  nrtgpu_search_bm25_shard_device_begin(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                           const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t k_stride, void* d_keys,
                                           void* d_counts, void* d_hits, int32_t spec_world, void* d_guess, nrtgpu_pending** out);
int  nrtgpu_note_shard_speculation(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, int32_t n_segs, int32_t n_queries, int32_t n_failed);
/* Cross-GPU bound exchange (the LazyMaxScoreAccumulator idea across processes, SURVEY 8e "optional cross-GPU
 * theta sharing").  When one search is sharded over `world` GPUs, every shard alone would converge on the
 * k-th best of ITS docs.  With an exchange open, each shard publishes a score that at least
 * ceil(k / (world - 1)) of its docs reach; the entries of any world - 1 shards then cover k docs, so a shard
 * needs to collect nothing below the smallest entry of the OTHER shards.  Results of the merged search are unchanged
 * (tests/test_exchange_gpu.py); shards return fewer low-ranked hits.
 * The table lives in POSIX shared memory `shm_name` (every rank passes the same name; rank 0 should
 * unlink stale files first), mapped into each process's GPU.  Ranks must synchronise once between
 * nrtgpu_exchange_open and their first search.  Batches are matched across ranks by `epoch` (same
 * queries in the same order on every rank, epochs increasing; ranks may run up to 6 epochs apart). */
int  nrtgpu_exchange_open(nrtgpu_ctx* ctx, const char* shm_name, int32_t world, int32_t rank);
void nrtgpu_exchange_close(nrtgpu_ctx* ctx);
int  nrtgpu_search_bm25_batch_device_epoch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases,
                                           int32_t n_segs, const nrtgpu_bm25_query* queries, int32_t n_queries,
                                           int32_t k_stride, void* d_keys, void* d_counts, void* d_hits, int64_t epoch);

/* The multi-GPU search with the collective INSIDE the library (a JVM caller has no torch): one process per GPU, every
 * rank holds a docid-range shard of the index.  nrtgpu_dist_unique_id on one rank (128 bytes: an ncclUniqueId), the bytes
 * travel to the other ranks by whatever means the deployment has, nrtgpu_dist_init on every rank (ncclCommInitRank on the
 * context's device; RCCL is bound at run time with dlopen, NRTGPU_ERR_UNSUPPORTED if it is not installed).  Then every
 * rank calls nrtgpu_dist_search_bm25_batch with the same queries in the same order (index-global statistics in the
 * weights) over ITS leaves: local search -> ONE grouped RCCL all-gather of keys / counts / hit totals over xGMI ->
 * TopDocs.merge on every rank; every rank receives every answer.  total_hits sums the shards' counts; the relation is
 * GREATER_THAN_OR_EQUAL_TO iff some shard's is.
 * Failure of ONE rank's part of a one-call search (nrtgpu_dist_search_bm25_batch[_mode], nrtgpu_dist_knn_exact,
 * nrtgpu_dist_search_hybrid_batch): a rank whose shard search fails -- its thread's deadline, a planner refusal, a device
 * error -- still enters the exchange, with empty lists and its status word in the same grouped collective, so that no peer is
 * left waiting in a collective this rank would never issue; EVERY rank then returns an error (the failed rank its own, the
 * others NRTGPU_ERR_STATE naming the first failed rank), and the communicator stays usable.  The re-run of queries whose
 * speculative threshold failed ignores the thread's deadline (it is the tail of a search that was launched in time).  A caller
 * that pipelines with the two halves below vouches for its own ranks: nothing of this travels there. */
int  nrtgpu_dist_unique_id(void* out128);
int  nrtgpu_dist_init(nrtgpu_ctx* ctx, int32_t world, int32_t rank, const void* id128);
int  nrtgpu_dist_search_bm25_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                   const nrtgpu_bm25_query* queries, int32_t n_queries, nrtgpu_topdocs* out);
/* The exchange stage on its own, for callers that pipeline (host threads run nrtgpu_search_bm25_batch_device[_epoch] ahead,
 * one thread issues the exchanges in batch order -- the same order on every rank): this rank's device-resident shard
 * results (keys n_queries x k_stride, counts, hit totals) -> grouped all-gather -> TopDocs.merge into `out`. */
int  nrtgpu_dist_allgather_merge(nrtgpu_ctx* ctx, int32_t n_queries, int32_t k_stride, const void* d_keys, const void* d_counts,
                                 const void* d_hits, const int32_t* ks, const int32_t* total_hits_thresholds, nrtgpu_topdocs* out);
/* Who receives which answer.  ALLGATHER (BASELINE.json's north star, and what the two calls above do): every rank merges every
 * query and holds every answer.  ALLTOALL: rank r receives every rank's lists for ITS slice of the batch -- queries
 * [r * n / world, (r + 1) * n / world) -- and merges only those: 1 / world of the bytes on every xGMI link (point-to-point
 * links: what an all-to-all wants), of the merge and of the host-side unpacking; the rank that owns a query answers its caller.
 * `out` always has n_queries entries, indexed like the batch; entries of queries this rank does not own come back with
 * n_hits = 0, total_hits = -1.  A batch the ranks cannot share evenly (n_queries % world != 0), or an RCCL without
 * ncclSend / ncclRecv, is gathered whole: nrtgpu_dist_owned_range says what a call will deliver. */
#define NRTGPU_EXCHANGE_ALLGATHER 0
#define NRTGPU_EXCHANGE_ALLTOALL 1
#define NRTGPU_EXCHANGE_NO_SPECULATION 0x100   /* or-ed into nrtgpu_dist_search_bm25_batch_mode's mode: no shard-level guesses */
int  nrtgpu_dist_owned_range(nrtgpu_ctx* ctx, int32_t n_queries, int32_t mode, int32_t* first_query, int32_t* n_owned);
int  nrtgpu_dist_search_bm25_batch_mode(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                        const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t mode, nrtgpu_topdocs* out);
int  nrtgpu_dist_exchange_merge(nrtgpu_ctx* ctx, int32_t n_queries, int32_t k_stride, const void* d_keys, const void* d_counts,
                                const void* d_hits, const int32_t* ks, const int32_t* total_hits_thresholds, int32_t mode,
                                nrtgpu_topdocs* out);
/* The same with the shards' speculative thresholds (d_guess of nrtgpu_search_bm25_shard_device_begin; NULL: the call above): the
 * guesses travel in the same grouped collective as the lists, and after the merge the k-th key of every merged list is checked
 * against the largest guess any shard published for the query (the k-th key is read from the MERGED keys, never from `out`'s
 * arrays: those may be NULL or shorter than k).  failed[n_queries] / *n_failed: the queries whose answer does
 * NOT stand -- the same verdicts on every rank (all-to-all: the owners' verdicts are all-gathered, one byte per query) -- to be
 * run again by EVERY rank without speculation (nrtgpu_dist_search_bm25_batch_mode with NRTGPU_EXCHANGE_NO_SPECULATION over
 * those queries, in the same order on every rank). */
int  nrtgpu_dist_exchange_merge_checked(nrtgpu_ctx* ctx, int32_t n_queries, int32_t k_stride, const void* d_keys, const void* d_counts,
                                        const void* d_hits, const void* d_guess, const int32_t* ks, const int32_t* total_hits_thresholds,
                                        int32_t mode, nrtgpu_topdocs* out, uint8_t* failed, int32_t* n_failed);
/* Exact vector search over a row-partitioned field (BASELINE config 4: 10M x 768 over 1..8 GPUs): every rank scores the rows of
 * ITS leaves (nrtgpu_knn_exact on its shard, results kept in HBM), the per-rank top-k lists are exchanged and merged like the
 * BM25 ones -- NrtKnnFloatVectorQuery's per-leaf merge (src/main/java/com/yelp/nrtsearch/server/query/vector/
 * NrtKnnFloatVectorQuery.java:60-64; request glue: search/KnnUtils.java:47-66) taken across GPUs.  Every rank passes the same
 * queries; total_hits = the live vectors of all shards. */
int  nrtgpu_dist_knn_exact(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs, int32_t field_id,
                           int32_t sim, const float* queries, int32_t n_queries, int32_t dim, int32_t k, float boost, int32_t mode,
                           nrtgpu_topdocs* out /* n_queries */);
/* The hybrid (BASELINE config 5: BM25 recall + vector rescore over 1..8 GPUs) on docid-range shards.  Every rank passes the same
 * queries and query vectors and ITS leaves: BM25 recall on every shard -> ONE all-gather + merge on every rank (the GLOBAL first
 * pass: a doc of a shard's list that did not make the merged list must not be rescored) -> every rank rescores ITS docs of the
 * merged lists against its resident vectors -> the rescored windows are exchanged (`mode`) and merged.  Same answers as
 * nrtgpu_search_hybrid_batch over the whole index; total_hits / relation are the first pass's. */
int  nrtgpu_dist_search_hybrid_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                     const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t field_id, int32_t sim,
                                     const float* query_vectors, int32_t dim, float boost, double query_weight, double rescore_weight,
                                     int32_t window, int32_t mode, nrtgpu_topdocs* out /* n_queries, capacity >= window */);
void nrtgpu_dist_close(nrtgpu_ctx* ctx);

/* TopDocs.merge of n_lists per-GPU results laid out as the all-gather leaves them:
 * d_keys_in[list][query][k_stride], d_counts_in[list][query], d_hits_in[list][query] (device).
 * Writes host-side topdocs (docs/scores/total_hits/relation) for each query.  total_hits = the sum of the lists' counts;
 * the relation is GREATER_THAN_OR_EQUAL_TO iff some list's count carries the tag (a slice of that shard passed the
 * threshold, or the shard pruned) and numHits hits are returned.  total_hits_thresholds is kept for the signature. */
int  nrtgpu_merge_topk_device(nrtgpu_ctx* ctx, int32_t n_lists, int32_t n_queries, int32_t k_stride,
                              const void* d_keys_in, const void* d_counts_in, const void* d_hits_in,
                              const int32_t* ks, const int32_t* total_hits_thresholds, nrtgpu_topdocs* out);

/* ---------------------------------------------------------------------------------------------
 * Exact (brute-force) float vector search.  Replaces ExactFloatVectorQuery's scorer loop
 * (src/main/java/com/yelp/nrtsearch/server/query/vector/ExactVectorQuery.java:137-173: score =
 * VectorSimilarityFunction.compare(query, docVector) * boost for every doc that has a vector) plus
 * the top-k collector behind it; also the exact reading of the `knn` request path (recall 1.0;
 * compare against ExactFloatVectorQuery, not HNSW).
 *   sim: 0 cosine, 1 dot_product (unit vectors; also "normalized_cosine"), 2 l2_norm, 3 max_inner_product
 *        (mapping: src/main/java/com/yelp/nrtsearch/server/field/VectorFieldDef.java:77-88)
 *   queries: n_queries * dim fp32, row-major (already normalised by the caller where the field asks
 *        for it, VectorFieldDef.java:564-573)
 * Scores are fp32 sums in MFMA order: equal to Lucene within 1e-5 relative (Lucene's own order
 * depends on the JVM's vector width); docids ranked by (score desc, doc asc).  l2_norm: the squared distance is taken
 * as |q|^2 + |v|^2 - 2 q.v (one pass over the rows for every query of the batch), whose rounding error is
 * ~1e-7 * (|q|^2 + |v|^2): for near-duplicate vectors of large norm that exceeds 1e-5 of a tiny distance -- the
 * tolerance claim holds for the score 1 / (1 + d^2) (absolute 1e-4, tests/test_vectors_gpu.py), not for d^2 itself.
 * --------------------------------------------------------------------------------------------- */
int  nrtgpu_knn_exact(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                      int32_t field_id, int32_t sim, const float* queries, int32_t n_queries, int32_t dim, int32_t k,
                      float boost, nrtgpu_topdocs* out /* n_queries */);

/* What a request thread calls with ONE exact vector query (ExactFloatVectorQuery, query/vector/ExactVectorQuery.java:179-196):
 * blocks; concurrent callers over the same leaves, field, similarity and boost are merged into panels of up to 64 queries that
 * share one pass over the rows (a pass costs the same for 1 query as for 64).  Leader / follower, no extra thread: a caller that
 * finds the device free runs at once with whatever is waiting (a lone caller pays no batching latency); while a panel runs,
 * arrivals accumulate and leave together when it finishes, or as a second panel in flight once 64 are waiting.  A panel is
 * searched with the largest k of its members; each gets the first k of its own.  Results, errors and deadlines
 * (NRTGPU_ERR_TIMEOUT for a request that expired while waiting) as nrtgpu_knn_exact with n_queries = 1. */
int  nrtgpu_knn_exact_coalesced(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                int32_t field_id, int32_t similarity, const float* query, int32_t dim, int32_t k, float boost,
                                nrtgpu_topdocs* out);
/* TotalHits.relation of an exact vector query as the reference reports it.  The count is exact either way (the scorer ignores
 * min competitive scores: every live doc with a vector is collected), but the reference's collector still flips its relation to
 * GREATER_THAN_OR_EQUAL_TO once a slice has collected more than max(totalHitsThreshold, numHits) hits with a full queue
 * (LazyQueueTopScoreDocCollector.java:176-199, one collector per slice: MyIndexSearcher.java:163-208).  Host only: the live
 * vectors of every leaf are known from upload and liveDocs.  Returns 1 = GREATER_THAN_OR_EQUAL_TO, 0 = EQUAL_TO, < 0 = error. */
int  nrtgpu_knn_exact_relation(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                               int32_t field_id, int32_t k, int32_t total_hits_threshold);
/* The `knn` request path (KnnQuery -> NrtKnnFloatVectorQuery, src/main/java/com/yelp/nrtsearch/server/field/
 * VectorFieldDef.java:564-594, executed at search/KnnUtils.java:56) answered exactly: the k nearest docs among
 * those the pre-filter accepts (filter_mask: a resident mask, 0 = none; liveDocs always apply), optionally only
 * docs whose unboosted score is >= min_score (the MinThresholdQuery wrapped around the knn query at
 * VectorFieldDef.java:591-594; the caller converts similarityThreshold with similarityToScore, :664-673;
 * 0 = none), scores multiplied by boost afterwards.  total_hits = hits returned, as for the rewritten
 * knn query.  Recall is 1.0 where the reference's HNSW walk is approximate. */
int  nrtgpu_knn_search(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                       int32_t field_id, int32_t sim, const float* queries, int32_t n_queries, int32_t dim, int32_t k,
                       float boost, int32_t filter_mask, float min_score, nrtgpu_topdocs* out /* n_queries */);

/* Vector rescorer: RescoreOperation.rescore(hits, ctx) of a QueryRescore whose rescoreQuery is an exact
 * vector query (src/main/java/com/yelp/nrtsearch/server/rescore/QueryRescore.java:40-57): every
 * first-pass hit gets combined = (float)(queryWeight * first + rescoreWeight * vectorScore) (double
 * arithmetic; hits without a vector keep queryWeight * first), hits are re-sorted by (score desc,
 * doc asc) and trimmed to `window`.  docs are global docids; doc_bases maps them to segments. */
int  nrtgpu_rescore_vectors(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                            int32_t field_id, int32_t sim, const float* query, int32_t dim, float boost,
                            const int32_t* docs, const float* first_scores, int32_t n, double query_weight,
                            double rescore_weight, int32_t window, nrtgpu_topdocs* out);

/* Hybrid tail (config C5): BM25 recall, then the vector rescorer over each query's hits, then the window --
 * on one stream, the first-pass hits never leave HBM (no host round trip between
 * SearchHandler.java:1412-1413 and RescoreTask.java:47-50).  Results are those of nrtgpu_search_bm25_batch
 * followed per query by nrtgpu_rescore_vectors(query_vectors[q]); total_hits / relation are the first
 * pass's (QueryRescorer keeps them).  Weights and boost must be >= 0 (else NRTGPU_ERR_UNSUPPORTED). */
int  nrtgpu_search_hybrid_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t field_id, int32_t sim,
                                const float* query_vectors /* n_queries * dim */, int32_t dim, float boost,
                                double query_weight, double rescore_weight, int32_t window,
                                nrtgpu_topdocs* out /* n_queries, capacity >= window */);

/* ---------------------------------------------------------------------------------------------
 * Host-side restatements the Java shim would otherwise take from Lucene objects
 * (BM25Similarity.scorer(boost, collectionStats, termStats); SmallFloat; slices()).
 * --------------------------------------------------------------------------------------------- */
/* The planner's fixed-point analysis of one clause, exposed for the tests: returns 1 and *out_scale = E when
 * every score the clause can produce over docs whose norm byte is <= max_norm -- weight - weight / (1 + freq *
 * norm_cache256[norm]), freq >= 1 -- is a positive integer below 2^32 after multiplication by 2^E (the
 * accumulators then add integers: exact, ds_add_u64); 0 when the range does not fit (the batch runs in fp64);
 * negative on bad arguments.  Needs no device. */
int  nrtgpu_fixed_point_scale(float weight, const float* norm_cache256, int32_t max_norm, int32_t* out_scale);
/* The planner's rule for cutting a batch's queries into work items (one item runs on one CU), exposed for the
 * tests: query_costs[q] = postings + 48 per 1024-doc sub-tile of the query (0: matches nothing), target_items =
 * CUs; out_items[q] = number of items.  Needs no device. */
int  nrtgpu_plan_item_counts(int32_t n_queries, const int64_t* query_costs, int32_t target_items, int64_t* out_items);
/* Multi-retriever blend (O(hits) host work, no device): BlenderOperation.blend =
 * mergeHits + sortAndPaginate (src/main/java/com/yelp/nrtsearch/server/search/multiretriever/blender/BlenderOperation.java:76-132).
 *   mode 0: WeightedRrfBlenderOperation (…/operation/WeightedRrfBlenderOperation.java:53-78): score = sum over retrievers, in
 *           declaration order, of boost / (rank_constant + rank), rank 1-based, float (…/score/WeightedRRFScoreDoc.java:62,75)
 *   mode 1: score order: score = sum of boost * the retriever's score
 * docs[r] / scores[r]: retriever r's hits in rank order (global docids), counts[r] of them; boosts NULL => 1.  Returns hits
 * [start_hit, top_hits) of the blended order; total_hits = distinct docs, relation GREATER_THAN_OR_EQUAL_TO as the
 * reference reports it.  Docs with EQUAL blended scores come out in the reference's order too: its java.util.HashMap
 * iteration order feeding a bounded java.util.PriorityQueue is restated (host_math.h). */
int  nrtgpu_blend(int32_t n_retrievers, const int32_t* const* docs, const float* const* scores, const int32_t* counts,
                  const float* boosts, int32_t mode, int32_t rank_constant, int32_t start_hit, int32_t top_hits, nrtgpu_topdocs* out);
int32_t nrtgpu_int_to_byte4(int32_t length);
int32_t nrtgpu_byte4_to_int(int32_t norm_byte);
float   nrtgpu_bm25_idf(int64_t doc_count, int64_t doc_freq);
float   nrtgpu_bm25_avgdl(int64_t sum_total_term_freq, int64_t doc_count);
void    nrtgpu_bm25_norm_cache(float avgdl, float k1, float b, float* out256);
/* MyIndexSearcher.slices / slicesForShards (src/main/java/com/yelp/nrtsearch/server/search/
 * MyIndexSearcher.java:79-208).  leaf i has max_docs[i] / num_docs[i] (live) docs and docBase
 * doc_bases[i].  Writes slice_of_leaf[i] = slice index (slices ordered as the reference orders
 * them) and returns the number of slices; with virtual_shards > 1 also writes shard_of_leaf[i]
 * (may be NULL). */
int32_t nrtgpu_slices(int32_t n_leaves, const int32_t* max_docs, const int32_t* num_docs, const int32_t* doc_bases,
                      int32_t virtual_shards, int32_t slice_max_docs, int32_t slice_max_segments,
                      int32_t* slice_of_leaf, int32_t* shard_of_leaf);

/* ---------------------------------------------------------------------------------------------
 * Deadlines and per-call diagnostics.
 * The reference checks the request's deadline between the phases of a search (src/main/java/com/yelp/nrtsearch/server/
 * handler/SearchHandler.java:158,194,263,277 -> grpc/DeadlineUtils.java:48-58) and bounds collection with timeoutSec
 * (search/SearchCutoffWrapper.java:149-159).  Here: the request thread states its deadline once -- absolute CLOCK_MONOTONIC
 * nanoseconds, 0 = none (the default) -- and every search / vector call it makes afterwards checks it on entry, again after
 * planning (i.e. after waiting for a workspace and for the device) and, in the vector search, between the passes over the
 * rows: past the deadline the call returns NRTGPU_ERR_TIMEOUT without launching (more) work.  A request waiting in
 * nrtgpu_search_bm25_coalesced carries its own thread's deadline: the batch leaves without it once it has expired.  Kernels
 * that are already launched run to completion (a batch is a few milliseconds); their results are returned as valid.
 * nrtgpu_last_diagnostics: what the calling thread's last completed search call cost -- the numbers behind
 * SearchResponse.Diagnostics.firstPassSearchTimeMs (SearchHandler.java:261) and a profile's per-phase times.
 * --------------------------------------------------------------------------------------------- */
void nrtgpu_set_thread_deadline_ns(int64_t deadline_ns);
int64_t nrtgpu_monotonic_ns(void);   /* the clock deadlines are on */
typedef struct {
  double  total_ms;        /* entry to return of the call (a coalesced request: of the batch it travelled in) */
  double  plan_ms;         /* host: queries -> launch plan */
  double  queue_ms;        /* waiting for a workspace and for the device (other batches' kernels) */
  double  device_ms;       /* scorer + merge kernels, HIP events; 0 unless nrtgpu_config.collect_timing */
  int64_t postings;        /* postings of the call's query terms (what an exhaustive scan streams) */
  int32_t queries;         /* queries of the batch */
  int32_t items_maxscore;  /* work items on the dynamic-pruning route */
  int32_t items_scan;      /* work items on the exhaustive route */
  int32_t reserved;
} nrtgpu_diagnostics;
int  nrtgpu_last_diagnostics(nrtgpu_diagnostics* out);

/* ---------------------------------------------------------------------------------------------
 * Diagnostics (maps onto SearchResponse.Diagnostics / profile fields).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t batches;            /* batch calls completed */
  int64_t queries;
  int64_t scan_launches;      /* postings-scan kernel launches */
  double  scan_ms;            /* sum of HIP-event durations of those launches (collect_timing) */
  int64_t scan_postings;      /* sum over launches of postings in the scanned term ranges */
  int64_t scan_items;
  double  merge_ms;
  double  host_plan_ms;       /* host time spent building launch plans */
  int64_t fixed_point_launches; /* scan launches that accumulated in exact fixed point (the others: fp64) */
  int64_t maxscore_launches;  /* launches of the MaxScore (dynamic pruning) kernel */
  double  maxscore_ms;        /* sum of their HIP-event durations (collect_timing) */
  int64_t maxscore_postings;  /* postings of the term ranges of the queries on that route (what an exhaustive scan streams) */
  int64_t maxscore_items;
  int64_t knn_panels;         /* exact vector searches: query panels (<= 32 queries) scored against every row */
  int64_t knn_score_launches; /* knn_score_kernel launches (a panel takes a few rounds, theta tightens in between) */
  double  knn_score_ms;       /* sum of their HIP-event durations (collect_timing) */
  int64_t knn_rows;           /* sum over panels of the rows scored */
  int64_t knn_second_passes;  /* panels that needed a second pass over the rows: a query's nominations did not certify its answer */
  int64_t knn_sketch_launches; /* of knn_score_launches: passes that nominated from the fp16 sketch (half the bytes per row) */
  int64_t spec_queries;       /* speculative thresholds (nrtgpu_set_speculation), since that call: queries run under them ... */
  int64_t spec_reruns;        /* ... queries whose guess failed the merge's check and were run again inside their call ... */
  int64_t spec_disabled;      /* ... 1 once the library has switched speculation off for one of the context's leaf sets: too many failed
                               * there even in the scattered order */
  int64_t spec_scattered;     /* ... 1 once a leaf set was moved to the scattered window order (its guesses failed in docid order: docids
                               * that are no sample of the index -- an index sort, time-ordered vocabulary) */
} nrtgpu_stats;
int  nrtgpu_get_stats(nrtgpu_ctx* ctx, nrtgpu_stats* out);
void nrtgpu_reset_stats(nrtgpu_ctx* ctx);
#ifdef __cplusplus
}
#endif
#endif /* NRTGPU_H */
