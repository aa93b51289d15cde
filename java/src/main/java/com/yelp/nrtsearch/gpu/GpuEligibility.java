package com.yelp.nrtsearch.gpu;

import static java.lang.foreign.ValueLayout.*;

import com.yelp.nrtsearch.server.query.vector.ExactVectorQuery;
import com.yelp.nrtsearch.server.search.SearchCollectorManager;
import com.yelp.nrtsearch.server.search.SearchCutoffWrapper;
import com.yelp.nrtsearch.server.search.SearchStatsWrapper;
import com.yelp.nrtsearch.server.search.collectors.DocCollector;
import com.yelp.nrtsearch.server.search.collectors.RelevanceCollector;
import java.io.IOException;
import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.util.ArrayList;
import java.util.List;
import org.apache.lucene.index.Term;
import org.apache.lucene.index.TermStates;
import org.apache.lucene.search.*;
import org.apache.lucene.search.similarities.BM25Similarity;
import org.apache.lucene.search.similarities.Similarity;

/**
 * The eligibility predicate of SURVEY 8b, Java half: recognise the shapes the native planner takes -- a (boosted)
 * TermQuery, a BooleanQuery of SHOULD and / or MUST (boosted) TermQuery clauses with FILTER and MUST_NOT clauses next to them
 * (built at query/QueryNodeMapper.java:257-283,360-395; the non-scoring clauses become resident doc-set masks: GpuMaskCache) or a
 * DisjunctionMaxQuery over such term queries with any tie breaker (QueryNodeMapper.java:350-358),
 * collected by a plain RelevanceCollector (search/collectors/RelevanceCollector.java:42-69) -- and marshal them into a
 * nrtgpu_bm25_query.  Whatever remains (clause counts, fields, fixed-point range, resident masks ...) is decided by the
 * library itself (nrtgpu_query_supported / NRTGPU_ERR_UNSUPPORTED), so this class carries no limits of its own.
 * NOT COMPILED here (no JDK).
 */
final class GpuEligibility {
  /** occur: 0 = SHOULD, 1 = MUST (nrtgpu_term.occur) */
  record Clause(Term term, float boost, int occur) {}

  /** The flattened query: scoring clauses, minimumNumberShouldMatch, DisjunctionMaxQuery?, and the non-scoring clauses -- any
   *  number of FILTER and MUST_NOT clauses, each a resident mask of its own (cached per clause like LRUQueryCache caches its
   *  DocIdSet); the library ANDs / AND-NOTs them at plan time (nrtgpu_bm25_query.more_filters / more_must_not). */
  record Shape(List<Clause> clauses, int minShouldMatch, int disjunctionMax, float tieBreaker, List<Query> filters, List<Query> mustNots) {}

  /** The collector behind the manager handed to search(), the request's timeoutSec (0 = none) and the wrapper that enforces it
   *  on the reference's path (null = none): a timeout on the device route is reported THROUGH it (GpuIndexSearcher.timedOut). */
  record Eligible(RelevanceCollector collector, double timeoutSec, SearchCutoffWrapper<?> cutoff) {}

  record Plan(MemorySegment query, MemorySegment out, MemorySegment docs, MemorySegment scores, int k) {
    TopDocs toTopDocs() {
      int n = out.get(JAVA_INT, NrtGpuLayouts.TOPDOCS_N_HITS);
      ScoreDoc[] hits = new ScoreDoc[n];
      for (int i = 0; i < n; i++) hits[i] = new ScoreDoc(docs.getAtIndex(JAVA_INT, i), scores.getAtIndex(JAVA_FLOAT, i));
      TotalHits.Relation rel = out.get(JAVA_INT, NrtGpuLayouts.TOPDOCS_TOTAL_HITS_IS_LOWER_BOUND) != 0 ? TotalHits.Relation.GREATER_THAN_OR_EQUAL_TO : TotalHits.Relation.EQUAL_TO;
      return new TopDocs(new TotalHits(out.get(JAVA_LONG, NrtGpuLayouts.TOPDOCS_TOTAL_HITS), rel), hits);
    }
  }

  /** A (boosted) ExactFloatVectorQuery (query/vector/ExactVectorQuery.java:179-196): field, query vector, boost. */
  record VectorShape(String field, float[] vector, float boost) {}

  /** null = not an exact float vector query. */
  static VectorShape vectorShape(Query q) {
    float boost = 1f;
    while (q instanceof BoostQuery b) {
      boost *= b.getBoost();
      q = b.getQuery();
    }
    if (q instanceof ExactVectorQuery.ExactFloatVectorQuery evq) return new VectorShape(evq.getField(), evq.getQueryVector(), boost);
    return null;
  }

  /** Flattens the rewritten query; null = not a shape the device takes. */
  static Shape shape(Query q) {
    List<Clause> out = new ArrayList<>();
    if (q instanceof DisjunctionMaxQuery dm) {                         // QueryNodeMapper.java:350-358
      float tb = dm.getTieBreakerMultiplier();                        // (float)(best + tieBreaker x the others): the kernel's second accumulator
      if (!(tb >= 0f && tb <= 1f)) return null;
      for (Query d : dm.getDisjuncts()) {
        Clause cl = term(d, 1f, 0);
        if (cl == null) return null;                                  // disjuncts that are not (boosted) term queries
        out.add(cl);
      }
      return out.isEmpty() ? null : new Shape(out, 0, 1, tb, List.of(), List.of());
    }
    if (q instanceof BooleanQuery bq) {
      List<Query> filters = new ArrayList<>(), mustNots = new ArrayList<>();
      int must = 0;
      for (BooleanClause c : bq.clauses()) {
        switch (c.occur()) {
          case SHOULD, MUST -> {                                       // MUST term clauses: every one required (MatchQuery operator MUST)
            Clause cl = term(c.query(), 1f, c.occur() == BooleanClause.Occur.MUST ? 1 : 0);
            if (cl == null) return null;
            out.add(cl);
            if (c.occur() == BooleanClause.Occur.MUST) must++;
          }
          case FILTER -> filters.add(c.query());                       // one resident mask per clause
          case MUST_NOT -> mustNots.add(c.query());
        }
      }
      if (out.isEmpty()) return null;                                 // filter-only queries score 0 for every match: Lucene's business
      if (filters.size() > NrtGpu.MAX_MASKS || mustNots.size() > NrtGpu.MAX_MASKS) return null;
      // mixed MUST / SHOULD: ReqOptSumScorer adds (float) required + (float) optional -- two separately rounded sums: the clauses
      // carry their occur and the kernel a second accumulator; with minimumNumberShouldMatch > 0 Lucene scores the SHOULD part as
      // one more required scorer of a conjunction (another sum structure): the caller's path
      boolean mixed = must != 0 && must != out.size();
      if (mixed && bq.getMinimumNumberShouldMatch() > 0) return null;
      int msm = mixed ? 0 : (must != 0 ? out.size() : bq.getMinimumNumberShouldMatch());
      if (!mixed) out.replaceAll(cl -> new Clause(cl.term(), cl.boost(), 0));   // all MUST == minimumNumberShouldMatch = n over SHOULD clauses
      if (!filters.isEmpty() && must == 0 && msm == 0) return null;   // SHOULD clauses next to a FILTER are optional: filter-only docs match with score 0
      return new Shape(out, msm, 0, 0f, filters, mustNots);
    }
    Clause cl = term(q, 1f, 0);
    if (cl == null) return null;
    out.add(cl);
    return new Shape(out, 0, 0, 0f, List.of(), List.of());
  }

  private static Clause term(Query q, float boost, int occur) {
    if (q instanceof BoostQuery b) return term(b.getQuery(), boost * b.getBoost(), occur);   // QueryNodeMapper.java:131-133
    if (q instanceof TermQuery t) return new Clause(t.getTerm(), boost, occur);
    return null;
  }

  /**
   * What SearchHandler hands to search() is DocCollector.getWrappedManager() (search/collectors/DocCollector.java:120-125,
   * :197-220): a SearchCollectorManager, possibly inside a SearchStatsWrapper (profile), a SearchCutoffWrapper (timeoutSec) and
   * a TerminateAfterWrapper.  The unwrapped doc collector must be a plain RelevanceCollector: no sort, no additional collectors
   * (facets ...), no terminateAfter (its early termination counts collected docs: the reference's business).  getWrapped() is
   * the reference's own (SearchCutoffWrapper.java:132, SearchStatsWrapper.java:90); getAdditionalCollectors() and
   * SearchCutoffWrapper.timedOutBeforeCollection() come with java/patches/nrtsearch-gpu-hook.diff.
   */
  static Eligible relevance(CollectorManager<?, ?> manager) {
    Object m = manager;
    double timeoutSec = 0.0;
    SearchCutoffWrapper<?> cutoffWrapper = null;
    for (;;) {
      if (m instanceof SearchStatsWrapper<?> stats) {
        m = stats.getWrapped();
      } else if (m instanceof SearchCutoffWrapper<?> cutoff) {
        timeoutSec = cutoff.getTimeoutSec();
        cutoffWrapper = cutoff;
        m = cutoff.getWrapped();
      } else {
        break;
      }
    }
    if (!(m instanceof SearchCollectorManager scm)) return null;
    DocCollector dc = scm.getDocCollector();
    if (!(dc instanceof RelevanceCollector rc) || !scm.getAdditionalCollectors().isEmpty()) return null;
    return new Eligible(rc, timeoutSec, cutoffWrapper);
  }

  static Plan marshal(Arena a, IndexSearcher searcher, GpuSegmentStore store, Shape shape, int[] filterMasks, int[] mustNotMasks,
      int k, int totalHitsThreshold, ScoreDoc after) throws IOException {
    List<Clause> clauses = shape.clauses();
    int msm = shape.minShouldMatch(), disjunctionMax = shape.disjunctionMax();
    Similarity sim = searcher.getSimilarity();
    if (!(sim instanceof BM25Similarity)) return null;                // IndexSimilarity.java:51-63: default similarity only
    List<String> fields = new ArrayList<>();
    MemorySegment terms = a.allocate(NrtGpu.TERM, clauses.size());
    for (int i = 0; i < clauses.size(); i++) {
      Clause c = clauses.get(i);
      if (!fields.contains(c.term().field())) fields.add(c.term().field());
      TermStates ts = TermStates.build(searcher, c.term(), true);      // index-global statistics, BlendedTermQuery.java:86-96
      CollectionStatistics cs = searcher.collectionStatistics(c.term().field());
      float idf = ts.docFreq() == 0 || cs == null ? 0f
          : (float) Math.log(1 + (cs.docCount() - ts.docFreq() + 0.5D) / (ts.docFreq() + 0.5D));   // BM25Similarity.idf
      MemorySegment t = terms.asSlice(i * NrtGpu.TERM.byteSize(), NrtGpu.TERM.byteSize());
      t.set(JAVA_INT, NrtGpuLayouts.TERM_FIELD_ID, store.fieldId(c.term().field()));
      t.set(JAVA_INT, NrtGpuLayouts.TERM_CACHE_SLOT, fields.indexOf(c.term().field()));
      t.set(JAVA_LONG, NrtGpuLayouts.TERM_TERM_HASH, GpuSegmentStore.termHash(c.term().bytes()));
      t.set(JAVA_FLOAT, NrtGpuLayouts.TERM_WEIGHT, c.boost() * idf);
      t.set(JAVA_INT, NrtGpuLayouts.TERM_OCCUR, c.occur());
    }
    MemorySegment cache = a.allocate(JAVA_FLOAT, fields.size() * 256L);
    for (int f = 0; f < fields.size(); f++) {
      CollectionStatistics cs = searcher.collectionStatistics(fields.get(f));
      float avgdl = cs == null ? 1f : (float) (cs.sumTotalTermFreq() / (double) cs.docCount());
      for (int i = 0; i < 256; i++)                                   // BM25Similarity.scorer: cache[i] = 1 / (k1 * ((1 - b) + b * LENGTH_TABLE[i] / avgdl))
        cache.setAtIndex(JAVA_FLOAT, f * 256L + i, 1f / (1.2f * ((1f - 0.75f) + 0.75f * org.apache.lucene.util.SmallFloat.byte4ToInt((byte) i) / avgdl)));
    }
    MemorySegment q = a.allocate(NrtGpu.QUERY);
    q.set(JAVA_INT, NrtGpuLayouts.QUERY_N_TERMS, clauses.size());
    q.set(ADDRESS, NrtGpuLayouts.QUERY_TERMS, terms);
    q.set(JAVA_INT, NrtGpuLayouts.QUERY_N_CACHES, fields.size());
    q.set(ADDRESS, NrtGpuLayouts.QUERY_NORM_CACHE, cache);
    q.set(JAVA_INT, NrtGpuLayouts.QUERY_K, k);
    q.set(JAVA_INT, NrtGpuLayouts.QUERY_TOTAL_HITS_THRESHOLD, totalHitsThreshold);
    q.set(JAVA_INT, NrtGpuLayouts.QUERY_HAS_AFTER, after != null ? 1 : 0);
    q.set(JAVA_INT, NrtGpuLayouts.QUERY_AFTER_DOC, after != null ? after.doc : 0);
    q.set(JAVA_FLOAT, NrtGpuLayouts.QUERY_AFTER_SCORE, after != null ? after.score : 0f);
    q.set(JAVA_INT, NrtGpuLayouts.QUERY_MIN_SHOULD_MATCH, msm);
    q.set(JAVA_INT, NrtGpuLayouts.QUERY_FILTER_MASK, filterMasks.length > 0 ? filterMasks[0] : 0);
    q.set(JAVA_INT, NrtGpuLayouts.QUERY_MUST_NOT_MASK, mustNotMasks.length > 0 ? mustNotMasks[0] : 0);
    if (filterMasks.length > 1) {        // further FILTER clauses: the library ANDs the masks at plan time
      MemorySegment more = a.allocate(JAVA_INT, filterMasks.length - 1);
      for (int i = 1; i < filterMasks.length; i++) more.setAtIndex(JAVA_INT, i - 1, filterMasks[i]);
      q.set(JAVA_INT, NrtGpuLayouts.QUERY_N_MORE_FILTERS, filterMasks.length - 1);
      q.set(ADDRESS, NrtGpuLayouts.QUERY_MORE_FILTERS, more);
    }
    if (mustNotMasks.length > 1) {
      MemorySegment more = a.allocate(JAVA_INT, mustNotMasks.length - 1);
      for (int i = 1; i < mustNotMasks.length; i++) more.setAtIndex(JAVA_INT, i - 1, mustNotMasks[i]);
      q.set(JAVA_INT, NrtGpuLayouts.QUERY_N_MORE_MUST_NOT, mustNotMasks.length - 1);
      q.set(ADDRESS, NrtGpuLayouts.QUERY_MORE_MUST_NOT, more);
    }
    q.set(JAVA_INT, NrtGpuLayouts.QUERY_DISJUNCTION_MAX, disjunctionMax);
    q.set(JAVA_FLOAT, NrtGpuLayouts.QUERY_TIE_BREAKER, shape.tieBreaker());
    MemorySegment docs = a.allocate(JAVA_INT, k), scores = a.allocate(JAVA_FLOAT, k), out = a.allocate(NrtGpu.TOPDOCS);
    out.set(JAVA_INT, NrtGpuLayouts.TOPDOCS_CAPACITY, k);
    out.set(ADDRESS, NrtGpuLayouts.TOPDOCS_DOCS, docs);
    out.set(ADDRESS, NrtGpuLayouts.TOPDOCS_SCORES, scores);
    return new Plan(q, out, docs, scores, k);
  }
}
