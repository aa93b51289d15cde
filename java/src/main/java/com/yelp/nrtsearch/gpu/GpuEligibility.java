package com.yelp.nrtsearch.gpu;

import static java.lang.foreign.ValueLayout.*;

import com.yelp.nrtsearch.server.search.SearchCollectorManager;
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
 * TermQuery, a BooleanQuery of SHOULD (boosted) TermQuery clauses (built at query/QueryNodeMapper.java:257-283,360-395) or a
 * DisjunctionMaxQuery over such term queries with tie breaker 0 (QueryNodeMapper.java:350-358),
 * collected by a plain RelevanceCollector (search/collectors/RelevanceCollector.java:42-69) -- and marshal them into a
 * nrtgpu_bm25_query.  Whatever remains (clause counts, fields, fixed-point range, resident masks ...) is decided by the
 * library itself (nrtgpu_query_supported / NRTGPU_ERR_UNSUPPORTED), so this class carries no limits of its own.
 * NOT COMPILED here (no JDK).
 */
final class GpuEligibility {
  record Clause(Term term, float boost) {}

  record Plan(MemorySegment query, MemorySegment out, MemorySegment docs, MemorySegment scores, int k) {
    TopDocs toTopDocs() {
      int n = out.get(JAVA_INT, 0);
      ScoreDoc[] hits = new ScoreDoc[n];
      for (int i = 0; i < n; i++) hits[i] = new ScoreDoc(docs.getAtIndex(JAVA_INT, i), scores.getAtIndex(JAVA_FLOAT, i));
      TotalHits.Relation rel = out.get(JAVA_INT, 32) != 0 ? TotalHits.Relation.GREATER_THAN_OR_EQUAL_TO : TotalHits.Relation.EQUAL_TO;
      return new TopDocs(new TotalHits(out.get(JAVA_LONG, 24), rel), hits);
    }
  }

  /** Flattens the rewritten query; null = not a shape the device takes.  shape[0] = minimumNumberShouldMatch,
   *  shape[1] = 1 for a DisjunctionMaxQuery (best clause instead of the sum). */
  static List<Clause> clauses(Query q, int[] shape) {
    int[] minShouldMatch = shape;
    List<Clause> out = new ArrayList<>();
    if (q instanceof DisjunctionMaxQuery dm) {                         // QueryNodeMapper.java:350-358
      if (dm.getTieBreakerMultiplier() != 0f) return null;            // the device keeps the best clause only
      for (Query d : dm.getDisjuncts()) {
        Clause cl = term(d, 1f);
        if (cl == null) return null;                                  // disjuncts that are not (boosted) term queries
        out.add(cl);
      }
      shape[1] = 1;
      return out.isEmpty() ? null : out;
    }
    if (q instanceof BooleanQuery bq) {
      for (BooleanClause c : bq.clauses()) {
        if (c.occur() != BooleanClause.Occur.SHOULD) return null;     // FILTER / MUST_NOT as masks: GpuMaskCache (not in this sketch)
        Clause cl = term(c.query(), 1f);
        if (cl == null) return null;
        out.add(cl);
      }
      minShouldMatch[0] = bq.getMinimumNumberShouldMatch();
      return out.isEmpty() ? null : out;
    }
    Clause cl = term(q, 1f);
    if (cl == null) return null;
    out.add(cl);
    return out;
  }

  private static Clause term(Query q, float boost) {
    if (q instanceof BoostQuery b) return term(b.getQuery(), boost * b.getBoost());   // QueryNodeMapper.java:131-133
    if (q instanceof TermQuery t) return new Clause(t.getTerm(), boost);
    return null;
  }

  /** The unwrapped doc collector must be a plain RelevanceCollector: no sort, no additional collectors, no terminateAfter. */
  static RelevanceCollector relevance(CollectorManager<?, ?> manager) {
    Object m = manager;
    while (m instanceof com.yelp.nrtsearch.server.search.collectors.additional.WrappedCollectorManager<?, ?> w) m = w.getWrapped();
    if (!(m instanceof SearchCollectorManager scm)) return null;
    DocCollector dc = scm.getDocCollector();
    if (!(dc instanceof RelevanceCollector rc) || !scm.getAdditionalCollectors().isEmpty()) return null;
    return rc;
  }

  static Plan marshal(Arena a, IndexSearcher searcher, GpuSegmentStore store, List<Clause> clauses, int msm, int disjunctionMax,
      int k, int totalHitsThreshold, ScoreDoc after) throws IOException {
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
      t.set(JAVA_INT, 0, store.fieldId(c.term().field()));
      t.set(JAVA_INT, 4, fields.indexOf(c.term().field()));
      t.set(JAVA_LONG, 8, GpuSegmentStore.termHash(c.term().bytes()));
      t.set(JAVA_FLOAT, 16, c.boost() * idf);
    }
    MemorySegment cache = a.allocate(JAVA_FLOAT, fields.size() * 256L);
    for (int f = 0; f < fields.size(); f++) {
      CollectionStatistics cs = searcher.collectionStatistics(fields.get(f));
      float avgdl = cs == null ? 1f : (float) (cs.sumTotalTermFreq() / (double) cs.docCount());
      for (int i = 0; i < 256; i++)                                   // BM25Similarity.scorer: cache[i] = 1 / (k1 * ((1 - b) + b * LENGTH_TABLE[i] / avgdl))
        cache.setAtIndex(JAVA_FLOAT, f * 256L + i, 1f / (1.2f * ((1f - 0.75f) + 0.75f * org.apache.lucene.util.SmallFloat.byte4ToInt((byte) i) / avgdl)));
    }
    MemorySegment q = a.allocate(NrtGpu.QUERY);
    q.set(JAVA_INT, 0, clauses.size());
    q.set(ADDRESS, 8, terms);
    q.set(JAVA_INT, 16, fields.size());
    q.set(ADDRESS, 24, cache);
    q.set(JAVA_INT, 32, k);
    q.set(JAVA_INT, 36, totalHitsThreshold);
    q.set(JAVA_INT, 40, after != null ? 1 : 0);
    q.set(JAVA_INT, 44, after != null ? after.doc : 0);
    q.set(JAVA_FLOAT, 48, after != null ? after.score : 0f);
    q.set(JAVA_INT, 52, msm);
    q.set(JAVA_INT, 68, disjunctionMax);
    MemorySegment docs = a.allocate(JAVA_INT, k), scores = a.allocate(JAVA_FLOAT, k), out = a.allocate(NrtGpu.TOPDOCS);
    out.set(JAVA_INT, 4, k);
    out.set(ADDRESS, 8, docs);
    out.set(ADDRESS, 16, scores);
    return new Plan(q, out, docs, scores, k);
  }
}
