package com.yelp.nrtsearch.gpu;

import static java.lang.foreign.ValueLayout.*;

import com.yelp.nrtsearch.server.search.MyIndexSearcher;
import com.yelp.nrtsearch.server.search.SearcherResult;
import com.yelp.nrtsearch.server.search.collectors.RelevanceCollector;
import java.io.IOException;
import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.util.List;
import java.util.Map;
import java.util.concurrent.Executor;
import org.apache.lucene.index.IndexReader;
import org.apache.lucene.index.LeafReaderContext;
import org.apache.lucene.search.*;

/**
 * Seam B2 (SURVEY 8b): the searcher nrtsearch installs in ShardSearcherFactory.newSearcher
 * (index/ShardState.java:506-527, :792-803).  search(Query, CollectorManager) -- the single entry of the hot path
 * (handler/SearchHandler.java:1412-1413, :556) -- goes to the device when the rewritten query and the collector are
 * eligible and every leaf is resident; anything else is super.search, i.e. the untouched Lucene path.
 * NOT COMPILED here (no JDK).
 */
public class GpuIndexSearcher extends MyIndexSearcher {
  private final MemorySegment ctx;
  private final GpuSegmentStore store;

  public GpuIndexSearcher(IndexReader reader, Executor executor, SlicingParams slicing, MemorySegment ctx, GpuSegmentStore store) {
    super(reader, executor, slicing);
    this.ctx = ctx;
    this.store = store;
  }

  @Override
  @SuppressWarnings("unchecked")
  public <C extends Collector, T> T search(Query query, CollectorManager<C, T> manager) throws IOException {
    RelevanceCollector rc = GpuEligibility.relevance(manager);
    if (rc == null) return super.search(query, manager);
    int[] msm = {0, 0};   // minimumNumberShouldMatch, DisjunctionMaxQuery?
    List<GpuEligibility.Clause> clauses = GpuEligibility.clauses(rewrite(query), msm);
    if (clauses == null) return super.search(query, manager);
    List<LeafReaderContext> leaves = getIndexReader().leaves();
    try (Arena a = Arena.ofConfined()) {
      MemorySegment segs = a.allocate(ADDRESS, leaves.size()), bases = a.allocate(JAVA_INT, leaves.size());
      for (int i = 0; i < leaves.size(); i++) {
        MemorySegment s = store.segmentOf(leaves.get(i));
        if (s == null) return super.search(query, manager);            // a leaf is not resident (yet): CPU path
        segs.setAtIndex(ADDRESS, i, s);
        bases.setAtIndex(JAVA_INT, i, leaves.get(i).docBase);
      }
      GpuEligibility.Plan plan = GpuEligibility.marshal(a, this, store, clauses, msm[0], msm[1], rc.getNumHitsToCollect(),
          rc.getTotalHitsThreshold(), rc.getSearchAfter());
      if (plan == null) return super.search(query, manager);
      int status;
      if (msm[0] > 1 || msm[1] != 0)   // clause counts / best-clause scores depend on the whole batch's accumulator mode: a batch of one
        status = (int) NrtGpu.SEARCH_BATCH.invokeExact(ctx, segs, bases, leaves.size(), plan.query(), 1, plan.out());
      else
        status = (int) NrtGpu.SEARCH1.invokeExact(ctx, segs, bases, leaves.size(), plan.query(), plan.out());   // blocks; batched inside
      if (status == NrtGpu.ERR_UNSUPPORTED) return super.search(query, manager);
      NrtGpu.check(status);
      return (T) new SearcherResult(plan.toTopDocs(), Map.of());       // search/SearcherResult.java:31-34
    } catch (IOException | RuntimeException e) {
      throw e;
    } catch (Throwable t) {
      throw new IOException(t);
    }
  }
}
