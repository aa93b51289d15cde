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
 * Seam B2 (SURVEY 8b): the searcher ShardSearcherFactory.newSearcher builds when the plugin has installed its
 * MyIndexSearcher.SearcherHook (java/patches/nrtsearch-gpu-hook.diff; index/ShardState.java:506-527).
 * search(Query, CollectorManager) -- the single entry of the hot path (handler/SearchHandler.java:1412-1413, :556) -- goes
 * to the device when the rewritten query and the collector are eligible.  PARTIAL RESIDENCY (SURVEY 8b; under NRT refresh a searcher
 * usually holds a young segment or two that are not resident yet, index/ShardState.java:506-527): the searcher's slices whose
 * leaves are all resident are searched on the device in one call, the other slices by Lucene's own collectors, and the parts are
 * reduced like the reference reduces its slices (TopDocs.merge, totalHits summed, GREATER_THAN_OR_EQUAL_TO if any part's is:
 * LazyQueueTopScoreDocCollectorManager.java:137-144) -- tests/test_partial_residency_gpu.py runs exactly this split through the C
 * ABI with the oracle in Lucene's place.  Anything else is super.search, i.e. the untouched Lucene path.
 * NOT COMPILED in this repository's image (no JDK); tests/test_java_shim_signatures.py checks every nrtsearch type, constructor
 * and method used here against the reference sources plus the patch.
 */
public class GpuIndexSearcher extends MyIndexSearcher {
  private final MemorySegment ctx;
  private final GpuSegmentStore store;
  private final GpuMaskCache masks;
  private static final ThreadLocal<double[]> LAST_DIAGNOSTICS = new ThreadLocal<>();

  /** MyIndexSearcher reads its slicing params from a static under a lock: construction goes through its factory. */
  public static GpuIndexSearcher create(IndexReader reader, Executor executor, MyIndexSearcher.SlicingParams slicing, MemorySegment ctx,
      GpuSegmentStore store, GpuMaskCache masks) {
    return MyIndexSearcher.create(reader, executor, slicing, (r, e) -> new GpuIndexSearcher(r, e, ctx, store, masks));
  }

  protected GpuIndexSearcher(IndexReader reader, Executor executor, MemorySegment ctx, GpuSegmentStore store, GpuMaskCache masks) {
    super(reader, executor);
    this.ctx = ctx;
    this.store = store;
    this.masks = masks;
  }

  /**
   * ExactFloatVectorQuery (query/vector/ExactVectorQuery.java:137-173: every doc with a vector is scored by
   * VectorSimilarityFunction.compare(query, vector) * boost) answered by nrtgpu_knn_exact_coalesced: the k best docs with the scalar
   * left-to-right fp32 similarity of each (DESIGN 4.3).  null = the caller's path (a leaf not resident, a field without
   * float vectors, a dimension the device does not take).
   */
  private TopDocs exactVectorSearch(GpuEligibility.VectorShape vs, int k, int totalHitsThreshold, double timeoutSec) throws IOException {
    List<LeafReaderContext> leaves = getIndexReader().leaves();
    int sim = -1;
    for (LeafReaderContext leaf : leaves) {
      org.apache.lucene.index.FieldInfo fi = leaf.reader().getFieldInfos().fieldInfo(vs.field());
      if (fi == null || fi.getVectorDimension() == 0) continue;         // a leaf without the field: nothing to score there
      if (fi.getVectorEncoding() != org.apache.lucene.index.VectorEncoding.FLOAT32 || fi.getVectorDimension() != vs.vector().length) return null;
      sim = switch (fi.getVectorSimilarityFunction()) {                 // field/VectorFieldDef.java:77-88
        case COSINE -> 0;
        case DOT_PRODUCT -> 1;
        case EUCLIDEAN -> 2;
        case MAXIMUM_INNER_PRODUCT -> 3;
      };
    }
    if (sim < 0 || k > NrtGpu.MAX_K) return null;
    try (Arena a = Arena.ofConfined()) {
      MemorySegment segs = a.allocate(ADDRESS, leaves.size()), bases = a.allocate(JAVA_INT, leaves.size());
      for (int i = 0; i < leaves.size(); i++) {
        MemorySegment s = store.segmentOf(leaves.get(i));
        if (s == null) return null;
        segs.setAtIndex(ADDRESS, i, s);
        bases.setAtIndex(JAVA_INT, i, leaves.get(i).docBase);
      }
      MemorySegment q = a.allocate(JAVA_FLOAT, vs.vector().length);
      MemorySegment.copy(vs.vector(), 0, q, JAVA_FLOAT, 0, vs.vector().length);
      MemorySegment docs = a.allocate(JAVA_INT, k), scores = a.allocate(JAVA_FLOAT, k), out = a.allocate(NrtGpu.TOPDOCS);
      out.set(JAVA_INT, NrtGpuLayouts.TOPDOCS_CAPACITY, k);
      out.set(ADDRESS, NrtGpuLayouts.TOPDOCS_DOCS, docs);
      out.set(ADDRESS, NrtGpuLayouts.TOPDOCS_SCORES, scores);
      long deadline = timeoutSec > 0.0 ? (long) NrtGpu.MONOTONIC_NS.invokeExact() + (long) (timeoutSec * 1e9) : 0L;
      int status;
      try {
        NrtGpu.SET_DEADLINE.invokeExact(deadline);
        status = (int) NrtGpu.KNN_EXACT1.invokeExact(ctx, segs, bases, leaves.size(), store.fieldId(vs.field()), sim, q,
            vs.vector().length, k, vs.boost(), out);                       // blocks; merged with concurrent callers inside
      } finally {
        NrtGpu.SET_DEADLINE.invokeExact(0L);
      }
      if (status == NrtGpu.ERR_TIMEOUT) return TIMED_OUT;               // the request's budget is spent: no second run on the CPU path
      if (status == NrtGpu.ERR_UNSUPPORTED) return null;
      NrtGpu.check(status);
      // the count is exact (every live doc with a vector is collected), the RELATION is the reference collector's: it flips to
      // GREATER_THAN_OR_EQUAL_TO once a slice has collected more than max(totalHitsThreshold, numHits) hits with a full queue,
      // although this query's scorer ignores min competitive scores (LazyQueueTopScoreDocCollector.java:176-199)
      int gte = (int) NrtGpu.KNN_RELATION.invokeExact(ctx, segs, bases, leaves.size(), store.fieldId(vs.field()), k, totalHitsThreshold);
      if (gte < 0) NrtGpu.check(gte);
      int n = out.get(JAVA_INT, NrtGpuLayouts.TOPDOCS_N_HITS);
      ScoreDoc[] hits = new ScoreDoc[n];
      for (int i = 0; i < n; i++) hits[i] = new ScoreDoc(docs.getAtIndex(JAVA_INT, i), scores.getAtIndex(JAVA_FLOAT, i));
      return new TopDocs(new TotalHits(out.get(JAVA_LONG, NrtGpuLayouts.TOPDOCS_TOTAL_HITS),   // every live doc with a vector matches
          gte != 0 && n == k ? TotalHits.Relation.GREATER_THAN_OR_EQUAL_TO : TotalHits.Relation.EQUAL_TO), hits);
    } catch (IOException | RuntimeException e) {
      throw e;
    } catch (Throwable t) {
      throw new IOException(t);
    }
  }

  private static final TopDocs TIMED_OUT = new TopDocs(new TotalHits(0, TotalHits.Relation.EQUAL_TO), new ScoreDoc[0]);

  /**
   * NRTGPU_ERR_TIMEOUT: the request spent its timeoutSec waiting for the device -- nothing was launched for it, so there are no
   * partial results.  Reported the reference's way, NOT by running the query again on the CPU path under a fresh budget (a
   * timed-out request would then take up to twice its timeoutSec, under exactly the overload that made it time out): what
   * SearchCutoffWrapper makes of a timeout at the first segment boundary -- CollectionTimeoutException when partial results are
   * not allowed, else the timeout action runs and the (empty) result stands (SearchCutoffWrapper.java:117-131, 160-172;
   * timedOutBeforeCollection(): java/patches/nrtsearch-gpu-hook.diff).
   */
  private static SearcherResult timedOut(GpuEligibility.Eligible el) {
    if (el.cutoff() != null) el.cutoff().timedOutBeforeCollection();
    return new SearcherResult(new TopDocs(new TotalHits(0, TotalHits.Relation.EQUAL_TO), new ScoreDoc[0]), Map.of());
  }

  /** {total_ms, plan_ms, queue_ms, device_ms, postings} of the calling thread's last device search (SearchResponse.Diagnostics). */
  public static double[] lastDiagnostics() {
    return LAST_DIAGNOSTICS.get();
  }

  @Override
  @SuppressWarnings("unchecked")
  public <C extends Collector, T> T search(Query query, CollectorManager<C, T> manager) throws IOException {
    GpuEligibility.Eligible el = GpuEligibility.relevance(manager);
    if (el == null) return super.search(query, manager);
    RelevanceCollector rc = el.collector();
    Query rewritten = rewrite(query);
    GpuEligibility.VectorShape vs = GpuEligibility.vectorShape(rewritten);
    if (vs != null) {
      TopDocs top = rc.getSearchAfter() == null
          ? exactVectorSearch(vs, rc.getNumHitsToCollect(), rc.getTotalHitsThreshold(), el.timeoutSec()) : null;   // (no paging on this route)
      if (top == TIMED_OUT) return (T) timedOut(el);
      return top == null ? super.search(query, manager) : (T) new SearcherResult(top, Map.of());
    }
    GpuEligibility.Shape shape = GpuEligibility.shape(rewritten);
    if (shape == null) return super.search(query, manager);
    // the searcher's slices (MyIndexSearcher.slices): a slice goes to the device iff ALL its leaves are resident
    LeafSlice[] slices = getSlices();
    java.util.ArrayList<LeafReaderContext> leaves = new java.util.ArrayList<>();          // the device's leaves, in docBase order
    java.util.ArrayList<Integer> sliceOfLeaf = new java.util.ArrayList<>();
    java.util.ArrayList<LeafSlice> cold = new java.util.ArrayList<>();
    for (int si = 0; si < slices.length; si++) {
      boolean resident = true;
      for (LeafReaderContextPartition p : slices[si].partitions) resident &= store.segmentOf(p.ctx) != null;
      if (!resident) {
        cold.add(slices[si]);
        continue;
      }
      for (LeafReaderContextPartition p : slices[si].partitions) {
        leaves.add(p.ctx);
        sliceOfLeaf.add(si);
      }
    }
    if (leaves.isEmpty()) return super.search(query, manager);          // nothing resident (yet): CPU path
    if (!cold.isEmpty() && (rc.getSearchAfter() != null || !shape.filters().isEmpty() || !shape.mustNots().isEmpty()))
      return super.search(query, manager);                              // (the split is built for the plain shapes; masks are keyed by the whole leaf set)
    Integer[] order = new Integer[leaves.size()];
    for (int i = 0; i < order.length; i++) order[i] = i;
    java.util.Arrays.sort(order, java.util.Comparator.comparingInt(i -> leaves.get(i).docBase));
    try (Arena a = Arena.ofConfined()) {
      MemorySegment segs = a.allocate(ADDRESS, leaves.size()), bases = a.allocate(JAVA_INT, leaves.size());
      MemorySegment sliceIds = a.allocate(JAVA_INT, leaves.size());
      List<LeafReaderContext> deviceLeaves = new java.util.ArrayList<>();
      for (int i = 0; i < order.length; i++) {
        LeafReaderContext leaf = leaves.get(order[i]);
        deviceLeaves.add(leaf);
        segs.setAtIndex(ADDRESS, i, store.segmentOf(leaf));
        bases.setAtIndex(JAVA_INT, i, leaf.docBase);
        sliceIds.setAtIndex(JAVA_INT, i, sliceOfLeaf.get(order[i]));
      }
      leaves.clear();
      leaves.addAll(deviceLeaves);
      // FILTER / MUST_NOT clauses next to the scoring ones: resident doc-set masks (nrtgpu_segment_set_mask)
      int[] filterMasks = new int[shape.filters().size()], mustNotMasks = new int[shape.mustNots().size()];
      for (int i = 0; i < filterMasks.length; i++)
        if ((filterMasks[i] = masks.maskOf(this, store, leaves, shape.filters().get(i))) < 0) return super.search(query, manager);
      for (int i = 0; i < mustNotMasks.length; i++)
        if ((mustNotMasks[i] = masks.maskOf(this, store, leaves, shape.mustNots().get(i))) < 0) return super.search(query, manager);
      GpuEligibility.Plan plan = GpuEligibility.marshal(a, this, store, shape, filterMasks, mustNotMasks, rc.getNumHitsToCollect(),
          rc.getTotalHitsThreshold(), rc.getSearchAfter());
      if (plan == null) return super.search(query, manager);
      // timeoutSec of the request (DocCollector's SearchCutoffWrapper) -> the thread's deadline inside the library
      long deadline = el.timeoutSec() > 0.0 ? (long) NrtGpu.MONOTONIC_NS.invokeExact() + (long) (el.timeoutSec() * 1e9) : 0L;
      int status;
      try {
        NrtGpu.SET_DEADLINE.invokeExact(deadline);
        if (!cold.isEmpty()) NrtGpu.check((int) NrtGpu.SET_THREAD_SLICES.invokeExact(sliceIds, leaves.size()));   // count by the WHOLE searcher's slices
        status = (int) NrtGpu.SEARCH1.invokeExact(ctx, segs, bases, leaves.size(), plan.query(), plan.out());   // blocks; batched inside
      } finally {
        NrtGpu.SET_DEADLINE.invokeExact(0L);
        int ignored = (int) NrtGpu.SET_THREAD_SLICES.invokeExact(MemorySegment.NULL, 0);
      }
      if (status == NrtGpu.ERR_TIMEOUT) return (T) timedOut(el);       // out of time before anything was launched
      if (status == NrtGpu.ERR_UNSUPPORTED) return super.search(query, manager);   // not a shape / size the device takes
      NrtGpu.check(status);
      MemorySegment d = a.allocate(NrtGpu.DIAGNOSTICS);
      if ((int) NrtGpu.LAST_DIAGNOSTICS.invokeExact(d) == NrtGpu.OK)
        LAST_DIAGNOSTICS.set(new double[] {d.get(JAVA_DOUBLE, NrtGpuLayouts.DIAGNOSTICS_TOTAL_MS), d.get(JAVA_DOUBLE, NrtGpuLayouts.DIAGNOSTICS_PLAN_MS),
            d.get(JAVA_DOUBLE, NrtGpuLayouts.DIAGNOSTICS_QUEUE_MS), d.get(JAVA_DOUBLE, NrtGpuLayouts.DIAGNOSTICS_DEVICE_MS),
            (double) d.get(JAVA_LONG, NrtGpuLayouts.DIAGNOSTICS_POSTINGS)});
      TopDocs top = plan.toTopDocs();
      if (!cold.isEmpty()) {
        // the cold slices on Lucene's own path: one collector per slice (IndexSearcher.search(leaves, weight, collector)), reduced by
        // the request's manager, then TopDocs.merge with the device's part -- (score desc, doc asc), totalHits summed, the relation
        // GREATER_THAN_OR_EQUAL_TO if either part's is
        Weight weight = createWeight(rewritten, ScoreMode.TOP_SCORES, 1.0f);
        java.util.ArrayList<C> collectors = new java.util.ArrayList<>();
        for (LeafSlice sl : cold) {
          C c = manager.newCollector();
          collectors.add(c);
          search(sl.partitions, weight, c);
        }
        TopDocs cpu = ((SearcherResult) manager.reduce(collectors)).getTopDocs();
        top = TopDocs.merge(0, rc.getNumHitsToCollect(), new TopDocs[] {top, cpu}, java.util.Comparator.comparingInt((ScoreDoc d) -> d.doc));
      }
      return (T) new SearcherResult(top, Map.of());       // search/SearcherResult.java:31-34
    } catch (IOException | RuntimeException e) {
      throw e;
    } catch (Throwable t) {
      throw new IOException(t);
    }
  }
}
