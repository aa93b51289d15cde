package com.yelp.nrtsearch.gpu;

import static java.lang.foreign.ValueLayout.*;

import com.yelp.nrtsearch.server.rescore.RescoreContext;
import com.yelp.nrtsearch.server.rescore.RescoreOperation;
import java.io.IOException;
import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import org.apache.lucene.search.ScoreDoc;
import org.apache.lucene.search.TopDocs;

/**
 * Seam B3: a RescorerPlugin operation (rescore/RescoreOperation.java:25-36, selected by Rescorer.pluginRescorer,
 * search.proto:1361-1367) that does what QueryRescore does with an exact vector rescoreQuery
 * (rescore/QueryRescore.java:40-57): combined = (float)(queryWeight * first + rescoreWeight * vectorScore), re-sorted
 * (score desc, doc asc), trimmed to windowSize.  The input array is not mutated (the reference clones as well:
 * rescore/ScriptRescore.java:75).  NOT COMPILED here (no JDK).
 */
final class GpuVectorRescorer implements RescoreOperation {
  private final MemorySegment ctx;
  private final GpuSegmentStore store;
  private final String field;
  private final int sim;
  private final float[] queryVector;
  private final double queryWeight, rescoreWeight;

  GpuVectorRescorer(MemorySegment ctx, GpuSegmentStore store, String field, int sim, float[] queryVector, double qw, double rw) {
    this.ctx = ctx; this.store = store; this.field = field; this.sim = sim; this.queryVector = queryVector;
    this.queryWeight = qw; this.rescoreWeight = rw;
  }

  @Override
  public TopDocs rescore(TopDocs hits, RescoreContext context) throws IOException {
    var leaves = context.getSearchContext().getSearcherAndTaxonomy().searcher().getIndexReader().leaves();
    int n = hits.scoreDocs.length, window = context.getWindowSize();
    try (Arena a = Arena.ofConfined()) {
      MemorySegment segs = a.allocate(ADDRESS, leaves.size()), bases = a.allocate(JAVA_INT, leaves.size());
      for (int i = 0; i < leaves.size(); i++) {
        MemorySegment s = store.segmentOf(leaves.get(i));
        if (s == null) throw new IOException("segment not resident");      // the provider registers a CPU fallback instead
        segs.setAtIndex(ADDRESS, i, s);
        bases.setAtIndex(JAVA_INT, i, leaves.get(i).docBase);
      }
      MemorySegment docs = a.allocate(JAVA_INT, n), first = a.allocate(JAVA_FLOAT, n), q = a.allocate(JAVA_FLOAT, queryVector.length);
      for (int i = 0; i < n; i++) { docs.setAtIndex(JAVA_INT, i, hits.scoreDocs[i].doc); first.setAtIndex(JAVA_FLOAT, i, hits.scoreDocs[i].score); }
      MemorySegment.copy(queryVector, 0, q, JAVA_FLOAT, 0, queryVector.length);
      MemorySegment od = a.allocate(JAVA_INT, window), os = a.allocate(JAVA_FLOAT, window), out = a.allocate(NrtGpu.TOPDOCS);
      out.set(JAVA_INT, NrtGpuLayouts.TOPDOCS_CAPACITY, window);
      out.set(ADDRESS, NrtGpuLayouts.TOPDOCS_DOCS, od);
      out.set(ADDRESS, NrtGpuLayouts.TOPDOCS_SCORES, os);
      NrtGpu.check((int) NrtGpu.RESCORE.invokeExact(ctx, segs, bases, leaves.size(), store.fieldId(field), sim, q, queryVector.length, 1.0f,
          docs, first, n, queryWeight, rescoreWeight, window, out));
      int m = out.get(JAVA_INT, NrtGpuLayouts.TOPDOCS_N_HITS);
      ScoreDoc[] res = new ScoreDoc[m];
      for (int i = 0; i < m; i++) res[i] = new ScoreDoc(od.getAtIndex(JAVA_INT, i), os.getAtIndex(JAVA_FLOAT, i));
      return new TopDocs(hits.totalHits, res);                            // QueryRescorer keeps the first pass's TotalHits
    } catch (IOException | RuntimeException e) {
      throw e;
    } catch (Throwable t) {
      throw new IOException(t);
    }
  }
}
