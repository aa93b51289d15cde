package com.yelp.nrtsearch.gpu;

import static java.lang.foreign.ValueLayout.*;

import java.io.IOException;
import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.util.LinkedHashMap;
import java.util.List;
import java.util.Map;
import org.apache.lucene.index.LeafReaderContext;
import org.apache.lucene.search.DocIdSetIterator;
import org.apache.lucene.search.IndexSearcher;
import org.apache.lucene.search.Query;
import org.apache.lucene.search.ScoreMode;
import org.apache.lucene.search.Scorer;
import org.apache.lucene.search.Weight;

/**
 * FILTER / MUST_NOT clauses (built at query/QueryNodeMapper.java:257-283) as resident doc-set masks: the clause's per-leaf
 * DocIdSet -- what Lucene's LRUQueryCache would cache -- is materialised once per (clause, leaf) as 64-bit words and registered
 * with nrtgpu_segment_set_mask under an id; queries then name the id (nrtgpu_bm25_query.filter_mask / must_not_mask) and the
 * MaxScore kernel probes the mask when a doc's score is complete.  Bounded like the query cache: the least recently used
 * clause is dropped (its masks are removed from the segments) when more than `capacity` are resident.
 * Handle lifetime: what this cache remembers per mask id is keyed by the handle's GENERATION (GpuSegmentStore.Handle.uid, never
 * reused), and GpuSegmentStore frees a handle only through {@link #releaseHandle} -- under this cache's lock, after the handle
 * has been forgotten here -- so an eviction can never call nrtgpu_segment_set_mask on freed memory, a handle allocated later at
 * a recycled address is not mistaken for a registered one, and the per-id sets shrink as segments go away.
 * NOT COMPILED here (no JDK).
 */
final class GpuMaskCache {
  private final int capacity;
  private int nextId = 1;
  private final LinkedHashMap<Query, Integer> ids = new LinkedHashMap<>(16, 0.75f, true);
  private final Map<Integer, Map<Long, MemorySegment>> registered = new java.util.HashMap<>();   // mask id -> live handles (by uid) that carry it

  GpuMaskCache(int capacity) { this.capacity = capacity; }

  /** The mask id of `clause` over these leaves (materialising what is missing), or -1 when a leaf cannot take it. */
  synchronized int maskOf(IndexSearcher searcher, GpuSegmentStore store, List<LeafReaderContext> leaves, Query clause) throws IOException {
    Integer id = ids.get(clause);
    if (id == null) {
      if (ids.size() >= capacity) evictOldest();
      id = nextId++;
      ids.put(clause, id);
      registered.put(id, new java.util.HashMap<>());
    }
    Weight w = null;
    for (LeafReaderContext lc : leaves) {
      GpuSegmentStore.Handle h = store.handleOf(lc);   // (live for the whole call: a release waits for this cache's lock)
      if (h == null) return -1;
      if (registered.get(id).containsKey(h.uid())) continue;
      MemorySegment seg = h.seg();
      if (w == null) w = searcher.createWeight(searcher.rewrite(clause), ScoreMode.COMPLETE_NO_SCORES, 1f);
      int maxDoc = lc.reader().maxDoc(), words = (maxDoc + 63) >>> 6;
      try (Arena a = Arena.ofConfined()) {
        MemorySegment bits = a.allocate((long) words * 8);
        Scorer sc = w.scorer(lc);
        if (sc != null) {
          DocIdSetIterator it = sc.iterator();
          for (int d = it.nextDoc(); d != DocIdSetIterator.NO_MORE_DOCS; d = it.nextDoc())
            bits.setAtIndex(JAVA_LONG, d >>> 6, bits.getAtIndex(JAVA_LONG, d >>> 6) | (1L << (d & 63)));
        }
        NrtGpu.check((int) NrtGpu.SET_MASK.invokeExact(seg, (int) id, bits, words));
        registered.get(id).put(h.uid(), seg);
      } catch (IOException | RuntimeException e) {
        throw e;
      } catch (Throwable t) {
        throw new IOException(t);
      }
    }
    return id;
  }

  private void evictOldest() {
    Map.Entry<Query, Integer> oldest = ids.entrySet().iterator().next();
    ids.remove(oldest.getKey());
    for (MemorySegment seg : registered.remove(oldest.getValue()).values()) {   // live handles only: releaseHandle removes the others first
      try {
        int ignored = (int) NrtGpu.SET_MASK.invokeExact(seg, (int) oldest.getValue(), MemorySegment.NULL, 0);   // bits == NULL drops the mask
      } catch (Throwable ignored) {
      }
    }
  }

  /** A segment handle goes away (merged away / reader version closed): forgotten here, THEN freed -- one critical section. */
  synchronized void releaseHandle(GpuSegmentStore.Handle h) {
    for (Map<Long, MemorySegment> s : registered.values()) s.remove(h.uid());
    try {
      NrtGpu.RELEASE.invokeExact(h.seg());
    } catch (Throwable ignored) {
    }
  }
}
