package com.yelp.nrtsearch.gpu;

import static java.lang.foreign.ValueLayout.*;

import java.io.IOException;
import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.util.*;
import java.util.concurrent.ConcurrentHashMap;
import org.apache.lucene.index.*;
import org.apache.lucene.search.DocIdSetIterator;
import org.apache.lucene.util.Bits;
import org.apache.lucene.util.BytesRef;

/**
 * One resident replica (nrtgpu_seg) per segment core, keyed like the reference keys its own per-segment caches
 * (IndexReader.CacheKey + addClosedListener, cf. field/TextBaseFieldDef.java:335-371).  Filled through Lucene's public
 * reader APIs -- TermsEnum / PostingsEnum (the per-leaf iteration idiom of query/MatchPhrasePrefixQuery.java:241-270),
 * getNormValues, getFloatVectorValues -- so nothing depends on Lucene104PostingsFormat internals.  Called from
 * GpuIndexSearcher's factory hook at refresh / warm time (index/ShardState.java:506-527), never on the query path.
 * NOT COMPILED here (no JDK).
 */
final class GpuSegmentStore {
  /** A native handle and its generation: `uid` is never reused, the handle's ADDRESS may be (the allocator recycles it), so
   *  whoever remembers handles across a release (GpuMaskCache) keys them by uid. */
  record Handle(MemorySegment seg, long uid) {}

  private final MemorySegment ctx;
  private final Map<IndexReader.CacheKey, Handle> resident = new ConcurrentHashMap<>();   // by segment core
  private final Map<IndexReader.CacheKey, Handle> versions = new ConcurrentHashMap<>();   // by leaf reader version: forks
  private final Map<String, Integer> fieldIds = new ConcurrentHashMap<>();
  private final java.util.concurrent.atomic.AtomicLong nextUid = new java.util.concurrent.atomic.AtomicLong(1);
  private volatile GpuMaskCache masks;    // told about every release BEFORE the handle is freed (GpuPlugin wires it)

  GpuSegmentStore(MemorySegment ctx) { this.ctx = ctx; }

  void attach(GpuMaskCache maskCache) { this.masks = maskCache; }

  /** nrtgpu_segment_release, after everybody who remembers the handle has forgotten it -- under the mask cache's lock, so that
   *  an eviction (which calls nrtgpu_segment_set_mask on the handles it remembers) never meets a freed one.  Searches still in
   *  flight over the handle are the library's business: the last of them frees it (include/nrtgpu.h). */
  private void release(Handle h) {
    GpuMaskCache m = masks;
    if (m != null) {
      m.releaseHandle(h);
    } else {
      try { NrtGpu.RELEASE.invokeExact(h.seg()); } catch (Throwable ignored) { }
    }
  }

  int fieldId(String field) { return fieldIds.computeIfAbsent(field, f -> fieldIds.size() + 1); }

  /** 64-bit id of (term bytes): the same at upload and at query time.  FNV-1a 64 is not injective: should two terms of one
   *  field ever collide, nrtgpu_segment_add_terms refuses the second (term ids must be unique per field) and sync() leaves that
   *  segment on the CPU path instead of letting the two posting lists alias. */
  static long termHash(BytesRef term) {
    long h = 0xcbf29ce484222325L;
    for (int i = 0; i < term.length; i++) h = (h ^ (term.bytes[term.offset + i] & 0xff)) * 0x100000001b3L;
    return h;
  }

  /**
   * Segments of `reader` that are not resident yet are uploaded (keyed by the segment CORE: postings, norms and vectors
   * are immutable).  A leaf reader with deletes gets its own handle per READER version: nrtgpu_segment_fork shares the
   * core's data and carries that version's liveDocs, so a search holding an older IndexSearcher keeps its point-in-time
   * view and a refresh never waits for running searches.  Forks go away with their reader, the data with the core.
   */
  void sync(DirectoryReader reader, Collection<String> textFields, Collection<String> vectorFields) throws IOException {
    for (LeafReaderContext lc : reader.leaves()) {
      LeafReader leaf = lc.reader();
      IndexReader.CacheHelper core = leaf.getCoreCacheHelper();
      if (core == null) continue;                       // not cacheable: this leaf stays on the CPU path
      Handle r = resident.get(core.getKey());
      if (r == null) {
        MemorySegment seg;
        try {
          seg = upload(leaf, textFields, vectorFields);
        } catch (IllegalArgumentException collision) {   // NRTGPU_ERR_INVALID_ARG: a term id added twice
          continue;                                       // this segment is searched by Lucene (segmentOf -> null)
        }
        r = new Handle(seg, nextUid.getAndIncrement());
        resident.put(core.getKey(), r);
        core.addClosedListener(key -> {                 // segment merged away / last reader closed
          Handle gone = resident.remove(key);            // (out of the map first: nobody finds it any more)
          if (gone != null) release(gone);
        });
      }
      IndexReader.CacheHelper version = leaf.getReaderCacheHelper();
      if (leaf.getLiveDocs() != null && version != null && !versions.containsKey(version.getKey())) {
        Handle fork = new Handle(fork(r.seg(), leaf.getLiveDocs(), leaf.maxDoc()), nextUid.getAndIncrement());
        versions.put(version.getKey(), fork);
        version.addClosedListener(key -> {              // this reader version is gone
          Handle gone = versions.remove(key);
          if (gone != null) release(gone);
        });
      }
    }
  }

  /** The handle a search over this leaf READER uses: its version's fork when it has deletes, else the core's handle. */
  Handle handleOf(LeafReaderContext lc) {
    LeafReader leaf = lc.reader();
    IndexReader.CacheHelper version = leaf.getReaderCacheHelper();
    if (leaf.getLiveDocs() != null)
      return version == null ? null : versions.get(version.getKey());   // (null: a version the store has not seen -- CPU path)
    IndexReader.CacheHelper core = leaf.getCoreCacheHelper();
    return core == null ? null : resident.get(core.getKey());
  }

  MemorySegment segmentOf(LeafReaderContext lc) {
    Handle h = handleOf(lc);
    return h == null ? null : h.seg();
  }

  private static MemorySegment fork(MemorySegment seg, Bits live, int maxDoc) throws IOException {
    try (Arena a = Arena.ofConfined()) {
      int words = (maxDoc + 63) >>> 6;
      MemorySegment bits = a.allocate((long) words * 8);
      for (int d = 0; d < maxDoc; d++)
        if (live.get(d)) bits.setAtIndex(JAVA_LONG, d >>> 6, bits.getAtIndex(JAVA_LONG, d >>> 6) | (1L << (d & 63)));
      MemorySegment out = a.allocate(ADDRESS);
      NrtGpu.check((int) NrtGpu.FORK.invokeExact(seg, bits, words, out));
      return out.get(ADDRESS, 0);
    } catch (IOException | RuntimeException e) {
      throw e;
    } catch (Throwable t) {
      throw new IOException(t);
    }
  }

  private MemorySegment upload(LeafReader leaf, Collection<String> textFields, Collection<String> vectorFields) throws IOException {
    try (Arena a = Arena.ofConfined()) {
      MemorySegment out = a.allocate(ADDRESS);
      NrtGpu.check((int) NrtGpu.SEG_BEGIN.invokeExact(ctx, leaf.maxDoc(), 0, out));
      MemorySegment seg = out.get(ADDRESS, 0);
      for (String field : textFields) {
        Terms terms = leaf.terms(field);
        if (terms == null) continue;
        int fid = fieldId(field);
        NumericDocValues norms = leaf.getNormValues(field);
        if (norms != null) {                            // norms omitted (ATOM fields, AtomFieldDef.java:123-126): NULL => 1
          MemorySegment nb = a.allocate(leaf.maxDoc());
          for (int d = norms.nextDoc(); d != DocIdSetIterator.NO_MORE_DOCS; d = norms.nextDoc()) nb.set(JAVA_BYTE, d, (byte) norms.longValue());
          NrtGpu.check((int) NrtGpu.ADD_NORMS.invokeExact(seg, fid, nb));
        } else {
          NrtGpu.check((int) NrtGpu.ADD_NORMS.invokeExact(seg, fid, MemorySegment.NULL));
        }
        // postings column-major, in groups of ~64 M postings per add_terms call (bounded staging memory)
        final long groupCap = 1L << 26;
        List<Long> hashes = new ArrayList<>();
        List<Long> offs = new ArrayList<>(List.of(0L));
        MemorySegment docs = a.allocate(groupCap * 4), freqs = a.allocate(groupCap * 4);
        long n = 0;
        TermsEnum te = terms.iterator();
        PostingsEnum pe = null;
        boolean hasFreqs = terms.hasFreqs();
        for (BytesRef t = te.next(); t != null; t = te.next()) {
          if (n + te.docFreq() > groupCap) { flushTerms(a, seg, fid, hashes, offs, docs, hasFreqs ? freqs : MemorySegment.NULL); n = 0; }
          pe = te.postings(pe, hasFreqs ? PostingsEnum.FREQS : PostingsEnum.NONE);
          for (int d = pe.nextDoc(); d != DocIdSetIterator.NO_MORE_DOCS; d = pe.nextDoc(), n++) {
            docs.setAtIndex(JAVA_INT, n, d);
            if (hasFreqs) freqs.setAtIndex(JAVA_INT, n, pe.freq());
          }
          hashes.add(termHash(t));
          offs.add(n);
        }
        flushTerms(a, seg, fid, hashes, offs, docs, hasFreqs ? freqs : MemorySegment.NULL);
      }
      for (String field : vectorFields) {
        FloatVectorValues vv = leaf.getFloatVectorValues(field);   // API use: VectorFieldDefTest.java:2301-2315
        if (vv == null || vv.size() == 0) continue;
        int dim = vv.dimension(), n = vv.size();
        MemorySegment rows = a.allocate((long) n * dim * 4), ord2doc = a.allocate((long) n * 4);
        KnnVectorValues.DocIndexIterator it = vv.iterator();
        for (int d = it.nextDoc(); d != DocIdSetIterator.NO_MORE_DOCS; d = it.nextDoc()) {
          int ord = it.index();
          ord2doc.setAtIndex(JAVA_INT, ord, d);
          MemorySegment.copy(vv.vectorValue(ord), 0, rows, JAVA_FLOAT, (long) ord * dim * 4, dim);
        }
        NrtGpu.check((int) NrtGpu.ADD_VECTORS.invokeExact(seg, fieldId(field), dim, n, n == leaf.maxDoc() ? MemorySegment.NULL : ord2doc, rows));
      }
      NrtGpu.check((int) NrtGpu.SEAL.invokeExact(seg));
      return seg;
    } catch (IOException | RuntimeException e) {
      throw e;
    } catch (Throwable t) {
      throw new IOException(t);
    }
  }

  private static void flushTerms(Arena a, MemorySegment seg, int fid, List<Long> hashes, List<Long> offs, MemorySegment docs,
      MemorySegment freqs) throws Throwable {
    if (hashes.isEmpty()) return;
    MemorySegment h = a.allocate((long) hashes.size() * 8), o = a.allocate((long) offs.size() * 8);
    for (int i = 0; i < hashes.size(); i++) h.setAtIndex(JAVA_LONG, i, hashes.get(i));
    for (int i = 0; i < offs.size(); i++) o.setAtIndex(JAVA_LONG, i, offs.get(i));
    NrtGpu.check((int) NrtGpu.ADD_TERMS.invokeExact(seg, fid, (long) hashes.size(), h, o, docs, freqs));
    hashes.clear();
    offs.clear();
    offs.add(0L);
  }

  private static void setLiveDocs(MemorySegment seg, Bits live, int maxDoc) throws IOException {
    try (Arena a = Arena.ofConfined()) {
      if (live == null) { NrtGpu.check((int) NrtGpu.SET_LIVE.invokeExact(seg, MemorySegment.NULL, 0)); return; }
      int words = (maxDoc + 63) >>> 6;
      MemorySegment bits = a.allocate((long) words * 8);
      for (int d = 0; d < maxDoc; d++)
        if (live.get(d)) bits.setAtIndex(JAVA_LONG, d >>> 6, bits.getAtIndex(JAVA_LONG, d >>> 6) | (1L << (d & 63)));
      NrtGpu.check((int) NrtGpu.SET_LIVE.invokeExact(seg, bits, words));
    } catch (IOException | RuntimeException e) {
      throw e;
    } catch (Throwable t) {
      throw new IOException(t);
    }
  }
}
