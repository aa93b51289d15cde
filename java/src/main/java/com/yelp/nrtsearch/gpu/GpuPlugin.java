package com.yelp.nrtsearch.gpu;

import static java.lang.foreign.ValueLayout.*;

import com.yelp.nrtsearch.server.config.NrtsearchConfig;
import com.yelp.nrtsearch.server.plugins.Plugin;
import com.yelp.nrtsearch.server.search.MyIndexSearcher;
import org.apache.lucene.index.DirectoryReader;
import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

/**
 * Lifecycle (plugins/PluginsService.java:195-201: ctor (NrtsearchConfig); close() at shutdown, :90-99): one native
 * context per process / GPU.  The constructor installs a MyIndexSearcher.SearcherHook (java/patches/nrtsearch-gpu-hook.diff):
 * ShardState.ShardSearcherFactory.newSearcher (index/ShardState.java:506-527) then asks it for the searcher of every new reader
 * -- the segments that are not resident yet are uploaded there, at refresh / warm time, never on the query path -- and the
 * index's slicing live settings travel on (sliceMaxDocs / sliceMaxSegments / virtualShards -> nrtgpu_set_slicing:
 * TotalHits.relation is decided per slice).  Fields to replicate: -Dnrtgpu.textFields=a,b -Dnrtgpu.vectorFields=v.
 * NOT COMPILED here (no JDK).
 */
public class GpuPlugin extends Plugin {
  private final MemorySegment ctx;
  private final GpuSegmentStore store;
  private final GpuMaskCache masks;

  public GpuPlugin(NrtsearchConfig config) throws Exception {
    try (Arena a = Arena.ofConfined()) {
      MemorySegment cfg = a.allocate(NrtGpu.CONFIG);
      cfg.set(JAVA_INT, NrtGpuLayouts.CONFIG_DEVICE_ID, Integer.getInteger("nrtgpu.device", 0));
      cfg.set(JAVA_INT, NrtGpuLayouts.CONFIG_MAX_BATCH, 1024);     // max_batch
      MemorySegment out = a.allocate(ADDRESS);
      try {
        NrtGpu.check((int) NrtGpu.CREATE.invokeExact(cfg, out));       // fails loudly without a gfx950 device: no CPU fallback inside
      } catch (Exception e) {
        throw e;
      } catch (Throwable t) {
        throw new RuntimeException(t);
      }
      ctx = out.get(ADDRESS, 0);
    }
    store = new GpuSegmentStore(ctx);
    masks = new GpuMaskCache(Integer.getInteger("nrtgpu.maskCache", 64));
    store.attach(masks);   // handles are freed through the cache: it forgets them first (GpuMaskCache.releaseHandle)
    java.util.List<String> text = java.util.Arrays.asList(System.getProperty("nrtgpu.textFields", "").split(","));
    java.util.List<String> vectors = java.util.Arrays.asList(System.getProperty("nrtgpu.vectorFields", "").split(","));
    MyIndexSearcher.setSearcherHook((reader, previousReader, executor, slicing) -> {
      if (!(reader instanceof DirectoryReader dr)) return null;       // default searcher
      try {
        store.sync(dr, text, vectors);
        setSlicing(slicing.sliceMaxDocs(), slicing.sliceMaxSegments(), slicing.virtualShards());
      } catch (Exception e) {
        return null;                                                    // a reader that cannot be replicated: the CPU path serves it
      }
      return GpuIndexSearcher.create(reader, executor, slicing, ctx, store, masks);
    });
  }

  public MemorySegment context() { return ctx; }
  public GpuSegmentStore segments() { return store; }

  public void setSlicing(int sliceMaxDocs, int sliceMaxSegments, int virtualShards) throws Exception {
    try {
      NrtGpu.check((int) NrtGpu.SET_SLICING.invokeExact(ctx, sliceMaxDocs, sliceMaxSegments, virtualShards));
    } catch (Exception e) {
      throw e;
    } catch (Throwable t) {
      throw new RuntimeException(t);
    }
  }

  @Override
  public void close() {
    MyIndexSearcher.setSearcherHook(null);
    try {
      NrtGpu.DESTROY.invokeExact(ctx);
    } catch (Throwable ignored) {
    }
  }
}
