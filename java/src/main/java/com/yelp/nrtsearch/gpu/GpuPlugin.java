package com.yelp.nrtsearch.gpu;

import static java.lang.foreign.ValueLayout.*;

import com.yelp.nrtsearch.server.config.NrtsearchConfig;
import com.yelp.nrtsearch.server.plugins.Plugin;
import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;

/**
 * Lifecycle (plugins/PluginsService.java:195-201: ctor (NrtsearchConfig); close() at shutdown, :90-99): one native
 * context per process / GPU.  The searcher factory hook (index/ShardState.java:506-527) asks this plugin for the
 * context and the segment store when it builds a GpuIndexSearcher, and passes the index's slicing live settings on
 * (sliceMaxDocs / sliceMaxSegments / virtualShards -> nrtgpu_set_slicing: TotalHits.relation is decided per slice).
 * NOT COMPILED here (no JDK).
 */
public class GpuPlugin extends Plugin {
  private final MemorySegment ctx;
  private final GpuSegmentStore store;

  public GpuPlugin(NrtsearchConfig config) throws Exception {
    try (Arena a = Arena.ofConfined()) {
      MemorySegment cfg = a.allocate(NrtGpu.CONFIG);
      cfg.set(JAVA_INT, 0, Integer.getInteger("nrtgpu.device", 0));
      cfg.set(JAVA_INT, 4, 1024);     // max_batch
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
    try {
      NrtGpu.DESTROY.invokeExact(ctx);
    } catch (Throwable ignored) {
    }
  }
}
