package com.yelp.nrtsearch.gpu;

import java.io.IOException;
import org.apache.lucene.search.*;
import org.apache.lucene.util.Bits;

/**
 * Seam B1 (SURVEY 8b, first row): the route that needs NO server patch.  A custom Query returned by a FieldTypePlugin's
 * FieldDef (TermQueryable, field/properties/TermQueryable.java:38-53) creates a Weight whose ScorerSupplier hands out
 * this BulkScorer; the device computes the LEAF's top-k and the scorer replays those hits, in ascending docid order,
 * into the caller's LeafCollector with a Scorable that returns the stored score -- so the reference's collector, its
 * wrappers and its totalHits logic stay untouched (API shape: query/multifunction/MultiFunctionScoreQuery.java:260-297).
 *
 * What B1 cannot give: batching across requests and multi-GPU sharding happen above the leaf (B2), and the replayed
 * stream holds only k docs per leaf, so totalHits becomes "hits replayed" -- the caller must run with
 * totalHitsThreshold <= k for the relation to stay meaningful.  Hence B2 is the preferred seam (INTEGRATION.md).
 * NOT COMPILED here (no JDK).
 */
final class GpuTermsBulkScorer extends BulkScorer {
  private final int[] docs;        // leaf docids, ascending
  private final float[] scores;    // the device's scores, same order
  private final long cost;

  GpuTermsBulkScorer(int[] docsAscending, float[] scores) {
    this.docs = docsAscending;
    this.scores = scores;
    this.cost = docsAscending.length;
  }

  @Override
  public int score(LeafCollector collector, Bits acceptDocs, int min, int max) throws IOException {
    final int[] cur = {-1};
    final float[] curScore = {0f};
    collector.setScorer(new Scorable() {
      @Override public float score() { return curScore[0]; }
      @Override public void setMinCompetitiveScore(float minScore) { /* the device already pruned; nothing to skip */ }
    });
    int i = java.util.Arrays.binarySearch(docs, min);
    if (i < 0) i = -i - 1;
    for (; i < docs.length && docs[i] < max; i++) {
      if (acceptDocs != null && !acceptDocs.get(docs[i])) continue;   // deletes were applied on the device as well
      cur[0] = docs[i];
      curScore[0] = scores[i];
      collector.collect(docs[i]);                                     // CollectionTerminatedException propagates
    }
    return i < docs.length ? docs[i] : DocIdSetIterator.NO_MORE_DOCS;
  }

  @Override
  public long cost() { return cost; }
}
