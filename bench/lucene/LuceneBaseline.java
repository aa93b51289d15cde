/*
 * LuceneBaseline -- the reference's CPU path on the same synthetic corpus and queries as bench.py (SURVEY.md 8d,
 * BASELINE.md 3): what nrtsearch executes below SearchHandler.java:1412 for a pure-SHOULD BooleanQuery of TermQuery
 * clauses under the default BM25Similarity with RelevanceCollector's TopScoreDocCollectorManager(k, null, 1000)
 * (src/main/java/com/yelp/nrtsearch/server/search/collectors/RelevanceCollector.java:63-68).
 *
 * This image has no JDK and no Lucene jars (SURVEY 0), so the file cannot be compiled here; bench.py probes
 * `java -version` + LUCENE_JARS at run time and, when both exist, compiles and runs it:
 *   javac -cp "$LUCENE_JARS" bench/lucene/LuceneBaseline.java -d /tmp/lb
 *   java  -cp "/tmp/lb:$LUCENE_JARS" LuceneBaseline <corpus-dump-dir> <index-dir> <threads> <out.json> [maxQueries]
 * Needs lucene-core 10.x only (the reference pins 10.4.0, gradle/libs.versions.toml:7).
 *
 * Input: the dump scripts/dump_corpus.py writes from nrtsearch_amd/synth.py (little-endian):
 *   meta.txt      n_docs n_terms n_queries terms_per_query k n_segments, then the segment sizes
 *   lengths.i32   n_docs doc lengths (tokens of the TEXT field)
 *   terms.i64     n_terms term ids (Zipf ranks), ascending
 *   offsets.i64   n_terms + 1 offsets into docids / freqs
 *   docids.i32, freqs.i32   postings, term-major, GLOBAL docids ascending per term
 *   queries.i64   n_queries x terms_per_query term ids
 * The index is built so that Lucene's postings, norms and docids equal the dump's: doc d gets, for every dumped term
 * with a posting in d, that term repeated freq times; its norm is SmallFloat.intToByte4(lengths[d]) through a
 * Similarity whose computeNorm returns the dumped length (the synthetic corpus draws freqs independently of the doc
 * length, so the token count cannot stand in for it; only the query set's terms are materialised anyway); documents
 * are added in docid order, one flush per dumped segment under NoMergePolicy, so segment i holds exactly the dump's
 * docid range i.  Scoring at search time is the stock BM25Similarity.
 *
 * Output (JSON): per query the top-k docids and Float.floatToIntBits(score), totalHits + relation, plus timings
 * (queries/s at `threads` threads through IndexSearcher's executor, p50/p99 per query single-threaded).  bench.py
 * reports it as cpu_baseline.kind = "lucene" and diffs docids / score bits against the device's answers: the only
 * route to parity against the reference itself for the four items SURVEY 8c lists as unpinned.
 */
import java.io.*;
import java.nio.*;
import java.nio.channels.FileChannel;
import java.nio.file.*;
import java.util.*;
import java.util.concurrent.*;

import org.apache.lucene.analysis.Analyzer;
import org.apache.lucene.analysis.TokenStream;
import org.apache.lucene.analysis.tokenattributes.CharTermAttribute;
import org.apache.lucene.document.Document;
import org.apache.lucene.document.Field;
import org.apache.lucene.document.FieldType;
import org.apache.lucene.index.*;
import org.apache.lucene.search.*;
import org.apache.lucene.search.similarities.BM25Similarity;
import org.apache.lucene.store.FSDirectory;

public final class LuceneBaseline {
  static final String FIELD = "body";

  /** Index-time similarity: BM25's scorer, but the norm encodes the dumped doc length. */
  static final class DumpLengthSimilarity extends org.apache.lucene.search.similarities.Similarity {
    private final BM25Similarity bm25 = new BM25Similarity();
    int currentLength = 1;
    @Override public long computeNorm(org.apache.lucene.index.FieldInvertState state) {
      return org.apache.lucene.util.SmallFloat.intToByte4(currentLength);
    }
    @Override public SimScorer scorer(float boost, CollectionStatistics collectionStats, TermStatistics... termStats) {
      return bm25.scorer(boost, collectionStats, termStats);
    }
  }

  /** One document's tokens: every dumped term of the doc, repeated freq times. */
  static final class DocTokens extends TokenStream {
    private final CharTermAttribute term = addAttribute(CharTermAttribute.class);
    long[] ids = new long[8];
    int[] freqs = new int[8];
    int n, length, ti, emittedOfTerm, emitted;

    void reset(int length) { n = 0; this.length = length; }
    void add(long id, int f) {
      if (n == ids.length) { ids = Arrays.copyOf(ids, 2 * n); freqs = Arrays.copyOf(freqs, 2 * n); }
      ids[n] = id; freqs[n] = f; n++;
    }
    @Override public void reset() throws IOException { super.reset(); ti = 0; emittedOfTerm = 0; emitted = 0; }
    @Override public boolean incrementToken() {
      clearAttributes();
      while (ti < n && emittedOfTerm == freqs[ti]) { ti++; emittedOfTerm = 0; }
      if (ti < n) { term.setEmpty().append('t').append(Long.toString(ids[ti])); emittedOfTerm++; emitted++; return true; }
      return false;
    }
  }

  static IntBuffer ints(Path p) throws IOException {
    try (FileChannel ch = FileChannel.open(p, StandardOpenOption.READ)) {
      return ch.map(FileChannel.MapMode.READ_ONLY, 0, ch.size()).order(ByteOrder.LITTLE_ENDIAN).asIntBuffer();
    }
  }
  static LongBuffer longs(Path p) throws IOException {
    try (FileChannel ch = FileChannel.open(p, StandardOpenOption.READ)) {
      return ch.map(FileChannel.MapMode.READ_ONLY, 0, ch.size()).order(ByteOrder.LITTLE_ENDIAN).asLongBuffer();
    }
  }

  public static void main(String[] args) throws Exception {
    Path dump = Paths.get(args[0]), indexDir = Paths.get(args[1]);
    int threads = Integer.parseInt(args[2]);
    Path out = Paths.get(args[3]);
    Scanner meta = new Scanner(dump.resolve("meta.txt"));
    int nDocs = meta.nextInt(), nTerms = meta.nextInt(), nQueries = meta.nextInt(), perQuery = meta.nextInt(), k = meta.nextInt();
    int nSegments = meta.nextInt();
    int[] segSize = new int[nSegments];
    for (int i = 0; i < nSegments; i++) segSize[i] = meta.nextInt();
    if (args.length > 4) nQueries = Math.min(nQueries, Integer.parseInt(args[4]));
    IntBuffer lengths = ints(dump.resolve("lengths.i32"));
    LongBuffer termIds = longs(dump.resolve("terms.i64")), offsets = longs(dump.resolve("offsets.i64"));
    LongBuffer queries = longs(dump.resolve("queries.i64"));
    // NOTE: postings beyond 2^31 ints need several mappings; C3 (71 M postings) fits one.
    IntBuffer docids = ints(dump.resolve("docids.i32")), freqs = ints(dump.resolve("freqs.i32"));

    // ---- index: invert the term-major dump into doc order with one cursor per term
    long tBuild = System.nanoTime();
    if (!DirectoryReader.indexExists(FSDirectory.open(indexDir))) {
      IndexWriterConfig cfg = new IndexWriterConfig((Analyzer) null);
      DumpLengthSimilarity indexSim = new DumpLengthSimilarity();
      cfg.setSimilarity(indexSim);
      cfg.setMergePolicy(NoMergePolicy.INSTANCE);
      cfg.setRAMBufferSizeMB(IndexWriterConfig.DISABLE_AUTO_FLUSH);
      cfg.setMaxBufferedDocs(IndexWriterConfig.DISABLE_AUTO_FLUSH);   // one flush per dumped segment, below
      FieldType ft = new FieldType();
      ft.setIndexOptions(IndexOptions.DOCS_AND_FREQS);
      ft.setTokenized(true);
      ft.setOmitNorms(false);
      ft.freeze();
      long[] cursor = new long[nTerms];
      for (int t = 0; t < nTerms; t++) cursor[t] = offsets.get(t);
      // min-heap of (next docid, term): every doc pops the terms that hold it
      PriorityQueue<long[]> heap = new PriorityQueue<>(Comparator.comparingLong(a -> a[0]));
      for (int t = 0; t < nTerms; t++)
        if (cursor[t] < offsets.get(t + 1)) heap.add(new long[] {docids.get((int) cursor[t]), t});
      try (IndexWriter w = new IndexWriter(FSDirectory.open(indexDir), cfg)) {
        DocTokens tokens = new DocTokens();
        Document doc = new Document();
        Field f = new Field(FIELD, tokens, ft);
        doc.add(f);
        int d = 0;
        for (int s = 0; s < nSegments; s++) {
          for (int i = 0; i < segSize[s]; i++, d++) {
            tokens.reset(lengths.get(d));
            indexSim.currentLength = lengths.get(d);
            while (!heap.isEmpty() && heap.peek()[0] == d) {
              long[] e = heap.poll();
              int t = (int) e[1];
              tokens.add(termIds.get(t), freqs.get((int) cursor[t]));
              if (++cursor[t] < offsets.get(t + 1)) { e[0] = docids.get((int) cursor[t]); heap.add(e); }
            }
            f.setTokenStream(tokens);
            w.addDocument(doc);
          }
          w.flush();   // segment boundary == the dump's
        }
        w.commit();
      }
    }
    double buildS = (System.nanoTime() - tBuild) / 1e9;

    // ---- search
    ExecutorService pool = threads > 1 ? Executors.newFixedThreadPool(threads) : null;
    try (DirectoryReader reader = DirectoryReader.open(FSDirectory.open(indexDir))) {
      IndexSearcher searcher = new IndexSearcher(reader, pool);   // Lucene's own slices(): 250k docs / 5 segments per slice
      searcher.setSimilarity(new BM25Similarity());
      Query[] qs = new Query[nQueries];
      for (int q = 0; q < nQueries; q++) {
        BooleanQuery.Builder b = new BooleanQuery.Builder();
        for (int j = 0; j < perQuery; j++)
          b.add(new TermQuery(new Term(FIELD, "t" + queries.get((long) q * perQuery + j))), BooleanClause.Occur.SHOULD);
        qs[q] = b.build();
      }
      TopDocs[] res = new TopDocs[nQueries];
      for (int q = 0; q < Math.min(nQueries, 64); q++) searcher.search(qs[q], new TopScoreDocCollectorManager(k, null, 1000));  // warm
      long[] lat = new long[nQueries];
      long t0 = System.nanoTime();
      for (int q = 0; q < nQueries; q++) {
        long a = System.nanoTime();
        res[q] = searcher.search(qs[q], new TopScoreDocCollectorManager(k, null, 1000));
        lat[q] = System.nanoTime() - a;
      }
      double secs = (System.nanoTime() - t0) / 1e9;
      Arrays.sort(lat);
      try (PrintWriter pw = new PrintWriter(Files.newBufferedWriter(out))) {
        pw.printf("{\"lucene\": \"%s\", \"java\": \"%s\", \"threads\": %d, \"index_build_s\": %.1f, \"queries\": %d, "
                + "\"queries_per_s\": %.2f, \"p50_ms\": %.3f, \"p99_ms\": %.3f, \"segments\": %d, \"results\": [",
            org.apache.lucene.util.Version.LATEST, System.getProperty("java.version"), threads, buildS, nQueries,
            nQueries / secs, lat[nQueries / 2] / 1e6, lat[(int) (nQueries * 0.99)] / 1e6, reader.leaves().size());
        for (int q = 0; q < nQueries; q++) {
          TopDocs td = res[q];
          pw.printf("%s{\"total\": %d, \"gte\": %b, \"docs\": [", q == 0 ? "" : ", ", td.totalHits.value(),
              td.totalHits.relation() == TotalHits.Relation.GREATER_THAN_OR_EQUAL_TO);
          for (int i = 0; i < td.scoreDocs.length; i++) pw.printf("%s%d", i == 0 ? "" : ",", td.scoreDocs[i].doc);
          pw.print("], \"score_bits\": [");
          for (int i = 0; i < td.scoreDocs.length; i++) pw.printf("%s%d", i == 0 ? "" : ",", Float.floatToIntBits(td.scoreDocs[i].score));
          pw.print("]}");
        }
        pw.println("]}");
      }
    } finally {
      if (pool != null) pool.shutdown();
    }
  }
}
