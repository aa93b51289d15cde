/*
 * LuceneGolden -- golden vectors from the reference's own engine (lucene-core 10.x, the dependency nrtsearch pins at 10.4.0:
 * gradle/libs.versions.toml:7) for every score structure SURVEY.md 8(c) lists as [Lucene-recall] only.  No JDK exists in the
 * build image, so this file has never been compiled there; it is kept ready so that ONE command on a box with a JDK and a
 * lucene-core jar turns each unpinned item into a committed fixture (scripts/make_lucene_goldens.sh):
 *   javac -cp "$LUCENE_JARS" bench/lucene/LuceneBaseline.java bench/lucene/LuceneGolden.java -d /tmp/lg
 *   java  -cp "/tmp/lg:$LUCENE_JARS" LuceneGolden <fixture-dump-dir> <index-dir> <out.json>
 *
 * Input: the dump scripts/dump_corpus.py --fixture writes (the layout LuceneBaseline reads, plus):
 *   filter.u8      n_docs bytes, 1 = the doc carries the filter term ("flt:1"): the doc set of the FILTER / MUST_NOT clauses
 *   shapes.txt     one query per line: <shape> <k> <totalHitsThreshold> <param> <n_terms> <term ids ...>
 *                  shapes: should (plain disjunction), dismax (param = tie breaker), must_should (param = number of leading MUST
 *                  clauses, the others SHOULD), msm (param = minimumNumberShouldMatch), filter (SHOULD clauses + FILTER flt:1),
 *                  must_not (SHOULD clauses + MUST_NOT flt:1), boost (param = boost of the first clause, BoostQuery)
 * The index is LuceneBaseline's (same DumpLengthSimilarity, same segment boundaries, NoMergePolicy) plus the untokenised field
 * "flt".  Searches run on ONE thread through IndexSearcher.search(query, TopScoreDocCollectorManager(k, null, threshold)) -- the
 * call RelevanceCollector makes (src/main/java/com/yelp/nrtsearch/server/search/collectors/RelevanceCollector.java:63-68) -- with
 * the query shapes QueryNodeMapper builds (src/main/java/com/yelp/nrtsearch/server/query/QueryNodeMapper.java:257-283, 350-358).
 *
 * Output (JSON): per query {shape, k, threshold, param, terms, total, gte, docs, score_bits} -- consumed by
 * tests/test_lucene_golden.py (oracle and device against them) once copied to tests/golden/lucene_shapes.json.
 */
import java.io.*;
import java.nio.*;
import java.nio.file.*;
import java.util.*;

import org.apache.lucene.analysis.Analyzer;
import org.apache.lucene.document.Document;
import org.apache.lucene.document.Field;
import org.apache.lucene.document.FieldType;
import org.apache.lucene.document.StringField;
import org.apache.lucene.index.*;
import org.apache.lucene.search.*;
import org.apache.lucene.search.similarities.BM25Similarity;
import org.apache.lucene.store.FSDirectory;

public final class LuceneGolden {
  static final String FIELD = LuceneBaseline.FIELD;

  static Query term(long id) {
    return new TermQuery(new Term(FIELD, "t" + id));
  }

  static Query build(String shape, double param, long[] terms) {
    BooleanQuery.Builder b = new BooleanQuery.Builder();
    switch (shape) {
      case "should":
        for (long t : terms) b.add(term(t), BooleanClause.Occur.SHOULD);
        return b.build();
      case "dismax": {
        List<Query> disjuncts = new ArrayList<>();
        for (long t : terms) disjuncts.add(term(t));
        return new DisjunctionMaxQuery(disjuncts, (float) param);
      }
      case "must_should": {
        int nMust = (int) param;
        for (int i = 0; i < terms.length; i++) b.add(term(terms[i]), i < nMust ? BooleanClause.Occur.MUST : BooleanClause.Occur.SHOULD);
        return b.build();
      }
      case "msm":
        for (long t : terms) b.add(term(t), BooleanClause.Occur.SHOULD);
        b.setMinimumNumberShouldMatch((int) param);
        return b.build();
      case "filter":
        for (long t : terms) b.add(term(t), BooleanClause.Occur.SHOULD);
        b.setMinimumNumberShouldMatch(1);   // (QueryNodeMapper.java:259-261: a FILTER next to SHOULD clauses alone keeps them required)
        b.add(new TermQuery(new Term("flt", "1")), BooleanClause.Occur.FILTER);
        return b.build();
      case "must_not":
        for (long t : terms) b.add(term(t), BooleanClause.Occur.SHOULD);
        b.add(new TermQuery(new Term("flt", "1")), BooleanClause.Occur.MUST_NOT);
        return b.build();
      case "boost":
        for (int i = 0; i < terms.length; i++) b.add(i == 0 ? new BoostQuery(term(terms[i]), (float) param) : term(terms[i]), BooleanClause.Occur.SHOULD);
        return b.build();
      default:
        throw new IllegalArgumentException("unknown shape " + shape);
    }
  }

  public static void main(String[] args) throws Exception {
    Path dump = Paths.get(args[0]), indexDir = Paths.get(args[1]), out = Paths.get(args[2]);
    Scanner meta = new Scanner(dump.resolve("meta.txt"));
    int nDocs = meta.nextInt(), nTerms = meta.nextInt();
    meta.nextInt();
    meta.nextInt();
    meta.nextInt();
    int nSegments = meta.nextInt();
    int[] segSize = new int[nSegments];
    for (int i = 0; i < nSegments; i++) segSize[i] = meta.nextInt();
    IntBuffer lengths = LuceneBaseline.ints(dump.resolve("lengths.i32"));
    LongBuffer termIds = LuceneBaseline.longs(dump.resolve("terms.i64")), offsets = LuceneBaseline.longs(dump.resolve("offsets.i64"));
    IntBuffer docids = LuceneBaseline.ints(dump.resolve("docids.i32")), freqs = LuceneBaseline.ints(dump.resolve("freqs.i32"));
    byte[] filter = Files.readAllBytes(dump.resolve("filter.u8"));

    if (!DirectoryReader.indexExists(FSDirectory.open(indexDir))) {
      IndexWriterConfig cfg = new IndexWriterConfig((Analyzer) null);
      LuceneBaseline.DumpLengthSimilarity indexSim = new LuceneBaseline.DumpLengthSimilarity();
      cfg.setSimilarity(indexSim);
      cfg.setMergePolicy(NoMergePolicy.INSTANCE);
      cfg.setRAMBufferSizeMB(IndexWriterConfig.DISABLE_AUTO_FLUSH);
      cfg.setMaxBufferedDocs(IndexWriterConfig.DISABLE_AUTO_FLUSH);
      FieldType ft = new FieldType();
      ft.setIndexOptions(IndexOptions.DOCS_AND_FREQS);
      ft.setTokenized(true);
      ft.setOmitNorms(false);
      ft.freeze();
      long[] cursor = new long[nTerms];
      for (int t = 0; t < nTerms; t++) cursor[t] = offsets.get(t);
      PriorityQueue<long[]> heap = new PriorityQueue<>(Comparator.comparingLong(a -> a[0]));
      for (int t = 0; t < nTerms; t++)
        if (cursor[t] < offsets.get(t + 1)) heap.add(new long[] {docids.get((int) cursor[t]), t});
      try (IndexWriter w = new IndexWriter(FSDirectory.open(indexDir), cfg)) {
        LuceneBaseline.DocTokens tokens = new LuceneBaseline.DocTokens();
        int d = 0;
        for (int s = 0; s < nSegments; s++) {
          for (int i = 0; i < segSize[s]; i++, d++) {
            tokens.reset(lengths.get(d));
            indexSim.currentLength = lengths.get(d);
            while (!heap.isEmpty() && heap.peek()[0] == d) {
              long[] e = heap.poll();
              int t = (int) e[1];
              tokens.add(termIds.get(t), freqs.get((int) cursor[t]));
              if (++cursor[t] < offsets.get(t + 1)) {
                e[0] = docids.get((int) cursor[t]);
                heap.add(e);
              }
            }
            Document doc = new Document();
            Field f = new Field(FIELD, tokens, ft);
            doc.add(f);
            if (filter[d] != 0) doc.add(new StringField("flt", "1", Field.Store.NO));
            w.addDocument(doc);
          }
          w.flush();
        }
        w.commit();
      }
    }

    List<String> lines = Files.readAllLines(dump.resolve("shapes.txt"));
    try (DirectoryReader reader = DirectoryReader.open(FSDirectory.open(indexDir));
         PrintWriter pw = new PrintWriter(Files.newBufferedWriter(out))) {
      IndexSearcher searcher = new IndexSearcher(reader);   // one thread: one collector per slice all the same (IndexSearcher.slices)
      searcher.setSimilarity(new BM25Similarity());
      pw.printf("{\"lucene\": \"%s\", \"java\": \"%s\", \"n_docs\": %d, \"segments\": %d, \"queries\": [", org.apache.lucene.util.Version.LATEST,
          System.getProperty("java.version"), nDocs, reader.leaves().size());
      boolean first = true;
      for (String line : lines) {
        line = line.trim();
        if (line.isEmpty() || line.startsWith("#")) continue;
        String[] p = line.split("\\s+");
        String shape = p[0];
        int k = Integer.parseInt(p[1]), threshold = Integer.parseInt(p[2]);
        double param = Double.parseDouble(p[3]);
        int n = Integer.parseInt(p[4]);
        long[] terms = new long[n];
        for (int i = 0; i < n; i++) terms[i] = Long.parseLong(p[5 + i]);
        TopDocs td = searcher.search(build(shape, param, terms), new TopScoreDocCollectorManager(k, null, threshold));
        pw.printf("%s{\"shape\": \"%s\", \"k\": %d, \"threshold\": %d, \"param\": %s, \"terms\": %s, \"total\": %d, \"gte\": %b, \"docs\": [",
            first ? "" : ", ", shape, k, threshold, Double.toString(param), Arrays.toString(terms), td.totalHits.value(),
            td.totalHits.relation() == TotalHits.Relation.GREATER_THAN_OR_EQUAL_TO);
        first = false;
        for (int i = 0; i < td.scoreDocs.length; i++) pw.printf("%s%d", i == 0 ? "" : ",", td.scoreDocs[i].doc);
        pw.print("], \"score_bits\": [");
        for (int i = 0; i < td.scoreDocs.length; i++) pw.printf("%s%d", i == 0 ? "" : ",", Float.floatToIntBits(td.scoreDocs[i].score));
        pw.print("]}");
      }
      pw.println("]}");
    }
  }
}
