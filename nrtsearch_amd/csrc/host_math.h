// host_math.h -- host-side restatements the Java shim would take from Lucene objects:
// SmallFloat, BM25Similarity statistics, MyIndexSearcher.slices.  Product code (not the oracle):
// kept independent of oracle/ on purpose; tests cross-check the two.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <unordered_map>
#include <vector>

namespace nrtgpu {
namespace hostmath {

// org.apache.lucene.util.SmallFloat (Lucene 10.4.0): 4-bit-mantissa float-like encoding of
// non-negative ints; the first NUM_FREE_VALUES (24) values are stored exactly.
inline int32_t long_to_int4(int64_t i) {
  const int num_bits = (i == 0) ? 0 : 64 - __builtin_clzll((unsigned long long)i);
  if (num_bits < 4) return (int32_t)i;
  const int shift = num_bits - 4;
  int32_t encoded = (int32_t)((uint64_t)i >> shift);
  encoded &= 0x07;
  encoded |= (shift + 1) << 3;
  return encoded;
}
inline int64_t int4_to_long(int32_t i) {
  const int64_t bits = i & 0x07;
  const int shift = (int)((uint32_t)i >> 3) - 1;
  return shift == -1 ? bits : ((bits | 0x08) << shift);
}
constexpr int32_t kNumFreeValues = 24;  // 255 - longToInt4(Integer.MAX_VALUE)
inline int32_t int_to_byte4(int32_t i) {
  if (i < 0) return -1;
  if (i < kNumFreeValues) return i;
  return (kNumFreeValues + long_to_int4((int64_t)i - kNumFreeValues)) & 0xFF;
}
inline int32_t byte4_to_int(int32_t b) {
  b &= 0xFF;
  if (b < kNumFreeValues) return b;
  return (int32_t)(kNumFreeValues + int4_to_long(b - kNumFreeValues));
}

// BM25Similarity (k1 = 1.2f, b = 0.75f by default; /root/reference/src/main/java/com/yelp/nrtsearch/
// server/similarity/SimilarityCreator.java:33).  Statistics are index-global.
inline float bm25_idf(int64_t doc_count, int64_t doc_freq) {
  return (float)std::log(1.0 + ((double)(doc_count - doc_freq) + 0.5) / ((double)doc_freq + 0.5));
}
inline float bm25_avgdl(int64_t sum_total_term_freq, int64_t doc_count) {
  return (float)((double)sum_total_term_freq / (double)doc_count);
}
// cache[i] = 1f / (k1 * ((1 - b) + b * LENGTH_TABLE[i] / avgdl)), fp32 in this association.
// (compiled with -ffp-contract=off; volatile keeps every intermediate a rounded float)
// BM25Similarity SimScorer.score(freq, norm) with Java's float op order (this header is compiled
// with -ffp-contract=off): weight - weight / (1f + freq * normInverse).
inline float bm25_score(float weight, float freq, float norm_inverse) {
  const float prod = freq * norm_inverse;
  const float den = 1.0f + prod;
  const float quo = weight / den;
  return weight - quo;
}

inline void bm25_norm_cache(float avgdl, float k1, float b, float* out256) {
  for (int i = 0; i < 256; ++i) {
    volatile float len = (float)byte4_to_int(i);
    volatile float omb = 1.0f - b;
    volatile float t = b * len;
    t = t / avgdl;
    t = omb + t;
    t = k1 * t;
    out256[i] = 1.0f / t;
  }
}

// java.util.PriorityQueue restated (binary heap, siftUp/siftDown exactly as OpenJDK) so that
// ties are broken the way the reference breaks them in MyIndexSearcher.slicesForShards.
template <class T, class Less>
class JavaPriorityQueue {
 public:
  explicit JavaPriorityQueue(Less less) : less_(less) {}
  bool empty() const { return q_.empty(); }
  size_t size() const { return q_.size(); }
  const T& peek() const { return q_[0]; }
  void add(const T& x) {
    q_.push_back(x);
    size_t k = q_.size() - 1;
    while (k > 0) {  // siftUp
      const size_t parent = (k - 1) >> 1;
      if (!less_(x, q_[parent])) break;  // comparator.compare(x, e) >= 0
      q_[k] = q_[parent];
      k = parent;
    }
    q_[k] = x;
  }
  T poll() {
    T result = q_[0];
    const size_t n = q_.size() - 1;
    T x = q_[n];
    q_.pop_back();
    if (n > 0) {  // siftDown(0, x)
      size_t k = 0;
      const size_t half = n >> 1;
      while (k < half) {
        size_t child = (k << 1) + 1;
        const size_t right = child + 1;
        if (right < n && less_(q_[right], q_[child])) child = right;  // compare(c, right) > 0
        if (!less_(q_[child], x)) break;                             // compare(x, c) <= 0
        q_[k] = q_[child];
        k = child;
      }
      q_[k] = x;
    }
    return result;
  }

 private:
  std::vector<T> q_;
  Less less_;
};

struct LeafInfo {
  int32_t index;     // position in the reader's leaf list
  int32_t max_doc;
  int32_t num_docs;  // live docs
  int32_t doc_base;
};

// MyIndexSearcher.slices(leaves, maxDocsPerSlice, maxSegmentsPerSlice)
// (/root/reference/src/main/java/com/yelp/nrtsearch/server/search/MyIndexSearcher.java:163-208).
// Returns slices as lists of leaf indices (each sorted by docBase, :202).
inline std::vector<std::vector<int32_t>> slices(std::vector<LeafInfo> leaves, int32_t max_docs_per_slice,
                                                int32_t max_segments_per_slice, const std::vector<LeafInfo>& all) {
  std::stable_sort(leaves.begin(), leaves.end(),
                   [](const LeafInfo& a, const LeafInfo& b) { return a.max_doc > b.max_doc; });
  std::vector<std::vector<int32_t>> grouped;
  int64_t doc_sum = 0;
  int cur = -1;
  for (const LeafInfo& l : leaves) {
    if (l.max_doc > max_docs_per_slice) {
      grouped.push_back({l.index});
    } else {
      if (cur < 0) {
        grouped.push_back({l.index});
        cur = (int)grouped.size() - 1;
      } else {
        grouped[cur].push_back(l.index);
      }
      doc_sum += l.max_doc;
      if ((int32_t)grouped[cur].size() >= max_segments_per_slice || doc_sum > max_docs_per_slice) {
        cur = -1;
        doc_sum = 0;
      }
    }
  }
  for (auto& g : grouped)
    std::stable_sort(g.begin(), g.end(), [&](int32_t a, int32_t b) { return all[a].doc_base < all[b].doc_base; });
  return grouped;
}

// MyIndexSearcher.slicesForShards (…MyIndexSearcher.java:117-160): LPT over live docs into
// `virtual_shards` containers, slices per shard, slices ordered by max docs descending.
inline std::vector<std::vector<int32_t>> slices_for_shards(const std::vector<LeafInfo>& all, int32_t virtual_shards,
                                                           int32_t max_docs_per_slice, int32_t max_segments_per_slice,
                                                           std::vector<int32_t>* shard_of_leaf) {
  std::vector<std::vector<int32_t>> out;
  if (all.empty()) return out;
  std::vector<LeafInfo> sorted = all;
  std::stable_sort(sorted.begin(), sorted.end(),
                   [](const LeafInfo& a, const LeafInfo& b) { return a.num_docs > b.num_docs; });
  struct Shard { int32_t id; int64_t num_docs; std::vector<LeafInfo> leaves; };
  std::vector<Shard> shards((size_t)virtual_shards);
  auto shard_less = [&shards](int32_t a, int32_t b) { return shards[a].num_docs < shards[b].num_docs; };
  JavaPriorityQueue<int32_t, decltype(shard_less)> pq(shard_less);
  for (int32_t i = 0; i < virtual_shards; ++i) {
    shards[i].id = i;
    shards[i].num_docs = 0;
    pq.add(i);
  }
  for (const LeafInfo& l : sorted) {
    const int32_t s = pq.poll();
    shards[s].leaves.push_back(l);
    shards[s].num_docs += l.num_docs;
    pq.add(s);
  }
  if (shard_of_leaf) shard_of_leaf->assign(all.size(), -1);
  struct SliceAndSize { std::vector<int32_t> leaves; int64_t num_docs; };
  std::vector<SliceAndSize> pool;
  auto slice_greater = [&pool](int32_t a, int32_t b) { return pool[a].num_docs > pool[b].num_docs; };
  JavaPriorityQueue<int32_t, decltype(slice_greater)> sorted_slices(slice_greater);
  while (!pq.empty()) {
    const int32_t s = pq.poll();
    if (shards[s].leaves.empty()) continue;
    if (shard_of_leaf)
      for (const LeafInfo& l : shards[s].leaves) (*shard_of_leaf)[l.index] = s;
    auto shard_slices = slices(shards[s].leaves, max_docs_per_slice, max_segments_per_slice, all);
    for (auto& sl : shard_slices) {
      int64_t md = 0;  // LeafSlice.getMaxDocs(): sum of partition maxDocs
      for (int32_t li : sl) md += all[li].max_doc;
      pool.push_back({sl, md});
      sorted_slices.add((int32_t)pool.size() - 1);
    }
  }
  while (!sorted_slices.empty()) out.push_back(pool[sorted_slices.poll()].leaves);
  return out;
}

// WeightedRrfBlenderOperation.mergeHits + BlenderOperation.sortAndPaginate
// (/root/reference/src/main/java/com/yelp/nrtsearch/server/search/multiretriever/blender/operation/
// WeightedRrfBlenderOperation.java:53-78, .../score/WeightedRRFScoreDoc.java:62,75, .../BlenderOperation.java:96-132):
// score(doc) = sum over retrievers, in declaration order, of boost / (k + rank) (float, rank 1-based); hits merged in a
// java.util.HashMap<Integer, ...> whose values() order feeds a size-bounded java.util.PriorityQueue (min-heap on score;
// an incoming doc displaces the root only if strictly greater); the heap is drained from the back.  Equal scores
// therefore come out in an order that depends on both containers -- restated here (HashMap: buckets of a power-of-two
// table in index order, insertion order inside a bucket, hash(key) = key ^ (key >>> 16), resize at 0.75 load; buckets
// that Java would treeify (>= 8 colliding keys in a table of >= 64) are not modelled).  weighted == false is the
// score-order blender's sibling with the same containers: score = sum of boost * retriever score.
struct BlendHit { int32_t doc; float score; };
inline int64_t blend_hits(int32_t n_retrievers, const int32_t* const* docs, const float* const* scores, const int32_t* counts,
                          const float* boosts, int32_t k, bool rrf, int32_t start_hit, int32_t top_hits, std::vector<BlendHit>* page) {
  page->clear();
  if (top_hits == 0 || start_hit > top_hits) return 0;
  // mergeHits: insertion-ordered entries + the HashMap's bucket structure
  struct Entry { int32_t doc; float score; };
  std::vector<Entry> entries;
  std::unordered_map<int32_t, size_t> where;
  for (int32_t r = 0; r < n_retrievers; ++r) {
    const float boost = boosts ? boosts[r] : 1.0f;
    for (int32_t i = 0; i < counts[r]; ++i) {
      const float add = rrf ? boost / (float)(k + i + 1) : boost * scores[r][i];
      auto it = where.find(docs[r][i]);
      if (it == where.end()) {
        where.emplace(docs[r][i], entries.size());
        entries.push_back({docs[r][i], add});
      } else {
        entries[it->second].score += add;
      }
    }
  }
  const size_t total = entries.size();
  size_t cap = 16;
  while ((double)total > 0.75 * (double)cap) cap <<= 1;  // HashMap.resize(): threshold = 0.75 * capacity, doubling
  std::vector<size_t> order(total);
  for (size_t i = 0; i < total; ++i) order[i] = i;
  auto bucket = [&](size_t i) {
    const uint32_t h = (uint32_t)entries[i].doc;
    return (size_t)((h ^ (h >> 16)) & (uint32_t)(cap - 1));
  };
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bucket(a) < bucket(b); });  // values(): bucket order, insertion order inside
  // sortAndPaginate
  const size_t capacity = std::min<size_t>((size_t)top_hits, total);
  auto less = [&](size_t a, size_t b) { return entries[a].score < entries[b].score; };   // Float.compare for the non-NaN, non-(-0.0 vs 0.0) scores here
  JavaPriorityQueue<size_t, decltype(less)> heap(less);
  for (size_t idx : order) {
    if (heap.size() < capacity) {
      heap.add(idx);
    } else if (capacity > 0 && entries[idx].score > entries[heap.peek()].score) {
      heap.poll();
      heap.add(idx);
    }
  }
  std::vector<size_t> topk(heap.size());
  for (size_t i = topk.size(); i-- > 0;) topk[i] = heap.poll();
  for (size_t i = std::min<size_t>((size_t)start_hit, topk.size()); i < topk.size(); ++i) page->push_back({entries[topk[i]].doc, entries[topk[i]].score});
  return (int64_t)total;
}

// How many work items each query of a batch is cut into (planner.cpp; exported as nrtgpu_plan_item_counts for
// the tests).  cost[q] = the query's postings + a per-sub-tile constant; 0 = matches nothing (no item).
// One item runs on one CU.  Default rule: a query is cut only when it alone exceeds the batch's fair share of
// a CU (cost / per_item, rounded), items never cheaper than min_item_cost.  A small batch (at most half as many
// queries as CUs) whose rounded counts overshoot the CUs is apportioned instead into EXACTLY target_items
// items by largest remainder: near-equal items one more than the CUs would run in two rounds with most CUs
// idle in the second.
inline void plan_item_counts(const int64_t* cost, int32_t n, int64_t target_items, int64_t min_item_cost, int64_t* items) {
  int64_t total = 0, n_live = 0, n_items = 0;
  for (int32_t q = 0; q < n; ++q) total += cost[q];
  const int64_t per_item = std::max<int64_t>(min_item_cost, total / std::max<int64_t>(1, target_items));
  for (int32_t q = 0; q < n; ++q) {
    items[q] = 0;
    if (cost[q] <= 0) continue;
    ++n_live;
    items[q] = std::max<int64_t>(1, (cost[q] + per_item / 2) / per_item);
    n_items += items[q];
  }
  if (n_live == 0 || n_live * 2 > target_items || n_items <= target_items || total < target_items * min_item_cost) return;
  std::vector<std::pair<double, int32_t>> frac;
  int64_t given = 0;
  for (int32_t q = 0; q < n; ++q) {
    if (cost[q] <= 0) continue;
    const double share = (double)cost[q] * (double)target_items / (double)total;
    items[q] = std::max<int64_t>(1, (int64_t)share);
    given += items[q];
    frac.emplace_back(share < 1.0 ? 0.0 : share - std::floor(share), q);  // (rounded up to one item already)
  }
  std::stable_sort(frac.begin(), frac.end(),
                   [](const std::pair<double, int32_t>& a, const std::pair<double, int32_t>& b) { return a.first > b.first; });
  for (size_t i = 0; i < frac.size() && given < target_items; ++i, ++given) items[frac[i].second]++;
}

}  // namespace hostmath
}  // namespace nrtgpu
