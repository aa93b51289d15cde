// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the BM25 hot path.
//
//   fold_norms_kernel  seal-time: folds each posting's field-norm byte into its freq word
//   bm25_scan_kernel   postings traversal + BM25Similarity + disjunction sum + per-item top-k
//   merge_topk_kernel  TopDocs.merge of per-item (or per-GPU) top-k lists + final ordering
//
// What they replace in the reference (all inside lucene-core 10.4.0, reached from
// /root/reference/src/main/java/com/yelp/nrtsearch/server/handler/SearchHandler.java:1412-1413):
// PostingsEnum traversal, BM25 SimScorer.score(freq, norm), the double-accumulated SHOULD sum of
// MaxScoreBulkScorer/BooleanScorer, TopScoreDocCollector (in-repo copy:
// src/main/java/org/apache/lucene/search/LazyQueueTopScoreDocCollector.java:103-199) and
// TopDocs.merge (LazyQueueTopScoreDocCollectorManager.java:137-144).
//
// Arithmetic contract (bit-exact with Java): every BM25 op is a separate
// IEEE fp32 operation (this TU is compiled with -ffp-contract=off and without fast-math, so
// `/` is the correctly rounded division), per-doc term scores are added in fp64 (ds_add_f64 into
// the LDS tile; the sum of a few fp32 values is exact in fp64, hence order-independent) and the
// sum is rounded once to fp32.
//
// HBM layout: per upload group two u32 columns, docid[] and fnorm[] = (freq << 8) | normByte, both
// read with coalesced 16 B/lane non-temporal loads; the per-doc norm gather of the reference
// (norms.longValue() per posting) is paid once at seal time instead of per query.
// Roofline: HBM.  Bytes the kernel must move per posting = 4 (docid) + 4 (freq|norm) = 8.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.h"
#include "topk.hiph"

namespace nrtgpu {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define NRT_GLOBAL __attribute__((address_space(1)))
typedef const NRT_GLOBAL u32x4* gvec_ptr;
typedef const NRT_GLOBAL uint32_t* gu32_ptr;
typedef const NRT_GLOBAL float* gf32_ptr;

constexpr uint64_t kUnmatched = 0x8000000000000000ull;  // -0.0: "no term matched this doc yet"
constexpr int kPerThread = kTileDocs / kScanThreads;     // 16 accumulator slots per thread
constexpr int kPrefetch = 2;                             // posting groups per thread loaded one tile ahead

struct ScanSmem {
  double   acc[kTileDocs];                 // fp64 score accumulators of the current doc tile (64 KiB)
  uint64_t cand[kCandCap];                 // competitive hits (packed keys), unordered (10 KiB)
  float    cache[kLdsCaches][256];         // BM25 normInverse tables of the query's first fields
  uint64_t t_docids[kMaxTerms];            // global addresses (kept as integers: LDS strips address spaces)
  uint64_t t_fnorm[kMaxTerms];
  uint64_t t_celloff[kMaxTerms];
  uint64_t t_cache[kMaxTerms];             // global normInverse table (fields beyond kLdsCaches)
  uint64_t t_lo[kMaxTerms];                // [t_lo, t_hi): the term's postings inside the columns
  uint64_t t_hi[kMaxTerms];
  float    t_weight[kMaxTerms];
  uint32_t t_shift[kMaxTerms];
  uint32_t t_slot[kMaxTerms];              // cache table index
  uint64_t g_start[2][kMaxTerms];          // first 4-posting group of the term in the tile (parity buffered)
  uint32_t g_prefix[2][kMaxTerms + 1];     // groups of terms 0..t-1 in the tile
  TopkScratch sc;
  uint64_t theta;      // packed key of the k-th best hit seen so far (0 = none)
  uint32_t cnt;        // valid entries in cand
  uint32_t tile_cand;  // competitive hits found in the current tile
  uint32_t hits;       // live matching docs of this item
  uint32_t pad;
};
static_assert(sizeof(ScanSmem) <= 80 * 1024, "two scan workgroups must fit in one CU's 160 KiB LDS");

__device__ __forceinline__ uint64_t dbl_bits(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ double unmatched_value() { return __longlong_as_double((long long)kUnmatched); }

// Keep the k best candidates, raise theta, publish it for the other items of the same query
// (the analogue of LazyMaxScoreAccumulator.accumulate, /root/reference/src/main/java/org/apache/
// lucene/search/LazyMaxScoreAccumulator.java:53-57).  Uniform call; ends synchronised.
// `n` must be the caller's barrier-protected snapshot of s.cnt; returns the new count (uniform).
__device__ __noinline__ uint32_t scan_compact(ScanSmem& s, uint32_t n, uint32_t k, unsigned long long* theta_g) {
  uint64_t thr = 0;
  const uint32_t m = topk_compact<kScanThreads, kCandCap>(s.cand, n, k, &s.sc, &thr);
  if (n > k && threadIdx.x == 0) {
    s.cnt = m;
    if (thr > s.theta) s.theta = thr;
    atomicMax(theta_g, (unsigned long long)thr);
  }
  __syncthreads();
  return m;
}

// One 4-posting group: which term, where, and (once loaded) its two column words.
struct Group {
  u32x4 d4, f4;
  uint32_t t;
  bool valid;
};

__device__ __forceinline__ void group_locate_load(const ScanSmem& s, uint32_t par, uint32_t n_terms, uint32_t v,
                                                  uint32_t total, Group& gr) {
  gr.valid = v < total;
  gr.t = 0;
  if (gr.valid) {
    uint32_t t = 0;
    while (v >= s.g_prefix[par][t + 1]) ++t;
    const uint64_t g = s.g_start[par][t] + (v - s.g_prefix[par][t]);
    gr.t = t;
    gr.d4 = __builtin_nontemporal_load((gvec_ptr)s.t_docids[t] + g);
    gr.f4 = __builtin_nontemporal_load((gvec_ptr)s.t_fnorm[t] + g);
  }
  (void)n_terms;
}

// BM25Similarity SimScorer.score(freq, norm) for the 4 postings of a group + fp64 accumulate.
__device__ __forceinline__ void group_score(ScanSmem& s, uint32_t par, const Group& gr, uint32_t v, uint32_t base,
                                            uint32_t tile_len) {
  if (!gr.valid) return;
  const uint32_t t = gr.t;
  const uint64_t g = s.g_start[par][t] + (v - s.g_prefix[par][t]);
  const uint64_t lo = s.t_lo[t], hi = s.t_hi[t];
  const float w = s.t_weight[t];
  const uint32_t slot = s.t_slot[t];
  const uint64_t idx0 = g << 2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t idx = idx0 + (uint64_t)j;
    const uint32_t rel = gr.d4[j] - base;  // unsigned: docs below the tile wrap to huge values
    if (idx >= lo && idx < hi && rel < tile_len) {
      const uint32_t fn = gr.f4[j];
      const uint32_t nb = fn & 255u;
      const float ninv = (slot < (uint32_t)kLdsCaches) ? s.cache[slot][nb] : ((gf32_ptr)s.t_cache[t])[nb];
      const float freq = (float)(int32_t)(fn >> 8);
      // BM25Similarity: weight - weight / (1f + freq * normInverse), one rounding per op
      const float prod = freq * ninv;
      const float den = 1.0f + prod;
      const float quo = w / den;
      const float sc = w - quo;
      unsafeAtomicAdd(&s.acc[rel], (double)sc);
    }
  }
}

// Posting ranges of every term for one doc tile, in two halves so the cell-table loads can be in
// flight while wave 0 does its share of the accumulate phase.  Wave 0 only (tid < 64).
__device__ __forceinline__ void tile_cells_load(const ScanSmem& s, uint32_t n_terms, uint32_t tile, bool in_range,
                                                uint32_t& lo, uint32_t& hi) {
  const uint32_t tid = threadIdx.x;
  lo = 0;
  hi = 0;
  if (in_range && tid < n_terms) {
    const gu32_ptr co = (gu32_ptr)s.t_celloff[tid];
    const uint32_t cell = tile >> s.t_shift[tid];
    lo = co[cell];
    hi = co[cell + 1];
  }
}
__device__ __forceinline__ void tile_tables_store(ScanSmem& s, uint32_t par, uint32_t lo, uint32_t hi) {
  const uint32_t tid = threadIdx.x;
  uint32_t ng = 0;
  if (hi > lo) {  // only lanes < n_terms can have hi > lo
    const uint64_t a = s.t_lo[tid] + lo, b = s.t_lo[tid] + hi;
    const uint64_t gs = a >> 2, ge = (b + 3) >> 2;
    s.g_start[par][tid] = gs;
    ng = (uint32_t)(ge - gs);
  }
  uint32_t incl = ng;
#pragma unroll
  for (int d = 1; d < kMaxTerms; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (tid >= (uint32_t)d) incl += o;
  }
  if (tid < (uint32_t)kMaxTerms) s.g_prefix[par][tid + 1] = incl;
  if (tid == 0) s.g_prefix[par][0] = 0;
}

// PIPE = true: the first kPrefetch posting groups of tile j+1 are loaded before tile j is swept, so
// their HBM latency hides under the LDS-bound sweep.  PIPE = false: plain per-tile phases (kept for
// A/B measurement; results are identical).
template <bool PIPE>
__global__ __launch_bounds__(kScanThreads, 4)
void bm25_scan_kernel(const DItem* __restrict__ items, const DTerm* __restrict__ terms,
                      const DQuery* __restrict__ queries, const float* __restrict__ caches,
                      unsigned long long* __restrict__ theta_g, uint64_t* __restrict__ item_keys,
                      uint32_t* __restrict__ item_counts, uint64_t* __restrict__ item_hits,
                      uint32_t k_stride) {
  __shared__ ScanSmem s;
  const uint32_t tid = threadIdx.x;
  const DItem item = items[blockIdx.x];
  const DQuery q = queries[item.query];
  const uint32_t n_terms = item.n_terms;
  const uint32_t k = q.k;
  unsigned long long* const my_theta_g = theta_g + item.query;
  const NRT_GLOBAL uint64_t* const live_bits = (const NRT_GLOBAL uint64_t*)item.live_bits;

  // ---- item prologue: clear the tile, stage the term table and the normInverse tables in LDS
  for (uint32_t i = tid; i < (uint32_t)kTileDocs; i += kScanThreads) s.acc[i] = unmatched_value();
  if (tid < n_terms) {
    const DTerm t = terms[item.term_begin + tid];
    s.t_docids[tid] = (uint64_t)t.docids;
    s.t_fnorm[tid] = (uint64_t)t.fnorm;
    s.t_celloff[tid] = (uint64_t)t.cell_off;
    s.t_cache[tid] = (uint64_t)(caches + t.cache_off);
    s.t_lo[tid] = t.start;
    s.t_hi[tid] = t.start + t.count;
    s.t_weight[tid] = t.weight;
    s.t_shift[tid] = t.shift;
    s.t_slot[tid] = t.cache_slot;
  }
  {
    const uint32_t n_lds = min(item.n_caches, (uint32_t)kLdsCaches) * 256u;
    for (uint32_t i = tid; i < n_lds; i += kScanThreads) (&s.cache[0][0])[i] = caches[item.cache_off + i];
  }
  if (tid == 0) {
    s.theta = 0;
    s.cnt = 0;
    s.tile_cand = 0;
    s.hits = 0;
  }
  uint32_t my_hits = 0;
  __syncthreads();
  if (tid < 64) {
    uint32_t lo, hi;
    tile_cells_load(s, n_terms, item.tile_begin, true, lo, hi);
    tile_tables_store(s, item.tile_begin & 1u, lo, hi);
  }
  __syncthreads();

  Group pf[kPrefetch];
#pragma unroll
  for (int r = 0; r < kPrefetch; ++r) pf[r].valid = false;
  if (PIPE) {
    const uint32_t par0 = item.tile_begin & 1u;
    const uint32_t total0 = s.g_prefix[par0][n_terms];
#pragma unroll
    for (int r = 0; r < kPrefetch; ++r) group_locate_load(s, par0, n_terms, tid + (uint32_t)r * kScanThreads, total0, pf[r]);
  }

  for (uint32_t tile = item.tile_begin; tile < item.tile_end; ++tile) {
    const uint32_t par = tile & 1u;
    const uint32_t base = tile << kTileShift;
    const uint32_t tile_len = min((uint32_t)kTileDocs, item.max_doc - base);
    const uint32_t total_groups = s.g_prefix[par][n_terms];  // written before the last barrier

    // ---- (1) wave 0: start fetching the posting ranges of the NEXT tile and the shared theta
    uint32_t next_lo = 0, next_hi = 0;
    uint64_t theta_shared = 0;
    if (tid < 64) {
      tile_cells_load(s, n_terms, tile + 1, tile + 1 < item.tile_end, next_lo, next_hi);
      if (tid == 0) theta_shared = __hip_atomic_load(my_theta_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- (2) stream the postings: coalesced 16 B/lane column loads, fp32 BM25, fp64 LDS accumulate
    if (total_groups != 0) {
      uint32_t v = tid;
      if (PIPE) {
#pragma unroll
        for (int r = 0; r < kPrefetch; ++r) group_score(s, par, pf[r], tid + (uint32_t)r * kScanThreads, base, tile_len);
        v = tid + (uint32_t)kPrefetch * kScanThreads;
      }
      for (; v < total_groups; v += 2 * kScanThreads) {
        Group a, b;
        group_locate_load(s, par, n_terms, v, total_groups, a);
        group_locate_load(s, par, n_terms, v + kScanThreads, total_groups, b);
        group_score(s, par, a, v, base, tile_len);
        group_score(s, par, b, v + kScanThreads, base, tile_len);
      }
    }
    if (tid < 64) {
      tile_tables_store(s, par ^ 1u, next_lo, next_hi);
      if (tid == 0 && theta_shared > s.theta) s.theta = theta_shared;
    }
    __syncthreads();  // (A) accumulators complete; next tile's tables visible

    if (PIPE) {  // next tile's first groups: in flight while this tile is swept
      const uint32_t total_next = s.g_prefix[par ^ 1u][n_terms];
#pragma unroll
      for (int r = 0; r < kPrefetch; ++r) group_locate_load(s, par ^ 1u, n_terms, tid + (uint32_t)r * kScanThreads, total_next, pf[r]);
    }
    if (total_groups == 0) continue;  // uniform: no posting of any term falls in this tile

    // ---- (3) sweep the tile: count hits, find competitive docs, reset non-competitive slots
    const uint64_t theta = s.theta;
    uint32_t cmask = 0, ncand = 0;
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
      const uint32_t i = tid + (uint32_t)j * kScanThreads;
      const double a = s.acc[i];
      const uint64_t bits = dbl_bits(a);
      if (bits != kUnmatched) {
        const uint32_t doc = base + i;
        bool live = true;
        if (live_bits) live = (live_bits[doc >> 6] >> (doc & 63u)) & 1ull;
        bool cand = false;
        if (live) {
          ++my_hits;  // totalHits counts every collected doc, also those skipped by `after`
          const float sc = (float)a;
          const uint32_t gdoc = (uint32_t)(item.doc_base + (int32_t)doc);
          const bool skip = q.has_after && (sc > q.after_score || (sc == q.after_score && (int32_t)gdoc <= q.after_doc));
          if (!skip) cand = pack_key(sc, gdoc) > theta;
        }
        if (cand) {
          cmask |= 1u << j;
          ++ncand;
        } else {
          s.acc[i] = unmatched_value();
        }
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) ncand += __shfl_xor(ncand, d, 64);
    if (lane_id() == 0 && ncand) atomicAdd(&s.tile_cand, ncand);
    __syncthreads();  // (B)

    // ---- (4) collect the competitive docs into the LDS candidate buffer
    const uint32_t tc = s.tile_cand;
    uint32_t cnt0 = s.cnt;
    __syncthreads();  // (B2) every thread holds the same (tc, cnt0) before anyone appends
    if (tc > 0) {     // uniform
      if (cnt0 + tc > (uint32_t)kCandCap) cnt0 = scan_compact(s, cnt0, k, my_theta_g);  // raises theta
      const bool flood = (cnt0 + tc > (uint32_t)kCandCap);                               // uniform
      if (!flood) {
#pragma unroll 1
        for (int j = 0; j < kPerThread; ++j) {
          const uint32_t i = tid + (uint32_t)j * kScanThreads;
          bool want = (cmask >> j) & 1u;
          uint64_t key = 0;
          if (want) {
            key = pack_key((float)s.acc[i], (uint32_t)(item.doc_base + (int32_t)(base + i)));
            s.acc[i] = unmatched_value();
            want = key > s.theta;
          }
          topk_append(s.cand, &s.cnt, want, key);
        }
      } else {
        // start of an item: more competitive docs than buffer space; collect kFloodStep docs at a
        // time and compact (raising theta) whenever the next step might not fit
#pragma unroll 1
        for (int jj = 0; jj < kPerThread * (kScanThreads / kFloodStep); ++jj) {
          const int j = jj / (kScanThreads / kFloodStep);
          const uint32_t part = (uint32_t)jj % (kScanThreads / kFloodStep);
          const uint32_t i = tid + (uint32_t)j * kScanThreads;
          bool want = ((cmask >> j) & 1u) && (tid / kFloodStep == part);
          uint64_t key = 0;
          if (want) {
            key = pack_key((float)s.acc[i], (uint32_t)(item.doc_base + (int32_t)(base + i)));
            s.acc[i] = unmatched_value();
            want = key > s.theta;
          }
          topk_append(s.cand, &s.cnt, want, key);
          __syncthreads();
          const uint32_t c = s.cnt;
          __syncthreads();  // everyone has read cnt before the next step appends
          if (c > (uint32_t)(kCandCap - kFloodStep)) scan_compact(s, c, k, my_theta_g);
        }
      }
    }
    __syncthreads();  // (C)
    if (tid == 0) s.tile_cand = 0;  // next reader is after the next tile's barriers
  }

  // ---- item epilogue: final top-k of the item, hit count
  __syncthreads();
  {
    const uint32_t c = s.cnt;
    __syncthreads();
    if (c > k) scan_compact(s, c, k, my_theta_g);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) my_hits += __shfl_xor(my_hits, d, 64);
  if (lane_id() == 0 && my_hits) atomicAdd(&s.hits, my_hits);
  __syncthreads();
  const uint32_t n = s.cnt;
  uint64_t* out = item_keys + (size_t)blockIdx.x * k_stride;
  for (uint32_t i = tid; i < n; i += kScanThreads) out[i] = s.cand[i];
  if (tid == 0) {
    item_counts[blockIdx.x] = n;
    item_hits[blockIdx.x] = s.hits;
  }
}

// ------------------------------------------------------------------------------------------------
// merge_topk_kernel: one workgroup per query.  List l of query q is record list_idx[q_base[q] + l]
// of in_keys/in_counts/in_hits: per-item outputs of the scan, or all-gathered per-GPU results.
// Output: the k best keys in (score desc, doc asc) order.
// ------------------------------------------------------------------------------------------------
struct MergeSmem {
  uint64_t cand[kMergeCap];
  TopkScratch sc;
  uint64_t theta;
  unsigned long long hits;
  uint32_t cnt;
  uint32_t pad;
};

__device__ __noinline__ void merge_compact(MergeSmem& s, uint32_t n, uint32_t k) {
  uint64_t thr = 0;
  const uint32_t m = topk_compact<kScanThreads, kMergeCap>(s.cand, n, k, &s.sc, &thr);
  if (n > k && threadIdx.x == 0) {
    s.cnt = m;
    if (thr > s.theta) s.theta = thr;
  }
  __syncthreads();
}

__global__ __launch_bounds__(kScanThreads)
void merge_topk_kernel(const uint64_t* __restrict__ in_keys, const uint32_t* __restrict__ in_counts,
                       const uint64_t* __restrict__ in_hits, const uint32_t* __restrict__ list_idx,
                       const uint32_t* __restrict__ q_base, const uint32_t* __restrict__ q_nlists, uint32_t k_stride_in,
                       const uint32_t* __restrict__ q_k, uint64_t* __restrict__ out_keys,
                       uint32_t* __restrict__ out_counts, uint64_t* __restrict__ out_hits,
                       uint32_t k_stride_out) {
  __shared__ MergeSmem s;
  const uint32_t tid = threadIdx.x;
  const uint32_t q = blockIdx.x;
  const uint32_t k = q_k[q];
  const uint32_t nl = q_nlists[q];
  const uint32_t base = q_base[q];
  if (tid == 0) {
    s.theta = 0;
    s.hits = 0;
    s.cnt = 0;
  }
  __syncthreads();
  unsigned long long h = 0;
  for (uint32_t l = tid; l < nl; l += kScanThreads) h += in_hits[list_idx[base + l]];
  if (h) atomicAdd(&s.hits, h);

  for (uint32_t l = 0; l < nl; ++l) {
    const uint32_t rec = list_idx[base + l];
    const uint32_t c = min(in_counts[rec], k_stride_in);
    const uint64_t* src = in_keys + (size_t)rec * k_stride_in;
    for (uint32_t off = 0; off < c; off += kScanThreads) {
      const uint32_t i = off + tid;
      const uint64_t key = (i < c) ? src[i] : 0;
      const bool want = (i < c) && (key > s.theta);
      topk_append(s.cand, &s.cnt, want, key);
      __syncthreads();
      const uint32_t cn = s.cnt;
      __syncthreads();
      if (cn > (uint32_t)(kMergeCap - kScanThreads)) merge_compact(s, cn, k);
    }
  }
  __syncthreads();
  {
    const uint32_t c = s.cnt;
    __syncthreads();
    if (c > k) merge_compact(s, c, k);
  }
  const uint32_t n = s.cnt;
  uint32_t n2 = 1;
  while (n2 < n) n2 <<= 1;
  for (uint32_t i = n + tid; i < n2; i += kScanThreads) s.cand[i] = 0;
  bitonic_sort_desc<kScanThreads>(s.cand, n2);
  uint64_t* out = out_keys + (size_t)q * k_stride_out;
  for (uint32_t i = tid; i < k_stride_out; i += kScanThreads) out[i] = (i < n) ? s.cand[i] : 0;
  if (tid == 0) {
    out_counts[q] = n;
    out_hits[q] = s.hits;
  }
}

// ------------------------------------------------------------------------------------------------
// fold_norms_kernel (seal time): fnorm[p] = (freq[p] << 8) | norms[docid[p]].
// freqs == nullptr => freq 1 (IndexOptions.DOCS); norms == nullptr => norm byte 1 (norms omitted,
// /root/reference/src/main/java/com/yelp/nrtsearch/server/field/AtomFieldDef.java:123-126).
// Sets *overflow when a freq does not fit 24 bits (the segment then stays on the CPU path).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void fold_norms_kernel(const uint32_t* __restrict__ docids, const uint32_t* __restrict__ freqs,
                       const uint8_t* __restrict__ norms, uint32_t* __restrict__ fnorm, uint64_t n,
                       uint32_t* __restrict__ overflow) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    const uint32_t f = freqs ? freqs[p] : 1u;
    const uint32_t nb = norms ? (uint32_t)norms[docids[p]] : 1u;
    if (f >= (1u << 24)) *overflow = 1u;
    fnorm[p] = (f << 8) | nb;
  }
}

// ---- launchers (called from runtime.cpp) ---------------------------------------------------------
void launch_bm25_scan(hipStream_t stream, bool pipelined, uint32_t n_items, const DItem* items, const DTerm* terms,
                      const DQuery* queries, const float* caches, unsigned long long* theta_g,
                      uint64_t* item_keys, uint32_t* item_counts, uint64_t* item_hits, uint32_t k_stride) {
  if (n_items == 0) return;
  if (pipelined)
    hipLaunchKernelGGL(bm25_scan_kernel<true>, dim3(n_items), dim3(kScanThreads), 0, stream, items, terms, queries,
                       caches, theta_g, item_keys, item_counts, item_hits, k_stride);
  else
    hipLaunchKernelGGL(bm25_scan_kernel<false>, dim3(n_items), dim3(kScanThreads), 0, stream, items, terms, queries,
                       caches, theta_g, item_keys, item_counts, item_hits, k_stride);
}

void launch_merge_topk(hipStream_t stream, uint32_t n_queries, const uint64_t* in_keys, const uint32_t* in_counts,
                       const uint64_t* in_hits, const uint32_t* list_idx, const uint32_t* q_base,
                       const uint32_t* q_nlists, uint32_t k_stride_in, const uint32_t* q_k, uint64_t* out_keys,
                       uint32_t* out_counts, uint64_t* out_hits, uint32_t k_stride_out) {
  if (n_queries == 0) return;
  hipLaunchKernelGGL(merge_topk_kernel, dim3(n_queries), dim3(kScanThreads), 0, stream, in_keys, in_counts, in_hits,
                     list_idx, q_base, q_nlists, k_stride_in, q_k, out_keys, out_counts, out_hits, k_stride_out);
}

void launch_fold_norms(hipStream_t stream, const uint32_t* docids, const uint32_t* freqs, const uint8_t* norms,
                       uint32_t* fnorm, uint64_t n, uint32_t* overflow) {
  if (n == 0) return;
  const uint64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(fold_norms_kernel, dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, docids,
                     freqs, norms, fnorm, n, overflow);
}

}  // namespace nrtgpu
