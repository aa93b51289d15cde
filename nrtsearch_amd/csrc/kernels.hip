// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the BM25 hot path.
//
//   fold_norms_kernel  seal-time: folds each posting's field-norm byte into its freq word
//   apply_live_kernel  per reader version: re-codes the postings of deleted docs to score the neutral element
//   bm25_scan_kernel   postings traversal + BM25Similarity + disjunction sum + per-item top-k
//                      (variants: clause counting for minimumNumberShouldMatch, doc-set masks for FILTER / MUST_NOT)
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
// `/` is the correctly rounded division), per-doc term scores are added exactly -- in fp64 (ds_add_f64
// into the LDS tile; the sum of a few fp32 values is exact in fp64, hence order-independent) or as
// fixed-point integers at a per-query scale (ds_add_u64; see "Accumulators" below) -- and the sum is
// rounded once to fp32.
//
// HBM layout: per upload group two u32 columns, docid[] and code[] (freq and the doc's norm byte
// folded into one word at seal time), both read with coalesced 16 B/lane non-temporal loads; the
// per-doc norm gather of the reference (norms.longValue() per posting) is paid once at seal.
// Scoring: (freq, norm byte) is a tiny domain, so each item builds the BM25 scores of its densest
// terms once (same float ops, bit-exact) into LDS tables and a posting costs one table read.
// Roofline: HBM.  Bytes the kernel must move per posting = 4 (docid) + 4 (freq|norm) = 8.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bm25_common.hiph"

namespace nrtgpu {

// Work decomposition: a workgroup is kScanWaves AUTONOMOUS waves and owns one CU (all 160 KiB of LDS).
// The sub-tiles (kTileDocs docs) of an item's parts form one sequence from which the waves help
// themselves; each wave on its own: load postings -> accumulate into its private LDS sub-tile ->
// collect the competitive docs into the workgroup's shared candidate buffer.  No workgroup barrier in
// steady state (not even between parts): the waves sit at different points of that chain and hide
// each other's LDS / HBM latency.  Per-term metadata of a sub-tile lives in lane registers (lane l <->
// term l) and in a small wave-private LDS table.  Barriers happen only at a rendezvous when the
// shared candidate buffer overflows (top-k compaction) and at the end of the item.
struct ScanSmem {
  uint64_t acc[kScanWaves][kTileDocs];     // score accumulators: one sub-tile per wave (96 KiB)
  uint64_t cand[kCandCap];                 // competitive hits of the item (packed keys), unordered (15 KiB)
  uint32_t tab[kTabTerms][kTabEntries];    // BM25 score of (freq, norm byte) for the item's densest terms:
                                           // fp32 bits, or the fixed-point integer at the term's scale
  float    cache[kLdsCaches][256];         // BM25 normInverse tables of the query's fields (division path)
  // wave-private view of the wave's current sub-tile, written by lanes 0..31 (lane l <-> term l):
  alignas(16) uint32_t w_incl[kScanWaves][kMaxTerms];     // 8-posting pairs of terms 0..l (inclusive prefix)
  alignas(16) uint32_t w_rec[kScanWaves][kMaxTerms][4];   // addr_d lo, addr_d hi, delta16, meta
  uint32_t w_before[kScanWaves][kMaxTerms];               // pairs of terms 0..l-1
  TopkScratch sc;
  uint64_t theta;        // packed key of the k-th best hit seen so far (0 = none)
  uint64_t thr;          // acc_threshold(theta): what the walk compares accumulators with
  uint32_t cnt;          // valid entries in cand
  uint32_t tile_cand;    // rendezvous: competitive hits still parked in the accumulators
  uint32_t slot_hits[kSliceSlots];   // live matching docs of this item, per searcher slice it touches (plan.h: DPart.slice)
  uint32_t slot_slice[kSliceSlots];  // which slice a slot stands for
  uint32_t rz_flag;      // a wave could not reserve candidate slots: everybody meet at the rendezvous
  uint32_t cnt_valid;    // entries of cand that are complete when cnt ran past kCandCap
  uint32_t next_tile;    // next unassigned sub-tile of the item (flattened over its parts)
  uint64_t prof[16];     // instrumented variant only (ABL == 7)
  uint64_t dummy[64];    // per-lane sink for invalid postings; always holds the marker (see group_prepare)
};
static_assert(sizeof(ScanSmem) <= 160 * 1024, "the scan workgroup owns one CU's 160 KiB LDS");

constexpr int kSlots = kTileDocs / 64;  // accumulator slots per lane (dense sweep)

// Per-term view of one sub-tile, built by lane l for term l and published in the wave's LDS table.
// The term's postings of the sub-tile are seen through a window of 16-byte groups (4 postings each):
//   meta: end (bits 0-21): postings [first, end) of the window belong to the sub-tile | first (22-23) |
//         score table (24-26, 7 = none) | coarse cell: postings may lie outside the sub-tile (27) |
//         fixed-point shift of the term (28-31)
// A lane processes a PAIR of consecutive groups (8 postings) per instruction.
constexpr uint32_t kMetaEnd = 0x3FFFFFu;

// Lane-as-term: posting range [lo, hi) of my term in a sub-tile -> table entry; returns the number of
// pair-instruction lanes of the whole sub-tile (wave-uniform).
__device__ __forceinline__ uint32_t subtile_build(ScanSmem& s, uint32_t wave, uint32_t lane, uint64_t my_docids,
                                                  uint32_t my_delta16, uint64_t my_lo, uint32_t my_flags, uint32_t lo,
                                                  uint32_t hi, bool use, uint32_t (&pre)[8]) {
  uint32_t end = 0, first = 0;
  uint64_t gs = my_lo >> 2;  // empty range: a group that is always safe to (pre)load
  if (use && hi > lo) {
    const uint64_t a = my_lo + lo;
    gs = a >> 2;
    first = (uint32_t)(a & 3u);
    end = first + (hi - lo);
  }
  const uint64_t addr_d = my_docids + gs * 16u;
  const uint32_t np = (end + 7u) >> 3;  // pairs
  const uint32_t incl = scan32_dpp(np);
  if (lane < (uint32_t)kMaxTerms) {
    s.w_incl[wave][lane] = incl;
    s.w_before[wave][lane] = incl - np;
    // my_delta16 == 0xFFFFFFFF: packed postings -- the slot holds the window's first group as a posting index instead
    u32x4 rec = {(uint32_t)addr_d, (uint32_t)(addr_d >> 32), my_delta16 == 0xFFFFFFFFu ? (uint32_t)(gs << 2) : my_delta16,
                 end | (first << 22) | (my_flags << 24)};
    *(u32x4*)&s.w_rec[wave][lane][0] = rec;
  }
  // the prefixes of the first 8 terms as wave-uniform scalars: locating a pair needs no LDS round trip
#pragma unroll
  for (int i = 0; i < 8; ++i) pre[i] = (uint32_t)__builtin_amdgcn_readlane((int)incl, i);
  return (uint32_t)__builtin_amdgcn_readlane((int)incl, kMaxTerms - 1);
}

// One pair of 4-posting groups: docids, score codes, and which (term, pair-in-term) it is.
struct Group {
  u32x4 d4[2], c4[2];
  uint32_t p;     // pair index inside the term's window of this sub-tile
  uint32_t meta;  // the term's meta
  uint32_t term;  // term index inside the part (division path only)
  uint32_t pidx;  // packed postings: index of the pair's first posting in its column (exception lookups)
};

// Locate flattened pair v of the sub-tile (clamped so the loads are always legal) through the wave's
// LDS table and load its column words (2 x 16 B per column).  No control flow around the loads.
template <bool PACKED>
__device__ __forceinline__ void group_locate_load(const ScanSmem& s, uint32_t wave, uint32_t n_terms, uint32_t v,
                                                  uint32_t total, const uint32_t (&pre)[8], Group& gr) {
  const uint32_t vc = min(v, max(total, 1u) - 1u);
  uint32_t t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += (vc >= pre[i]) ? 1u : 0u;  // terms 0..7, branchless, scalar operands
  if (n_terms > 8) {  // uniform; long disjunctions continue with a scalar-style walk
    while (t < n_terms - 1u && vc >= s.w_incl[wave][t]) ++t;
  }
  t = min(t, n_terms - 1u);  // total == 0: every prefix compares true
  const u32x4 rec = *(const u32x4*)&s.w_rec[wave][t][0];
  const uint32_t p = (total != 0u) ? vc - s.w_before[wave][t] : 0u;
  const uint64_t ad = (((uint64_t)rec[1] << 32) | rec[0]) + (uint64_t)p * 32u;
  const uint64_t ac = ad + (uint64_t)rec[2] * 16u;
  gr.p = p;
  gr.meta = rec[3];
  gr.term = t;
  gr.pidx = PACKED ? rec[2] + p * 8u : 0u;  // (packed: rec[2] is the window's first group as a posting index, not the column distance)
  gr.d4[0] = __builtin_nontemporal_load((gvec_ptr)ad);
  gr.d4[1] = __builtin_nontemporal_load((gvec_ptr)ad + 1);
  if (!PACKED) {  // (packed postings: the 8 words of the pair ARE d4 -- doc offset and code in one)
    gr.c4[0] = __builtin_nontemporal_load((gvec_ptr)ac);
    gr.c4[1] = __builtin_nontemporal_load((gvec_ptr)ac + 1);
  }
}

// Scoring a pair of groups happens in two steps so that the registers holding the loaded column
// words can be recycled for the next sub-tile's loads in between:
//   group_prepare:    per posting the LDS address of its doc's accumulator (off) and the value it adds
//                     (val: one LDS table read; division only for postings / terms no table serves)
//   group_commit_add: one LDS atomic per posting.
// Invalid postings of a pair (outside the term's window / the sub-tile) are redirected instead of
// predicated: they add a neutral value (-0.0 / 0) to the lane's dummy slot, which therefore holds the
// "unmatched" marker forever, so neither the adds nor the collecting swaps need per-posting control
// flow (a conditionally executed returning LDS op makes the compiler wait for each result at the end
// of its branch) and a dummy never looks like a matched doc.
template <bool FX, int ABL, bool PACKED>
__device__ __forceinline__ void group_prepare(const ScanSmem& s, const Group& gr, bool valid, uint32_t acc_addr, uint32_t base,
                                              uint32_t tile_len, uint32_t dummy_addr, const DTerm* __restrict__ part_terms,
                                              uint32_t (&off)[8], uint32_t (&val)[8]) {
  const uint32_t meta = gr.meta;
  // valid postings of my pair: window positions [first, end) intersected with [8p, 8p + 8)
  const int idx0 = (int)(gr.p * 8u);
  const int lo_cut = (int)((meta >> 22) & 3u) - idx0, hi_cut = (int)(meta & kMetaEnd) - idx0;
  const uint32_t hm = (1u << (uint32_t)min(max(hi_cut, 0), 8)) - 1u;
  const uint32_t lm = (1u << (uint32_t)min(max(lo_cut, 0), 8)) - 1u;
  uint32_t vmask = valid ? (hm & ~lm) : 0u;
  const uint32_t tab = (meta >> 24) & 7u;
  // LDS address of the doc's accumulator in the wave's sub-tile: acc + (doc - base) * 8 in one op
  // (packed postings: the word's upper 20 bits are the doc's offset inside the sub-tile's 2^20-doc super-window)
  const uint32_t abase = acc_addr - (PACKED ? (base & kPackDocMask) : base) * 8u;
  uint32_t cw[8];  // the postings' score codes as table byte offsets; sign bit: escape
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t w = gr.d4[j >> 2][j & 3];
    off[j] = ((PACKED ? (w >> kPackCodeBits) : w) << 3) + abase;
    cw[j] = PACKED ? (w & kPackCodeMask) << 2 : gr.c4[j >> 2][j & 3];
  }
  if (__any((meta >> 27) & 1u)) {
    // sparse terms share one posting range between several sub-tiles (coarse cells): doc-range filter
    // (unsigned: docs below the sub-tile wrap to huge values)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (off[j] - acc_addr >= tile_len * 8u) vmask &= ~(1u << j);
  }
  // postings the score table cannot serve: freq > kTabMaxFreq / norm >= kTabNorms (sign bit of the
  // code) or a term without a table.  Rare: one OR-reduction decides whether anybody in the wave
  // needs the division at all.
  bool any_esc;
  if (PACKED) {  // escape codes are the ones past the table's last entry
    const uint32_t mx = max(max(max(cw[0], cw[1]), max(cw[2], cw[3])), max(max(cw[4], cw[5]), max(cw[6], cw[7])));
    any_esc = mx >= (kPackEscBase << 2);
  } else {
    uint32_t cor = cw[0] | cw[1] | cw[2];
    cor |= cw[3] | cw[4];
    cor |= cw[5] | cw[6];
    cor |= cw[7];
    any_esc = (cor >> 31) != 0u;
  }
  const bool special = vmask != 0u && (any_esc || tab == 7u);
  const char* tb = (const char*)&s.tab[0][0] + (tab == 7u ? 0u : tab) * (uint32_t)(kTabEntries * 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (ABL == 1) val[j] = (cw[j] & 0x1FFCu) | 0x3F000000u;  // timing ablation: no table read
    else val[j] = *(const uint32_t*)(tb + (cw[j] & 0x1FFCu));  // masked: idle lanes stay in LDS
  }
  if (__any(special)) {
    // long docs / high freqs / terms without a score table (a few lanes)
    const float w = part_terms[gr.term].weight;
    const int fx_scale = part_terms[gr.term].fx_scale;
    const float* cache = &s.cache[part_terms[gr.term].cache_slot][0];
    const gu32_ptr esc_list = (gu32_ptr)part_terms[gr.term].fnorm;  // packed postings: the group's exception list
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t c = cw[j];
      if (PACKED && c >= (kPackEscBase << 2)) c = ((vmask >> j) & 1u) ? packed_escape_word(esc_list, gr.pidx + (uint32_t)j, c >> 2) : 0x80000100u;
      const bool esc = (c >> 31) != 0u;
      const uint32_t f = esc ? ((c >> 8) & 0x3FFFFFu) : ((c >> 9) & 15u);
      const bool dead = esc ? ((c >> 30) & 1u) != 0u : (c >> 20) != 0u;  // posting of a deleted doc (apply_live_kernel)
      const uint32_t nb = esc ? (c & 255u) : ((c >> 2) & 127u);
      if (((vmask >> j) & 1u) && (esc || tab == 7u))
        val[j] = dead ? (FX ? 0u : 0x80000000u) : score_value<FX>(bm25_score(w, (float)(int32_t)f, cache[nb]), fx_scale);
    }
  }
  if (!__all(vmask == 0xFFu)) {  // wave-uniform
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // three instructions per posting: bit-field extract (all ones: valid), bit-field insert, and / insert
      const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe((int)vmask, j, 1);
      off[j] = bfi(t, off[j], dummy_addr);
      val[j] = FX ? (val[j] & t) : bfi(t, val[j], 0x80000000u);  // neutral element: 0 / -0.0f
    }
  }
}

// One LDS atomic per posting (invalid ones were redirected by group_prepare): ds_add_f64 of the fp32
// score widened to double, or ds_add_u64 of the term's fixed-point integer shifted into the query's scale.
// cnt_hi != 0 (minimumNumberShouldMatch variant, FX only): every valid posting (its entry is a positive
// integer; redirected ones add 0) also adds one to the clause count kept above the score sum.
// use_max (query-shapes variant, FX only; wave-uniform): DisjunctionMaxQuery with tie breaker 0 -- the doc keeps its best
// clause's entry instead of the sum (ds_max_u64; the unmatched marker 0 and the redirected zeros behave as under add).
template <bool FX>
__device__ __forceinline__ void group_commit_add(const uint32_t (&off)[8], const uint32_t (&val)[8], uint32_t fx_shift,
                                                 uint32_t cnt_hi = 0, bool use_max = false) {
  const uint32_t fx_mult = 1u << fx_shift;  // entry << shift as one 32 x 32 -> 64 multiply (no 64-bit operand to set up)
  if (FX && use_max) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicMax((unsigned long long*)lds_ptr(off[j]), (unsigned long long)val[j] * (unsigned long long)fx_mult);
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (FX) {
      const unsigned long long c = (unsigned long long)(val[j] != 0u ? cnt_hi : 0u) << 32;  // the multiply's addend
      atomicAdd((unsigned long long*)lds_ptr(off[j]), (unsigned long long)val[j] * (unsigned long long)fx_mult + c);  // v_mad_u64_u32
    } else {
      unsafeAtomicAdd((double*)lds_ptr(off[j]), (double)__uint_as_float(val[j]));
    }
  }
}

// Quantile exchange between the items of one query (its slices on this GPU).  max(k-th best of each
// item) -- what LazyMaxScoreAccumulator shares -- is a weak bound when a query is cut m ways: every
// item converges on its own 1/m of the docs.  Instead each item publishes a score that at least
// ceil(k / m) of ITS docs reach; once all m have published, at least k docs of the query reach the
// smallest of them, so nothing below it can enter the merged top-k.  With three or more items the
// quantile is ceil(k / (m - 1)) instead and an item bounds itself by the OTHER items (`skip`): it needs
// no compaction of its own first.  One wave: returns that bound as a theta key (low word 0: a doc
// scoring exactly the bound still passes), or 0 while a peer is silent.
__device__ __forceinline__ uint64_t peers_bound(const unsigned long long* peers, uint32_t n_peers, uint32_t lane, uint32_t skip) {
  uint32_t inv = 0;  // max over peers of ~score word; an unpublished peer (0) saturates it
  for (uint32_t j = lane; j < n_peers; j += 64u)
    if (j != skip) inv = max(inv, ~(uint32_t)(__hip_atomic_load(peers + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32));
  inv = (uint32_t)__builtin_amdgcn_readlane((int)wave_max_u32(inv), 63);
  return inv == 0xFFFFFFFFu ? 0ull : (uint64_t)(~inv) << 32;
}

// Workgroup-wide rendezvous body: keep the k best of (candidate buffer UNION the candidates a wave
// still has parked in its sub-tile), publish theta.  Every thread calls it.  A wave whose reservation
// failed (`parked`) has reset every non-competitive slot of its sub-tile, so its parked candidates are
// exactly the slots still matched; other waves have none.  Contains barriers; returns with the buffer
// consistent and nothing parked.
template <bool FX>
__device__ __forceinline__ void rendezvous_compact(ScanSmem& s, uint64_t* acc, bool parked, uint32_t gdoc0, uint32_t k,
                                                   int fx_E, unsigned long long* theta_g, unsigned long long* peers,
                                                   uint32_t n_peers, uint32_t my_peer, const DExchange* xch, uint32_t query,
                                                   bool prof) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  // how many slices must together cover k docs: with an exchange this query's items here times the
  // world - 1 OTHER ranks whose entries bound a rank; without, the OTHER items (all of them if only two)
  const bool others_only = !xch && n_peers >= 3u;
  const uint32_t n_shares = xch ? n_peers * (xch->world - 1u) : (others_only ? n_peers - 1u : n_peers);
  uint64_t t0 = 0, t1 = 0, t2 = 0;
  if (prof) t0 = __builtin_readcyclecounter();
  uint32_t cmask = 0;
  if (parked) {  // wave-uniform
#pragma unroll
    for (int j = 0; j < kSlots; ++j) cmask |= (uint32_t)(acc[lane + 64u * (uint32_t)j] != acc_marker<FX>()) << j;
  }
  const uint32_t ncand = (uint32_t)__builtin_amdgcn_readlane((int)scan64_dpp((uint32_t)__popc(cmask)), 63);
  if (lane == 0 && ncand) atomicAdd(&s.tile_cand, ncand);
  const uint32_t cnt_raw = s.cnt;
  const uint32_t cnt0 = cnt_raw > (uint32_t)kCandCap ? s.cnt_valid : cnt_raw;  // failed reservations inflate cnt
  __syncthreads();  // every thread holds the same cnt0; tile_cand complete
  const uint32_t tc = s.tile_cand;
  uint64_t thr = 0;
  if (cnt0 + tc > k) {  // uniform
    thr = topk_kth_union<kScanThreads>(s.cand, cnt0, k, &s.sc, [&](auto&& f) {
      uint32_t m = cmask;
      while (m) {
        const int j = __ffs((int)m) - 1;
        m &= m - 1u;
        f(pack_key(acc_score<FX>(acc[lane + 64u * (uint32_t)j], fx_E), gdoc0 + lane + 64u * (uint32_t)j));
      }
    }, n_shares > 1u ? (k + n_shares - 1u) / n_shares : 0u);
    const uint32_t q2_hi = s.sc.q2_hi;  // (stable until the next selection)
    if (prof) t1 = __builtin_readcyclecounter();
    const uint32_t kept = topk_keep_ge<kScanThreads, kCandCap>(s.cand, cnt0, thr, &s.sc);
    if (prof) t2 = __builtin_readcyclecounter();
    if (tid == 0) {
      s.cnt = kept;
      if (thr > s.theta) {
        s.theta = thr;
        s.thr = acc_threshold<FX>(thr, fx_E);
      }
      atomicMax(theta_g, (unsigned long long)thr);  // LazyMaxScoreAccumulator.accumulate analogue
      if (n_shares > 1u && q2_hi != 0u) atomicMax(peers + my_peer, ((unsigned long long)q2_hi << 32) | 1ull);
      s.prof[6] += 1;
    }
    if ((n_shares > 1u || xch) && tid < 64u) {  // wave 0 (uniform inside it): what the search's slices know together
      // (one share = an unsplit query on two GPUs: the share is the item's own k-th best -- k of this rank's docs reach it)
      uint64_t pb = n_shares > 1u ? peers_bound(peers, n_peers, lane, others_only ? my_peer : ~0u)  // this GPU's items
                                  : (thr & 0xFFFFFFFF00000000ull);
      if (xch) pb = exchange_bound(*xch, query, pb, lane);  // publish it, bound myself by the other GPUs' entries
      if (tid == 0 && pb > s.theta) {
        s.theta = pb;
        s.thr = acc_threshold<FX>(pb, fx_E);
      }
      // The bound holds for the whole query (at least k of its docs reach it), so the query's other items get it
      // right away through theta_g -- every wave reads that once per sub-tile -- instead of at their own next
      // compaction, which comes the later the better their bound already is.
      if (tid == 0 && pb != 0ull) atomicMax(theta_g, (unsigned long long)pb);
    }
    __syncthreads();
  } else {
    if (tid == 0) s.cnt = cnt0;
    __syncthreads();
  }
#pragma unroll 1
  for (int j = 0; j < kSlots; ++j) {  // parked candidates that made the cut (fits: <= k in total)
    bool want = (cmask >> j) & 1u;
    uint64_t key = 0;
    if (want) {
      key = pack_key(acc_score<FX>(acc[lane + 64u * (uint32_t)j], fx_E), gdoc0 + lane + 64u * (uint32_t)j);
      acc[lane + 64u * (uint32_t)j] = acc_marker<FX>();
      want = key >= thr;
    }
    topk_append(s.cand, &s.cnt, want, key);
  }
  __syncthreads();
  if (tid == 0) {
    s.tile_cand = 0;
    s.rz_flag = 0;
    if (prof) {  // instrumented runs: phases of the rendezvous replace three event counters
      s.prof[8] += t1 - t0;                                // parked scan + k-th selection
      s.prof[10] += t2 - t1;                               // keep the survivors
      s.prof[11] += __builtin_readcyclecounter() - t2;     // publish, append parked, reset
    }
  }
  __syncthreads();
}

// Out-of-line entry: the compaction needs many registers of its own; as a call it costs the walk
// nothing (live values are saved around the call, a few times per item) instead of inflating the
// register demand of the whole kernel.  The LDS pointer keeps its address space across the call.
typedef __attribute__((address_space(3))) ScanSmem* lds_smem_ptr;
__device__ __noinline__ void rendezvous_call(lds_smem_ptr sp, uint32_t wave, bool parked, uint32_t gdoc0, uint32_t k,
                                             bool fixed, int fx_E, unsigned long long* theta_g, unsigned long long* peers,
                                             uint32_t n_peers, uint32_t my_peer, const DExchange* xch, uint32_t query, bool prof) {
  ScanSmem& s = *(ScanSmem*)sp;
  if (fixed) rendezvous_compact<true>(s, &s.acc[wave][0], parked, gdoc0, k, fx_E, theta_g, peers, n_peers, my_peer, xch, query, prof);
  else rendezvous_compact<false>(s, &s.acc[wave][0], parked, gdoc0, k, fx_E, theta_g, peers, n_peers, my_peer, xch, query, prof);
}

// Reserve room for the wave's `mine`-per-lane candidates in the shared buffer: one DPP scan and ONE
// LDS atomic.  Returns the lane's first slot, or kCandCap when the wave's candidates do not fit
// (rz_flag raised: the caller leaves them parked in its sub-tile and the whole workgroup meets at
// the rendezvous).  cnt only grows between rendezvous, so exactly the first reservation that crosses
// the end has base <= kCandCap: everything below its base is completely written -> cnt_valid.
__device__ __forceinline__ uint32_t reserve_candidates(ScanSmem& s, uint32_t lane, uint32_t mine) {
  const uint32_t incl = scan64_dpp(mine);
  const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  uint32_t wbase = 0;
  if (lane == 0) {
    wbase = atomicAdd(&s.cnt, wave_total);
    if (wbase + wave_total > (uint32_t)kCandCap) {
      if (wbase <= (uint32_t)kCandCap) s.cnt_valid = wbase;
      __hip_atomic_store(&s.rz_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
  if (wbase + wave_total > (uint32_t)kCandCap) return (uint32_t)kCandCap;
  return wbase + incl - mine;
}

// Write `value` into the slots of my sub-tile whose docs are outside the mask, without spending vector
// registers or vector-memory waits on it (the walk has neither to spare: a vector load here would make
// the wave wait for its prefetched postings as well).  Slot lane + 64 j belongs to doc 64 j + lane of the
// sub-tile, i.e. to bit `lane` of the mask's 64-bit word j: the complement of that word IS the execution
// mask of the store.  The 16 words arrive by scalar loads (two of 64 bytes); per word: three scalar
// instructions and one LDS store.  `mask_tile` points at the sub-tile's 128 mask bytes (uniform; the
// arrays are padded, segment.cpp, so the last sub-tile of a segment may read past max_doc).
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
template <int OFF>
__device__ __forceinline__ void store_where_clear(uint64_t word, uint32_t addr, uint64_t value) {
  uint64_t saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_andn2_b64 exec, exec, %1\n\tds_write_b64 %2, %3 offset:%4\n\ts_mov_b64 exec, %0"
               : "=&s"(saved) : "s"(word), "v"(addr), "v"(value), "n"(OFF) : "memory", "scc");
}
template <int CHUNK>
__device__ __forceinline__ void mark_outside_half(const void* mask_tile, uint32_t addr, uint64_t value) {
  u32x16 w;
  asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(mask_tile), "n"(CHUNK * 64) : "memory");
#define NRT_WORD(i) (((uint64_t)w[2 * (i) + 1] << 32) | (uint64_t)w[2 * (i)])
  store_where_clear<(CHUNK * 8 + 0) * 512>(NRT_WORD(0), addr, value);
  store_where_clear<(CHUNK * 8 + 1) * 512>(NRT_WORD(1), addr, value);
  store_where_clear<(CHUNK * 8 + 2) * 512>(NRT_WORD(2), addr, value);
  store_where_clear<(CHUNK * 8 + 3) * 512>(NRT_WORD(3), addr, value);
  store_where_clear<(CHUNK * 8 + 4) * 512>(NRT_WORD(4), addr, value);
  store_where_clear<(CHUNK * 8 + 5) * 512>(NRT_WORD(5), addr, value);
  store_where_clear<(CHUNK * 8 + 6) * 512>(NRT_WORD(6), addr, value);
  store_where_clear<(CHUNK * 8 + 7) * 512>(NRT_WORD(7), addr, value);
#undef NRT_WORD
}
__device__ __forceinline__ void mark_outside(const void* mask_tile, uint32_t acc_addr, uint32_t lane, uint64_t value) {
  const uint32_t addr = acc_addr + lane * 8u;
  // the stored value as an opaque register pair: as a plain constant the compiler merges it with the walk's
  // other uses of the marker, keeps that pair alive through the whole loop, spills it and reloads it
  // from scratch (with a full vector-memory wait) at every use
  uint32_t lo, hi;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(lo), "=v"(hi) : "s"((uint32_t)value), "s"((uint32_t)(value >> 32)));
  value = ((uint64_t)hi << 32) | (uint64_t)lo;
  mark_outside_half<0>(mask_tile, addr, value);
  mark_outside_half<1>(mask_tile, addr, value);  // (sub-tiles of 1024 docs: 16 words; other shapes do not use the masked variant)
}

// Sparse collect of one pair's swapped-out slot values a[j] (the marker where this posting is not its
// doc's collector): count the hits, send the competitive docs to the shared candidate buffer.
// Returns true when they did not fit and were parked back into the sub-tile.
// ABL == 8 (clause counting): a total is a hit when it reaches hit_floor (minimumNumberShouldMatch in the count
// bits; 1 for a query of the batch that does not count), and its score sum is what score_hi_mask leaves.
template <bool FX, int ABL>
__device__ __forceinline__ bool collect_swapped(ScanSmem& s, uint32_t acc_addr, uint32_t lane, uint64_t (&a)[8],
                                                const uint32_t (&off)[8], uint64_t thr, uint64_t theta, int fx_E,
                                                uint32_t gdoc0, uint32_t& wave_hits, uint64_t after_key, uint64_t hit_floor = 0,
                                                uint32_t score_hi_mask = 0xFFFFFFFFu) {
  unsigned long long any_maybe = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (ABL == 8) {
      const bool hit = a[j] >= hit_floor;
      wave_hits += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(hit));
      a[j] = hit ? (a[j] & (((uint64_t)score_hi_mask << 32) | 0xFFFFFFFFull)) : 0ull;  // from here on: bare sums, non-hits gone
    } else {
      wave_hits += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ABL == 9 ? acc_is_hit<FX>(a[j]) : a[j] != acc_marker<FX>()));
    }
    any_maybe |= __builtin_amdgcn_ballot_w64(acc_reaches<FX>(a[j], thr));
  }
  if (any_maybe == 0ull || ABL == 6) return false;  // wave-uniform; the steady state once theta has converged
  if (ABL == 7 && threadIdx.x == 0) s.prof[13] += 1;
  uint32_t cmask = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (acc_reaches<FX>(a[j], thr)) {
      const uint64_t key = pack_key(acc_score<FX>(a[j], fx_E), gdoc0 + ((off[j] - acc_addr) >> 3));
      if (key > theta && key < after_key) cmask |= 1u << j;  // (at or above after_key: already on an earlier page)
    }
  if (!__any(cmask != 0)) return false;
  if (ABL == 7 && threadIdx.x == 0) s.prof[12] += 1;
  uint32_t pos = reserve_candidates(s, lane, (uint32_t)__popc(cmask));
  if (pos < (uint32_t)kCandCap) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if ((cmask >> j) & 1u) s.cand[pos++] = pack_key(acc_score<FX>(a[j], fx_E), gdoc0 + ((off[j] - acc_addr) >> 3));
    return false;
  }
  // back into my sub-tile: the rendezvous that follows this iteration takes them from there
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if ((cmask >> j) & 1u) *(uint64_t*)lds_ptr(off[j]) = a[j];
  return true;
}

// FX: accumulator representation (see above).  PIPE = true: the first posting pair per lane of the
// wave's next sub-tile is loaded before the current one is collected.  ABL selects the variant:
// 0 = the scan; 8 = clause counting (minimumNumberShouldMatch > 1, fixed point only); 9 = doc-set masks
// (FILTER / MUST_NOT, liveDocs that are not folded into the postings); 7 = instrumented (event counters
// per item); 1-4, 6 = timing ablations (wrong results).
template <bool FX, bool PIPE, int ABL, bool PACKED>
__global__ __launch_bounds__(kScanThreads, kScanWaves / 4)
void bm25_scan_kernel(const DItem* __restrict__ items, const DPart* __restrict__ parts,
                      const DTerm* __restrict__ terms, const DQuery* __restrict__ queries,
                      const float* __restrict__ caches, unsigned long long* __restrict__ theta_g,
                      unsigned long long* __restrict__ quant_g, const DExchange* __restrict__ xch, uint32_t* __restrict__ slice_sum,
                      uint64_t* __restrict__ item_keys, uint32_t* __restrict__ item_counts,
                      uint64_t* __restrict__ item_hits, uint32_t k_stride, uint64_t* __restrict__ item_prof) {
  __shared__ ScanSmem s;
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  uint64_t* const acc = &s.acc[wave][0];
  const uint32_t acc_addr = lds_addr(acc), dummy_addr = lds_addr(&s.dummy[lane]);
  const DItem item = items[blockIdx.x];
  const DQuery q = queries[item.query];
  const uint32_t k = q.k;
  const int fx_E = item.fx_E;
  unsigned long long* const my_theta_g = theta_g + item.query;
  const bool multi_item = q.n_items > 1;  // uniform: only then is there anybody to share theta with
  // searchAfter (LazyQueueTopScoreDocCollector.java:112-120): a hit is skipped when score > afterScore or
  // (score == afterScore and doc <= afterDoc) -- in key order exactly "key >= key(afterScore, afterDoc)".  It is
  // still a hit (totalHits counts it), so paging only narrows the candidate test and needs no path of its own.
  const uint64_t after_key = q.has_after ? pack_key(q.after_score, (uint32_t)q.after_doc) : ~0ull;
  // ABL == 8 (fixed point only): the variant for batches with minimumNumberShouldMatch > 1 queries.  Such a
  // query's postings also count clauses in the accumulator's top bits; all its sub-tiles take the general
  // sweep, whose exact path drops docs with too few clauses and strips the count.
  constexpr bool kMsm = FX && ABL == 8;
  const uint32_t msm = kMsm ? q.min_should_match : 0u;
  const uint32_t cnt_hi = (kMsm && msm > 1u) ? (1u << (kMsmCountShift - 32)) : 0u;
  const uint64_t hit_floor = cnt_hi ? (uint64_t)msm << kMsmCountShift : 1ull;
  const uint32_t score_hi_mask = cnt_hi ? (1u << (kMsmCountShift - 32)) - 1u : 0xFFFFFFFFu;
  const bool use_max = kMsm && q.combine_max != 0u;  // DisjunctionMaxQuery: best clause instead of the sum

  uint64_t t_start = 0, t_walk = 0;
  if (ABL == 7) t_start = __builtin_readcyclecounter();
  // ---- item prologue: clear the sub-tiles, stage the normInverse tables, build the score tables
  for (int j = 0; j < kSlots; ++j) acc[lane + 64u * (uint32_t)j] = acc_marker<FX>();
  {
    const uint32_t n_lds = min(item.n_caches, (uint32_t)kLdsCaches) * 256u;
    for (uint32_t i = tid; i < n_lds; i += kScanThreads) (&s.cache[0][0])[i] = caches[item.cache_off + i];
  }
  if (tid == 0) {
    // what is known about the query's k-th best hit before this item starts: the caller's
    // min_competitive_score and whatever the query's other items have published since
    const uint64_t theta0 = __hip_atomic_load(my_theta_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s.theta = theta0;
    s.thr = acc_threshold<FX>(theta0, fx_E);
    s.cnt = 0;
    s.tile_cand = 0;
    for (int i = 0; i < kSliceSlots; ++i) s.slot_hits[i] = s.slot_slice[i] = 0u;
    s.rz_flag = 0;
    s.cnt_valid = 0;
    s.next_tile = 3u * (uint32_t)kScanWaves;  // every wave starts with three sub-tiles
    for (int i = 0; i < 16; ++i) s.prof[i] = 0;
    s.prof[15] = ~0ull;
  }
  if (tid < 64) s.dummy[tid] = acc_marker<FX>();
  if ((multi_item || xch) && tid < 64u) {  // wave 0: what the search's other slices have published together so far
    uint64_t pb = peers_bound(quant_g + q.item_begin, q.n_items, lane, (!xch && q.n_items >= 3u) ? item.peer_slot - q.item_begin : ~0u);
    if (xch) pb = exchange_bound(*xch, item.query, pb, lane);
    if (tid == 0 && pb > s.theta) {
      s.theta = pb;
      s.thr = acc_threshold<FX>(pb, fx_E);
    }
  }
  __syncthreads();
  for (uint32_t slot = 0; slot < item.n_tabs; ++slot) {
    const float w = items[blockIdx.x].tab_weight[slot];  // (indexing the register copy would spill it)
    const int scale = items[blockIdx.x].tab_scale[slot];
    const float* cache = &s.cache[items[blockIdx.x].tab_cache[slot]][0];
    // row 0 (freq 0) serves the postings of deleted docs (apply_live_kernel): the neutral element of the sum
    for (uint32_t e = tid; e < (uint32_t)kTabEntries; e += kScanThreads)
      s.tab[slot][e] = e < (uint32_t)kTabNorms ? (FX ? 0u : 0x80000000u)
                                               : score_value<FX>(bm25_score(w, (float)(int32_t)(e >> 7), cache[e & 127u]), scale);
  }
  __syncthreads();  // tables complete; from here on the waves run on their own
  if (ABL == 7) {
    t_walk = __builtin_readcyclecounter();
    if (tid == 0) s.prof[0] += t_walk - t_start;
  }
  uint32_t my_hits = 0;    // per-lane count (general sweep)
  uint32_t wave_hits = 0;  // wave-uniform count, added by lane 0 when the wave moves on to a part of another slice (and at the end)
  uint32_t cur_slot = 0;   // the slot (searcher slice) of the part the wave is in
  // The sub-tiles of the item's parts form one sequence g = 0, 1, ... (part.tile_offset + tile index in the
  // part).  Waves take them DYNAMICALLY from a shared counter: the hardware favours the oldest waves,
  // with a static split the youngest ones finish ~20% later while the others idle.  A wave always holds
  // three indices: g_cur (table built, first pairs loaded), g_nxt (cells loaded), g_nxt2 (just taken).
  // A rendezvous interrupts the walk: the wave leaves its loops (so that the compaction code is not
  // inside them, holding every loop register live) and afterwards resumes at g_cur by re-running the
  // part prologue -- the same path as entering a new part.
  uint32_t g_cur = wave, g_nxt = wave + (uint32_t)kScanWaves, g_nxt2 = wave + 2u * (uint32_t)kScanWaves;
  uint32_t pi = 0;
  bool parked = false;     // wave-uniform: my candidates did not fit the shared buffer (they wait in my sub-tile)
  uint32_t gdoc0 = 0;      // global docid of slot 0 of my current sub-tile

  for (;;) {  // epochs between rendezvous
  bool interrupted = false;
  for (;;) {  // parts
    DPart part;
    for (;; ++pi) {  // the part that holds g_cur (indices only grow)
      if (pi >= item.n_parts) break;
      part = parts[item.part_begin + pi];
      if (g_cur < part.tile_offset + (part.tile_end - part.tile_begin)) break;
    }
    if (pi >= item.n_parts) break;  // this wave is out of sub-tiles
    {  // hits are counted per searcher slice: a part of another slice closes my count of the previous one
      const uint32_t p_slot = part.slice >> 24;
      if (p_slot != cur_slot) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) my_hits += __shfl_xor(my_hits, d, 64);
        if (lane == 0 && (my_hits + wave_hits)) atomicAdd(&s.slot_hits[cur_slot], my_hits + wave_hits);
        my_hits = wave_hits = 0;
        cur_slot = p_slot;
      }
      if (lane == 0) s.slot_slice[p_slot] = part.slice & 0xFFFFFFu;
    }
    const uint32_t g0 = part.tile_offset, gn = g0 + (part.tile_end - part.tile_begin);
    // tile index inside the segment of a flattened index of this part: g - g0 + tile_begin
    const uint32_t tile_bias = part.tile_begin - g0;
    const uint32_t n_terms = part.n_terms;
    const DTerm* const part_terms = terms + part.term_begin;
    const NRT_GLOBAL uint64_t* const live_bits = (const NRT_GLOBAL uint64_t*)part.live_bits;
    // uniform: no searchAfter, no clause counting, and no doc-set mask -- unless this is the masked variant
    // (ABL == 9), which poisons the slots outside the mask instead of checking every matched doc
    constexpr bool kMask = ABL == 9 && kTileDocs == 1024;
    const bool simple = live_bits == nullptr || kMask;  // (searchAfter only filters candidates: after_key)
    const bool masked = kMask && live_bits != nullptr && simple;
    auto mask_tile_of = [&](uint32_t g) -> const void* {  // the 128 mask bytes of sub-tile g, as a uniform address
      const uint64_t a = (uint64_t)part.live_bits + (uint64_t)(g + tile_bias) * (uint64_t)(kTileDocs / 8);
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
      return (const void*)(((uint64_t)hi << 32) | (uint64_t)lo);
    };
    if (masked)  // (entering a part, or resuming after a rendezvous: the sub-tile holds markers only)
      mark_outside(mask_tile_of(g_cur), acc_addr, lane, acc_dead<FX>());
    // lane l looks after term min(l, n_terms - 1) of this part (registers)
    const DTerm mt = part_terms[min(lane, n_terms - 1u)];
    const uint64_t my_docids = (uint64_t)mt.docids, my_lo = mt.start;
    const uint32_t my_delta16 = PACKED ? 0xFFFFFFFFu : (uint32_t)(((uint64_t)mt.fnorm - (uint64_t)mt.docids) >> 4);  // same allocation
    const gu32_ptr my_cells = (gu32_ptr)mt.cell_off;
    const uint32_t my_shift = mt.shift;
    const uint32_t my_flags = ((mt.tab_slot & 0xFFFFu) < (uint32_t)kTabTerms ? (mt.tab_slot & 0xFFFFu) : 7u) | ((mt.shift != 0 ? 1u : 0u) << 3) |
                              ((FX ? mt.fx_shift : 0u) << 4);
    const bool has_term = lane < n_terms;

    const uint32_t last_tile = part.tile_end - 1u;
    // software pipeline state: the wave's LDS table and `pre` describe sub-tile g_cur (total_groups
    // pairs), (nlo, nhi) are the cell values of g_nxt, pf holds the first 64 pairs of g_cur
    uint32_t nlo, nhi;
    uint32_t total_groups;
    uint32_t pre[8];  // wave-uniform: pair prefixes of terms 0..7 of the table's sub-tile
    {
      const uint32_t c0 = (g_cur + tile_bias) >> my_shift;
      total_groups = subtile_build(s, wave, lane, my_docids, my_delta16, my_lo, my_flags, my_cells[c0], my_cells[c0 + 1], has_term, pre);
      const uint32_t c1 = min(g_nxt + tile_bias, last_tile) >> my_shift;  // (clamped: g_nxt may belong to a later part)
      nlo = my_cells[c1];
      nhi = my_cells[c1 + 1];
    }
    Group pf;
    group_locate_load<PACKED>(s, wave, n_terms, lane, total_groups, pre, pf);
    // theta of the query's other items (LazyMaxScoreAccumulator analogue): read one sub-tile ahead of its
    // use -- it is only a filter, a stale value costs a few extra candidates, never a result
    uint64_t theta_other = 0, theta_other_next = 0;

    for (;;) {  // sub-tiles of this part
      const uint32_t base = (g_cur + tile_bias) * (uint32_t)kTileDocs;
      const uint32_t tile_len = min((uint32_t)kTileDocs, part.max_doc - base);
      gdoc0 = (uint32_t)(part.doc_base + (int32_t)base);
      if (ABL == 7 && tid == 0) s.prof[7] += 1;

      // ---- (1) score the postings: coalesced 32 B/lane column loads (already in flight), table-lookup BM25
      const uint32_t cur_groups = total_groups;
      // wave-uniform: sub-tiles of up to one (16 waves: registers) or two pair-instructions per lane are
      // collected through the postings instead of a sweep
      constexpr bool kTwoGroups = kScanWaves <= 12 && ABL != 9;  // (the masked variant has no registers for a second pair instruction)
      const bool sparse = simple && cur_groups <= (kTwoGroups ? 128u : 64u);
      const bool second = kTwoGroups && cur_groups > 64u;  // the sub-tile has a second instruction's worth of pairs
      const bool act = lane < cur_groups, act2 = second && 64u + lane < cur_groups;
      uint32_t off[8], off2[8];
      uint32_t val[8], val2[8];
      uint32_t sh = 0, sh2 = 0;
      if (cur_groups != 0) {
        if (sparse) {
          if (second) {
            Group a;
            group_locate_load<PACKED>(s, wave, n_terms, 64u + lane, cur_groups, pre, a);
            group_prepare<FX, ABL, PACKED>(s, a, 64u + lane < cur_groups, acc_addr, base, tile_len, dummy_addr, part_terms, off2, val2);
            sh2 = a.meta >> 28;
          }
        } else {
          for (uint32_t vb = 64u; vb < cur_groups; vb += 64u) {  // dense sub-tile: pairs beyond the first 64 (wave-uniform trip count)
            Group a;
            group_locate_load<PACKED>(s, wave, n_terms, vb + lane, cur_groups, pre, a);
            group_prepare<FX, ABL, PACKED>(s, a, vb + lane < cur_groups, acc_addr, base, tile_len, dummy_addr, part_terms, off2, val2);
            if (vb + lane < cur_groups) group_commit_add<FX>(off2, val2, a.meta >> 28, cnt_hi, use_max);
          }
        }
        group_prepare<FX, ABL, PACKED>(s, pf, act, acc_addr, base, tile_len, dummy_addr, part_terms, off, val);
        sh = pf.meta >> 28;
      }

      // ---- (2a) the next sub-tile's table (the column words in pf are consumed), the cells after it, theta
      total_groups = subtile_build(s, wave, lane, my_docids, my_delta16, my_lo, my_flags, nlo, nhi, has_term && g_nxt < gn, pre);
      {
        const uint32_t c2 = min(g_nxt2 + tile_bias, last_tile) >> my_shift;
        nlo = my_cells[c2];
        nhi = my_cells[c2 + 1];
      }
      uint32_t g_new = 0;  // take one more sub-tile; the counter's answer is needed at the end of this iteration
      if (lane == 0) g_new = atomicAdd(&s.next_tile, 1u);
      theta_other = theta_other_next;
      if (multi_item) theta_other_next = __hip_atomic_load(my_theta_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // ---- (2b) the next sub-tile's first posting pairs: requested here, in front of this sub-tile's LDS adds and swaps (through
      //      round 5 behind them: "in flight while this one is collected"; in front of them the loads also cover the sixteen LDS
      //      atomics per lane -- same registers, 10.30 -> 10.22 ms per 1024 C3 queries, profiles/r06_scan_early_load_ab.log)
      group_locate_load<PACKED>(s, wave, n_terms, lane, total_groups, pre, pf);
      if (!PIPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // A/B: no overlap of the column loads

      // ---- (3) LDS accumulate; sparse sub-tiles: then each posting swaps the "unmatched" marker into its
      //      doc's slot.  LDS executes a wave's operations in order, so the first posting of a doc to do
      //      so receives the doc's complete score and is its collector; the others (and invalid
      //      postings, on the dummy slot) receive the marker.  No sweep.  Idle lanes stay out of the
      //      LDS pipe; every add of the sub-tile precedes every swap.
      uint64_t a[8], a2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = a2[j] = acc_marker<FX>();
      if (cur_groups != 0) {
        if (act && ABL != 2 && ABL != 4) group_commit_add<FX>(off, val, sh, cnt_hi, use_max);
        if (sparse && ABL != 2 && ABL != 3) {
          if (act2) group_commit_add<FX>(off2, val2, sh2, cnt_hi, use_max);
          if (act) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = atomicExch((unsigned long long*)lds_ptr(off[j]), (unsigned long long)acc_marker<FX>());
          }
          if (act2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a2[j] = atomicExch((unsigned long long*)lds_ptr(off2[j]), (unsigned long long)acc_marker<FX>());
          }
        }
      }


      if (cur_groups != 0) {
        const uint64_t theta_l = s.theta;
        const uint64_t theta = theta_other > theta_l ? theta_other : theta_l;
        const uint64_t thr = s.thr;  // acc_threshold(theta_l), kept next to theta by the rendezvous
        if (sparse) {
          // ---- (4s) collect through the postings
          if (ABL == 7 && tid == 0) s.prof[9] += 1;
          parked = collect_swapped<FX, ABL>(s, acc_addr, lane, a, off, thr, theta, fx_E, gdoc0, wave_hits, after_key, hit_floor, score_hi_mask);
          if (second) parked |= collect_swapped<FX, ABL>(s, acc_addr, lane, a2, off2, thr, theta, fx_E, gdoc0, wave_hits, after_key, hit_floor, score_hi_mask);
        } else {
          // ---- (4d) dense sweep of my sub-tile: count hits, reset every slot that cannot be competitive.
          //      A slot whose fp32 score reaches theta's score stays in place (mmask) for the exact path.
          uint32_t mmask = 0;
          unsigned long long any_maybe = 0;  // wave-uniform
          if (simple) {
#pragma unroll
            for (int h = 0; h < kSlots / 4; ++h) {
              uint64_t v[4];
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) v[jj] = acc[lane + 64u * (uint32_t)(h * 4 + jj)];
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const int j = h * 4 + jj;
                if (kMsm) {  // clause counting: a hit needs its count; what stays in place is the bare sum
                  const bool touched = v[jj] != 0ull, hit = v[jj] >= hit_floor;
                  const uint64_t sum = v[jj] & (((uint64_t)score_hi_mask << 32) | 0xFFFFFFFFull);
                  wave_hits += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(hit));
                  const bool maybe = hit && acc_reaches<FX>(sum, thr);
                  any_maybe |= __builtin_amdgcn_ballot_w64(maybe);
                  if (touched) acc[lane + 64u * (uint32_t)j] = maybe ? sum : 0ull;
                  continue;
                }
                const bool matched = ABL == 9 ? acc_is_hit<FX>(v[jj]) : v[jj] != acc_marker<FX>();
                wave_hits += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(matched));
                const bool maybe = acc_reaches<FX>(v[jj], thr);  // implies matched
                any_maybe |= __builtin_amdgcn_ballot_w64(maybe);
                if (matched & !maybe) acc[lane + 64u * (uint32_t)j] = acc_marker<FX>();
              }
            }
          } else {
            any_maybe = ~0ull;  // deletes / searchAfter: every slot goes through the exact path
          }
          if (any_maybe != 0ull) {  // wave-uniform: which of my slots were left in place? (one batched re-read)
            if (ABL == 7 && tid == 0) s.prof[13] += 1;
            if (simple) {
              uint64_t v[kSlots];
#pragma unroll
              for (int j = 0; j < kSlots; ++j) v[j] = acc[lane + 64u * (uint32_t)j];
#pragma unroll
              for (int j = 0; j < kSlots; ++j) mmask |= (uint32_t)(ABL == 9 ? acc_is_hit<FX>(v[j]) : v[j] != acc_marker<FX>()) << j;
            } else {
              mmask = (1u << kSlots) - 1u;
            }
          }
          uint32_t cmask = 0;
          while (__any(mmask != 0)) {  // exact path: a few iterations once theta has converged
            if (mmask) {
              const int j = __ffs((int)mmask) - 1;
              mmask &= mmask - 1u;
              const uint32_t i = lane + 64u * (uint32_t)j;
              uint64_t v = acc[i];
              if (v != acc_marker<FX>()) {
                const uint32_t doc = base + i;
                bool live = true;
                if (!simple && live_bits) live = (live_bits[doc >> 6] >> (doc & 63u)) & 1ull;
                if (kMsm && cnt_hi != 0u && !simple) {  // (the simple sweep has done this already) a hit needs msm clauses; then v is the bare sum
                  live = live && (uint32_t)(v >> kMsmCountShift) >= msm;
                  v &= (1ull << kMsmCountShift) - 1ull;
                }
                bool cand = false;
                if (live) {
                  if (!simple) ++my_hits;  // totalHits counts every collected doc, also those skipped by `after`
                  const float sc = acc_score<FX>(v, fx_E);
                  const uint32_t gdoc = gdoc0 + i;
                  const uint64_t key = pack_key(sc, gdoc);
                  cand = key > theta && key < after_key;  // at or above after_key: collected on an earlier page
                }
                if (cand) {
                  cmask |= 1u << j;
                  if (kMsm && cnt_hi != 0u && !simple) acc[i] = v;  // stays in place for the candidate copy / the rendezvous: without the count
                } else {
                  acc[i] = acc_marker<FX>();
                }
              }
            }
          }
          if (__any(cmask != 0)) {
            if (ABL == 7 && tid == 0) s.prof[12] += 1;
            uint32_t pos = reserve_candidates(s, lane, (uint32_t)__popc(cmask));
            if (pos < (uint32_t)kCandCap) {
              while (cmask) {
                const int j = __ffs((int)cmask) - 1;
                cmask &= cmask - 1u;
                const uint32_t i = lane + 64u * (uint32_t)j;
                s.cand[pos++] = pack_key(acc_score<FX>(acc[i], fx_E), gdoc0 + i);
                acc[i] = acc_marker<FX>();
              }
            } else {
              parked = true;
            }
          }
        }
      }

      // ---- (5) advance.  Somebody's candidates did not fit (seen at the next sub-tile boundary at the
      //      latest): leave the walk for the rendezvous
      if (masked) mark_outside(mask_tile_of(g_cur), acc_addr, lane, acc_marker<FX>());  // un-poison: only parked candidates may stay
      g_cur = g_nxt;
      g_nxt = g_nxt2;
      g_nxt2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_new);
      if (__hip_atomic_load(&s.rz_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {  // wave-uniform
        interrupted = true;
        break;
      }
      if (g_cur >= gn) break;  // the next sub-tile lies in a later part (or past the item)
      if (masked) mark_outside(mask_tile_of(g_cur), acc_addr, lane, acc_dead<FX>());  // poison the next sub-tile: nothing of it has been added yet
    }
    if (interrupted) break;
  }

  // ---- rendezvous point: waves that are out of sub-tiles wait here for the others
  uint64_t t_r0 = 0;
  if (ABL == 7) {
    t_r0 = __builtin_readcyclecounter();
    if (!interrupted && lane == 0) {  // this wave is out of work: spread of the waves' finish times
      atomicMax((unsigned long long*)&s.prof[14], (unsigned long long)(t_r0 - t_walk));
      atomicMin((unsigned long long*)&s.prof[15], (unsigned long long)(t_r0 - t_walk));
    }
  }
  __syncthreads();  // R1: all waves -- interrupted ones and finished ones
  if (ABL == 7 && tid == 0) s.prof[1] += __builtin_readcyclecounter() - t_r0;  // wave 0 waiting for the others
  if (!__hip_atomic_load(&s.rz_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;  // nobody asked: everybody is finished
  rendezvous_call((lds_smem_ptr)&s, wave, parked, gdoc0, k, FX, fx_E, my_theta_g, quant_g + q.item_begin, q.n_items,
                  item.peer_slot - q.item_begin, xch, item.query, ABL == 7);  // ends with barriers: the flag is re-read safely
  parked = false;
  if (ABL == 7 && tid == 0) {
    s.prof[5] += 1;
    s.prof[2] += __builtin_readcyclecounter() - t_r0;  // rendezvous incl. the wait
  }
  }
  uint64_t t_epi = 0;
  if (ABL == 7) {
    t_epi = __builtin_readcyclecounter();
    if (tid == 0) s.prof[3] += t_epi - t_walk;  // walk + rendezvous
  }

  // ---- item epilogue: final top-k of the item, hit count
  {
    const uint32_t c = s.cnt;
    __syncthreads();
    if (c > k) {
      uint64_t thr = 0;
      const uint32_t m = topk_compact<kScanThreads, kCandCap>(s.cand, c, k, &s.sc, &thr);
      if (tid == 0) s.cnt = m;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) my_hits += __shfl_xor(my_hits, d, 64);
  if (lane == 0 && (my_hits + wave_hits)) atomicAdd(&s.slot_hits[cur_slot], my_hits + wave_hits);
  __syncthreads();
  const uint32_t n = s.cnt;
  uint64_t* out = item_keys + (size_t)blockIdx.x * k_stride;
  for (uint32_t i = tid; i < n; i += kScanThreads) out[i] = s.cand[i];
  if (tid == 0) {
    item_counts[blockIdx.x] = n;
    // the item's hits (exact: the scan skips nothing), per slice into the query's sums (slice_relation_kernel) and in total
    uint32_t hits = 0;
    for (int i = 0; i < kSliceSlots; ++i) {
      const uint32_t h = s.slot_hits[i];
      hits += h;
      if (h != 0u && q.gte_floor != 0xFFFFFFFFu) atomicAdd(&slice_sum[q.slice_base + s.slot_slice[i]], h);
    }
    item_hits[blockIdx.x] = hits;
    if (ABL == 7 && item_prof) {
      s.prof[4] += __builtin_readcyclecounter() - t_epi;  // epilogue
      for (int i = 0; i < 16; ++i) item_prof[(size_t)blockIdx.x * 16 + i] = s.prof[i];
    }
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
  // a block of the query's lists, one per thread: its record, where its keys begin among the block's (exclusive prefix of the
  // counts; [kScanThreads] = the block's total), the waves' totals of the scan
  uint32_t lrec[kScanThreads];
  uint32_t loff[kScanThreads + 1];
  uint32_t wtot[kScanThreads / 64];
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
                       uint32_t k_stride_out, const uint32_t* __restrict__ help_query, uint32_t n_help,
                       uint32_t help_slot_base, const unsigned long long* __restrict__ spec_g) {
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
  // The query's lists: its items' records, then the slots of the helpers that worked for it (maxscore.hip, plan.h: DHelp --
  // help_query[h] = the query of helper slot h, + 1; slot = help_slot_base + h).  Through round 5 the helpers' slots were a linked
  // list per query, walked one dependent load after the other, and every list cost the workgroup two barriers: with ONE query per
  // call and 250 helpers the merge took as long as the scorer (profiles/r06_single_query_timeline.txt).  Now a thread per list:
  // record and count at once, an exclusive scan of the counts, and the lists' keys as ONE sequence -- barriers per 768 keys.
  const uint32_t n_lists = nl + (help_query ? n_help : 0u);
  for (uint32_t l0 = 0; l0 < n_lists; l0 += kScanThreads) {
    const uint32_t l = l0 + tid;
    uint32_t rec = 0xFFFFFFFFu, c = 0;
    if (l < nl) {
      rec = list_idx[base + l];
    } else if (l < n_lists && help_query[l - nl] == q + 1u) {
      rec = help_slot_base + (l - nl);
    }
    if (rec != 0xFFFFFFFFu) {
      c = min(in_counts[rec], k_stride_in);
      const unsigned long long h = in_hits[rec];
      if (h) atomicAdd(&s.hits, h);
    }
    // exclusive scan of c over the workgroup: DPP scan per wave, the waves' totals through LDS
    const uint32_t incl = scan64_dpp(c);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    if (lane == 63u) s.wtot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < (uint32_t)(kScanThreads / 64); ++w) {
      const uint32_t t = s.wtot[w];
      before += w < wave ? t : 0u;
      total += t;
    }
    s.lrec[tid] = rec;
    s.loff[tid] = before + incl - c;
    if (tid == 0) s.loff[kScanThreads] = total;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < total; i0 += kScanThreads) {
      const uint32_t i = i0 + tid;
      uint64_t key = 0;
      bool want = false;
      if (i < total) {
        // the last list whose keys begin at or before i (lists without keys share their successor's offset: the last of such a
        // run is the one that has keys)
        uint32_t lo = 0, hi = kScanThreads;   // loff[lo] <= i < loff[hi] (loff[kScanThreads] = total > i)
        while (hi - lo > 1u) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s.loff[mid] <= i) lo = mid; else hi = mid;
        }
        key = in_keys[(size_t)s.lrec[lo] * k_stride_in + (i - s.loff[lo])];
        want = key > s.theta;
      }
      topk_append(s.cand, &s.cnt, want, key);
      __syncthreads();
      const uint32_t cn = s.cnt;
      __syncthreads();
      if (cn > (uint32_t)(kMergeCap - kScanThreads)) merge_compact(s, cn, k);
    }
    __syncthreads();   // (lrec / loff / wtot are rewritten by the next block of lists)
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
    // Speculative thresholds (plan.h: kHitsSpecInvalid): whatever the MaxScore walk skipped scores below the largest guess
    // published for the query.  The merged list stands iff its k-th key reaches that guess -- then nothing skipped could have
    // entered; otherwise the query is tagged and the host runs it again without speculation.
    const unsigned long long guess = spec_g ? spec_g[q] : 0ull;
    const bool failed = guess != 0ull && (n < k || s.cand[k - 1u] < guess);
    out_hits[q] = s.hits | (failed ? kHitsSpecInvalid : 0ull);
  }
}

// ------------------------------------------------------------------------------------------------
// fold_norms_kernel (seal time): per posting the score code of (freq, norm byte of the doc):
//   freq <= kTabMaxFreq && norm < kTabNorms : ((freq << 7) | norm) << 2   (byte offset into a score table)
//   otherwise                               : 0x80000000 | freq << 8 | norm
// freqs == nullptr => freq 1 (IndexOptions.DOCS); norms == nullptr => norm byte 1 (norms omitted,
// /root/reference/src/main/java/com/yelp/nrtsearch/server/field/AtomFieldDef.java:123-126).
// Sets *overflow when a freq does not fit 23 bits (the segment then stays on the CPU path).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void fold_norms_kernel(const uint32_t* __restrict__ docids, const uint32_t* __restrict__ freqs,
                       const uint8_t* __restrict__ norms, uint32_t* __restrict__ fnorm, uint64_t n,
                       uint32_t* __restrict__ overflow) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    const uint32_t f = freqs ? freqs[p] : 1u;
    const uint32_t nb = norms ? (uint32_t)norms[docids[p]] : 1u;
    if (f >= (1u << 22)) *overflow = 1u;
    fnorm[p] = (f >= 1u && f <= (uint32_t)kTabMaxFreq && nb < (uint32_t)kTabNorms) ? (((f << 7) | nb) << 2)
                                                                                  : (0x80000000u | (f << 8) | nb);
  }
}

// ------------------------------------------------------------------------------------------------
// apply_live_kernel (when liveDocs change): the postings of deleted docs are re-coded so that they score
// the neutral element -- the scan then needs no per-doc liveness test at all.  A posting of a doc whose
// bit in `live` is clear becomes
//   table form   ((f << 7) | nb) << 2          ->  (f << 20) | (nb << 2)      (row 0 of every score table)
//   escape form  0x80000000 | f << 8 | nb      ->  the same with bit 30 set
// and reverts when the doc is live again (live == nullptr: every doc).  freq keeps its place in the word,
// so nothing else has to be stored.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void apply_live_kernel(const uint32_t* __restrict__ docids, uint32_t* __restrict__ fnorm, uint64_t n,
                       const uint64_t* __restrict__ live) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    uint32_t c = fnorm[p];
    const bool esc = (c >> 31) != 0u;
    // back to the live form
    if (esc) c &= ~0x40000000u;
    else if ((c >> 20) != 0u) c = (((c >> 20) & 15u) << 9) | (c & 0x1FCu);
    bool is_live = true;
    if (live) {
      const uint32_t d = docids[p];
      is_live = (live[d >> 6] >> (d & 63u)) & 1ull;
    }
    if (!is_live) c = esc ? (c | 0x40000000u) : ((((c >> 9) & 15u) << 20) | (c & 0x1FCu));
    fnorm[p] = c;
  }
}

// ------------------------------------------------------------------------------------------------
// Seal time, NRTGPU_FLAG_PACKED_POSTINGS: (docid, 32-bit score code) -> ONE word per posting (plan.h: kPack*).
// pack_count_kernel: one workgroup per block of 2048 postings counts the block's exceptions (escape words: freq >
// kTabMaxFreq or norm >= kTabNorms); the host turns the counts into the directory (exclusive prefix).
// pack_write_kernel: the same blocks write the packed words; exceptions are numbered in posting order (directory entry +
// rank inside the block: wave ballots + an LDS prefix over the four waves x eight rounds) and their escape words
// stored at that number.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void pack_count_kernel(const uint32_t* __restrict__ fnorm, uint64_t n, uint32_t* __restrict__ counts) {
  __shared__ uint32_t total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  const uint64_t p0 = (uint64_t)blockIdx.x << kPackEscBlockShift;
  uint32_t mine = 0;
  for (uint32_t i = threadIdx.x; i < (1u << kPackEscBlockShift); i += 256u) {
    const uint64_t p = p0 + i;
    if (p < n && (fnorm[p] >> 31) != 0u) ++mine;
  }
  if (mine) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(256)
void pack_write_kernel(const uint32_t* __restrict__ docids, const uint32_t* __restrict__ fnorm, uint64_t n,
                       const uint32_t* __restrict__ dir, uint32_t* __restrict__ exceptions, uint32_t* __restrict__ packed) {
  __shared__ uint32_t wave_cnt[8][4];  // [round][wave]: exceptions of that wave's 64 postings
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t p0 = (uint64_t)blockIdx.x << kPackEscBlockShift;
  uint32_t c[8];
  unsigned long long ball[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {  // posting (r, wave, lane) = p0 + r * 256 + wave * 64 + lane: posting order = (r, wave, lane) order
    const uint64_t p = p0 + (uint64_t)r * 256u + threadIdx.x;
    c[r] = p < n ? fnorm[p] : 0u;
    ball[r] = __builtin_amdgcn_ballot_w64((c[r] >> 31) != 0u);
    if (lane == 0) wave_cnt[r][wave] = (uint32_t)__popcll(ball[r]);
  }
  __syncthreads();
  const uint32_t e_block = dir[blockIdx.x];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint64_t p = p0 + (uint64_t)r * 256u + threadIdx.x;
    if (p >= n) continue;
    uint32_t code = c[r] >> 2;
    if ((c[r] >> 31) != 0u) {
      uint32_t before = 0;
      for (int rr = 0; rr < 8; ++rr)
        for (int ww = 0; ww < 4; ++ww)
          if (rr < r || (rr == r && (uint32_t)ww < wave)) before += wave_cnt[rr][ww];
      const uint32_t e = e_block + before + (uint32_t)__popcll(ball[r] & ((1ull << lane) - 1ull));
      exceptions[e] = c[r];
      code = kPackEscBase + (e & kPackEscLowMask);
    }
    packed[p] = ((docids[p] & kPackDocMask) << kPackCodeBits) | code;
  }
}

// expand_terms_kernel: the compact plan -> the DTerm records of every (query, leaf).  One thread per (query, leaf):
// the clauses the leaf holds, ordered densest first (exhaustive scan) or heaviest first, ties sparsest first (MaxScore
// route), written at the offset the host reserved (out_begin; ~0 = the leaf holds none of the query's terms).  The
// position of a clause is its rank among the leaf's clauses -- at most 32, so ranking by comparison needs no scratch.
__global__ __launch_bounds__(256)
void expand_terms_kernel(const DQExpand* __restrict__ qx, const DQTerm* __restrict__ qterms, const uint32_t* __restrict__ out_begin,
                         uint32_t n_queries, uint32_t n_leaves, DTerm* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_queries * n_leaves) return;
  const uint32_t ob = out_begin[i];
  if (ob == 0xFFFFFFFFu) return;
  const uint32_t q = i / n_leaves, leaf = i - q * n_leaves;
  const DQExpand x = qx[q];
  const DQTerm* const qt = qterms + x.term_begin;
  for (uint32_t t = 0; t < x.n_terms; ++t) {
    DTerm d = qt[t].table[leaf];
    if (d.docids == nullptr) continue;
    const uint32_t cnt = __float_as_uint(d.weight);
    const float w = qt[t].weight;
    uint32_t rank = 0;
    for (uint32_t u = 0; u < x.n_terms; ++u) {
      if (u == t) continue;
      const DTerm& o = qt[u].table[leaf];
      if (o.docids == nullptr) continue;
      const uint32_t oc = __float_as_uint(o.weight);
      const float ow = qt[u].weight;
      const bool before = x.by_weight ? (ow > w || (ow == w && (oc < cnt || (oc == cnt && u < t))))
                                      : (oc > cnt || (oc == cnt && u < t));
      rank += before ? 1u : 0u;
    }
    d.weight = w;
    d.cache_slot = qt[t].cache_slot;
    d.tab_slot = qt[t].tab_slot;
    d.fx_scale = qt[t].fx_scale;
    d.fx_shift = qt[t].fx_shift;
    out[ob + rank] = d;
  }
}
void launch_expand_terms(hipStream_t stream, const DQExpand* qx, const DQTerm* qterms, const uint32_t* out_begin, uint32_t n_queries,
                         uint32_t n_leaves, DTerm* out) {
  const uint64_t n = (uint64_t)n_queries * n_leaves;
  if (n == 0) return;
  hipLaunchKernelGGL(expand_terms_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, qx, qterms, out_begin, n_queries,
                     n_leaves, out);
}

// slice_relation_kernel: TotalHits.relation by the reference's rule -- GREATER_THAN_OR_EQUAL_TO iff some slice's
// collector saw more than max(totalHitsThreshold, numHits) hits (one collector per slice, MyIndexSearcher.java:163-208;
// LazyQueueTopScoreDocCollector.java:176-199; reduce: LazyQueueTopScoreDocCollectorManager.java:137-144).  The scorers' items
// have added their per-slice counts into slice_sum (exact for every item that skipped nothing; an item that did skip has
// tagged its own count -- it only starts once a slice of its own has passed the floor); one thread per query tags the
// merged count with kHitsPrunedUnit, the same tag such items carry through the merge's sum.
__global__ __launch_bounds__(256)
void slice_relation_kernel(const uint32_t* __restrict__ slice_sum, const DQuery* __restrict__ queries, uint32_t n_slices,
                           uint64_t* __restrict__ out_hits, uint32_t n) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  if ((out_hits[q] >> 48) != 0ull) return;  // already tagged (pruned items)
  const uint32_t floor_ = queries[q].gte_floor;
  if (floor_ == 0xFFFFFFFFu) return;        // ScoreMode.COMPLETE: the count is the count
  const uint32_t* sums = slice_sum + queries[q].slice_base;
  bool gte = false;
  for (uint32_t i = 0; i < n_slices; ++i) gte = gte || sums[i] > floor_;
  if (gte) out_hits[q] += kHitsPrunedUnit;
}
void launch_slice_relation(hipStream_t stream, const uint32_t* slice_sum, const DQuery* queries, uint32_t n_slices, uint64_t* out_hits,
                           uint32_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(slice_relation_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, slice_sum, queries, n_slices, out_hits, n);
}

// Device-resident results (multi-GPU path): a query whose kernel SKIPPED work (its count carries the tag, plan.h:
// kHitsPrunedUnit) and for which the planner knew a certain lower bound reports that bound instead of the number of docs
// the kernel happened to evaluate -- the same value on every run.  A count without the tag is exact (nothing was
// skipped: e.g. the last page of a searchAfter walk) and stays.
__global__ __launch_bounds__(256)
void patch_hits_kernel(const uint64_t* __restrict__ lower, uint64_t* __restrict__ hits, uint32_t n) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  // (the merge's tag of a failed speculative threshold, kHitsSpecInvalid, stays: nrtgpu_pending_wait reads it)
  if (q < n && lower[q] != 0ull && (hits[q] >> 48) != 0ull) hits[q] = kHitsPrunedUnit + lower[q] + (hits[q] & kHitsSpecInvalid);
}
void launch_patch_hits(hipStream_t stream, const uint64_t* lower, uint64_t* hits, uint32_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(patch_hits_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, lower, hits, n);
}

// ---- launchers (called from the host runtime) ---------------------------------------------------------
void launch_pack_count(hipStream_t stream, const uint32_t* fnorm, uint64_t n, uint32_t n_blocks, uint32_t* counts) {
  if (n_blocks == 0) return;
  hipLaunchKernelGGL(pack_count_kernel, dim3(n_blocks), dim3(256), 0, stream, fnorm, n, counts);
}
void launch_pack_write(hipStream_t stream, const uint32_t* docids, const uint32_t* fnorm, uint64_t n, uint32_t n_blocks,
                       const uint32_t* dir, uint32_t* exceptions, uint32_t* packed) {
  if (n_blocks == 0) return;
  hipLaunchKernelGGL(pack_write_kernel, dim3(n_blocks), dim3(256), 0, stream, docids, fnorm, n, dir, exceptions, packed);
}
void launch_apply_live(hipStream_t stream, const uint32_t* docids, uint32_t* fnorm, uint64_t n, const uint64_t* live) {
  if (n == 0) return;
  const uint64_t blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(apply_live_kernel, dim3((uint32_t)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, docids, fnorm, n, live);
}
void launch_bm25_scan(hipStream_t stream, bool fixed_point, bool pipelined, bool packed, int ablation, uint32_t n_items, const DItem* items,
                      const DPart* parts, const DTerm* terms, const DQuery* queries, const float* caches, unsigned long long* theta_g,
                      unsigned long long* quant_g, const DExchange* xch, uint32_t* slice_sum, uint64_t* item_keys, uint32_t* item_counts,
                      uint64_t* item_hits, uint32_t k_stride, uint64_t* item_prof) {
  if (n_items == 0) return;
#define NRT_LAUNCH_K(F, P, A, K)                                                                                  \
  hipLaunchKernelGGL((bm25_scan_kernel<F, P, A, K>), dim3(n_items), dim3(kScanThreads), 0, stream, items, parts, terms, queries, \
                     caches, theta_g, quant_g, xch, slice_sum, item_keys, item_counts, item_hits, k_stride, item_prof)
#define NRT_LAUNCH(F, P, A)                 \
  do {                                      \
    if (packed) NRT_LAUNCH_K(F, P, A, true);  \
    else NRT_LAUNCH_K(F, P, A, false);      \
  } while (0)
#define NRT_LAUNCH_FX(P, A)            \
  do {                                 \
    if (fixed_point) NRT_LAUNCH(true, P, A); \
    else NRT_LAUNCH(false, P, A);      \
  } while (0)
  if (ablation == 8) { NRT_LAUNCH(true, true, 8); return; }  // minimumNumberShouldMatch > 1 / DisjunctionMaxQuery somewhere in the batch (fixed point only)
  if (ablation == 9) { NRT_LAUNCH_FX(true, 9); return; }     // some part carries a doc-set mask (deletes / FILTER / MUST_NOT)
  if (!pipelined) { NRT_LAUNCH_FX(false, 0); return; }
  switch (ablation) {
#ifdef NRTGPU_DEV  // timing ablations (wrong results): development build only, nrtgpu_create rejects the flag values otherwise
    case 1: NRT_LAUNCH_K(false, true, 1, false); break;  // 1-4: of the fp64 kernel
    case 2: NRT_LAUNCH_K(false, true, 2, false); break;
    case 3: NRT_LAUNCH_K(false, true, 3, false); break;
    case 4: NRT_LAUNCH_K(false, true, 4, false); break;
    case 6: NRT_LAUNCH_K(false, true, 6, false); break;  // no candidate handling in sparse sub-tiles
    case 7: NRT_LAUNCH_FX(true, 7); break;               // instrumented (same results; nrtgpu_get_scan_profile)
#endif
    default: NRT_LAUNCH_FX(true, 0); break;
  }
#undef NRT_LAUNCH_FX
#undef NRT_LAUNCH
#undef NRT_LAUNCH_K
}

void launch_merge_topk(hipStream_t stream, uint32_t n_queries, const uint64_t* in_keys, const uint32_t* in_counts,
                       const uint64_t* in_hits, const uint32_t* list_idx, const uint32_t* q_base,
                       const uint32_t* q_nlists, uint32_t k_stride_in, const uint32_t* q_k, uint64_t* out_keys,
                       uint32_t* out_counts, uint64_t* out_hits, uint32_t k_stride_out, const uint32_t* help_query,
                       uint32_t n_help, uint32_t help_slot_base, const unsigned long long* spec_g) {
  if (n_queries == 0) return;
  hipLaunchKernelGGL(merge_topk_kernel, dim3(n_queries), dim3(kScanThreads), 0, stream, in_keys, in_counts, in_hits,
                     list_idx, q_base, q_nlists, k_stride_in, q_k, out_keys, out_counts, out_hits, k_stride_out, help_query, n_help,
                     help_slot_base, spec_g);
}

void launch_fold_norms(hipStream_t stream, const uint32_t* docids, const uint32_t* freqs, const uint8_t* norms,
                       uint32_t* fnorm, uint64_t n, uint32_t* overflow) {
  if (n == 0) return;
  const uint64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(fold_norms_kernel, dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, docids,
                     freqs, norms, fnorm, n, overflow);
}

}  // namespace nrtgpu
