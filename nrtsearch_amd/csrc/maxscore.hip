// maxscore.hip -- the MaxScore route of the BM25 hot path on gfx950 (CDNA4, wave64): dynamic pruning on the device.
//
// What it replaces in the reference: the block-max / MaxScore skipping of lucene-core 10.4.0's
// MaxScoreBulkScorer (pure-SHOULD disjunctions under ScoreMode.TOP_SCORES), driven by the collector's
// Scorable.setMinCompetitiveScore calls
// (/root/reference/src/main/java/org/apache/lucene/search/LazyQueueTopScoreDocCollector.java:168-171,176-199).
// SURVEY.md 8a row a5.  Same contract as Lucene's: the top-k (docids, ranks, score bits) is exactly the
// exhaustive one -- only docs that provably cannot enter it are skipped -- and totalHits becomes a lower bound
// (relation GREATER_THAN_OR_EQUAL_TO), which is why the host takes this route only for queries whose hit count
// certainly exceeds totalHitsThreshold (planner.cpp) and never in ScoreMode.COMPLETE.
//
// Algorithm (the MaxScore idea, organised for a GPU: no per-doc priority queue of clause iterators, no
// per-sub-tile accumulators).  The clauses of a (query, segment) are ordered by weight, rarest first:
// t_0 .. t_{n-1}, each with its exact maximum score ub_j in this segment under the query's statistics (from
// the term's impact frontier, DTermAux) and the suffix sums S_j = ub_j + ... + ub_{n-1}.  A doc is evaluated at
// the FIRST clause (in that order) that matches it:
//   * clause t_i streams its postings (coalesced 32 B/lane column loads, score = one LDS table read);
//   * a posting whose score s_i + S_{i+1} cannot reach theta is dropped: if the doc matched an earlier clause it
//     was that clause's business, otherwise S_{i+1} bounds everything else it can get;
//   * a surviving posting tests-and-sets its doc's bit in the wave's LDS window ("already evaluated": the doc
//     matched an earlier clause and survived there); the winner looks the doc up in the later clauses
//     t_{i+1} .. one by one -- a 16-byte membership + rank record per 64 docs for dense terms, the cell
//     table + a short binary search for sparse ones -- re-checking running + S_j before each lookup, and ends
//     with the doc's exact score (the same fixed-point integers the exhaustive scan adds: identical bits);
//   * once S_i < theta the clauses t_i .. are non-essential: their postings are never streamed.
// A doc dropped at clause i is never evaluated later: under a later clause m its bound s_m + S_{m+1} <= S_{i+1}
// is below the theta it was dropped at, and theta only grows.  So every doc that can reach the final theta is
// evaluated exactly once, with its complete score.
//
// Work decomposition: a work item is a query x doc range; a workgroup is kMsWaves autonomous waves that own a CU (160 KB of
// LDS).  A wave takes windows of kMsWinDocs docs and walks the clauses of its window on its own, so the ordering argument
// above holds per wave with no barrier.  Competitive docs go to the workgroup's shared LDS candidate buffer; when it overflows
// all waves meet, a bucket select (topk.hiph) keeps the k best and raises theta -- the collector's pqTop /
// minCompetitiveScore.
// The launch (round 4; plan.h: MsArgs, DHelp; the head of the kernel): the items of a batch are a QUEUE; the launch has one
// PERSISTENT workgroup per CU (minus a few spare CUs) that chooses work round after round -- start the next item, or HELP a
// running one: an item's windows come from a counter in global memory, so its owner and any number of helpers share them; a
// helper keeps its own candidate list and output slot, which the merge walks next to the item's.  Which doc is evaluated by
// whom changes nothing of the argument above: windows partition an item's docs, theta only filters.  The launch's slowest
// queries get windows of a quarter the size (DItem.flags bits 2-3): a window is the grain at which work is shared.
// Speculative thresholds (plan.h: kHitsSpecInvalid; ms_compact): besides the guaranteed theta -- the k-th best of what has been
// seen -- a workgroup publishes a GUESS at the final k-th key from the best of the docs it has walked; the merge checks every
// guess against the merged list and the host runs a query whose guess failed again.  A guess therefore changes what is skipped
// and, when it fails, how often a query is run -- never what is returned.
// Roofline: HBM.  Reported both ways: effective = 9 B x the postings of the query's terms, physical
// = what the kernel fetches (a few percent of that).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bm25_common.hiph"

namespace nrtgpu {

constexpr int kMsWinWords = kMsWinDocs / 32;
constexpr int kSl = kMsSlots;   // postings a lane holds per instruction
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const NRT_GLOBAL u32x2* gvec2_ptr;

// What a lane needs to stream or look up one clause of the wave's current part / window (wave-private LDS table,
// written by lane c for clause c): lanes of one instruction may work for different clauses.
struct alignas(16) WClause {
  uint64_t docids, fnorm;   // column bases
  uint64_t begin;           // postings of the clause inside the window: first (absolute index into the columns) ...
  uint32_t count, pad0;     // ... and how many
  float    weight;
  int32_t  fx_scale;
  uint32_t flags;           // score table (0-2, 7 = none) | MUST clause << 3 | fx_shift << 4 | normInverse table << 8 | cell shift << 16 |
                            // lookup kind (plan.h: kLook*) << 24 | log2 docs per lookup cell << 27
  uint32_t pad;
  uint64_t u_after;         // what the later clauses can add at most: S_{c+1}
  uint64_t look;            // the term's lookup structure (plan.h: DTermAux.look): records / lookup cells, 0 = none
  uint64_t cells, start;    // cell table, first posting of the term in the columns
};
static_assert(sizeof(WClause) == 80, "WClause layout");

struct MsSmem {
  uint64_t cand[kMsCandCap];               // competitive hits of the item (packed keys), unordered
  uint32_t tab[kTabTerms][kTabEntries];    // fixed-point BM25 score of (freq, norm byte) for the item's densest terms
  float    cache[kLdsCaches][256];         // BM25 normInverse tables of the query's fields (division path, bounds)
  uint32_t seen[kMsWaves][kMsWinWords];    // per wave: docs of its window that have been evaluated
  WClause  wc[kMsWaves][kMsMaxTerms];      // per wave: the clauses of its part / window
  TopkScratch sc;
  uint64_t theta;        // packed key of the k-th best hit seen so far (0 = none)
  uint64_t thr;          // acc_threshold(theta): what running sums are compared with
  uint32_t cnt;          // entries in cand
  uint32_t cnt_valid;    // entries of cand that are complete when cnt ran past kMsCandCap
  uint32_t rz_flag;      // a wave could not reserve candidate slots (or a new speculative theta is due): everybody meet
  uint32_t wins_started; // doc windows this workgroup's waves have taken so far (of the item it works on)
  uint32_t spec_at;      // speculation: the next estimate is due when wins_started reaches this (0: never)
  uint32_t q_wins;       // doc windows of ALL items of the query
  uint32_t spec_z16;     // speculation (plan.h: kHitsSpecInvalid): safety margin in standard deviations x 16; 0: off
  uint32_t spec_grow16;  // ... the next estimate is due when wins_started has grown by this factor x 16
  unsigned long long* spec_slot;   // ... and where the query's largest speculative theta is published
  uint64_t pick;         // a helper's choice: float bits of its key (expected time left / an exponential variate) << 32 | item + 1
  uint32_t role[4];      // the workgroup's first decision (start the next item / help one): scratch values every thread reads
  uint32_t prune_on;     // bounds may skip work (kMsModeCount: raised once a slice's count has passed the query's gte_floor)
  uint32_t slot_hits[kSliceSlots];   // live matching docs evaluated, per searcher slice the item touches (plan.h: DPart.slice)
  uint32_t slot_slice[kSliceSlots];  // which slice a slot stands for
  uint64_t prof[16];
};
static_assert(sizeof(MsSmem) <= 160 * 1024, "the MaxScore workgroup owns one CU's 160 KiB LDS");

// a value every lane holds, moved into scalar registers (the compiler cannot know that an LDS read is uniform)
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | (uint64_t)lo;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, uint32_t l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
  return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// The value postings add for 8 score codes of ONE term (wave-uniform term): a table read, or -- for codes the
// table cannot serve (freq > kTabMaxFreq / norm >= kTabNorms, sign bit set) and for terms without a table -- the
// BM25 formula itself.  0 = posting of a deleted doc (apply_live_kernel) -- every live posting scores >= 1.
// PACKED: c[j] = the packed word's 12-bit code << 2 (codes from kPackEscBase on: exceptions, looked up in the group's
// exception list by the posting's index pidx[j] in its column).
template <bool PACKED, int NS>
__device__ __forceinline__ void values_of_codes(const MsSmem& s, const uint32_t (&c)[NS], uint32_t need, uint32_t tab_slot,
                                                float w, int fx_scale, uint32_t cache_slot, uint64_t esc_list,
                                                const uint32_t (&pidx)[NS], uint32_t (&val)[NS]) {
  const uint32_t tab = tab_slot < (uint32_t)kTabTerms ? tab_slot : 7u;
  const char* tb = (const char*)&s.tab[tab == 7u ? 0u : tab][0];
  uint32_t cor = 0;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    val[j] = *(const uint32_t*)(tb + (c[j] & 0x1FFCu));
    const uint32_t cn = ((need >> j) & 1u) ? c[j] : 0u;
    cor = PACKED ? max(cor, cn) : (cor | cn);
  }
  const bool special = need != 0u && ((PACKED ? cor >= (kPackEscBase << 2) : (cor >> 31) != 0u) || tab == 7u);
  if (__any(special)) {
    const float* cache = &s.cache[cache_slot][0];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      uint32_t cj = c[j];
      if (PACKED && cj >= (kPackEscBase << 2)) cj = ((need >> j) & 1u) ? packed_escape_word((gu32_ptr)esc_list, pidx[j], cj >> 2) : 0x80000100u;
      const bool esc = (cj >> 31) != 0u;
      const uint32_t f = esc ? ((cj >> 8) & 0x3FFFFFu) : ((cj >> 9) & 15u);
      const bool dead = esc ? ((cj >> 30) & 1u) != 0u : (cj >> 20) != 0u;
      const uint32_t nb = esc ? (cj & 255u) : ((cj >> 2) & 127u);
      if (((need >> j) & 1u) && (esc || tab == 7u))
        val[j] = dead ? 0u : score_value<true>(bm25_score(w, (float)(int32_t)f, cache[nb]), fx_scale);
    }
  }
}

// -DNRT_MS_PHASE_CLOCKS (a measurement build, instrumented kernel only): where a wave's walk time goes.  At each mark the wave
// waits for everything it has requested, reads the cycle counter and books the time since the last mark to a phase; the sums
// over all waves replace the event counters in slots 0-8 of the item's profile row (scripts/gpu_phase_clocks.py names them).
// The waits serialise what the product build overlaps within one phase -- the split is of THIS build's walk, which runs a few
// percent longer.
#ifdef NRT_MS_PHASE_CLOCKS
#define NRT_PH_MARK(i)                                                          \
  do {                                                                          \
    if (PROF) {                                                                 \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               \
      const uint64_t t_ph_ = __builtin_readcyclecounter();                      \
      ph[i] += t_ph_ - ph_t;                                                    \
      ph_t = t_ph_;                                                             \
    }                                                                           \
  } while (0)
#else
#define NRT_PH_MARK(i) do {} while (0)
#endif
// The launch record's pointer, laundered: what is loaded through it from here on is loaded again (see the kernel's head).
// (Through a vector register and back: an asm output counts as divergent, readfirstlane makes it a scalar again.)
// The record is read through the CONSTANT address space (scalar loads); the pointers it holds are global memory and are cast so
// where they are loaded (NRT_GLOBAL): a pointer that comes out of memory is a generic one to the compiler, and every access
// through it a flat vector instruction.
typedef const __attribute__((address_space(4))) MsArgs* ms_args_ptr;
// "this pointer, which came out of memory, points to GLOBAL memory": through the global address space and back, so that the
// compiler's address-space inference turns what is accessed through it into global_* instructions
template <class T>
__device__ __forceinline__ T* as_global(T* p) { return (T*)(NRT_GLOBAL T*)p; }
__device__ __forceinline__ ms_args_ptr ms_fresh(const MsArgs* p) {
  uint32_t lo = (uint32_t)(uintptr_t)p, hi = (uint32_t)((uintptr_t)p >> 32);
  asm volatile("" : "+v"(lo), "+v"(hi));
  lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
  hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
  return (ms_args_ptr)(((uintptr_t)hi << 32) | (uintptr_t)lo);
}

// All waves: keep the k best candidates, raise theta.  Contains barriers.
// xch (multi-GPU, nrtgpu_exchange_open): the item also publishes a score that ceil(k / (world - 1)) of ITS docs reach -- for free
// from the selection's histogram -- and bounds itself by the smallest entry of the OTHER ranks (bm25_common.hiph:
// exchange_bound): nothing below it can enter the merged top-k.
__device__ __noinline__ void ms_compact(__attribute__((address_space(3))) MsSmem* sp, uint32_t k, int fx_E,
                                        unsigned long long* theta_g, const DExchange* xch, uint32_t query) {
  MsSmem& s = *(MsSmem*)sp;
  const uint32_t tid = threadIdx.x;
  const uint32_t cnt_raw = s.cnt;
  const uint32_t cnt0 = cnt_raw > (uint32_t)kMsCandCap ? s.cnt_valid : cnt_raw;  // failed reservations inflate cnt
  __syncthreads();
  // Speculation (plan.h: kHitsSpecInvalid): this workgroup has taken wins_started of the query's q_wins doc windows, so about
  // m = k x that fraction of the query's final top-k are among the keys it holds -- their (m + z sqrt(m) + 1)-th best is a
  // guess at the final k-th key, z standard deviations on the safe side.  Windows that are only begun count as walked: the
  // fraction errs high, the rank deep, the guess low.
  const bool exchanging = xch && xch->world > 1u;
  uint32_t r = 0;
  // (only once bounds may skip: by then more than max(totalHitsThreshold, numHits) docs are known to match -- a query with
  //  fewer than k hits would fail every guess)
  if (s.spec_z16 != 0u && !exchanging && __hip_atomic_load(&s.prune_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
    const float m = (float)k * fminf(1.0f, (float)s.wins_started / (float)max(s.q_wins, 1u));
    const float rr = m + (float)s.spec_z16 * (1.0f / 16.0f) * sqrtf(m) + 2.0f;
    r = rr < (float)k ? (uint32_t)rr : 0u;
  }
  uint64_t guess = 0;
  if (cnt0 > k) {  // uniform
    const uint32_t k2 = exchanging ? (k + xch->world - 2u) / (xch->world - 1u) : r;
    const uint64_t thr = topk_kth_union<kMsThreads>(s.cand, cnt0, k, &s.sc, [](auto&&) {}, k2);
    const uint32_t q2_hi = s.sc.q2_hi;  // (stable until the next selection)
    const uint32_t kept = topk_keep_ge<kMsThreads, kMsCandCap>(s.cand, cnt0, thr, &s.sc);
    if (tid == 0) {
      s.cnt = kept;
      if (thr > s.theta) {
        s.theta = thr;
        s.thr = acc_threshold<true>(thr, fx_E);
      }
      atomicMax(theta_g, (unsigned long long)thr);  // LazyMaxScoreAccumulator.accumulate analogue
      s.prof[1] += 1;
    }
    if (!exchanging && r != 0u && q2_hi != 0u) guess = (uint64_t)q2_hi << 32;   // a score at least r of my keys reach (the histogram's bucket edge)
    if (xch && tid < 64u) {   // wave 0: publish my quantile, bound myself by the other ranks' entries
      const uint64_t pb = exchange_bound(*xch, query, q2_hi != 0u ? (uint64_t)q2_hi << 32 : 0ull, tid);
      if (tid == 0 && pb > s.theta) {
        s.theta = pb;
        s.thr = acc_threshold<true>(pb, fx_E);
        atomicMax(theta_g, (unsigned long long)pb);   // (the query's other items on this GPU get it through theta_g)
      }
    }
  } else {
    if (r != 0u && cnt0 > r) guess = topk_kth_union<kMsThreads>(s.cand, cnt0, r, &s.sc, [](auto&&) {}, 0u);   // the r-th best key itself; nothing is dropped
    if (tid == 0) s.cnt = cnt0;
  }
  if (tid == 0 && s.spec_z16 != 0u) {
    if (guess > s.theta) {
      s.theta = guess;
      s.thr = acc_threshold<true>(guess, fx_E);
      atomicMax(theta_g, (unsigned long long)guess);
      atomicMax(s.spec_slot, (unsigned long long)guess);
    }
    // the next estimate: when (by default) twice as many windows have been taken -- the guess moves with the fraction's square root
    const uint32_t ws = s.wins_started;
    s.spec_at = ws >= s.q_wins ? 0u : max(ws * s.spec_grow16 / 16u, ws + (uint32_t)kMsWaves);
  }
  __syncthreads();
  if (tid == 0) s.rz_flag = 0;
  __syncthreads();
}

// Meeting point of the workgroup's waves: returns false when nobody asked for a compaction (every wave is out
// of work), else runs it.
__device__ __forceinline__ bool ms_meet(MsSmem& s, uint32_t k, int fx_E, unsigned long long* theta_g, const DExchange* xch = nullptr,
                                        uint32_t query = 0) {
  __syncthreads();
  if (!__hip_atomic_load(&s.rz_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
  ms_compact((__attribute__((address_space(3))) MsSmem*)&s, k, fx_E, theta_g, xch, query);
  return true;
}

// Reserve room for the wave's `mine`-per-lane candidates in the shared buffer: one DPP scan and ONE LDS atomic.
// Returns whether they fit -- a WAVE-UNIFORM answer, computed from scalars: the caller's failure path contains
// a workgroup barrier, so no lane may disagree (a lane past the wave's last candidate holds pos == end of the
// reservation, which for an exactly filled buffer equals its capacity).  On failure rz_flag is raised; cnt only
// grows between compactions, so exactly the first reservation that crosses the end has base <= kMsCandCap:
// everything below its base is completely written -> cnt_valid.
__device__ __forceinline__ bool ms_reserve(MsSmem& s, uint32_t lane, uint32_t mine, uint32_t& pos) {
  const uint32_t incl = scan64_dpp(mine);
  const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  uint32_t wbase = 0;
  if (lane == 0) {
    wbase = atomicAdd(&s.cnt, wave_total);
    if (wbase + wave_total > (uint32_t)kMsCandCap) {
      if (wbase <= (uint32_t)kMsCandCap) s.cnt_valid = wbase;
      __hip_atomic_store(&s.rz_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
  pos = wbase + incl - mine;
  return wbase + wave_total <= (uint32_t)kMsCandCap;
}

// PROF: per-item event counters (nrtgpu_get_maxscore_profile): [0] windows, [1] compactions, [2] posting chunks,
// [3] postings streamed, [4] postings surviving the bound, [5] docs evaluated, [6] lookups, [7] candidates; shader-clock
// cycles: [8] item prologue (tables), [9] the whole item, [10] sum over waves of the cycles spent in meetings (waiting +
// compaction), [11] sum over waves of the cycles between running out of windows and the item's end, [12] sum over waves of
// part prologues (clause maxima), [13] sum over waves of the window walk (stream + lookups + candidates), [14] the last
// wave's cycle of running out of windows, [15] item epilogue.
// PACKED: the segments keep one 32-bit word per posting (plan.h: kPack*).  liveDocs that are not folded into the
// postings (part.live_bits != nullptr) are tested when a doc's score is complete.
// SHAPES: the batch holds queries with a doc-set mask next to their scoring clauses (FILTER / MUST_NOT,
// QueryNodeMapper.java:257-283), minimumNumberShouldMatch > 1 (:259-261) or a DisjunctionMaxQuery (:350-358, tie breaker 0):
//   * mask: a doc outside part.live_bits (liveDocs & filter & ~must_not) is no hit -- tested when its score is complete;
//   * minimumNumberShouldMatch: the clauses that matched a doc are counted next to its sum (4 bits per posting slot); a doc
//     with fewer is no hit.  The MaxScore argument is untouched: a doc that matches non-essential clauses only cannot be
//     competitive however many of them it matches;
//   * DisjunctionMaxQuery: a doc scores its BEST clause -- `max` where the sum has `+`, and the clauses after c can lift a
//     doc to max(ub_c+1 ..) instead of their sum.
// SHAPES == 2: some query's score is not ONE sum (plan.h: kMsSec*) -- a DisjunctionMaxQuery with a tie breaker > 0
//   (QueryNodeMapper.java:350-358), MUST next to SHOULD clauses (:257-283).  Every posting slot carries a SECOND accumulator
//   (sec[]: the best clause / the SHOULD clauses' sum) that enters the doc's final key only:
//     (float)(best + (sum - best) * tieBreaker) in double -- DisjunctionMaxScorer;  (float)mustSum + (float)shouldSum --
//     ReqOptSumScorer [Lucene-recall].  Both are at most what the plain sum scores (tieBreaker <= 1; the two-float addition up
//     to 2 ulp, which its thresholds are lowered by: loosen()), so every bound of the walk stays a bound.
//   MUST clauses: a doc first reached at clause c lacks every streamed clause before c -- a MUST clause among them and it is
//   no hit; clauses behind the first MUST clause start no doc and are never streamed; a MUST clause whose lookup misses ends
//   the doc.  (A leaf that lacks a MUST term is not planned at all.)
//   Sixteen more registers than the kernel has: this instantiation spills inside the walk, the other two are untouched.
template <bool PROF, bool PACKED, int SHAPES>
__global__ __launch_bounds__(kMsThreads)
void bm25_maxscore_kernel(const MsArgs* __restrict__ launch) {
  constexpr bool TWO = SHAPES == 2;
  __shared__ MsSmem s;
  // The launch record (plan.h: MsArgs) lives next to the plan; ONE pointer is the kernel's argument.  Its fields are scalar loads
  // made where they are used -- the workgroup's round, then once more in front of the item's epilogue (ms_fresh: the compiler
  // cannot tell that the pointer it returns is the one it was given) -- instead of fifteen kernel arguments held in scalar registers from the first
  // instruction to the last: the walk between prologue and epilogue sits on the register edge (168 VGPRs, SGPR spills go to
  // VGPR lanes), and what only the epilogue needs has no business being live there.
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  // ---- what does this workgroup do?  The launch has n_own + n_help workgroups and a QUEUE of n_own items in launch order
  //      (plan.h: DHelp.item_next).  A workgroup that gets a CU either starts the next item or HELPS one that is running: a
  //      running item's windows come from a counter in global memory, so its owner and any number of helpers simply share them.
  //        * While items are queued, it helps only an item on the launch's CRITICAL PATH: one whose expected time left -- the
  //          time it has run so far x unassigned windows / windows handed out, divided among the helpers already there --
  //          exceeds what is left of the whole launch (elapsed x windows not handed out / windows handed out over ALL items,
  //          x alpha).  That item would end after everything else; a CU is better spent on it than on starting one more item.
  //        * Once the queue is empty, it helps any item with min_rem windows left (the tail), or leaves.
  //      At most n_help workgroups help (they reserve an output slot each), so at least n_own start items: the queue drains.
  //      Helpers come in herds (an item that ends frees its owner and its helpers at once) whose members cannot see each
  //      other's choices: each picks item i with PROBABILITY proportional to left_i -- the largest left_i / -ln(u_i), u_i
  //      uniform and its own -- so a herd spreads over the candidates in proportion to the time they have left.
  // A workgroup is PERSISTENT (hp.persistent; the launch then has one workgroup per CU): having finished what it chose to do it
  // chooses again -- start the next item of the queue, or help -- until there is nothing left.  A workgroup that owns a whole CU
  // (160 KB of LDS) is not replaced the moment it ends: measured with one workgroup per item, ~9 % of the CUs sat between two
  // workgroups at any time while items were still queued (profiles/r04_makespan_*.log: 215-235 of 256 running).
  for (uint32_t round = 0;; ++round) {
    const uint64_t wall_entry = PROF ? wall_clock64() : 0ull;
    const ms_args_ptr ap = ms_fresh(launch);
    const __attribute__((address_space(4))) DHelp& hp = ap->help;
    const DItem* const items = as_global(ap->items);
    const DPart* const parts = as_global(ap->parts);
    const DTerm* const terms = as_global(ap->terms);
    const DExchange* const xch = as_global(ap->xch);
    bool helper = false;
    uint32_t my_item = 0, out_slot = 0;
    {
      const uint32_t min_rem = hp.min_rem & 0xFFFFu;
      const bool greedy = ((hp.min_rem >> 16) & 1u) != 0u;
      for (int attempt = 0; attempt < 2; ++attempt) {   // (the second: the queue ran dry between the look and the pop)
        if (tid == 0) {
          s.pick = 0ull;
          s.role[0] = __hip_atomic_load(hp.item_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s.role[1] = __hip_atomic_load(hp.help_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s.role[2] = 0u;   // windows handed out, all items
          s.role[3] = __hip_atomic_load(hp.help_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const uint32_t q_next = s.role[0];
        const bool queue_empty = attempt == 1 || q_next >= hp.n_own;
        const bool slots_left = s.role[1] < hp.n_help;
        if (queue_empty && (!slots_left || s.role[3] != 0u)) return;   // (uniform) nothing to start, nothing to help
        // worth a look?  Not while the first items have hardly begun (nothing is known about anybody's pace yet)
        const bool look = slots_left && (queue_empty || (hp.alpha16 != 0u && q_next >= hp.n_cus + hp.n_cus / 4u));
        if (look) {
          const uint64_t now = wall_clock64();
          float t_rem = 0.f;   // what is left of the launch (queue not empty: the bar an item's own time left must pass)
          if (!queue_empty) {
            uint32_t mine = 0;
            for (uint32_t i = tid; i < hp.n_own; i += kMsThreads) {
              const uint64_t t0 = __hip_atomic_load(hp.item_t0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (t0 == 0ull) continue;
              mine += min(items[i].flags >> 8, (uint32_t)kMsWaves + __hip_atomic_load(hp.win_next + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            mine = (uint32_t)__builtin_amdgcn_readlane((int)scan64_dpp(mine), 63);
            if (lane == 0 && mine != 0u) atomicAdd(&s.role[2], mine);
            __syncthreads();
            const uint32_t handed = s.role[2];
            const uint64_t ts = __hip_atomic_load(hp.t_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t_rem = (handed == 0u || ts == 0ull || now <= ts) ? 3.0e38f
                    : (float)(now - ts) * (float)(hp.total_wins > handed ? hp.total_wins - handed : 0u) / (float)handed * ((float)hp.alpha16 * (1.0f / 16.0f));
          }
          for (uint32_t i = tid; i < hp.n_own; i += kMsThreads) {
            const uint32_t fl = items[i].flags;
            // Prune-mode items, and (round 5) COUNTING items -- masks, minimumNumberShouldMatch, DisjunctionMax, uncertain counts: a
            // helper is to its item what a second item is to its query: it counts the live matching docs of ITS windows exactly, per
            // slice, until its own count passes the floor or the query's q_prune flag is up (checked at every window's head), and adds
            // its per-slice counts to the query's sums at its end -- the argument that makes a counting query of several items right
            // covers it.  (Exact-mode items are small by construction: nobody needs help there.)
            if ((fl & 3u) == kMsModeExact) continue;
            const uint32_t nw = fl >> 8;
            const uint32_t taken = (uint32_t)kMsWaves + __hip_atomic_load(hp.win_next + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (taken + min_rem > nw) continue;
            const uint64_t t0 = __hip_atomic_load(hp.item_t0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t0 == 0ull || now <= t0) continue;     // its owner has not started yet
            const uint32_t hc = __hip_atomic_load(hp.help_cnt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float left = (float)(now - t0) * (float)(nw - taken) / ((float)(taken - (uint32_t)kMsWaves + 1u) * (float)(1u + hc));
            if (!queue_empty && !(left > t_rem)) continue;
            uint32_t hsh = ((blockIdx.x + round * gridDim.x) * 0x9E3779B9u) ^ (i * 0x85EBCA6Bu) ^ (uint32_t)now;
            hsh ^= hsh >> 16; hsh *= 0x7FEB352Du; hsh ^= hsh >> 15; hsh *= 0x846CA68Bu; hsh ^= hsh >> 16;
            const float u = ((float)(hsh >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float key = greedy ? left : left / -__logf(u);   // (greedy: A/B -- everybody takes the largest left_i)
            atomicMax((unsigned long long*)&s.pick, ((unsigned long long)__float_as_uint(key) << 32) | (unsigned long long)(i + 1u));
          }
        }
        __syncthreads();
        if (tid == 0) {
          const uint64_t pick = s.pick;
          uint32_t r_item = 0xFFFFFFFFu, r_slot = 0xFFFFFFFFu;
          if ((uint32_t)pick != 0u) {
            const uint32_t slot = atomicAdd(hp.help_used, 1u);
            if (slot < hp.n_help) {
              r_slot = slot;
              r_item = (uint32_t)pick - 1u;
              atomicAdd(hp.help_cnt + r_item, 1u);
            }
          } else if (queue_empty && look) {
            __hip_atomic_store(hp.help_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (it stays that way: nothing starts any more)
          }
          if (r_slot == 0xFFFFFFFFu && !queue_empty) {
            const uint32_t it = atomicAdd(hp.item_next, 1u);
            if (it < hp.n_own) r_item = it;
          }
          s.role[0] = r_item;
          s.role[1] = r_slot;
        }
        __syncthreads();
        const uint32_t r_item = s.role[0], r_slot = s.role[1];
        __syncthreads();   // (s.role is rewritten by the next attempt)
        if (r_item != 0xFFFFFFFFu) {
          my_item = r_item;
          helper = r_slot != 0xFFFFFFFFu;
          out_slot = helper ? hp.slot_base + r_slot : r_item;
          break;
        }
        if (queue_empty) return;   // (uniform)
      }
    }
    NRT_GLOBAL uint32_t* const win_next_g = (NRT_GLOBAL uint32_t*)(hp.win_next + my_item);   // (a global address: no flat instruction in the window loop)
    // a helper wave's first window (an owner's waves start with windows 0 .. kMsWaves - 1): asked for now, read behind the tables
    uint32_t first_win = 0;
    if (helper && lane == 0) first_win = __hip_atomic_fetch_add(win_next_g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const DItem item = items[my_item];
    const DQuery q = as_global(ap->queries)[item.query];
    const uint32_t k = q.k;
    const int fx_E = item.fx_E;
    unsigned long long* const my_theta_g = as_global(ap->theta_g) + item.query;
    const bool multi_item = q.n_items > 1 || hp.n_help != 0u;   // (theta_g is how owner and helpers share theta as well)
    // When may bounds skip work (plan.h: kMsMode*)?  Exact: never.  Count: once a slice has collected more than gte_floor hits --
    // until then every live matching doc is evaluated and counted, as the reference's collector does before it first
    // publishes a min competitive score.  theta filters the candidates in every mode.
    const uint32_t mode = item.flags & 3u;
    const uint32_t win_tiles = (uint32_t)kMsWinTiles >> ((item.flags >> 2) & 3u);   // (uniform) sub-tiles per doc window of this item
    const uint32_t msm = SHAPES ? q.min_should_match : 0u;   // (> 1: clause counting)
    const bool use_max = SHAPES && q.combine_max != 0u;       // DisjunctionMaxQuery, tie breaker 0
    const uint32_t sec_mode = TWO ? q.sec_mode : kMsSecNone;  // (uniform) what the second accumulator holds
    const float tie_breaker = TWO ? q.tie_breaker : 0.0f;
    // kMsSecReqOpt: (float)a + (float)b may exceed (float)(a + b) by 2 float ulps: sums are compared with a threshold 2^-21 lower
    auto loosen = [&](uint64_t t) { return (TWO && sec_mode == kMsSecReqOpt) ? max(t - (t >> 21), (uint64_t)2) - (uint64_t)1 : t; };   // (both uint64_t: max(uint64_t, unsigned long long) is HIP's DOUBLE overload)
    unsigned int* const my_prune_g = as_global(ap->q_prune) + item.query;    // set by the first item of the query whose slice passed the floor
    const uint64_t after_key = q.has_after ? pack_key(q.after_score, (uint32_t)q.after_doc) : ~0ull;

    // ---- item prologue: normInverse tables, score tables
    {
      const uint32_t n_lds = min(item.n_caches, (uint32_t)kLdsCaches) * 256u;
      const float* const caches = as_global(ap->caches);
      for (uint32_t i = tid; i < n_lds; i += kMsThreads) (&s.cache[0][0])[i] = caches[item.cache_off + i];
    }
    if (tid == 0) {
      const uint64_t theta0 = __hip_atomic_load(my_theta_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s.theta = theta0;
      s.thr = acc_threshold<true>(theta0, fx_E);
      s.cnt = 0;
      s.cnt_valid = 0;
      s.rz_flag = 0;
      if (!helper) {   // when this item began; the launch's first item: when the launch began
        const unsigned long long now0 = (unsigned long long)wall_clock64();
        __hip_atomic_store(hp.item_t0 + my_item, now0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (my_item == 0u) __hip_atomic_store(hp.t_start, now0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s.prune_on = mode == kMsModePrune ? 1u : 0u;
      // speculation (plan.h: kHitsSpecInvalid): off unless the launch has somebody who re-runs a query whose guess failed
      const bool spec = hp.spec_g != nullptr && hp.spec_z16 != 0u;
      s.wins_started = 0;
      s.q_wins = spec ? as_global(ap->q_wins)[item.query] : 0u;
      s.spec_z16 = spec ? hp.spec_z16 : 0u;
      s.spec_slot = spec ? as_global(hp.spec_g) + item.query : nullptr;
      s.spec_at = spec ? max(hp.spec_sched & 255u, 1u) : 0u;   // the first estimate (default: when every wave has begun its second window)
      s.spec_grow16 = max((hp.spec_sched >> 8) & 255u, 17u);
      for (int i = 0; i < kSliceSlots; ++i) s.slot_hits[i] = s.slot_slice[i] = 0u;
      for (int i = 0; i < 16; ++i) s.prof[i] = 0;
    }
    const uint64_t t_item0 = PROF ? __builtin_readcyclecounter() : 0ull;
    const uint64_t wall0 = PROF ? wall_clock64() : 0ull;
    uint64_t tc_meet = 0, tc_part = 0, tc_walk = 0;
    __syncthreads();
    if (xch && tid < 64u) {   // wave 0: what the other GPUs' shards have published for this query so far (nothing of mine yet)
      const uint64_t pb = exchange_bound(*xch, item.query, 0ull, tid);
      if (tid == 0 && pb > s.theta) {
        s.theta = pb;
        s.thr = acc_threshold<true>(pb, fx_E);
      }
    }
    for (uint32_t slot = 0; slot < item.n_tabs; ++slot) {
      const float w = items[my_item].tab_weight[slot];
      const int scale = items[my_item].tab_scale[slot];
      const float* cache = &s.cache[items[my_item].tab_cache[slot]][0];
      for (uint32_t e = tid; e < (uint32_t)kTabEntries; e += kMsThreads)  // row 0: postings of deleted docs score 0
        s.tab[slot][e] = e < (uint32_t)kTabNorms ? 0u : score_value<true>(bm25_score(w, (float)(int32_t)(e >> 7), cache[e & 127u]), scale);
    }
    __syncthreads();  // from here on the waves run on their own
  #ifndef NRT_MS_PHASE_CLOCKS
    if (PROF && tid == 0) s.prof[8] = __builtin_readcyclecounter() - t_item0;
  #endif

    uint32_t* const seen = &s.seen[wave][0];
    WClause* const wcl = &s.wc[wave][0];
    const uint32_t wcl_addr = lds_addr(wcl);
    uint32_t wave_hits = 0;    // hits of my current slot not yet added to s.slot_hits
    uint32_t cur_slot = 0;
    uint64_t pc_post = 0, pc_surv = 0, pc_look = 0, pc_cand = 0, pc_chunks = 0, pc_wins = 0;
  #ifdef NRT_MS_PHASE_CLOCKS
    uint64_t ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = 0;   // 0 window head, 1 columns, 2 values + bounds, 3 test-and-set, 4 records, 5 codes, 6 sums, 7 sparse clause, 8 hits + candidates
  #endif
  #ifdef NRT_MS_COUNT_ROUNDS   // experiment build: event counts instead of four of the cycle counters
    uint64_t pc_dense = 0, pc_sparse = 0, pc_steps = 0, pc_tas = 0, pc_crounds = 0;
  #endif
    // The item's windows (flattened over its parts, item_wins of them) are handed out in a SCATTERED order: the i-th window taken
    // (by anybody: the counter is shared with the item's helpers) is window (i x a) mod item_wins, a coprime to item_wins and near
    // 0.618 item_wins -- consecutive takes land far apart and any prefix of the takes is spread evenly over the item's docs, whatever
    // its parts are.  The speculative thresholds (ms_compact) read "the docs of the windows begun so far" as a sample of the query's
    // docs: true of independently drawn docids in any order; true of time-ordered docids, of terms that come in bursts, of an index
    // sorted by a field the score follows only in an order like this one.  A wave changes its part more often than in docid order (a
    // part prologue each time); with a dozen windows per wave and item it met most parts anyway.
    // (a = P mod n for a prime P > n: coprime; of four primes the one whose a / n is nearest the golden section.)
    const uint32_t item_wins = item.flags >> 8;
    uint32_t sc_mul = 1u;
    if (ap->scatter != 0u && item_wins > 3u && item_wins < 65536u) {   // (uniform; i x a stays below 2^32)
      const uint32_t primes[4] = {2654435761u, 2246822519u, 3266489917u, 668265263u};
      uint32_t best = 0xFFFFFFFFu;
  #pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t a = primes[i] % item_wins;
        const uint32_t x = a * 1000u, y = item_wins * 618u;
        const uint32_t dist = x > y ? x - y : y - x;
        if (a > 1u && dist < best) {
          best = dist;
          sc_mul = a;
        }
      }
    }
    auto window_of_take = [&](uint32_t h) -> uint32_t {   // (uniform) the h-th take's window; takes past the item stay past it
      return (sc_mul == 1u || h >= item_wins) ? h : (h * sc_mul) % item_wins;
    };
    uint32_t g = window_of_take(helper ? (uint32_t)kMsWaves + (uint32_t)__builtin_amdgcn_readfirstlane((int)first_win) : wave);   // my current window (flattened over the item's parts)
    uint32_t pi = 0;        // its part ...
    uint32_t win_base = 0;  // ... and the windows of the parts before that one

    for (;;) {
      // ---- the part that holds window g
      DPart part;
      uint32_t part_wins = 0;
      if (sc_mul != 1u) {   // (scattered order: the next window may lie before the current part)
        pi = 0;
        win_base = 0;
      }
      for (;; ++pi) {
        if (pi >= item.n_parts) break;
        part = parts[item.part_begin + pi];
        // windows start on win_tiles boundaries of the SEGMENT (the first one of a part may be short): a window never
        // spans two 2^20-doc super-windows, which is what packed doc offsets are relative to
        part_wins = (part.tile_end - (part.tile_begin & ~(win_tiles - 1u)) + win_tiles - 1u) / win_tiles;
        if (g < win_base + part_wins) break;
        win_base += part_wins;
      }
      if (pi >= item.n_parts) break;
      {  // hits are counted per searcher slice: a part of another slice closes my count of the previous one
        const uint32_t p_slot = part.slice >> 24;
        if (p_slot != cur_slot && wave_hits != 0u) {
          if (lane == 0) atomicAdd(&s.slot_hits[cur_slot], wave_hits);
          wave_hits = 0;
        }
        cur_slot = p_slot;
        if (lane == 0) s.slot_slice[p_slot] = part.slice & 0xFFFFFFu;
      }
      const uint32_t n_terms = part.n_terms;  // <= kMsMaxTerms (planner)
      const DTerm* const part_terms = terms + part.term_begin;

      // ---- per part: lane l looks after clause l: its exact maximum score in this segment, suffix sums, and the
      //      clause's record in the wave's LDS table (what a lane needs to stream or look up clause l)
      uint64_t my_ub = 0, my_suf = 0;
      const uint64_t t_part0 = PROF ? __builtin_readcyclecounter() : 0ull;
      const DTerm mt = part_terms[min(lane, n_terms - 1u)];
      {
        // 16 lanes per clause, four clauses per pass: lane (g, i) evaluates frontier entry i of clause 4 * pass + g -- two
        // dependent loads per PASS (the clause's record, then its frontier byte) instead of two per clause
        uint32_t ub_raw = 0;
        for (uint32_t pass = 0; pass * 4u < n_terms; ++pass) {  // uniform
          const uint32_t tt = pass * 4u + (lane >> 4), i = lane & 15u;
          const DTerm* Tp = part_terms + min(tt, n_terms - 1u);
          const DTermAux* ax = Tp->aux;
          const float w = Tp->weight;
          const int scale = Tp->fx_scale;
          const float* cache = &s.cache[Tp->cache_slot][0];
          uint32_t v = 0;
          if (tt < n_terms) {
            if (i < 12u) {
              const uint32_t nb = ax->min_norm[i];
              if (nb != 0xFFu) v = score_value<true>(bm25_score(w, (float)(int32_t)(i + 1u), cache[nb]), scale);
            } else if (i == 12u) {
              const uint32_t mf = ax->esc_max_freq;
              if (mf != 0u) v = score_value<true>(bm25_score(w, (float)(int32_t)mf, cache[ax->esc_min_norm]), scale);
            }
          }
  #pragma unroll
          for (int dlt = 8; dlt > 0; dlt >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, dlt, 64));  // maximum of each 16-lane group
          const uint32_t got = (uint32_t)__shfl((int)v, (int)((lane & 3u) * 16u), 64);            // lane L = clause L: group L & 3 of pass L >> 2
          if ((lane >> 2) == pass) ub_raw = got;
        }
        if (lane < n_terms) my_ub = (uint64_t)ub_raw << mt.fx_shift;
      }
      uint64_t my_after = 0;   // what the clauses after mine can add (sum) / lift a doc to (DisjunctionMaxQuery): S_{lane+1}
      {
        uint64_t run = 0;
        for (int m = (int)n_terms - 1; m >= 0; --m) {
          const uint64_t ub = readlane_u64(my_ub, (uint32_t)m);
          if (lane == (uint32_t)m) my_after = run;
          run = use_max ? max(run, ub) : run + ub;
          if (lane == (uint32_t)m) my_suf = run;
        }
      }
      const gu32_ptr my_cells = (gu32_ptr)mt.cell_off;
      const uint32_t my_shift = mt.shift;
      if (lane < n_terms) {
        WClause w;
        w.docids = (uint64_t)mt.docids;
        w.fnorm = (uint64_t)mt.fnorm;
        w.begin = 0;
        w.count = w.pad0 = 0;
        w.weight = mt.weight;
        w.fx_scale = mt.fx_scale;
        const DTermAux* const ax = mt.aux;
        const uint32_t look_kind = ax->look_kind;
        w.flags = ((mt.tab_slot & 0xFFFFu) < (uint32_t)kTabTerms ? (mt.tab_slot & 0xFFFFu) : 7u) | (TWO && (mt.tab_slot & kTabSlotRequired) ? 8u : 0u) |
                  (mt.fx_shift << 4) | (mt.cache_slot << 8) | ((mt.shift & 31u) << 16) | ((look_kind & 7u) << 24) | (((uint32_t)ax->look_shift & 31u) << 27);
        w.pad = 0;
        w.u_after = my_after;
        w.look = look_kind != kLookNone ? (uint64_t)ax->look : 0ull;
        w.cells = (uint64_t)mt.cell_off;
        w.start = mt.start;
        wcl[lane] = w;
      }
      // (uniform) the part's MUST clauses, bit c = clause c; first_req: the last clause that may start a doc
      const uint32_t req_mask = TWO ? (uint32_t)__builtin_amdgcn_ballot_w64(lane < n_terms && (mt.tab_slot & kTabSlotRequired) != 0u) : 0u;
      const uint32_t first_req = req_mask != 0u ? (uint32_t)__builtin_ctz(req_mask) : 0xFFu;
      if (PROF) tc_part += __builtin_readcyclecounter() - t_part0;

      for (;;) {  // windows of this part
        const uint64_t t_win0 = PROF ? __builtin_readcyclecounter() : 0ull;
        const uint32_t ta = (part.tile_begin & ~(win_tiles - 1u)) + (g - win_base) * win_tiles;
        const uint32_t t0 = max(ta, part.tile_begin);
        const uint32_t t1 = min(ta + win_tiles, part.tile_end);
        const uint32_t doc_lo = t0 * (uint32_t)kTileDocs;
        const uint32_t doc_span = min(t1 * (uint32_t)kTileDocs, part.max_doc) - doc_lo;
        if (PROF) pc_wins += 1;
        if (lane == 0) {   // speculation: one more window begun; a new estimate is due every time their number has doubled
          const uint32_t ws = atomicAdd(&s.wins_started, 1u) + 1u, at = s.spec_at;
          if (at != 0u && ws >= at) __hip_atomic_store(&s.rz_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // my next window: taken now, so that the counter's answer is there when this one is done
        uint32_t g_new = 0;
        if (lane == 0) g_new = (uint32_t)kMsWaves + __hip_atomic_fetch_add(win_next_g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (global: shared with the item's helpers)
        // theta of the query's other items (LazyMaxScoreAccumulator analogue), once per window
        uint64_t theta_other = 0, thr_other = 0;
        if (multi_item) {
          theta_other = __hip_atomic_load(my_theta_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          thr_other = acc_threshold<true>(theta_other, fx_E);
          // (another item's slice has passed the floor: the relation is decided, this item may skip as well)
          if (mode == kMsModeCount && lane == 0 && __hip_atomic_load(my_prune_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
            __hip_atomic_store(&s.prune_on, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // posting range of every clause in this window (cells may be coarser than the window: doc-range filter below)
        uint32_t my_lo = 0, my_hi = 0;
        if (lane < n_terms) {
          my_lo = my_cells[t0 >> my_shift];
          my_hi = my_cells[((t1 - 1u) >> my_shift) + 1u];
        }
        if (n_terms > 1u) {
  #pragma unroll
          for (int j = 0; j < kMsWinWords / 64 / 4; ++j) *(u32x4*)&seen[(lane + 64u * (uint32_t)j) * 4u] = u32x4{0u, 0u, 0u, 0u};
        }
        // The essential clauses of the window -- S_c >= theta, a prefix of the order -- are streamed as ONE sequence of
        // 8-posting groups (16-byte aligned in the columns): a lane takes one group, so sparse clauses share an
        // instruction instead of taking one each.  Clause c's groups come before clause c + 1's.
        uint32_t ng = 0;
        {
          const uint64_t thr_w = __hip_atomic_load(&s.prune_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? loosen(max(s.thr, thr_other)) : 0ull;
          const uint64_t pb = mt.start + my_lo, pe = mt.start + my_hi;
          if (lane < n_terms) {
            // (minimumNumberShouldMatch: a doc is evaluated at the first clause that holds it, so one first met at clause c matches
            //  at most n_terms - c clauses: the last msm - 1 clauses cannot start a hit and are never streamed)
            if (my_suf >= thr_w && pe > pb && (!SHAPES || msm <= 1u || lane + msm <= n_terms) && (!TWO || lane <= first_req)) ng = (uint32_t)((pe - (pb & ~3ull) + (uint64_t)(kSl - 1)) / (uint64_t)kSl);
            *(u32x4*)&wcl[lane].begin = u32x4{(uint32_t)pb, (uint32_t)(pb >> 32), (uint32_t)(pe - pb), 0u};
          }
        }
        const uint32_t incl = scan32_dpp(ng);
        uint32_t pre[kMsMaxTerms];
  #pragma unroll
        for (int i = 0; i < kMsMaxTerms; ++i) pre[i] = (uint32_t)__builtin_amdgcn_readlane((int)incl, i);
        const uint32_t n_groups = pre[kMsMaxTerms - 1];

  #ifdef NRT_MS_PHASE_CLOCKS
        if (PROF) ph_t = t_win0;
  #endif
        NRT_PH_MARK(0);
        for (uint32_t v0 = 0; v0 < n_groups; v0 += 64u) {  // 64 groups = up to 512 postings per instruction
          const uint32_t v = v0 + lane;
          const bool act = v < n_groups;
          uint32_t c = 0, before = 0;
  #pragma unroll
          for (int i = 0; i < kMsMaxTerms - 1; ++i) {
            c += (v >= pre[i]) ? 1u : 0u;
            before = (v >= pre[i]) ? pre[i] : before;
          }
          const uint32_t c_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
          // theta as of now (it only grows: a stale value costs work, never a result).  Has it passed what this
          // instruction's first clause and everything after it can reach?  Then the rest of the window is non-essential.
          const uint64_t theta = max(s.theta, theta_other), thr = loosen(max(s.thr, thr_other));
          const bool pruning = __hip_atomic_load(&s.prune_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;   // (uniform)
          const uint64_t thr_p = pruning ? thr : 0ull;   // what the bounds are compared with
          if (readlane_u64(my_suf, c_first) < thr_p) break;
          if (PROF) pc_chunks += 1;
          // what my clause is: column bases, posting range, score table, scale, what the later clauses can still add
          const uint32_t rec = wcl_addr + c * (uint32_t)sizeof(WClause);
          const u32x4 r0 = *(const u32x4*)lds_ptr(rec), r1 = *(const u32x4*)lds_ptr(rec + 16u), r2 = *(const u32x4*)lds_ptr(rec + 32u);
          const uint64_t u_after = *(const uint64_t*)lds_ptr(rec + 48u);
          const uint64_t col_d = ((uint64_t)r0[1] << 32) | r0[0], col_c = ((uint64_t)r0[3] << 32) | r0[2];
          const uint64_t p_begin = ((uint64_t)r1[1] << 32) | r1[0];
          const uint32_t flags = r2[2];
          const uint32_t q0 = (v - before) * (uint32_t)kSl;  // my group's first posting, counted from the clause's 16-byte aligned begin
          const uint64_t mine0 = (p_begin & ~3ull) + (uint64_t)q0;
          uint32_t d[kSl], cd[kSl];
  #pragma unroll
          for (int j = 0; j < kSl; ++j) d[j] = cd[j] = 0u;
          if (act) {  // (the columns are padded: a partly valid group may read past the term)
            const u32x4 d0 = __builtin_nontemporal_load((gvec_ptr)(col_d + mine0 * 4u));
            const u32x4 d1 = kSl > 4 ? __builtin_nontemporal_load((gvec_ptr)(col_d + mine0 * 4u) + 1) : d0;
            if (PACKED) {  // one word per posting: doc offset inside the window's super-window | code
              const uint32_t sw = doc_lo & ~kPackDocMask;
  #pragma unroll
              for (int j = 0; j < 4; ++j) {
                d[j] = (d0[j] >> kPackCodeBits) | sw;
                cd[j] = (d0[j] & kPackCodeMask) << 2;
                if (kSl > 4) {
                  d[(4 + j) % kSl] = (d1[j] >> kPackCodeBits) | sw;
                  cd[(4 + j) % kSl] = (d1[j] & kPackCodeMask) << 2;
                }
              }
            } else {
              const u32x4 c0 = __builtin_nontemporal_load((gvec_ptr)(col_c + mine0 * 4u));
              const u32x4 c1 = kSl > 4 ? __builtin_nontemporal_load((gvec_ptr)(col_c + mine0 * 4u) + 1) : c0;
  #pragma unroll
              for (int j = 0; j < 4; ++j) {
                d[j] = d0[j];
                cd[j] = c0[j];
                if (kSl > 4) {
                  d[(4 + j) % kSl] = d1[j];
                  cd[(4 + j) % kSl] = c1[j];
                }
              }
            }
          }
          NRT_PH_MARK(1);
          uint32_t vmask = 0;  // my postings inside the clause's range (one unsigned compare: positions before the range wrap) and the window
          {
            const uint32_t rel = q0 - ((uint32_t)p_begin & 3u), cnt = act ? r1[2] : 0u;
  #pragma unroll
            for (int j = 0; j < kSl; ++j)
              if (rel + (uint32_t)j < cnt && d[j] - doc_lo < doc_span) vmask |= 1u << j;
          }
          // the values my postings add: per-lane table (lanes of one instruction may belong to different clauses)
          uint32_t val[kSl];
          {
            const uint32_t tab = flags & 7u;
            const char* tb = (const char*)&s.tab[tab == 7u ? 0u : tab][0];
            uint32_t cor = 0;
  #pragma unroll
            for (int j = 0; j < kSl; ++j) {
              val[j] = *(const uint32_t*)(tb + (cd[j] & 0x1FFCu));
              const uint32_t cn = ((vmask >> j) & 1u) ? cd[j] : 0u;
              cor = PACKED ? max(cor, cn) : (cor | cn);
            }
            const bool special = vmask != 0u && ((PACKED ? cor >= (kPackEscBase << 2) : (cor >> 31) != 0u) || tab == 7u);
            if (__any(special)) {  // long docs / high freqs / clauses without a score table
              const float w = __uint_as_float(r2[0]);
              const int fx_scale = (int)r2[1];
              const float* cache = &s.cache[(flags >> 8) & 255u][0];
  #pragma unroll
              for (int j = 0; j < kSl; ++j) {
                uint32_t cj = cd[j];
                if (PACKED && cj >= (kPackEscBase << 2))   // (col_c: the group's exception list; mine0 + j: the posting's index in its column)
                  cj = ((vmask >> j) & 1u) ? packed_escape_word((gu32_ptr)col_c, (uint32_t)mine0 + (uint32_t)j, cj >> 2) : 0x80000100u;
                const bool esc = (cj >> 31) != 0u;
                const uint32_t f = esc ? ((cj >> 8) & 0x3FFFFFu) : ((cj >> 9) & 15u);
                const bool dead = esc ? ((cj >> 30) & 1u) != 0u : (cj >> 20) != 0u;
                const uint32_t nb = esc ? (cj & 255u) : ((cj >> 2) & 127u);
                if (((vmask >> j) & 1u) && (esc || tab == 7u))
                  val[j] = dead ? 0u : score_value<true>(bm25_score(w, (float)(int32_t)f, cache[nb]), fx_scale);
              }
            }
          }
          uint64_t run[kSl];
          uint32_t alive = 0;
          {
            const uint32_t mult = 1u << ((flags >> 4) & 15u);  // entry << shift as one 32 x 32 -> 64 multiply
            // "entry * mult + u_after >= thr_p" (DisjunctionMaxQuery: max instead of +) as ONE 32-bit compare per posting: the
            // smallest entry that passes, computed once per lane (>= 1: a posting of a deleted doc scores 0 and never passes)
            uint32_t need_v = 1u;
            if (u_after < thr_p) {
              const uint64_t gap = use_max ? thr_p : thr_p - u_after;
              const uint64_t nv = (gap + (uint64_t)(mult - 1u)) >> ((flags >> 4) & 15u);
              need_v = nv > 0xFFFFFFFFull ? 0xFFFFFFFFu : max((uint32_t)nv, 1u);
              // (an entry of 2^32 - 1 that still falls short: the exact test below is the rare fallback)
            }
            uint32_t pass = 0;
  #pragma unroll
            for (int j = 0; j < kSl; ++j) {
              run[j] = (uint64_t)val[j] * (uint64_t)mult;
              pass |= val[j] >= need_v ? (1u << j) : 0u;
            }
            alive = vmask & pass;
            if (need_v == 0xFFFFFFFFu) {   // (per lane, practically never)
  #pragma unroll
              for (int j = 0; j < kSl; ++j)
                if ((use_max ? max(run[j], u_after) : run[j] + u_after) < thr_p) alive &= ~(1u << j);
            }
          }
          uint64_t sec[TWO ? kSl : 1];
          if (TWO) {
            // a doc first reached at clause c is in no streamed clause before c: a MUST clause among them and it is no hit
            if ((req_mask & ((1u << c) - 1u)) != 0u) alive = 0u;
            const bool mine_counts = sec_mode == kMsSecTieBreaker || (sec_mode == kMsSecReqOpt && ((req_mask >> c) & 1u) == 0u);
  #pragma unroll
            for (int j = 0; j < kSl; ++j) sec[j] = mine_counts ? run[j] : 0ull;
          }
          if (PROF) {
            pc_post += (uint64_t)__popc(vmask);
            pc_surv += (uint64_t)__popc(alive);
          }
          NRT_PH_MARK(2);
          // first clause to reach the doc?  Test-and-set, clause by clause in order: LDS executes a wave's operations
          // in order, so of two postings of one doc in this instruction the earlier clause's wins.  (Lanes without a
          // survivor OR a zero into a word of their own.)
          const uint32_t c_last = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)(min(n_groups - v0, 64u) - 1u));
          if (n_terms > 1u && __any(alive != 0u)) {
            for (uint32_t cc = c_first; cc <= c_last; ++cc) {
              const uint32_t am = c == cc ? alive : 0u;
              if (!__any(am != 0u)) continue;
  #ifdef NRT_MS_COUNT_ROUNDS
              if (PROF) pc_tas += 1;
  #endif
              uint32_t old[kSl];
  #pragma unroll
              for (int j = 0; j < kSl; ++j) {
                const bool a = (am >> j) & 1u;
                const uint32_t w = a ? ((d[j] - doc_lo) >> 5) : lane;
                old[j] = atomicOr(&seen[w], a ? (1u << (d[j] & 31u)) : 0u);
              }
  #pragma unroll
              for (int j = 0; j < kSl; ++j)
                if (((am >> j) & 1u) && ((old[j] >> (d[j] & 31u)) & 1u)) alive &= ~(1u << j);
            }
          }
          NRT_PH_MARK(3);
          uint32_t ccnt = 0x11111111u;   // SHAPES, minimumNumberShouldMatch: clauses that matched the doc, 4 bits per posting slot

          // ---- the later clauses of the surviving docs, one clause at a time (a lane of clause c takes part from c + 1 on)
          for (uint32_t j2 = c_first + 1u; j2 < n_terms; ++j2) {
            if (!__any(alive != 0u)) break;
            const uint64_t S_j = readlane_u64(my_suf, j2);
            uint32_t am = c < j2 ? alive : 0u;
            {
              // "sum + S_j < thr_p" (DisjunctionMaxQuery: max(sum, S_j)) against a UNIFORM value: sum < thr_p - S_j -- one 64-bit
              // compare per doc, no add; nothing is dropped while S_j alone reaches thr_p
              const uint64_t short_of = S_j < thr_p ? (use_max ? thr_p : thr_p - S_j) : 0ull;
              uint32_t kill = 0;
  #pragma unroll
              for (int j = 0; j < kSl; ++j) kill |= run[j] < short_of ? (1u << j) : 0u;
              if (SHAPES && msm > 1u) {   // (uniform) too few clauses left to reach minimumNumberShouldMatch: no hit, whatever it scores
                const uint32_t left = n_terms - j2;
  #pragma unroll
                for (int j = 0; j < kSl; ++j) kill |= ((ccnt >> (4 * j)) & 15u) + left < msm ? (1u << j) : 0u;
              }
              kill &= am;
              alive &= ~kill;
              am &= ~kill;
            }
            if (!__any(am != 0u)) continue;
            if (PROF) pc_look += (uint64_t)__popc(am);
            const WClause& w2 = wcl[j2];  // uniform reads; the pointers as scalars: a gather is then base + 32-bit lane offset
            const uint64_t look2 = uniform_u64(w2.look);
            const uint32_t flags2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w2.flags);
            const uint32_t kind2 = (flags2 >> 24) & 7u;   // (uniform) how a doc is looked up in this clause (plan.h: kLook*)
            const uint64_t start2 = uniform_u64(w2.start);
            const gu32_ptr codes2 = (gu32_ptr)(uniform_u64(PACKED ? w2.docids : w2.fnorm) + start2 * 4u);  // packed: the code rides in the posting's word
            uint32_t c2[kSl];
            uint32_t pi2[kSl];  // packed postings: the looked-up postings' indices in their column (exception lookups)
            uint32_t present = 0;
  #ifdef NRT_MS_COUNT_ROUNDS
            if (PROF) { if (kind2 == kLookBits) pc_dense += 1; else pc_sparse += 1; }
  #endif
            if (kind2 == kLookBits) {
              // RECORDS: one 8-byte record per 32 docs {doc bits, postings of the term before the block} says whether the doc is
              // there and where its posting is; the code is a second, dependent gather
              const gvec2_ptr recs = (gvec2_ptr)look2;
              u32x2 r[kSl];
  #pragma unroll
              for (int j = 0; j < kSl; ++j) r[j] = recs[((am >> j) & 1u) ? (d[j] >> 5) : 0u];
              __builtin_amdgcn_sched_barrier(0);   // every record load is issued before the first one is waited for
              NRT_PH_MARK(4);
              uint32_t idx[kSl];
  #pragma unroll
              for (int j = 0; j < kSl; ++j) {
                const uint32_t bb = d[j] & 31u;
                const bool there = ((am >> j) & 1u) && ((r[j][0] >> bb) & 1u);
                idx[j] = there ? r[j][1] + (uint32_t)__popc(r[j][0] & ((1u << bb) - 1u)) : 0u;
                present |= (there ? 1u : 0u) << j;
              }
  #pragma unroll
              for (int j = 0; j < kSl; ++j) c2[j] = codes2[idx[j]];
              __builtin_amdgcn_sched_barrier(0);   // (the same for the code loads)
              NRT_PH_MARK(5);
  #pragma unroll
              for (int j = 0; j < kSl; ++j) pi2[j] = (uint32_t)start2 + idx[j];
            } else {
              // SEARCH: the doc's cell -- a lookup cell of 2^look_shift docs holding 0.5 - 1 posting on average (kLookCells), else
              // its cell of the tile-granular table (~4 - 8 postings) -- then a binary search among the cell's postings; the 8
              // searches of a lane advance in lockstep, so every step is one round of loads in flight instead of eight
              const gu32_ptr cells2 = (gu32_ptr)(kind2 == kLookCells ? look2 : uniform_u64(w2.cells));
              const gu32_ptr docs2 = (gu32_ptr)(uniform_u64(w2.docids) + start2 * 4u);
              const uint32_t cshift = kind2 == kLookCells ? (flags2 >> 27) & 31u : 10u + ((flags2 >> 16) & 31u);
              uint32_t a[kSl], b[kSl];
  #pragma unroll
              for (int j = 0; j < kSl; ++j) {
                const uint32_t cell = ((am >> j) & 1u) ? (d[j] >> cshift) : 0u;
                a[j] = cells2[cell];
                b[j] = cells2[cell + 1u];
              }
              uint32_t open = 0;
  #pragma unroll
              for (int j = 0; j < kSl; ++j) {
                if (!((am >> j) & 1u)) b[j] = a[j];
                open |= (a[j] < b[j] ? 1u : 0u) << j;
              }
              while (__any(open != 0u)) {  // lower bound of d[j] in [a, b): b stays the first index known to hold a docid >= d[j]
  #ifdef NRT_MS_COUNT_ROUNDS
                if (PROF) pc_steps += 1;
  #endif
                uint32_t mid[kSl], vv[kSl];
  #pragma unroll
                for (int j = 0; j < kSl; ++j) {
                  mid[j] = (a[j] + b[j]) >> 1;
                  vv[j] = docs2[((open >> j) & 1u) ? mid[j] : 0u];
                }
  #pragma unroll
                for (int j = 0; j < kSl; ++j)
                  if ((open >> j) & 1u) {
                    // (packed: doc offsets inside the cell's super-window -- the doc's own, a cell never spans two)
                    const uint32_t dv = PACKED ? vv[j] >> kPackCodeBits : vv[j], dd = PACKED ? d[j] & kPackDocMask : d[j];
                    if (dv < dd) a[j] = mid[j] + 1u;
                    else b[j] = mid[j];
                    if (dv == dd) {  // found: close the search on it
                      a[j] = b[j] = mid[j];
                      present |= 1u << j;
                    }
                    if (!(a[j] < b[j])) open &= ~(1u << j);
                  }
              }
  #pragma unroll
              for (int j = 0; j < kSl; ++j) {
                c2[j] = codes2[((present >> j) & 1u) ? a[j] : 0u];   // (packed: the posting's word again -- keeping the probe's word alive
                                                                      //  through the search loop cost more than this gather)
                pi2[j] = (uint32_t)start2 + a[j];
              }
              NRT_PH_MARK(7);
            }
            if (__any(present != 0u)) {
              uint32_t v2[kSl];
              if (PACKED) {
  #pragma unroll
                for (int j = 0; j < kSl; ++j) c2[j] = (c2[j] & kPackCodeMask) << 2;
              }
              values_of_codes<PACKED, kSl>(s, c2, present, flags2 & 7u, w2.weight, w2.fx_scale, (flags2 >> 8) & 255u, w2.fnorm, pi2, v2);
              const uint32_t mult2 = 1u << ((flags2 >> 4) & 15u);
              if (use_max) {   // (uniform)
  #pragma unroll
                for (int j = 0; j < kSl; ++j) run[j] = max(run[j], (uint64_t)(((present >> j) & 1u) ? v2[j] : 0u) * (uint64_t)mult2);
              } else if (TWO && sec_mode != kMsSecNone) {   // (uniform) the sum, and the best clause / the SHOULD clauses' sum next to it
                const bool counts = sec_mode == kMsSecReqOpt && (flags2 & 8u) == 0u;
  #pragma unroll
                for (int j = 0; j < kSl; ++j) {
                  const uint64_t add = (uint64_t)(((present >> j) & 1u) ? v2[j] : 0u) * (uint64_t)mult2;
                  run[j] += add;
                  sec[j] = sec_mode == kMsSecTieBreaker ? max(sec[j], add) : sec[j] + (counts ? add : 0ull);
                }
              } else {
  #pragma unroll
                for (int j = 0; j < kSl; ++j) run[j] += (uint64_t)(((present >> j) & 1u) ? v2[j] : 0u) * (uint64_t)mult2;  // v_mad_u64_u32
              }
              if (SHAPES && msm > 1u) {   // (uniform) bit j of `present` -> nibble j
                uint32_t x = present;
                x = (x | (x << 12)) & 0x000F000Fu;
                x = (x | (x << 6)) & 0x03030303u;
                x = (x | (x << 3)) & 0x11111111u;
                ccnt += x;
              }
            }
            if (TWO && (flags2 & 8u) != 0u) alive &= ~(am & ~present);   // (uniform) a MUST clause the doc lacks: no hit
            NRT_PH_MARK(6);
          }

          {
          // ---- complete scores: the competitive ones go to the shared candidate buffer.  Rare once theta has
          //      converged, so the key (a double conversion) is built only for sums that reach theta's score, one
          //      posting per lane and round.
          uint32_t maybe = 0;
  #pragma unroll
          for (int j = 0; j < kSl; ++j)
            if (((alive >> j) & 1u) && run[j] >= thr) maybe |= 1u << j;
          // ---- hits.  While nothing is being skipped every live matching doc reaches this point exactly once: the count is
          //      exact.  Once bounds skip it is a lower bound, and only the docs that reach theta's score are looked at.
          //      A hit lies inside the part's doc set (liveDocs that are not folded into the postings -- packed layout, forked
          //      reader versions -- and FILTER / MUST_NOT masks: one dword gather per doc) and matches enough clauses.
          {
            uint32_t pool = pruning ? maybe : alive;
            if (SHAPES && msm > 1u) {   // (uniform)
  #pragma unroll
              for (int j = 0; j < kSl; ++j)
                if (((ccnt >> (4 * j)) & 15u) < msm) pool &= ~(1u << j);
            }
            if (part.live_bits != nullptr && __any(pool != 0u)) {   // (uniform)
              uint32_t lw[kSl];
  #pragma unroll
              for (int j = 0; j < kSl; ++j) lw[j] = ((const NRT_GLOBAL uint32_t*)part.live_bits)[((pool >> j) & 1u) ? (d[j] >> 5) : 0u];
  #pragma unroll
              for (int j = 0; j < kSl; ++j)
                if (!((lw[j] >> (d[j] & 31u)) & 1u)) pool &= ~(1u << j);
            }
            maybe &= pool;
            uint32_t h = (uint32_t)__popc(pool);
            h = (uint32_t)__builtin_amdgcn_readlane((int)scan64_dpp(h), 63);
            if (mode == kMsModeCount && !pruning) {
              // counting towards the floor: into the slot at once, so that the item notices when a slice has passed it
              if (lane == 0 && h != 0u) {
                const uint32_t tot = atomicAdd(&s.slot_hits[cur_slot], h) + h;
                if (tot > q.gte_floor) {
                  __hip_atomic_store(&s.prune_on, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  if (multi_item) __hip_atomic_store(my_prune_g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
              }
            } else {
              wave_hits += h;
            }
          }
          uint64_t theta_now = theta;
          while (__any(maybe != 0u)) {
  #ifdef NRT_MS_COUNT_ROUNDS
            if (PROF) pc_crounds += 1;
  #endif
            const uint32_t low = maybe & (0u - maybe);  // my lowest pending posting
            uint64_t rsel = run[0], ssel = TWO ? sec[0] : 0ull;
            uint32_t dsel = d[0];
  #pragma unroll
            for (int j = 1; j < kSl; ++j)
              if (low == (1u << j)) {
                rsel = run[j];
                dsel = d[j];
                if (TWO) ssel = sec[j];
              }
            float score = acc_score<true>(rsel, fx_E);
            if (TWO && sec_mode == kMsSecTieBreaker)        // DisjunctionMaxScorer.score(): (float)(scoreMax + otherScoreSum * tieBreaker), doubles
              score = (float)(ldexp((double)ssel, -fx_E) + ldexp((double)(rsel - ssel), -fx_E) * (double)tie_breaker);
            else if (TWO && sec_mode == kMsSecReqOpt)       // ReqOptSumScorer.score(): req.score() + opt.score(), floats (an absent opt adds 0.0f)
              score = acc_score<true>(rsel - ssel, fx_E) + acc_score<true>(ssel, fx_E);
            const uint64_t key = pack_key(score, (uint32_t)(part.doc_base + (int32_t)dsel));
            const bool want = low != 0u && key > theta_now && key < after_key;
            uint32_t pos = 0;
            if (!__any(want)) {
              maybe &= ~low;
              continue;
            }
            if (ms_reserve(s, lane, want ? 1u : 0u, pos)) {  // wave-uniform
              if (want) s.cand[pos] = key;
              if (PROF) pc_cand += want ? 1u : 0u;
              maybe &= ~low;
              continue;
            }
            // no room: everybody meets, the k best stay, theta rises; then the same postings again under the new theta
            const uint64_t t_m0 = PROF ? __builtin_readcyclecounter() : 0ull;
            (void)ms_meet(s, k, fx_E, my_theta_g, xch, item.query);
            if (PROF) tc_meet += __builtin_readcyclecounter() - t_m0;
            theta_now = max(theta_now, s.theta);
          }
          }
          NRT_PH_MARK(8);
          // somebody else asked for a compaction: join it between two instructions
          if (__hip_atomic_load(&s.rz_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            const uint64_t t_m0 = PROF ? __builtin_readcyclecounter() : 0ull;
            (void)ms_meet(s, k, fx_E, my_theta_g, xch, item.query);
            if (PROF) tc_meet += __builtin_readcyclecounter() - t_m0;
          }
        }

        // ---- next window
        if (PROF) tc_walk += __builtin_readcyclecounter() - t_win0;
        g = window_of_take((uint32_t)__builtin_amdgcn_readfirstlane((int)g_new));
        if (g < win_base || g >= win_base + part_wins) break;  // another part (or past the item)
      }
    }
    // ---- out of work: stay available for the others' compactions until everybody is done
    const uint64_t t_idle0 = PROF ? __builtin_readcyclecounter() : 0ull;
    while (ms_meet(s, k, fx_E, my_theta_g, xch, item.query)) {
    }
    const uint64_t t_epi0 = PROF ? __builtin_readcyclecounter() : 0ull;
    if (PROF && lane == 0) {
      atomicAdd((unsigned long long*)&s.prof[10], (unsigned long long)tc_meet);
      atomicAdd((unsigned long long*)&s.prof[11], (unsigned long long)(t_epi0 - t_idle0));
      atomicAdd((unsigned long long*)&s.prof[12], (unsigned long long)tc_part);
      atomicAdd((unsigned long long*)&s.prof[13], (unsigned long long)(tc_walk - tc_meet));
      atomicMax((unsigned long long*)&s.prof[14], (unsigned long long)(t_idle0 - t_item0));
  #ifdef NRT_MS_COUNT_ROUNDS
      atomicAdd((unsigned long long*)&s.prof[11], (unsigned long long)pc_tas - (unsigned long long)(t_epi0 - t_idle0));
      atomicAdd((unsigned long long*)&s.prof[12], (unsigned long long)pc_dense - (unsigned long long)tc_part);
      atomicAdd((unsigned long long*)&s.prof[10], (unsigned long long)pc_sparse - (unsigned long long)tc_meet);
      atomicAdd((unsigned long long*)&s.prof[13], (unsigned long long)pc_steps - (unsigned long long)(tc_walk - tc_meet));
      atomicAdd((unsigned long long*)&s.prof[8], (unsigned long long)pc_crounds);
  #endif
    }

    // ---- item epilogue
    {
      const uint32_t c = s.cnt;
      __syncthreads();
      if (c > k) {
        uint64_t thr = 0;
        const uint32_t m = topk_compact<kMsThreads, kMsCandCap>(s.cand, c, k, &s.sc, &thr);
        if (tid == 0) s.cnt = m;
      }
    }
    if (lane == 0 && wave_hits) atomicAdd(&s.slot_hits[cur_slot], wave_hits);
  #ifdef NRT_MS_PHASE_CLOCKS
    if (PROF && lane == 0) {
  #pragma unroll
      for (int i = 0; i < 9; ++i) atomicAdd((unsigned long long*)&s.prof[i], (unsigned long long)ph[i]);
    }
    if (false) {
  #else
    if (PROF && lane == 0) {
      atomicAdd((unsigned long long*)&s.prof[0], (unsigned long long)pc_wins);
      atomicAdd((unsigned long long*)&s.prof[2], (unsigned long long)pc_chunks);
    }
    if (PROF) {
  #endif
      uint64_t v3 = pc_post, v4 = pc_surv, v6 = pc_look, v7 = pc_cand;
  #pragma unroll
      for (int dlt = 32; dlt > 0; dlt >>= 1) {
        v3 += __shfl_xor(v3, dlt, 64);
        v4 += __shfl_xor(v4, dlt, 64);
        v6 += __shfl_xor(v6, dlt, 64);
        v7 += __shfl_xor(v7, dlt, 64);
      }
      if (lane == 0) {
        atomicAdd((unsigned long long*)&s.prof[3], (unsigned long long)v3);
        atomicAdd((unsigned long long*)&s.prof[4], (unsigned long long)v4);
        atomicAdd((unsigned long long*)&s.prof[6], (unsigned long long)v6);
        atomicAdd((unsigned long long*)&s.prof[7], (unsigned long long)v7);
      }
    }
    __syncthreads();
    const uint32_t n_held = s.cnt;
    const ms_args_ptr ape = ms_fresh(launch);   // (the epilogue's fields of the launch record: loaded here, not held through the walk)
    const __attribute__((address_space(4))) DHelp& hpe = ape->help;
    uint64_t* out = as_global(ape->item_keys) + (size_t)out_slot * ape->k_stride;
    // What leaves the workgroup: the keys that still reach the query's threshold AS IT STANDS NOW (theta_g: what the query's other
    // items and this item's other workgroups have published since these keys were collected -- a key below it is in nobody's
    // top-k).  A helper that walked a few early windows holds up to k keys from when theta was low; with one query per call and
    // 250 helpers the merge read them all (43 - 88 us of a 230 us call, profiles/r06_single_query_timeline.txt).
    if (tid == 0) {
      uint64_t th_end = s.theta;   // (compared as 64-bit INTEGERS: no max() overload that would take them through a double)
      if (multi_item) {
        const uint64_t tg = (uint64_t)__hip_atomic_load(my_theta_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tg > th_end) th_end = tg;
      }
      s.thr = th_end;     // (s.thr: free after the walk)
      s.cnt_valid = 0u;   // (free since the walk's last compaction: the output's counter)
    }
    __syncthreads();
    for (uint32_t i = tid; i < n_held; i += kMsThreads) {
      const uint64_t key = s.cand[i];
      if (key >= s.thr) out[atomicAdd(&s.cnt_valid, 1u)] = key;
    }
    __syncthreads();
    const uint32_t n = s.cnt_valid;
    if (tid == 0) {
      as_global(ape->item_counts)[out_slot] = n;
      if (helper)   // my slot is one of the query's lists (merge_topk_kernel reads the helpers' slots behind the items')
        as_global(hpe.help_query)[out_slot - hpe.slot_base] = item.query + 1u;
      // the item's hits, per slice into the query's sums (slice_relation_kernel) and in total
      uint32_t hits = 0;
      for (int i = 0; i < kSliceSlots; ++i) {
        const uint32_t h = s.slot_hits[i];
        hits += h;
        if (h != 0u && q.gte_floor != 0xFFFFFFFFu) atomicAdd(as_global(ape->slice_sum) + q.slice_base + s.slot_slice[i], h);
      }
      // anything skipped?  Only a theta can skip, and only once the item may prune; with none the walk evaluated every live
      // matching doc exactly once.
      const bool pruned = s.prune_on != 0u && (s.theta != 0ull || __hip_atomic_load(my_theta_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull);
      as_global(ape->item_hits)[out_slot] = (uint64_t)hits + (pruned ? kHitsPrunedUnit : 0ull);
      if (PROF && ape->item_prof) {
  #ifndef NRT_MS_PHASE_CLOCKS
        s.prof[5] = hits;
  #endif
        const uint64_t t_end = __builtin_readcyclecounter();
        s.prof[9] = t_end - t_item0;
        s.prof[15] = t_end - t_epi0;
        for (int i = 0; i < 16; ++i) as_global(ape->item_prof)[(size_t)out_slot * 16 + i] = s.prof[i];
        if (hpe.walls) {
          unsigned long long* const wr = as_global(hpe.walls) + (size_t)out_slot * 8;
          wr[0] = wall0;                 // the item's prologue begins (role chosen, plan records read)
          wr[1] = wall_clock64();        // the item is done
          wr[2] = (unsigned long long)my_item | ((unsigned long long)item.flags << 32);   // (+ the item's flags: mode, window size, windows)
          wr[3] = s.prof[0];             // windows walked here
          wr[4] = wall_entry;            // this round began (a fresh workgroup: its first instruction)
          // which CU: XCC_ID[3:0] | HW_ID's se_id[15:13], sh_id[12], cu_id[11:8]
          wr[5] = ((unsigned long long)((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 8) |
                  (unsigned long long)(((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 255u);
          wr[6] = (unsigned long long)round | ((unsigned long long)item.query << 32);
          wr[7] = blockIdx.x;
        }
      }
    }
    if (hpe.persistent == 0u) return;
    __syncthreads();   // (everybody has read what the next round's prologue rewrites)
  }
}

// ------------------------------------------------------------------------------------------------
// Seal-time kernels of the MaxScore route.
//
// term_frontier_kernel: per term its DTermAux record: the impact frontier (min_norm / esc_*) from the score codes
// fold_norms_kernel wrote (before any liveDocs are folded in), and where its lookup structure is
// (t_look[t] = byte offset inside the group's lookup buffer, ~0: none; t_meta[t] = kind | log2 docs per cell << 8).
// One workgroup per term.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void term_frontier_kernel(const uint32_t* __restrict__ fnorm, const uint64_t* __restrict__ t_start,
                          const uint32_t* __restrict__ t_count, const uint64_t* __restrict__ t_look,
                          const uint32_t* __restrict__ t_meta, const char* __restrict__ look_base, DTermAux* __restrict__ out) {
  __shared__ uint32_t mn[13];
  __shared__ uint32_t mf;
  const uint32_t t = blockIdx.x;
  if (threadIdx.x < 13) mn[threadIdx.x] = 0xFFu;
  if (threadIdx.x == 13) mf = 0u;
  __syncthreads();
  const uint64_t st = t_start[t];
  const uint32_t n = t_count[t];
  uint32_t lmn[13];
  uint32_t lmf = 0;
#pragma unroll
  for (int i = 0; i < 13; ++i) lmn[i] = 0xFFu;
  for (uint32_t p = threadIdx.x; p < n; p += 256u) {
    const uint32_t c = fnorm[st + p];
    if (c >> 31) {
      lmn[12] = min(lmn[12], c & 255u);
      lmf = max(lmf, (c >> 8) & 0x3FFFFFu);
    } else {
      const uint32_t f = (c >> 9) & 15u, nb = (c >> 2) & 127u;
#pragma unroll
      for (int i = 0; i < 12; ++i)
        if (f == (uint32_t)(i + 1)) lmn[i] = min(lmn[i], nb);
    }
  }
#pragma unroll
  for (int i = 0; i < 13; ++i)
    if (lmn[i] != 0xFFu) atomicMin(&mn[i], lmn[i]);
  if (lmf) atomicMax(&mf, lmf);
  __syncthreads();
  if (threadIdx.x == 0) {
    DTermAux a;
    const bool has = t_look[t] != ~0ull;
    a.look = has ? (const void*)(look_base + t_look[t]) : nullptr;
    for (int i = 0; i < 12; ++i) a.min_norm[i] = (uint8_t)mn[i];
    a.esc_min_norm = (uint8_t)mn[12];
    a.look_kind = has ? (uint8_t)(t_meta[t] & 255u) : (uint8_t)kLookNone;
    a.look_shift = has ? (uint8_t)((t_meta[t] >> 8) & 255u) : (uint8_t)0;
    a.pad = 0;
    a.esc_max_freq = mf;
    a.pad2 = 0;
    out[t] = a;
  }
}

// term_bits_kernel: MEMBERSHIP + RANK RECORDS (plan.h: kLookBits): per 32 docs {doc bits, postings of the term before the
// block}.  Grid: (chunks, terms with records: `which`); the records were zeroed.  Postings are ascending in docid, so the first posting of a block is the one
// whose predecessor lies in an earlier block: it records its index.
__global__ __launch_bounds__(256)
void term_bits_kernel(const uint32_t* __restrict__ docids, const uint64_t* __restrict__ t_start, const uint32_t* __restrict__ t_count,
                      const uint64_t* __restrict__ t_look, const uint32_t* __restrict__ which, char* __restrict__ look_base) {
  const uint32_t t = which[blockIdx.y];
  const uint64_t st = t_start[t];
  const uint32_t n = t_count[t];
  uint32_t* const r = (uint32_t*)(look_base + t_look[t]);
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    const uint32_t d = docids[st + p];
    const uint32_t blk = d >> 5;
    atomicOr(&r[(size_t)blk * 2u], 1u << (d & 31u));
    if (p == 0u || (docids[st + p - 1u] >> 5) != blk) r[(size_t)blk * 2u + 1u] = p;
  }
}

// term_cells_kernel: the LOOKUP CELLS of the other terms (plan.h: kLookCells): entry i = the term's postings with a docid
// below i << shift (a lower bound in its docid column), for i = 0 .. cells.  Grid: (chunks, terms with cells).
__global__ __launch_bounds__(256)
void term_cells_kernel(const uint32_t* __restrict__ docids, const uint64_t* __restrict__ t_start, const uint32_t* __restrict__ t_count,
                       const uint64_t* __restrict__ t_look, const uint32_t* __restrict__ t_meta, const uint32_t* __restrict__ which,
                       uint32_t max_doc, char* __restrict__ look_base) {
  const uint32_t t = which[blockIdx.y];
  const uint32_t* const dd = docids + t_start[t];
  const uint32_t n = t_count[t];
  const uint32_t shift = (t_meta[t] >> 8) & 255u;
  const uint32_t n_cells = ((max_doc - 1u) >> shift) + 1u;
  uint32_t* const out = (uint32_t*)(look_base + t_look[t]);
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n_cells; i += stride) {
    const uint64_t bound = (uint64_t)i << shift;   // (the last entry's bound may pass 2^32)
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if ((uint64_t)dd[mid] < bound) lo = mid + 1u;
      else hi = mid;
    }
    out[i] = lo;
  }
}

// ---- launchers ---------------------------------------------------------------------------------------
void launch_bm25_maxscore(hipStream_t stream, bool profile, bool packed, int shapes, const MsArgs& args, const MsArgs* args_d) {
  const uint32_t n_items = args.help.n_own;
  if (n_items == 0) return;
  // (args: the host's copy of *args_d, the record the kernel reads)
  // persistent: one workgroup per CU, each choosing work until none is left; else one workgroup per item + the helpers behind them
  const uint32_t grid = args.help.persistent ? std::min(n_items + args.help.n_help, std::max(args.help.n_cus, 1u)) : n_items + args.help.n_help;
#define NRT_MS_LAUNCH(P, K, S) hipLaunchKernelGGL((bm25_maxscore_kernel<P, K, S>), dim3(grid), dim3(kMsThreads), 0, stream, args_d)
#define NRT_MS_LAUNCH_S(P, K)          \
  do {                                 \
    if (shapes == 2) NRT_MS_LAUNCH(P, K, 2);      \
    else if (shapes == 1) NRT_MS_LAUNCH(P, K, 1); \
    else NRT_MS_LAUNCH(P, K, 0);                  \
  } while (0)
#ifdef NRTGPU_DEV   // the instrumented instantiations exist in the development build only (include/nrtgpu_dev.h)
  if (profile) {
    if (packed) NRT_MS_LAUNCH_S(true, true);
    else NRT_MS_LAUNCH_S(true, false);
    return;
  }
#else
  (void)profile;
#endif
  if (packed) NRT_MS_LAUNCH_S(false, true);
  else NRT_MS_LAUNCH_S(false, false);
#undef NRT_MS_LAUNCH_S
#undef NRT_MS_LAUNCH
}

void launch_term_frontier(hipStream_t stream, const uint32_t* fnorm, const uint64_t* t_start, const uint32_t* t_count,
                          const uint64_t* t_look, const uint32_t* t_meta, const void* look_base, uint32_t n_terms, DTermAux* out) {
  if (n_terms == 0) return;
  hipLaunchKernelGGL(term_frontier_kernel, dim3(n_terms), dim3(256), 0, stream, fnorm, t_start, t_count, t_look, t_meta, (const char*)look_base, out);
}

void launch_term_bits(hipStream_t stream, const uint32_t* docids, const uint64_t* t_start, const uint32_t* t_count, const uint64_t* t_look,
                      const uint32_t* which, uint32_t n_which, uint32_t max_count, void* look_base) {
  if (n_which == 0) return;
  uint32_t chunks = (max_count + 256u * 16u - 1u) / (256u * 16u);
  chunks = chunks < 1u ? 1u : (chunks > 1024u ? 1024u : chunks);
  for (uint32_t off = 0; off < n_which; off += 65535u) {   // (gridDim.y holds 65535 at most)
    const uint32_t n = n_which - off < 65535u ? n_which - off : 65535u;
    hipLaunchKernelGGL(term_bits_kernel, dim3(chunks, n), dim3(256), 0, stream, docids, t_start, t_count, t_look, which + off, (char*)look_base);
  }
}

void launch_term_cells(hipStream_t stream, const uint32_t* docids, const uint64_t* t_start, const uint32_t* t_count, const uint64_t* t_look,
                       const uint32_t* t_meta, const uint32_t* which, uint32_t n_which, uint32_t max_cells, uint32_t max_doc, void* look_base) {
  if (n_which == 0) return;
  uint32_t chunks = (max_cells + 256u * 4u - 1u) / (256u * 4u);
  chunks = chunks < 1u ? 1u : (chunks > 256u ? 256u : chunks);
  for (uint32_t off = 0; off < n_which; off += 65535u) {
    const uint32_t n = n_which - off < 65535u ? n_which - off : 65535u;
    hipLaunchKernelGGL(term_cells_kernel, dim3(chunks, n), dim3(256), 0, stream, docids, t_start, t_count, t_look, t_meta, which + off, max_doc,
                       (char*)look_base);
  }
}

}  // namespace nrtgpu
