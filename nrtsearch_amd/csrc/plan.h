// plan.h -- launch-plan records shared by the host runtime and the gfx950 kernels.
// A "plan" is what one batch call uploads: per query, per (query, doc-range) work item, per
// (item, term) posting columns.  Names follow the reference's domain: leaves/segments, postings,
// slices (here: items), collectors.
#pragma once
#include <stdint.h>

namespace nrtgpu {

// Workgroup shape of the scan: waves x docs-per-sub-tile must fit 96 KiB of fp64 accumulators.
// 12 x 1024 (3 waves per SIMD, 168 VGPRs each) measured faster than 16 x 768 (4 per SIMD, 128 VGPRs:
// spills) on MI355X; a build-time choice (-DNRT_SCAN_WAVES / -DNRT_TILE_DOCS).
#ifndef NRT_SCAN_WAVES
#define NRT_SCAN_WAVES 12
#endif
#ifndef NRT_TILE_DOCS
#define NRT_TILE_DOCS 1024
#endif
constexpr int kTileDocs = NRT_TILE_DOCS;        // docs per wave-private LDS accumulator sub-tile (fp64)
constexpr int kScanWaves = NRT_SCAN_WAVES;      // autonomous wave64 per workgroup, 1 workgroup per CU (whole 160 KiB LDS)
constexpr int kScanThreads = kScanWaves * 64;
static_assert(kScanWaves % 4 == 0 && kTileDocs % 64 == 0 && kTileDocs / 64 <= 16, "scan workgroup shape");
static_assert(kScanWaves * kTileDocs * 8 <= 96 * 1024, "accumulators exceed their LDS share");
constexpr int kCandCap = (kScanWaves == 12) ? 1920 : 1664;  // scan: LDS candidate slots (what the LDS budget leaves next to four normInverse tables: 15 / 13 KiB; 2176 / 1920 with two tables measured the same)
constexpr int kMergeCap = 2048;     // merge: candidate slots = kMaxK + kScanThreads
constexpr int kLdsCaches = 4;       // normInverse tables kept in LDS per item (one per field)
// Score tables: for the kTabTerms densest terms of a query the BM25 score of every
// (freq <= kTabMaxFreq, norm byte < kTabNorms) pair is computed once per item into LDS, so scoring a
// posting is one LDS read.  Other (freq, norm) pairs / terms take the division path.
constexpr int kTabTerms = 5;
constexpr int kTabMaxFreq = 12;
constexpr int kTabNorms = 128;
constexpr int kTabEntries = (kTabMaxFreq + 1) * kTabNorms;  // row 0 (freq 0) unused
constexpr int kMaxK = 1024;
constexpr int kMaxTerms = 32;
// Packed postings (NRTGPU_FLAG_PACKED_POSTINGS): ONE 32-bit word per posting instead of the docid and
// code columns -- the doc's offset inside its 2^20-doc super-window (bits 12-31; the super-window is known from the
// posting's cell: cell shifts are capped so that a cell never spans two) | a 12-bit score code:
//   code <  kPackEscBase : (freq << 7) | norm byte (freq <= kTabMaxFreq, norm < kTabNorms): table byte offset = code << 2
//   code >= kPackEscBase : an EXCEPTION (freq > kTabMaxFreq or norm >= kTabNorms): the low 11 bits of its number e in
//                          the upload group's exception list (exceptions numbered in posting order).  The list's
//                          directory holds, per block of 2^kPackEscBlockShift = 2048 postings, the number of
//                          exceptions before the block (e0): e = e0 + ((low bits - e0) & 2047) -- unique, a block
//                          holds at most 2048.  exceptions[e] is the 32-bit escape word 0x80000000 | freq << 8 | norm
//                          of the unpacked layout.  Memory: 4 B per exception + 4 B per 2048 postings.
// The list (DTerm.fnorm of a packed term): u32 header[4] = {n_blocks, n_exceptions, 0, 0}, u32 dir[n_blocks + 1],
// u32 exceptions[n_exceptions].
// liveDocs are NOT folded into packed postings (no spare bit): deletes stay a mask the scorers test.
constexpr uint32_t kPackDocBits = 20;
constexpr uint32_t kPackCodeBits = 12;
constexpr uint32_t kPackCodeMask = (1u << kPackCodeBits) - 1u;
constexpr uint32_t kPackDocMask = (1u << kPackDocBits) - 1u;
constexpr uint32_t kPackEscBase = (uint32_t)kTabEntries;                    // 1664
constexpr uint32_t kPackEscBlockShift = 11;
constexpr uint32_t kPackEscLowMask = (1u << kPackEscBlockShift) - 1u;
static_assert(kPackEscBase + kPackEscLowMask < (1u << kPackCodeBits), "exception codes must fit the 12-bit code");
constexpr uint32_t kPackMaxCellShift = kPackDocBits - 10;                   // cell = tile >> shift with 1024-doc tiles: <= 2^20 docs
static_assert(kTileDocs == 1024 || kPackMaxCellShift > 0, "packed postings assume 1024-doc sub-tiles");

struct DTermAux;
// One query term inside one segment: where its posting columns live in HBM.
struct alignas(16) DTerm {
  const uint32_t* docids;    // docid column (all terms of the upload group, concatenated)
  const uint32_t* fnorm;     // score-code column: byte offset ((freq << 7 | norm) << 2) into a score table when
                             // freq <= kTabMaxFreq and norm < kTabNorms, else 0x80000000 | freq << 8 | norm.
                             // Packed postings: `docids` is the packed column and this is the group's exception list
  const uint32_t* cell_off;  // (n_cells + 1) posting offsets relative to `start`, one per doc-range cell
  uint64_t start;            // index of the term's first posting in the columns
  const DTermAux* aux;       // seal-time facts of the term the MaxScore route uses (resident next to the columns)
  uint32_t shift;            // cell = tile >> shift (0 for dense terms: one cell per tile)
  float    weight;           // boost * idf
  uint32_t cache_slot;       // per-query normInverse table index (< kLdsCaches)
  uint32_t tab_slot;         // bits 0-15: score table of this term in the item's LDS (< kTabTerms) or kTabSlotNone; bit 16: MUST clause
  int32_t  fx_scale;         // fixed-point batches: the term's scores are integers < 2^32 after * 2^fx_scale ...
  uint32_t fx_shift;         // ... and enter the query's common scale 2^-fx_E shifted left by fx_shift
};
static_assert(sizeof(DTerm) == 64, "DTerm layout");

// The host uploads a COMPACT plan: per query clause one DQTerm naming the term's resident per-leaf table (the static
// half of a DTerm for every leaf of the leaf set, written once per term: planner.cpp, LeafSetCache) plus the query's
// side of it; expand_terms_kernel writes the DTerm records of every (query, leaf) the scorers read.  The host touches
// one cache line per clause instead of one per (clause, leaf).
struct alignas(16) DQTerm {
  const DTerm* table;        // n_leaves entries; docids == nullptr: the leaf lacks the term; `weight` holds the posting count (bits)
  float    weight;           // boost * idf
  uint32_t cache_slot;
  uint32_t tab_slot;
  int32_t  fx_scale;
  uint32_t fx_shift;
  uint32_t pad;
};
static_assert(sizeof(DQTerm) == 32, "DQTerm layout");
struct alignas(16) DQExpand {
  uint32_t term_begin;       // the query's DQTerms
  uint32_t n_terms;
  uint32_t by_weight;        // order of a (query, leaf)'s DTerms: 1 = heaviest clause first (MaxScore route), 0 = densest first
  uint32_t pad;
};
static_assert(sizeof(DQExpand) == 16, "DQExpand layout");

// What the MaxScore route (maxscore.hip) knows about a term besides its columns (one record per term of a segment,
// written at seal): the term's impact frontier -- per freq the smallest norm byte it occurs with -- from which the kernel
// takes the term's exact maximum score under the query's statistics (the role of Lucene's competitive (freq, norm) impacts,
// SURVEY 8a row a5) -- and a doc -> posting LOOKUP structure for the walk's later clauses:
//   kLookBits  : MEMBERSHIP + RANK RECORDS, per 32 docs {doc bits, postings of the term before the block}: whether the doc is
//                there and where its posting is; its score code is a second, dependent gather.  0.25 B per doc -- 16 KB per
//                65 536-doc window, and 2.5 MB for a whole 10 M-doc segment: the records of the terms that hundreds of a
//                batch's queries share stay in L2 / the Infinity Cache.
//   kLookCells : LOOKUP CELLS, one posting offset per 2^look_shift docs with look_shift chosen so that a cell holds
//                0.5 - 1 posting on average (4 - 8 B per POSTING): a search in the doc's cell, usually over no posting or one.
//   kLookNone  : the doc is searched for in its cell of the tile-granular table DTerm.cell_off (~4 - 8 postings per cell).
// Which term gets what: segment.cpp: build_term_aux (the segment's lookup budget, nrtgpu_config.lookup_budget_pct).
// Measured and dropped (round 5, profiles/r05_look_policy_matrix.log; kernel ms per 1024 C3 queries, same box): structures that
// answer in ONE gather instead of two -- a 16-bit code map per doc (2 B per doc) 2.97, a 4-bit freq map + the field's norm bytes
// (0.5 B per doc) 3.56 -- against 2.11 for the records: what a lookup costs is decided by how much of the structure the caches
// hold (a map of a frequent term is 8x / 2x the records plus nothing saved on the absent docs, the common answer), not by the
// number of dependent gathers.  No structure at all: 6.72; lookup cells for every term: 3.20.
// Deleted docs need no care here: a doc whose postings were re-coded to score 0 (apply_live_kernel) never survives the stream
// phase of the walk, so it is never looked up.
constexpr uint32_t kLookNone = 0, kLookBits = 1, kLookCells = 2;
struct alignas(16) DTermAux {
  const void* look;        // the records (uint32_t[2][max_doc / 32 + pad]) / the lookup cells (uint32_t[cells + 1], offsets
                           // relative to the term's first posting); nullptr: kLookNone
  uint8_t  min_norm[12];   // postings a score table can serve (freq f = 1..12, norm byte < 128): smallest norm byte
                           // seen with freq f at [f - 1]; 0xFF = no such posting
  uint8_t  esc_min_norm;   // the other postings (freq > 12 or norm byte >= 128): smallest norm byte ...
  uint8_t  look_kind;      // kLook*
  uint8_t  look_shift;     // kLookCells: log2 of the docs per cell
  uint8_t  pad;
  uint32_t esc_max_freq;   // ... and largest freq among them; 0 = none
  uint32_t pad2;
};
static_assert(sizeof(DTermAux) == 32, "DTermAux layout");

// Workgroup shape of the MaxScore route: 12 autonomous waves (168 VGPRs each); a wave owns a window of kMsWinTiles
// sub-tiles at a time (its docs' "already evaluated" bits: kMsWinDocs / 8 bytes of LDS).
#ifndef NRT_MS_WIN_TILES
#define NRT_MS_WIN_TILES 64
#endif
#ifndef NRT_MS_WAVES
#define NRT_MS_WAVES 12
#endif
#ifndef NRT_MS_SLOTS
#define NRT_MS_SLOTS 8
#endif
constexpr int kMsWaves = NRT_MS_WAVES;
constexpr int kMsSlots = NRT_MS_SLOTS;   // postings per lane and instruction: 8 (two 16-byte loads per column) or 4
static_assert(kMsSlots == 4 || kMsSlots == 8, "MaxScore kernel: 4 or 8 postings per lane");
constexpr int kMsThreads = kMsWaves * 64;
static_assert(kMsThreads <= 1024, "a workgroup holds at most 16 waves");
constexpr int kMsWinTiles = NRT_MS_WIN_TILES;
constexpr int kMsWinDocs = kMsWinTiles * kTileDocs;
constexpr int kMsMaxTerms = 8;      // clauses of a query on the MaxScore route (longer disjunctions are scanned exhaustively)
// A query whose hit count is not certain to pass the threshold (or that wants the exact count: ScoreMode.COMPLETE) may still
// run in the MaxScore kernel when it is SMALL (<= this many postings over all clauses and leaves): in EXACT mode the kernel
// skips nothing, so every matching doc is evaluated once and counted -- the exhaustive scan's answer, without walking every
// 1024-doc sub-tile of the shard for a handful of postings.
constexpr int64_t kMsExactMaxPostings = 1 << 18;
#ifndef NRT_MS_CAND_CAP
#define NRT_MS_CAND_CAP (NRT_MS_WIN_TILES > 32 ? 2304 : 3072)
#endif
constexpr int kMsCandCap = NRT_MS_CAND_CAP;  // LDS candidate slots (>= kMaxK + 512: a wave's retry always fits)
static_assert(kMsCandCap >= kMaxK + 512, "candidate buffer too small");
// item_hits of a MaxScore item: the docs it evaluated, plus kHitsPrunedUnit when it skipped anything (the count is
// then a lower bound).  The merge kernel's plain sum keeps both: low 48 bits = docs, high 16 = pruned items.
constexpr uint64_t kHitsPrunedUnit = 1ull << 48;
// Speculative thresholds (maxscore.hip: "speculation"; search.cpp: the re-run): a workgroup that has walked a fraction g of a
// query's doc windows holds the exact top-k of THOSE docs; if docs are spread over windows like a random sample, about k x g of
// the query's final top-k lie among them, so the (k g + z sqrt(k g))-th best key it holds is -- z standard deviations deep --
// below the final k-th key, and a far better theta than the k-th best of the docs seen so far.  It is a GUESS: everything it
// skips is only known to score below the guess.  The merge therefore checks it -- the merged list's k-th key must reach the
// largest guess published for the query (then nothing skipped could have entered) -- and tags the query's count with
// kHitsSpecInvalid otherwise; the host runs tagged queries again without speculation.  The results are exact either way.
constexpr uint64_t kHitsSpecInvalid = 1ull << 47;

// One part of a work item: a contiguous tile range of one segment (a LeafReaderContextPartition).
struct alignas(16) DPart {
  const uint64_t* live_bits;  // nullptr => every doc live
  uint32_t term_begin;        // first DTerm of (query, segment); terms sorted by postings, densest first
  uint32_t n_terms;
  uint32_t tile_begin, tile_end;
  uint32_t max_doc;
  int32_t  doc_base;
  uint32_t tile_offset;       // sub-tiles of the item's earlier parts: the item's sub-tiles form one sequence
  uint32_t slice;             // bits 0-23: the searcher slice (MyIndexSearcher.slices) the part's leaf belongs to, an index into the
                              // query's per-slice hit sums (DQuery.slice_base); bits 24-31: the slice's slot among the item's
                              // slices (< kSliceSlots): the item counts its hits per slot
};
static_assert(sizeof(DPart) == 48, "DPart layout");

// One work item == one workgroup == one collector: a query over a list of parts visited in docBase
// order, sharing one candidate buffer / theta -- the analogue of a Lucene LeafSlice
// (/root/reference/src/main/java/com/yelp/nrtsearch/server/search/MyIndexSearcher.java:163-208).
struct alignas(16) DItem {
  uint32_t query;             // batch-local query index
  uint32_t part_begin;
  uint32_t n_parts;
  uint32_t cache_off;         // offset in floats of the query's first normInverse table
  uint32_t n_caches;
  uint32_t n_tabs;            // score tables to build (<= kTabTerms)
  float    tab_weight[kTabTerms];
  uint32_t tab_cache[kTabTerms];  // normInverse table of each score table
  int32_t  tab_scale[kTabTerms];  // fixed-point batches: fx_scale of each score table's term
  int32_t  fx_E;                  // fixed-point batches: accumulators hold score * 2^fx_E
  uint32_t peer_slot;             // this item's slot among the query's items [DQuery.item_begin, + n_items)
  uint32_t flags;                 // MaxScore kernel, bits 0-1: when bounds may skip work (kMsMode*); bits 2-3: the item's doc windows hold
                                  // kMsWinTiles >> that many sub-tiles (planner.cpp: the launch's heaviest queries get finer ones);
                                  // bits 8-31: the item's doc windows (summed over its parts): what helpers share with the owner
};
static_assert(sizeof(DItem) == 96, "DItem layout");

struct alignas(16) DQuery {
  uint32_t k;
  uint32_t has_after;
  int32_t  after_doc;      // global docid
  float    after_score;
  uint32_t item_begin;     // items of this query are [item_begin, item_begin + n_items)
  uint32_t n_items;
  uint32_t min_should_match;  // > 1: only docs matched by that many clauses are hits (count-carrying kernel variant)
  uint32_t combine_max;       // 1: DisjunctionMaxQuery (tie breaker 0): a doc scores its best clause, not the sum (same variant)
  uint32_t gte_floor;         // max(totalHitsThreshold, numHits): a slice that collects more hits makes the relation
                              // GREATER_THAN_OR_EQUAL_TO (LazyQueueTopScoreDocCollector.java:176-199); ~0: never (ScoreMode.COMPLETE)
  uint32_t slice_base;        // the query's per-slice hit sums: slice_sum[slice_base + slice]
  uint32_t sec_mode;          // MaxScore kernel, SHAPES == 2: the doc's score needs a SECOND accumulator next to the sum (kMsSec*)
  float    tie_breaker;       // kMsSecTieBreaker: DisjunctionMaxQuery.tieBreakerMultiplier
};
static_assert(sizeof(DQuery) == 48, "DQuery layout");
// Hit counting (both scorers): an item counts the live matching docs of each searcher slice it touches (at most kSliceSlots
// of them: the planner cuts items there) in LDS and adds them to the query's per-slice sums at its end; slice_relation_kernel
// then applies the reference's rule -- one collector per slice, MyIndexSearcher.java:163-208 -- to the sums.
constexpr int kSliceSlots = 8;
// MaxScore kernel, DItem.flags bits 0-1: when may bounds skip work?
//   kMsModePrune : from the start (the planner KNOWS some slice passes max(totalHitsThreshold, numHits): what is reported is
//                  its certain lower bound)
//   kMsModeExact : never (ScoreMode.COMPLETE on a small query): every live matching doc is evaluated and counted
//   kMsModeCount : like Lucene's collector (LazyQueueTopScoreDocCollector.java:176-199: no min competitive score before
//                  totalHits has passed the threshold) -- exact counting until some slice's count passes gte_floor, from
//                  then on bounds skip and the count is a lower bound (relation GREATER_THAN_OR_EQUAL_TO)
constexpr uint32_t kMsModePrune = 0, kMsModeExact = 1, kMsModeCount = 2;
// DQuery.sec_mode: scores that are not ONE sum of the matching clauses' scores.  The walk's bounds stay bounds of the plain sum
// (either score is at most what the sum scores, up to the float rounding kMsSecReqOpt's threshold allows for); the second
// accumulator only enters the final key.
//   kMsSecTieBreaker : DisjunctionMaxQuery with a tie breaker > 0: second = the best clause; score = (float)(best + (sum - best) * tb)
//   kMsSecReqOpt     : MUST next to SHOULD clauses: second = the sum of the SHOULD clauses; score = (float)(sum - second) + (float)second;
//                      a doc that lacks a MUST clause (DTerm.tab_slot bit 16) is no hit
constexpr uint32_t kMsSecNone = 0, kMsSecTieBreaker = 1, kMsSecReqOpt = 2;
// DTerm.tab_slot: bits 0-15 the score table (0xFFFF: none), bit 16: a MUST clause
constexpr uint32_t kTabSlotNone = 0xFFFFu, kTabSlotRequired = 1u << 16;
// minimumNumberShouldMatch > 1: the fixed-point accumulator carries the number of matching clauses above
// the score sum (sum < 2^52: 32 clauses x 2^32 x 2^15)
constexpr int kMsmCountShift = 56;

// Helping (maxscore.hip): the windows of an item are handed out by a counter in GLOBAL memory, so a workgroup other than the
// item's own -- a HELPER -- can take windows of an item that is running.  The launch has n_own + n_help workgroups; the items
// themselves are a queue in launch order (item_next).  A workgroup that gets a CU starts the next item, or helps: an item on
// the launch's critical path while the queue still holds items, any unfinished item once it is empty -- the tail of a launch
// (1024 items of very different cost on 256 CUs, one workgroup per CU) is shared instead of waited for.  A helper has its own
// score tables, candidate list and output slot (slot_base + its number), starts from the theta the query's items have
// published (theta_g) and notes its query beside its slot (help_query) for the merge.
struct DHelp {
  uint32_t* win_next;            // [n_own]   windows handed out beyond the owner's first kMsWaves (zeroed per launch)
  uint32_t* help_cnt;            // [n_own]   helpers that joined the item
  unsigned long long* item_t0;   // [n_own]   wall clock (100 MHz) at which the item's owner started; 0: not yet
  uint32_t* help_head;           // [queries] (unused since round 6: the helpers' slots were a linked list per query)
  uint32_t* help_query;          // [n_help]  the query helper slot h worked for, + 1 (0: the slot was not used): what the merge reads
  uint32_t* help_off;            // [1]       a helper found nothing left worth joining: the later ones leave at once
  uint32_t* item_next;           // [1]       the next item of the launch order to be started
  uint32_t* help_used;           // [1]       helper slots handed out
  unsigned long long* t_start;   // [1]       wall clock at which item 0 started
  uint32_t n_own, n_help;        // items of the launch, workgroups launched beyond them
  uint32_t slot_base;            // output slot (item_keys / item_counts / item_hits) of helper 0
  uint32_t min_rem;              // bits 0-15: an item with fewer unassigned windows is not joined; bit 16: A/B, greedy choice
  uint32_t total_wins;           // windows of all items (the launch's progress = windows handed out / this)
  uint32_t alpha16;              // critical path: help while items are queued when an item's time left > alpha16 / 16 x the launch's; 0: never
  uint32_t n_cus;
  uint32_t persistent;           // 1: the launch has one workgroup per CU and each chooses work until none is left (maxscore.hip)
  unsigned long long* spec_g;    // [queries] the largest SPECULATIVE theta a workgroup has published for the query (kMsSpec*), 0 = none;
                                 // nullptr: no speculation in this launch
  uint32_t spec_z16, spec_sched; // the estimate's safety margin in standard deviations x 16; when estimates are due: bits 0-7 the
                                 // first one (doc windows begun by the workgroup), bits 8-15 the factor x 16 between one and the next
  unsigned long long* walls;     // instrumented kernel only, else nullptr: per output slot 8 words -- {start, end} on the 100 MHz
                                 // wall clock, item, windows walked, when the workgroup's round began, CU id, round, workgroup --
                                 // when every piece of the launch ran, on one time base (nrtgpu_get_maxscore_item_walls)
};

// What one launch of bm25_maxscore_kernel works on: ONE record next to the plan, the kernel's only argument (maxscore.hip reads
// its fields where it uses them).
struct DExchange;
struct MsArgs {
  const DItem* items;            // [help.n_own] in launch order
  const DPart* parts;
  const DTerm* terms;
  const DQuery* queries;
  const float* caches;           // the queries' normInverse tables
  unsigned long long* theta_g;   // per query: the best theta any of its items (or their helpers) has published
  uint32_t* slice_sum;           // per (query, searcher slice): hits counted
  uint32_t* q_prune;             // per query: a slice has passed the floor (kMsModeCount)
  const DExchange* xch;          // cross-GPU bound exchange, or nullptr
  uint64_t* item_keys;           // per output slot k_stride keys ...
  uint32_t* item_counts;         // ... how many ...
  uint64_t* item_hits;           // ... and the hits counted there
  uint64_t* item_prof;           // instrumented kernel: 16 counters per output slot, else nullptr
  const uint32_t* q_wins;        // per query: the doc windows of all its items (speculation: the denominator of "how much have I seen")
  uint32_t k_stride;
  uint32_t scatter;              // != 0: an item's doc windows are handed out in a SCATTERED order (maxscore.hip): whatever a workgroup
                                 // has walked so far is spread over the item's docs like a sample -- what the speculative thresholds
                                 // assume -- also where the docid order follows time or a sort key
  DHelp help;
};

// Cross-GPU bound exchange of one batch (nrtgpu_exchange_open): entry (rank r, query q) of the batch's
// slot holds (tag << 32) | score word that at least ceil(k / world) docs of rank r's shard reach.
constexpr int kExchangeSlots = 8;
struct alignas(16) DExchange {
  unsigned long long* slot;  // this batch's [world][stride] entries (host-mapped, system scope)
  uint32_t world, rank;
  uint32_t stride;           // entries per rank (max_batch)
  uint32_t tag;              // epoch tag, != 0
};
static_assert(sizeof(DExchange) == 32, "DExchange layout");

// Packed hit: larger key == better hit under Lucene's HitQueue order (score desc, doc asc).
// Scores are >= 0 (BM25 weights are non-negative) so float bits order like the floats.
// One leaf's vector field as the hybrid tail sees it (doc -> row lookups happen on the device).
struct DVecSeg {
  const float* vecs;          // n_vec x dim fp32, row-major; nullptr: the leaf has no vectors for the field
  const float* vnorm2;        // |v|^2 per row
  const int32_t* ord_to_doc;  // ascending; nullptr: row == docid
  int32_t doc_base, max_doc, n_vec, pad;
};
static_assert(sizeof(DVecSeg) == 40, "DVecSeg layout");

// A leaf as the sketch kernel sees it (knn.hip: knn_sketch_kernel walks ALL leaves of a search in one launch: an NRT index is
// dozens of segments, and a launch per leaf cost 17 % of a pass at 40 leaves, 84 % at 160).  The leaves' tiles of 16 rows are
// numbered through: leaf i holds tiles [tile_begin, tile_begin + ceil(n_rows / 16)).
struct alignas(16) DKnnLeaf {
  const void* sketch;          // the leaf's fp16 sketch (tile 0 first)
  const float* vnorm2;         // |v|^2 per row (allocation padded by 64 B: a tile's 16 norms are read whole)
  const int32_t* ord_to_doc;   // nullptr: row == docid
  const uint64_t* accept;      // liveDocs (& filter) bits of the leaf; nullptr: every doc
  int64_t tile_begin;
  int32_t n_rows, doc_base;
  float inv_rows_scale;        // 1 / the power of two the rows were multiplied by
  int32_t pad[3];
};
static_assert(sizeof(DKnnLeaf) == 64, "DKnnLeaf layout");

// Exact vector search: the matrix-core ESTIMATE of a score against the RESULT (the same similarity summed in the oracle's
// order).  e_abs bounds |estimate - result| -- for EUCLIDEAN of the squared distance (the score 1 / (1 + d2) flattens with d2: a
// bound in score units would be useless for far rows), else of the score with the boost applied; e_rel covers the roundings of
// the map to a score.  DESIGN 4.5.
//   knn_result_upper: no row whose estimate is <= m has a result above this.
//   knn_estimate_lower: every row whose result is >= s has an estimate of at least this.
__host__ __device__ inline double knn_result_upper(int sim, double m, double e_abs, double e_rel, double boost) {
  if (sim == 2) {
    if (!(m > 0.0)) return 0.0;
    const double d2 = boost / m - 1.0;
    const double lo = d2 - (e_abs + e_rel * (1.0 + d2));
    return boost / (1.0 + (lo > 0.0 ? lo : 0.0)) * (1.0 + 1e-6);
  }
  return (m + e_abs + e_rel * (m < 0.0 ? -m : m)) * (1.0 + 1e-6);
}
__host__ __device__ inline double knn_estimate_lower(int sim, double s, double e_abs, double e_rel, double boost) {
  if (sim == 2) {
    if (!(s > 0.0)) return 0.0;
    const double d2 = boost / s - 1.0;
    return boost / (1.0 + (d2 > 0.0 ? d2 : 0.0) + e_abs + e_rel * (1.0 + d2)) * (1.0 - 1e-6);
  }
  return (s - (e_abs + e_rel * (s < 0.0 ? -s : s))) * (1.0 - 1e-6);
}

__host__ __device__ inline uint64_t pack_key(float score, uint32_t global_doc) {
  union { float f; uint32_t u; } c;
  c.f = score;
  return ((uint64_t)c.u << 32) | (uint64_t)(0xFFFFFFFFu - global_doc);
}
__host__ __device__ inline float key_score(uint64_t key) {
  union { float f; uint32_t u; } c;
  c.u = (uint32_t)(key >> 32);
  return c.f;
}
__host__ __device__ inline uint32_t key_doc(uint64_t key) { return 0xFFFFFFFFu - (uint32_t)key; }

}  // namespace nrtgpu
