// dist.cpp -- the multi-GPU exchange behind the C ABI: one process per GPU, every rank searches its docid-range shard,
// one RCCL all-gather of the per-shard top-k (keys, counts, hit totals) over xGMI, TopDocs.merge on every rank
// (the reduce the reference does in LazyQueueTopScoreDocCollectorManager.java:137-144 over slices, here over
// GPUs).  A JVM caller has no torch: the collective lives in the library.  RCCL is bound at run time (dlopen), so the
// library carries no link-time dependency on it and single-GPU deployments never load it.
#include <dlfcn.h>

#include "runtime_internal.h"

namespace {
struct NcclUniqueId { char internal[128]; };   // rccl.h: ncclUniqueId
typedef void* NcclComm;
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclUniqueId, int);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, NcclComm, hipStream_t);
typedef int (*SendFn)(const void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*RecvFn)(void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*GroupFn)(void);
typedef const char* (*ErrStrFn)(int);
const int kNcclInt8 = 0, kNcclInt32 = 2, kNcclInt64 = 4;   // rccl.h: ncclDataType_t

struct Rccl {
  void* lib = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllGatherFn all_gather = nullptr;
  SendFn send = nullptr;   // (point-to-point: the all-to-all form of the exchange; absent in very old builds -> all-gather only)
  RecvFn recv = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GroupFn group_start = nullptr, group_end = nullptr;
  ErrStrFn err_str = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a process that already holds RCCL (e.g. torch's bundled copy) resolves to that one.  NRTGPU_RCCL_LIB (development build
    // only): the file to bind instead -- tests/mockrccl under a process that holds torch's RCCL as well (a path with a slash is
    // opened as that file, whatever library of the same soname is loaded)
    if (const char* forced = dev_env_str("NRTGPU_RCCL_LIB", nullptr)) r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (r.lib) break;
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.lib) return;
    r.get_unique_id = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
    r.all_gather = (AllGatherFn)dlsym(r.lib, "ncclAllGather");
    r.send = (SendFn)dlsym(r.lib, "ncclSend");
    r.recv = (RecvFn)dlsym(r.lib, "ncclRecv");
    r.comm_destroy = (CommDestroyFn)dlsym(r.lib, "ncclCommDestroy");
    r.group_start = (GroupFn)dlsym(r.lib, "ncclGroupStart");
    r.group_end = (GroupFn)dlsym(r.lib, "ncclGroupEnd");
    r.err_str = (ErrStrFn)dlsym(r.lib, "ncclGetErrorString");
  });
  return (r.lib && r.get_unique_id && r.comm_init_rank && r.all_gather && r.comm_destroy && r.group_start && r.group_end) ? &r : nullptr;
}
int nccl_fail(const char* what, int rc) {
  Rccl* r = rccl();
  return fail(NRTGPU_ERR_HIP, "%s failed: %s", what, (r && r->err_str) ? r->err_str(rc) : "RCCL error");
}
}  // namespace

struct nrtgpu_dist {
  NcclComm comm = nullptr;
  int32_t world = 0, rank = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_wait = nullptr;   // NRTGPU_FLAG_BLOCKING_WAIT
  DevBuf local, gathered;   // [keys | hits | counts] of this rank / of every rank
  DevBuf stage;             // hybrid: the merged first pass and this rank's rescored windows
  DevBuf my_guess;          // BM25: this rank's largest speculative threshold per query (nrtgpu_dist_search_bm25_batch_mode)
  DevBuf guess;             // ... the verdicts on them: mine, then every rank's (all-to-all form)
  DevBuf status;            // this rank's status word of an exchange (exchange_lists), then every rank's
  std::mutex mu;            // one collective (one user of `gathered`) at a time per communicator
  std::mutex call_mu;       // one whole search call (one user of `local`) at a time: taken before `mu`
};

extern "C" int nrtgpu_dist_unique_id(void* out128) {
  if (!out128) return fail(NRTGPU_ERR_INVALID_ARG, "NULL argument");
  Rccl* r = rccl();
  if (!r) return fail(NRTGPU_ERR_UNSUPPORTED, "librccl.so could not be loaded");
  NcclUniqueId id;
  if (int rc = r->get_unique_id(&id)) return nccl_fail("ncclGetUniqueId", rc);
  memcpy(out128, &id, sizeof(id));
  return NRTGPU_OK;
}

extern "C" int nrtgpu_dist_init(nrtgpu_ctx* ctx, int32_t world, int32_t rank, const void* id128) {
  if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return fail(NRTGPU_ERR_INVALID_ARG, "bad dist arguments");
  if (ctx->dist) return fail(NRTGPU_ERR_STATE, "a communicator is already open on this context");
  Rccl* r = rccl();
  if (!r) return fail(NRTGPU_ERR_UNSUPPORTED, "librccl.so could not be loaded");
  HIP_TRY(hipSetDevice(ctx->device));
  auto d = std::make_unique<nrtgpu_dist>();
  d->world = world;
  d->rank = rank;
  NcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  if (int rc = r->comm_init_rank(&d->comm, world, id, rank)) return nccl_fail("ncclCommInitRank", rc);
  hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_wait, hipEventBlockingSync | hipEventDisableTiming);
  if (e != hipSuccess) {   // (the communicator must not outlive a failed init)
    (void)r->comm_destroy(d->comm);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    return fail(NRTGPU_ERR_HIP, "nrtgpu_dist_init: %s", hipGetErrorString(e));
  }
  ctx->dist = d.release();
  return NRTGPU_OK;
}

extern "C" void nrtgpu_dist_close(nrtgpu_ctx* ctx) {
  if (!ctx || !ctx->dist) return;
  (void)hipSetDevice(ctx->device);
  nrtgpu_dist* d = ctx->dist;
  ctx->dist = nullptr;
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (Rccl* r = rccl())
    if (d->comm) (void)r->comm_destroy(d->comm);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  if (d->ev_wait) (void)hipEventDestroy(d->ev_wait);
  d->local.release();
  d->gathered.release();
  d->stage.release();
  d->guess.release();
  d->my_guess.release();
  d->status.release();
  delete d;
}

// ------------------------------------------------------------------------------------------------
// The exchange stage: this rank's device-resident shard results (keys n_queries x k_stride, counts, hit totals: what
// nrtgpu_search_bm25_batch_device[_epoch] or the vector search left in HBM) -> the other ranks -> TopDocs.merge.
//   mode NRTGPU_EXCHANGE_ALLGATHER : ONE grouped RCCL all-gather; every rank merges every query and holds every answer
//                                    (BASELINE.json's north star)
//   mode NRTGPU_EXCHANGE_ALLTOALL  : grouped ncclSend / ncclRecv -- rank r receives every rank's lists for ITS slice of the
//                                    batch, queries [r * n / W, (r + 1) * n / W), and merges only those: 1 / W of the bytes on
//                                    every link (xGMI is point-to-point: exactly what an all-to-all wants), 1 / W of the merge
//                                    and of the host-side unpacking.  The rank that owns a query answers its caller.
// Gathered layout [array][list][query]: what nrtgpu_merge_topk_device reads.
// ------------------------------------------------------------------------------------------------
struct Gathered {
  char* keys = nullptr;
  char* cnt = nullptr;
  char* hits = nullptr;
  int32_t first_q = 0, n_q = 0;   // the queries this rank merges
  std::vector<uint64_t> guess;    // [world][n_q] the shards' largest speculative thresholds for them (host; empty: none travelled)
  std::vector<int32_t> status;    // [world] what every rank said about ITS part of the call (0: fine; empty: no status travelled)
};

static void owned_range(int32_t world, int32_t rank, int32_t n_queries, int32_t mode, int32_t* first, int32_t* count) {
  if (mode == NRTGPU_EXCHANGE_ALLTOALL && world > 0 && n_queries % world == 0) {
    *count = n_queries / world;
    *first = rank * *count;
  } else {   // (a batch the ranks cannot share evenly is gathered whole)
    *first = 0;
    *count = n_queries;
  }
}

extern "C" int nrtgpu_dist_owned_range(nrtgpu_ctx* ctx, int32_t n_queries, int32_t mode, int32_t* first_query, int32_t* n_owned) {
  if (!ctx || !ctx->dist) return fail(NRTGPU_ERR_STATE, "nrtgpu_dist_init has not been called on this context");
  if (!first_query || !n_owned || n_queries <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad arguments");
  const bool p2p = rccl() && rccl()->send && rccl()->recv;
  owned_range(ctx->dist->world, ctx->dist->rank, n_queries, (mode == NRTGPU_EXCHANGE_ALLTOALL && p2p) ? mode : NRTGPU_EXCHANGE_ALLGATHER, first_query,
              n_owned);
  return NRTGPU_OK;
}

// (d->mu held by the caller)
// d_guess (may be NULL): this rank's largest speculative threshold per query (search.cpp: search_bm25_shard_device); they travel in
// the same group as the lists -- no collective, no wait of their own -- and arrive on the host in g->guess.
// my_status (kNoStatus: none travels): how this rank's part of the call went BEFORE the exchange -- the one-call entries
// (nrtgpu_dist_search_bm25_batch_mode ...) enter the exchange even when their local search failed (with empty lists), so that no peer
// is left waiting in a collective this rank would never issue; one more 4-byte all-gather in the same group, read on the host
// behind the same wait.  Every rank then sees every status and all of them return an error together (peers_failed).
const int32_t kNoStatus = INT32_MIN;
static int exchange_lists(nrtgpu_ctx* ctx, nrtgpu_dist* d, int32_t n_queries, int32_t k_stride, const void* d_keys, const void* d_counts,
                          const void* d_hits, const void* d_guess, int32_t mode, Gathered* g, int32_t my_status = kNoStatus) {
  Rccl* r = rccl();
  const size_t W = (size_t)d->world;
  int32_t* d_status = nullptr;   // [mine | every rank's]
  g->status.clear();
  if (my_status != kNoStatus) {
    if (int rc = d->status.reserve((W + 1) * 4)) return rc;
    d_status = (int32_t*)d->status.p;
    HIP_TRY(hipMemcpyAsync(d_status, &my_status, 4, hipMemcpyHostToDevice, d->stream));
  }
  if (mode == NRTGPU_EXCHANGE_ALLTOALL && !(r->send && r->recv)) mode = NRTGPU_EXCHANGE_ALLGATHER;
  owned_range(d->world, d->rank, n_queries, mode, &g->first_q, &g->n_q);
  const bool sliced = g->n_q != n_queries;
  const size_t nq = (size_t)g->n_q, kb = nq * (size_t)k_stride * 8, hb = nq * 8, cb = nq * 4;
  const size_t cb8 = (cb + 7) & ~(size_t)7;
  if (int rc = d->gathered.reserve((kb + hb + cb8 + hb) * W)) return rc;
  char* gb = (char*)d->gathered.p;
  g->keys = gb;
  g->hits = gb + W * kb;
  g->cnt = gb + W * (kb + hb);
  char* gg = gb + W * (kb + hb + cb8);   // the guesses: [world][n_q] like the hit totals
  if (!sliced) {
    if (int rc = r->group_start()) return nccl_fail("ncclGroupStart", rc);
    int rc1 = r->all_gather(d_keys, g->keys, kb / 8, kNcclInt64, d->comm, d->stream);
    int rc2 = r->all_gather(d_hits, g->hits, hb / 8, kNcclInt64, d->comm, d->stream);
    int rc3 = r->all_gather(d_counts, g->cnt, cb / 4, kNcclInt32, d->comm, d->stream);
    int rc4 = d_guess ? r->all_gather(d_guess, gg, hb / 8, kNcclInt64, d->comm, d->stream) : 0;
    int rc5 = d_status ? r->all_gather(d_status, d_status + 1, 1, kNcclInt32, d->comm, d->stream) : 0;
    if (int rc = r->group_end()) return nccl_fail("ncclGroupEnd", rc);
    if (rc1 || rc2 || rc3 || rc4 || rc5) return nccl_fail("ncclAllGather", rc1 ? rc1 : (rc2 ? rc2 : (rc3 ? rc3 : (rc4 ? rc4 : rc5))));
  } else {
    // my lists for peer p's slice go to p; p's lists for my slice arrive as list p.  My own slice: a device copy.
    const char* lk = (const char*)d_keys;
    const char* lh = (const char*)d_hits;
    const char* lc = (const char*)d_counts;
    const char* lg = (const char*)d_guess;
    int bad = 0;
    if (W > 1) {
      if (int rc = r->group_start()) return nccl_fail("ncclGroupStart", rc);
      for (int p = 0; p < d->world; ++p) {
        if (p == d->rank) continue;
        bad |= r->send(lk + (size_t)p * kb, kb / 8, kNcclInt64, p, d->comm, d->stream);
        bad |= r->recv(g->keys + (size_t)p * kb, kb / 8, kNcclInt64, p, d->comm, d->stream);
        bad |= r->send(lh + (size_t)p * hb, hb / 8, kNcclInt64, p, d->comm, d->stream);
        bad |= r->recv(g->hits + (size_t)p * hb, hb / 8, kNcclInt64, p, d->comm, d->stream);
        bad |= r->send(lc + (size_t)p * cb, cb / 4, kNcclInt32, p, d->comm, d->stream);
        bad |= r->recv(g->cnt + (size_t)p * cb, cb / 4, kNcclInt32, p, d->comm, d->stream);
        if (lg) {
          bad |= r->send(lg + (size_t)p * hb, hb / 8, kNcclInt64, p, d->comm, d->stream);
          bad |= r->recv(gg + (size_t)p * hb, hb / 8, kNcclInt64, p, d->comm, d->stream);
        }
        if (d_status) {
          bad |= r->send(d_status, 1, kNcclInt32, p, d->comm, d->stream);
          bad |= r->recv(d_status + 1 + p, 1, kNcclInt32, p, d->comm, d->stream);
        }
      }
      if (int rc = r->group_end()) return nccl_fail("ncclGroupEnd", rc);
      if (bad) return nccl_fail("ncclSend / ncclRecv", bad);
    }
    const size_t me = (size_t)d->rank;
    HIP_TRY(hipMemcpyAsync(g->keys + me * kb, lk + me * kb, kb, hipMemcpyDeviceToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(g->hits + me * hb, lh + me * hb, hb, hipMemcpyDeviceToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(g->cnt + me * cb, lc + me * cb, cb, hipMemcpyDeviceToDevice, d->stream));
    if (lg) HIP_TRY(hipMemcpyAsync(gg + me * hb, lg + me * hb, hb, hipMemcpyDeviceToDevice, d->stream));
    if (d_status) HIP_TRY(hipMemcpyAsync(d_status + 1 + me, d_status, 4, hipMemcpyDeviceToDevice, d->stream));
  }
  if (d_status) {
    g->status.resize(W);
    HIP_TRY(hipMemcpyAsync(g->status.data(), d_status + 1, W * 4, hipMemcpyDeviceToHost, d->stream));
  }
  g->guess.clear();
  if (d_guess) {
    g->guess.resize(W * nq);
    HIP_TRY(hipMemcpyAsync(g->guess.data(), gg, W * hb, hipMemcpyDeviceToHost, d->stream));
  }
  HIP_TRY(wait_for_stream((ctx->cfg.flags & NRTGPU_FLAG_BLOCKING_WAIT) != 0, d->stream, d->ev_wait));
  return NRTGPU_OK;
}

// `out` has n_queries entries (indexed like the batch); the entries of the queries this rank does not own (all-to-all) are
// marked n_hits = 0, total_hits = -1.
static void mark_not_owned(nrtgpu_topdocs* out, int32_t n_queries, const Gathered& g) {
  for (int32_t qi = 0; qi < n_queries; ++qi)
    if (qi < g.first_q || qi >= g.first_q + g.n_q) {
      out[qi].n_hits = 0;
      out[qi].total_hits = -1;
      out[qi].total_hits_is_lower_bound = 0;
    }
}

// After an exchange that carried the ranks' statuses: did anybody's part of the call fail?  Then EVERY rank returns an error --
// its own where it has one, else one that names the first failed peer -- and nobody goes on to a collective the failed rank
// would not issue.
static int peers_failed(const nrtgpu_dist* d, const Gathered& g, int32_t my_status, const std::string& my_error) {
  for (size_t w = 0; w < g.status.size(); ++w)
    if (g.status[w] != 0) {
      if (my_status != 0) return fail(my_status, "%s", my_error.c_str());
      return fail(NRTGPU_ERR_STATE, "rank %d of %d failed its part of the search (status %d): the call fails on every rank", (int)w, d->world, g.status[w]);
    }
  return NRTGPU_OK;
}

// The exchange, TopDocs.merge of the shards' lists, and -- with d_guess -- the check of the shards' speculative thresholds against
// the MERGED lists: the k-th key of a query's merged list must reach the largest guess any shard published for it; then nothing
// any shard skipped could have entered (kernels.hip: merge_topk_kernel applies the same rule to one call's list).  failed
// [n_queries]: the same verdicts on every rank (all-gather form: every rank holds every list and every guess; all-to-all: the
// owners' verdicts are all-gathered, one byte per query).
// my_status / my_error: how this rank's part of the call went before the exchange (kNoStatus: the caller vouches for it, nothing
// travels): see exchange_lists.
static int exchange_merge_checked(nrtgpu_ctx* ctx, int32_t n_queries, int32_t k_stride, const void* d_keys, const void* d_counts, const void* d_hits,
                                  const void* d_guess, const int32_t* ks, const int32_t* total_hits_thresholds, int32_t mode, nrtgpu_topdocs* out,
                                  uint8_t* failed, int32_t* n_failed, int32_t my_status, const std::string& my_error) {
  if (!ctx || !ctx->dist) return fail(NRTGPU_ERR_STATE, "nrtgpu_dist_init has not been called on this context");
  if (!d_keys || !d_counts || !d_hits || !ks || !total_hits_thresholds || !out || n_queries <= 0 || k_stride <= 0 || k_stride % 16 != 0)
    return fail(NRTGPU_ERR_INVALID_ARG, "bad arguments");
  if (d_guess && (!failed || !n_failed)) return fail(NRTGPU_ERR_INVALID_ARG, "guesses without a place for the verdicts");
  if (mode != NRTGPU_EXCHANGE_ALLGATHER && mode != NRTGPU_EXCHANGE_ALLTOALL) return fail(NRTGPU_ERR_INVALID_ARG, "unknown exchange mode %d", mode);
  nrtgpu_dist* d = ctx->dist;
  HIP_TRY(hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> lk(d->mu);   // one collective (and one user of the gathered buffer) at a time per communicator
  Gathered g;
  if (int rc = exchange_lists(ctx, d, n_queries, k_stride, d_keys, d_counts, d_hits, d_guess, mode, &g, my_status)) return rc;
  if (int rc = peers_failed(d, g, my_status, my_error)) return rc;
  mark_not_owned(out, n_queries, g);
  // (the verdicts below are taken from the MERGED keys -- the same on every rank that holds the same lists -- never from the caller's
  //  output arrays: those may be absent or shorter than k, and then differ from rank to rank.  ADVICE round 5.)
  std::vector<uint64_t> kth(d_guess ? (size_t)g.n_q : 0);
  if (int rc = merge_topk_device_kth(ctx, d->world, g.n_q, k_stride, g.keys, g.cnt, g.hits, ks + g.first_q, total_hits_thresholds + g.first_q,
                                     out + g.first_q, d_guess ? kth.data() : nullptr))
    return rc;
  if (n_failed) *n_failed = 0;
  if (!d_guess) return NRTGPU_OK;
  const size_t W = (size_t)d->world, nq = (size_t)g.n_q;
  memset(failed, 0, (size_t)n_queries);
  for (size_t q = 0; q < nq; ++q) {
    uint64_t gmax = 0;
    for (size_t w = 0; w < W; ++w) gmax = std::max(gmax, g.guess[w * nq + q]);
    if (gmax == 0) continue;   // (nobody guessed)
    failed[(size_t)g.first_q + q] = kth[q] < gmax ? 1 : 0;   // (kth 0: fewer than k hits in the merged list)
  }
  if (g.n_q != n_queries) {   // (sliced: every rank learns every owner's verdicts; the slices are disjoint, so their OR is their union)
    Rccl* r = rccl();
    if (int rc = d->guess.reserve((size_t)n_queries * (W + 1))) return rc;
    unsigned char* d_mine = (unsigned char*)d->guess.p;
    unsigned char* d_all = d_mine + (size_t)n_queries;
    std::vector<unsigned char> all((size_t)n_queries * W);
    HIP_TRY(hipMemcpyAsync(d_mine, failed, (size_t)n_queries, hipMemcpyHostToDevice, d->stream));
    if (int rc = r->group_start()) return nccl_fail("ncclGroupStart", rc);
    const int rc1 = r->all_gather(d_mine, d_all, (size_t)n_queries, kNcclInt8, d->comm, d->stream);
    if (int rc = r->group_end()) return nccl_fail("ncclGroupEnd", rc);
    if (rc1) return nccl_fail("ncclAllGather", rc1);
    HIP_TRY(hipMemcpyAsync(all.data(), d_all, all.size(), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(wait_for_stream((ctx->cfg.flags & NRTGPU_FLAG_BLOCKING_WAIT) != 0, d->stream, d->ev_wait));
    for (size_t w = 0; w < W; ++w)
      for (size_t q = 0; q < (size_t)n_queries; ++q) failed[q] |= all[w * (size_t)n_queries + q];
  }
  int32_t nf = 0;
  for (int32_t q = 0; q < n_queries; ++q) nf += failed[q] != 0;
  *n_failed = nf;
  return NRTGPU_OK;
}

extern "C" int nrtgpu_dist_exchange_merge_checked(nrtgpu_ctx* ctx, int32_t n_queries, int32_t k_stride, const void* d_keys,
                                                  const void* d_counts, const void* d_hits, const void* d_guess, const int32_t* ks,
                                                  const int32_t* total_hits_thresholds, int32_t mode, nrtgpu_topdocs* out,
                                                  uint8_t* failed, int32_t* n_failed) {
  return exchange_merge_checked(ctx, n_queries, k_stride, d_keys, d_counts, d_hits, d_guess, ks, total_hits_thresholds, mode, out, failed, n_failed,
                                kNoStatus, std::string());
}

extern "C" int nrtgpu_dist_exchange_merge(nrtgpu_ctx* ctx, int32_t n_queries, int32_t k_stride, const void* d_keys, const void* d_counts,
                                          const void* d_hits, const int32_t* ks, const int32_t* total_hits_thresholds, int32_t mode,
                                          nrtgpu_topdocs* out) {
  return nrtgpu_dist_exchange_merge_checked(ctx, n_queries, k_stride, d_keys, d_counts, d_hits, nullptr, ks, total_hits_thresholds, mode, out, nullptr,
                                            nullptr);
}

extern "C" int nrtgpu_dist_allgather_merge(nrtgpu_ctx* ctx, int32_t n_queries, int32_t k_stride, const void* d_keys,
                                           const void* d_counts, const void* d_hits, const int32_t* ks,
                                           const int32_t* total_hits_thresholds, nrtgpu_topdocs* out) {
  return nrtgpu_dist_exchange_merge(ctx, n_queries, k_stride, d_keys, d_counts, d_hits, ks, total_hits_thresholds, NRTGPU_EXCHANGE_ALLGATHER, out);
}

// A rank whose part of the call failed enters the exchange with EMPTY lists (counts, hit totals and guesses zeroed; the keys behind
// a zero count are never read) next to its status word.
static int empty_lists(nrtgpu_ctx* ctx, int32_t n_queries, void* d_counts, void* d_hits, void* d_guess) {
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemset(d_counts, 0, (size_t)n_queries * 4));
  HIP_TRY(hipMemset(d_hits, 0, (size_t)n_queries * 8));
  if (d_guess) HIP_TRY(hipMemset(d_guess, 0, (size_t)n_queries * 8));
  return NRTGPU_OK;
}

// This rank's buffer for a whole call: [keys | hits | counts] of n_queries queries.  Calls on one communicator are serialised
// (`call_mu`): the buffer belongs to the call from the local search to the end of the exchange.
static int local_lists(nrtgpu_dist* d, int32_t n_queries, int32_t k_stride, char** keys, char** hits, char** cnts) {
  const size_t nq = (size_t)n_queries, kb = nq * (size_t)k_stride * 8, hb = nq * 8, cb = nq * 4;
  if (int rc = d->local.reserve(kb + hb + ((cb + 7) & ~(size_t)7))) return rc;
  *keys = (char*)d->local.p;
  *hits = *keys + kb;
  *cnts = *hits + hb;
  return NRTGPU_OK;
}

// Every rank calls this with the same queries in the same order (index-global statistics in the weights) over ITS
// leaves.  mode: who receives which answer (above).
extern "C" int nrtgpu_dist_search_bm25_batch_mode(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                                  const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t mode, nrtgpu_topdocs* out) {
  if (!ctx || !ctx->dist) return fail(NRTGPU_ERR_STATE, "nrtgpu_dist_init has not been called on this context");
  if (!queries || !out || n_queries <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad arguments");
  const bool no_spec = (mode & NRTGPU_EXCHANGE_NO_SPECULATION) != 0;
  mode &= ~NRTGPU_EXCHANGE_NO_SPECULATION;
  nrtgpu_dist* d = ctx->dist;
  HIP_TRY(hipSetDevice(ctx->device));
  int32_t kmax = 1;
  for (int qi = 0; qi < n_queries; ++qi) kmax = std::max(kmax, queries[qi].k);
  const int32_t k_stride = (int32_t)round_up((uint32_t)std::min(kmax, NRTGPU_MAX_K), 16);
  std::lock_guard<std::mutex> call(d->call_mu);   // the local buffer is this call's until its exchange is over
  char *lk = nullptr, *lh = nullptr, *lc = nullptr;
  if (int rc = local_lists(d, n_queries, k_stride, &lk, &lh, &lc)) return rc;
  std::vector<int32_t> ks((size_t)n_queries), thr((size_t)n_queries);
  for (int32_t q = 0; q < n_queries; ++q) {
    ks[(size_t)q] = queries[q].k;
    thr[(size_t)q] = queries[q].total_hits_threshold;
  }
  // 1. this rank's shard: top-k per query stays in HBM (synchronous: complete when it returns).  Its speculative thresholds are
  //    guesses at the k-th score of the WHOLE search (a shard's docs are a 1 / world sample of the index; search.cpp:
  //    enqueue_search, spec_world): every rank converges on the global threshold from its own docs, collects about k / world
  //    candidates instead of k, and nothing is exchanged for it.  A guess is only a guess: the largest one of any rank is checked
  //    against the list merged over all ranks (step 2), and a query whose guess failed is run again on every rank without
  //    speculation (step 3) -- the answer is exact either way.
  const bool may_speculate = d->world > 1 && !no_spec;
  uint64_t* d_guess = nullptr;
  if (may_speculate) {
    if (int rc = d->my_guess.reserve((size_t)n_queries * 8)) return rc;
    d_guess = (uint64_t*)d->my_guess.p;
  }
  bool speculated = false;
  // (a rank whose shard search fails -- its thread's deadline, a planner refusal, a device error -- still enters the exchange, with
  //  empty lists and its status: the peers are already on their way into that collective and would wait in it for ever.  Every rank
  //  then returns an error from step 2.  ADVICE round 5.)
  int32_t my_status = 0;
  std::string my_error;
  if (int rc = search_bm25_shard_device(ctx, segs, doc_bases, n_segs, queries, n_queries, k_stride, lk, lc, lh, may_speculate ? d->world : 0, d_guess,
                                        &speculated)) {
    my_status = rc;
    my_error = g_last_error;
    if (int rc2 = empty_lists(ctx, n_queries, lc, lh, d_guess)) return rc2;
  }
  // 2. the exchange over xGMI (the guesses and the statuses ride in the same group), TopDocs.merge of the shards' lists, the check
  std::vector<uint8_t> failed((size_t)n_queries, 0);
  int32_t n_failed = 0;
  if (int rc = exchange_merge_checked(ctx, n_queries, k_stride, lk, lc, lh, d_guess, ks.data(), thr.data(), mode, out, d_guess ? failed.data() : nullptr,
                                      d_guess ? &n_failed : nullptr, my_status, my_error))
    return rc;
  if (speculated) note_shard_speculation(ctx, segs, n_segs, n_queries, n_failed);   // (this rank's verdict on ITS leaf set)
  if (n_failed == 0) return NRTGPU_OK;
  // 3. the queries whose guess failed (the same ones on every rank): every rank runs them again on its shard without speculation;
  //    the lists are gathered whole (a handful of queries: no point in slicing them) and the answers replace the first ones where
  //    this rank holds them
  std::vector<int32_t> again;
  for (int32_t q = 0; q < n_queries; ++q)
    if (failed[(size_t)q]) again.push_back(q);
  std::vector<nrtgpu_bm25_query> rq(again.size());
  std::vector<nrtgpu_topdocs> ro(again.size());
  std::vector<int32_t> rks(again.size()), rthr(again.size());
  for (size_t i = 0; i < again.size(); ++i) {
    rq[i] = queries[again[i]];
    ro[i] = out[again[i]];
    rks[i] = rq[i].k;
    rthr[i] = rq[i].total_hits_threshold;
  }
  // (the re-run is the tail of a search that was launched in time: it ignores the thread's deadline, as the second pass of a single
  //  GPU's speculative call does -- a rank that timed out HERE would leave between two collectives; and its status travels with
  //  the lists like the first pass's)
  struct DeadlineOff {
    int64_t saved;
    DeadlineOff() : saved(g_deadline_ns) { g_deadline_ns = 0; }
    ~DeadlineOff() { g_deadline_ns = saved; }
  } deadline_off;
  if (int rc = local_lists(d, (int32_t)again.size(), k_stride, &lk, &lh, &lc)) return rc;
  my_status = 0;
  if (int rc = search_bm25_shard_device(ctx, segs, doc_bases, n_segs, rq.data(), (int32_t)rq.size(), k_stride, lk, lc, lh, 0, nullptr, nullptr)) {
    my_status = rc;
    my_error = g_last_error;
    if (int rc2 = empty_lists(ctx, (int32_t)again.size(), lc, lh, nullptr)) return rc2;
  }
  if (int rc = exchange_merge_checked(ctx, (int32_t)rq.size(), k_stride, lk, lc, lh, nullptr, rks.data(), rthr.data(), NRTGPU_EXCHANGE_ALLGATHER, ro.data(),
                                      nullptr, nullptr, my_status, my_error))
    return rc;
  for (size_t i = 0; i < again.size(); ++i)
    if (out[again[i]].total_hits >= 0) {   // (mine: held after the first exchange)
      nrtgpu_topdocs& o = out[again[i]];
      o.n_hits = ro[i].n_hits;
      o.total_hits = ro[i].total_hits;
      o.total_hits_is_lower_bound = ro[i].total_hits_is_lower_bound;
    }
  return NRTGPU_OK;
}

extern "C" int nrtgpu_dist_search_bm25_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                             const nrtgpu_bm25_query* queries, int32_t n_queries, nrtgpu_topdocs* out) {
  return nrtgpu_dist_search_bm25_batch_mode(ctx, segs, doc_bases, n_segs, queries, n_queries, NRTGPU_EXCHANGE_ALLGATHER, out);
}

// Exact vector search over a row-partitioned field (BASELINE config 4): every rank scores ITS rows (the leaves of its docid
// range), the per-rank top-k lists are exchanged like the BM25 ones and merged -- NrtKnnFloatVectorQuery's per-leaf merge
// (src/main/java/com/yelp/nrtsearch/server/query/vector/NrtKnnFloatVectorQuery.java:60-64; the per-request glue:
// search/KnnUtils.java:47-66) taken across GPUs.  total_hits = the live vectors of all shards.
extern "C" int nrtgpu_dist_knn_exact(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs, int32_t field_id,
                                     int32_t sim, const float* queries, int32_t n_queries, int32_t dim, int32_t k, float boost, int32_t mode,
                                     nrtgpu_topdocs* out) {
  if (!ctx || !ctx->dist) return fail(NRTGPU_ERR_STATE, "nrtgpu_dist_init has not been called on this context");
  if (!queries || !out || n_queries <= 0 || k <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad arguments");
  if (k > NRTGPU_MAX_K) return fail(NRTGPU_ERR_UNSUPPORTED, "k %d > %d", k, NRTGPU_MAX_K);
  nrtgpu_dist* d = ctx->dist;
  HIP_TRY(hipSetDevice(ctx->device));
  const int32_t k_stride = (int32_t)round_up((uint32_t)k, 16);
  std::lock_guard<std::mutex> call(d->call_mu);
  char *lk = nullptr, *lh = nullptr, *lc = nullptr;
  if (int rc = local_lists(d, n_queries, k_stride, &lk, &lh, &lc)) return rc;
  int32_t my_status = 0;
  std::string my_error;
  if (int rc = knn_exact_device(ctx, segs, doc_bases, n_segs, field_id, sim, queries, n_queries, dim, k, boost, k_stride, lk, lc, lh)) {
    my_status = rc;   // (enters the exchange with empty lists and its status: nrtgpu_dist_search_bm25_batch_mode)
    my_error = g_last_error;
    if (int rc2 = empty_lists(ctx, n_queries, lc, lh, nullptr)) return rc2;
  }
  std::vector<int32_t> ks((size_t)n_queries, k), thr((size_t)n_queries, INT32_MAX);
  return exchange_merge_checked(ctx, n_queries, k_stride, lk, lc, lh, nullptr, ks.data(), thr.data(), mode, out, nullptr, nullptr, my_status, my_error);
}

// The hybrid (BASELINE config 5) over docid-range shards: BM25 recall on every shard -> ONE all-gather + merge on every rank
// (the GLOBAL first pass: a doc that made its shard's list but not the merged one must not be rescored) -> every rank rescores
// ITS docs of the merged lists against its resident vectors -> the rescored windows are exchanged (mode) and merged.  Same
// answers as nrtgpu_search_hybrid_batch over the whole index; TotalHits are the first pass's (QueryRescorer keeps them).
// Reference: multi-retriever / rescorer chain of SearchHandler.java:556 -> RescoreTask.java:47-50 -> QueryRescore.java:39-57.
extern "C" int nrtgpu_dist_search_hybrid_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                               const nrtgpu_bm25_query* queries, int32_t n_queries, int32_t field_id, int32_t sim,
                                               const float* query_vectors, int32_t dim, float boost, double query_weight,
                                               double rescore_weight, int32_t window, int32_t mode, nrtgpu_topdocs* out) {
  if (!ctx || !ctx->dist) return fail(NRTGPU_ERR_STATE, "nrtgpu_dist_init has not been called on this context");
  if (!queries || !out || !query_vectors || n_queries <= 0 || (n_segs > 0 && (!segs || !doc_bases))) return fail(NRTGPU_ERR_INVALID_ARG, "bad arguments");
  if (dim <= 0 || sim < 0 || sim > 3 || window <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad rescore arguments");
  if (!(query_weight >= 0.0) || !(rescore_weight >= 0.0) || !(boost >= 0.0f))
    return fail(NRTGPU_ERR_UNSUPPORTED, "hybrid tail: negative weights (combined scores must stay >= 0)");
  nrtgpu_dist* d = ctx->dist;
  HIP_TRY(hipSetDevice(ctx->device));
  int32_t kmax = 1;
  for (int qi = 0; qi < n_queries; ++qi) kmax = std::max(kmax, queries[qi].k);
  const int32_t k_stride = (int32_t)round_up((uint32_t)std::min(kmax, NRTGPU_MAX_K), 16);
  const int32_t win = std::min<int32_t>(window, NRTGPU_MAX_K);
  const uint32_t w_stride = round_up((uint32_t)win, 16);
  const size_t nq = (size_t)n_queries;
  std::lock_guard<std::mutex> call(d->call_mu);
  char *lk = nullptr, *lh = nullptr, *lc = nullptr;
  if (int rc = local_lists(d, n_queries, k_stride, &lk, &lh, &lc)) return rc;
  // 1. the first pass on this shard
  int32_t my_status = 0;
  std::string my_error;
  if (int rc = nrtgpu_search_bm25_batch_device(ctx, segs, doc_bases, n_segs, queries, n_queries, k_stride, lk, lc, lh)) {
    my_status = rc;   // (enters the exchange with empty lists and its status: nrtgpu_dist_search_bm25_batch_mode)
    my_error = g_last_error;
    if (int rc2 = empty_lists(ctx, n_queries, lc, lh, nullptr)) return rc2;
  }
  std::vector<int32_t> ks(nq), thr(nq), wins(nq, win);
  for (size_t q = 0; q < nq; ++q) {
    ks[q] = queries[q].k;
    thr[q] = queries[q].total_hits_threshold;
  }
  // merged first pass [keys | hits | counts], this rank's windows [keys | hits | counts]
  const size_t kb = nq * (size_t)k_stride * 8, wb = nq * (size_t)w_stride * 8, hb = nq * 8, cb = (nq * 4 + 7) & ~(size_t)7, qb = cb;
  if (int rc = d->stage.reserve(kb + hb + cb + wb + hb + cb + qb)) return rc;
  char* mk = (char*)d->stage.p;
  char* mh = mk + kb;
  char* mc = mh + hb;
  char* wk = mc + cb;
  char* wh = wk + wb;
  char* wc = wh + hb;
  char* qk = wc + cb;   // numHits per query, for hybrid_hits_kernel
  Slot* slot = nullptr;
  acquire_slot(ctx, &slot);
  struct Guard { nrtgpu_ctx* c; Slot* s; ~Guard() { release_slot(c, s); } } guard{ctx, slot};
  {
    std::lock_guard<std::mutex> lk_(d->mu);
    // 2. the global first pass: every rank needs every merged list (its docs may be anywhere in it)
    Gathered g;
    if (int rc = exchange_lists(ctx, d, n_queries, k_stride, lk, lc, lh, nullptr, NRTGPU_EXCHANGE_ALLGATHER, &g, my_status)) return rc;
    if (int rc = peers_failed(d, g, my_status, my_error)) return rc;
    if (int rc = merge_lists_on_device(ctx, slot, d->world, n_queries, k_stride, g.keys, g.cnt, g.hits, ks.data(), mk, mc, mh)) return rc;
    HIP_TRY(hipStreamSynchronize(slot->stream));   // (`gathered` is free again)
  }
  // 3. this rank's docs of the merged lists, rescored; rank 0 carries the first pass's hit totals
  SegReadLocks content(segs, n_segs);
  auto tail = [&]() -> int {
    if (int rc = hybrid_tail_on_device(ctx, slot, segs, doc_bases, n_segs, field_id, sim, query_vectors, dim, boost, query_weight, rescore_weight,
                                       win, n_queries, mk, mc, k_stride, d->world > 1 ? 1 : 0, wk, wc, w_stride))
      return rc;
    std::vector<uint32_t> hk(nq);
    for (size_t q = 0; q < nq; ++q) hk[q] = (uint32_t)ks[q];
    HIP_TRY(hipMemcpyAsync(qk, hk.data(), nq * 4, hipMemcpyHostToDevice, slot->stream));
    launch_hybrid_hits(slot->stream, (const uint64_t*)mh, (const uint32_t*)mc, (const uint32_t*)qk, d->rank == 0 ? 1 : 0, (uint64_t*)wh, (uint32_t)n_queries);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(slot->stream));   // (hk is a stack vector; the windows must be complete before the exchange reads them)
    return NRTGPU_OK;
  };
  if (int rc = tail()) {   // (a tail that failed on this rank still enters step 4: the peers are on their way into it)
    my_status = rc;
    my_error = g_last_error;
    (void)hipStreamSynchronize(slot->stream);
    if (int rc2 = empty_lists(ctx, n_queries, wc, wh, nullptr)) return rc2;
  }
  // 4. the windows: exchanged and merged like any per-rank top-k
  return exchange_merge_checked(ctx, n_queries, (int32_t)w_stride, wk, wc, wh, nullptr, wins.data(), thr.data(), mode, out, nullptr, nullptr, my_status,
                                my_error);
}
