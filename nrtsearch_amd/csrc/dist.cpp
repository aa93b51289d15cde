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
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*GroupFn)(void);
typedef const char* (*ErrStrFn)(int);
const int kNcclInt32 = 2, kNcclInt64 = 4;   // rccl.h: ncclDataType_t

struct Rccl {
  void* lib = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllGatherFn all_gather = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GroupFn group_start = nullptr, group_end = nullptr;
  ErrStrFn err_str = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a process that already holds RCCL (e.g. torch's bundled copy) resolves to that one
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.get_unique_id = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
    r.all_gather = (AllGatherFn)dlsym(r.lib, "ncclAllGather");
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
  std::mutex mu;            // one collective at a time per communicator
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
  HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&d->ev_wait, hipEventBlockingSync | hipEventDisableTiming));
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
  delete d;
}

// The exchange stage alone: this rank's device-resident shard results (what nrtgpu_search_bm25_batch_device[_epoch] left
// in HBM: keys n_queries x k_stride, counts, hit totals) -> ONE grouped all-gather over xGMI -> TopDocs.merge on this
// rank.  A caller that pipelines (scan threads ahead of the exchange) issues these in batch order on every rank.
extern "C" int nrtgpu_dist_allgather_merge(nrtgpu_ctx* ctx, int32_t n_queries, int32_t k_stride, const void* d_keys,
                                           const void* d_counts, const void* d_hits, const int32_t* ks,
                                           const int32_t* total_hits_thresholds, nrtgpu_topdocs* out) {
  if (!ctx || !ctx->dist) return fail(NRTGPU_ERR_STATE, "nrtgpu_dist_init has not been called on this context");
  if (!d_keys || !d_counts || !d_hits || !ks || !total_hits_thresholds || !out || n_queries <= 0 || k_stride <= 0 || k_stride % 16 != 0)
    return fail(NRTGPU_ERR_INVALID_ARG, "bad arguments");
  Rccl* r = rccl();
  nrtgpu_dist* d = ctx->dist;
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t nq = (size_t)n_queries, kb = nq * (size_t)k_stride * 8, hb = nq * 8, cb = nq * 4;
  const size_t W = (size_t)d->world;
  std::lock_guard<std::mutex> lk(d->mu);
  if (int rc = d->gathered.reserve((kb + hb + ((cb + 7) & ~(size_t)7)) * W)) return rc;
  // gathered[array][rank][query]: the layout nrtgpu_merge_topk_device reads
  char* gb = (char*)d->gathered.p;
  char* g_keys = gb;
  char* g_hits = gb + W * kb;
  char* g_cnt = gb + W * (kb + hb);
  if (int rc = r->group_start()) return nccl_fail("ncclGroupStart", rc);
  int rc1 = r->all_gather(d_keys, g_keys, kb / 8, kNcclInt64, d->comm, d->stream);
  int rc2 = r->all_gather(d_hits, g_hits, hb / 8, kNcclInt64, d->comm, d->stream);
  int rc3 = r->all_gather(d_counts, g_cnt, cb / 4, kNcclInt32, d->comm, d->stream);
  if (int rc = r->group_end()) return nccl_fail("ncclGroupEnd", rc);
  if (rc1 || rc2 || rc3) return nccl_fail("ncclAllGather", rc1 ? rc1 : (rc2 ? rc2 : rc3));
  HIP_TRY(wait_for_stream((ctx->cfg.flags & NRTGPU_FLAG_BLOCKING_WAIT) != 0, d->stream, d->ev_wait));
  return nrtgpu_merge_topk_device(ctx, d->world, n_queries, k_stride, g_keys, g_cnt, g_hits, ks, total_hits_thresholds, out);
}

// Every rank calls this with the same queries in the same order (index-global statistics in the weights) over ITS
// leaves; every rank receives every answer.
extern "C" int nrtgpu_dist_search_bm25_batch(nrtgpu_ctx* ctx, const nrtgpu_seg* const* segs, const int32_t* doc_bases, int32_t n_segs,
                                             const nrtgpu_bm25_query* queries, int32_t n_queries, nrtgpu_topdocs* out) {
  if (!ctx || !ctx->dist) return fail(NRTGPU_ERR_STATE, "nrtgpu_dist_init has not been called on this context");
  if (!queries || !out || n_queries <= 0) return fail(NRTGPU_ERR_INVALID_ARG, "bad arguments");
  nrtgpu_dist* d = ctx->dist;
  HIP_TRY(hipSetDevice(ctx->device));
  int32_t kmax = 1;
  for (int qi = 0; qi < n_queries; ++qi) kmax = std::max(kmax, queries[qi].k);
  const int32_t k_stride = (int32_t)round_up((uint32_t)std::min(kmax, NRTGPU_MAX_K), 16);
  const size_t nq = (size_t)n_queries, kb = nq * (size_t)k_stride * 8, hb = nq * 8, cb = nq * 4;
  const size_t o_k = 0, o_h = kb, o_c = kb + hb, block = kb + hb + ((cb + 7) & ~(size_t)7);
  char* lb = nullptr;
  {
    std::lock_guard<std::mutex> lk(d->mu);
    if (int rc = d->local.reserve(block)) return rc;
    lb = (char*)d->local.p;
  }
  // 1. this rank's shard: top-k per query stays in HBM (synchronous: complete when it returns)
  if (int rc = nrtgpu_search_bm25_batch_device(ctx, segs, doc_bases, n_segs, queries, n_queries, k_stride, lb + o_k, lb + o_c, lb + o_h)) return rc;
  // 2 + 3. all-gather over xGMI, TopDocs.merge of the shards' lists on this rank
  std::vector<int32_t> ks(nq), thr(nq);
  for (size_t q = 0; q < nq; ++q) {
    ks[q] = queries[q].k;
    thr[q] = queries[q].total_hits_threshold;
  }
  return nrtgpu_dist_allgather_merge(ctx, n_queries, k_stride, lb + o_k, lb + o_c, lb + o_h, ks.data(), thr.data(), out);
}
