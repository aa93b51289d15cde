// knn.hip -- exact (brute-force) float vector search and vector rescoring on gfx950.
//
// Replaces, for ExactFloatVectorQuery / the brute-force reading of KnnFloatVectorQuery:
//   /root/reference/src/main/java/com/yelp/nrtsearch/server/query/vector/ExactVectorQuery.java:137-173
//   (VectorValuesScorer.score() = VectorSimilarityFunction.compare(query, doc vector) * boost for
//   EVERY doc that has a vector) + the TopScoreDocCollector behind it, and for the rescore tail
//   /root/reference/src/main/java/com/yelp/nrtsearch/server/rescore/QueryRescore.java:40-57.
// Similarity -> score mapping: VectorFieldDef.java:77-88 / docs/field_types/vector.rst:26-35.
//
// Who computes what: a pass over ALL rows only NOMINATES -- it keeps, per query, the k + max(32, k / 2) rows
// with the best ESTIMATE of the score: knn_sketch_kernel over the fp16 sketch of the rows (2 bytes per element, <= 64 queries per
// pass, HBM roofline: N * dim * 2 bytes), or knn_score_kernel over the fp32 rows (v_mfma_f32_16x16x4_f32, <= 32 queries per
// workgroup, N * dim * 4 bytes; segments without a sketch, second passes).  knn_select_kernel<true> then RESCORES every nomination in
// the oracle's order of summation (knn_score_seq: scalar, left to right, every op rounded to fp32) and CERTIFIES the answer against
// the rows left outside with a worst-case rounding bound of the estimate; what it cannot certify goes through a second pass.
// The answer is the oracle's docids and score bits for all four similarities (tests/test_vectors_gpu.py compares with ==).
// The vector RESCORER (rescore_vectors_kernel, hybrid_rescore_kernel: one wave per hit, lanes striding the dimensions) keeps a
// tolerance of 1e-5 relative against the oracle.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.h"
#include "topk.hiph"

namespace nrtgpu {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kKnnThreads = 1024;  // 16 waves (one workgroup per CU: the query panel takes up to 160 KiB of LDS), each owns 16 docs per step
constexpr int kKnnDepth = 8;       // 16-byte row chunks in flight per lane (HBM latency x bandwidth needs ~100 B per lane)

// out[i] = sum_k v[i][k]^2 (fp32, sequential chunks) -- used by cosine.
__global__ __launch_bounds__(256) void knn_row_norms_kernel(const float* __restrict__ vecs, int32_t dim, int64_t n,
                                                            float* __restrict__ norm2) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (row >= n) return;
  const float* v = vecs + row * dim;
  float s = 0.f;
  for (int32_t k = (int32_t)lane; k < dim; k += 64) s += v[k] * v[k];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  if (lane == 0) norm2[row] = s;
}

// *out = max(*out, max_i norm2[i]) as float bits (squared norms are >= 0: their bits order like the values).
__global__ __launch_bounds__(256) void knn_norm_max_kernel(const float* __restrict__ norm2, int64_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = max(m, __float_as_uint(fabsf(norm2[i])));
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
  if ((threadIdx.x & 63u) == 0 && m) atomicMax(out, m);
}

__device__ __forceinline__ float knn_map_score(int sim, float dot, float nq, float nv, float boost) {
  float s;
  if (sim == 0) {  // COSINE: max((1 + cos) / 2, 0), cos = (float)(dot / sqrt((double)nq * nv))
    const float c = (float)((double)dot / sqrt((double)nq * (double)nv));
    s = fmaxf((1.0f + c) / 2.0f, 0.0f);
  } else if (sim == 1) {  // DOT_PRODUCT: max((1 + dot) / 2, 0)
    s = fmaxf((1.0f + dot) / 2.0f, 0.0f);
  } else if (sim == 2) {  // EUCLIDEAN: 1 / (1 + |q - v|^2), |q - v|^2 = nq + nv - 2 dot
    const float d2 = fmaxf(nq + nv - 2.0f * dot, 0.0f);
    s = 1.0f / (1.0f + d2);
  } else {  // MAXIMUM_INNER_PRODUCT
    s = dot < 0.0f ? 1.0f / (1.0f - dot) : dot + 1.0f;
  }
  return s * boost;
}

// The similarity in the ORACLE's order (oracle/nrt_oracle.c nrt_oracle_vector_score: scalar, left to right, every product
// and every sum rounded to fp32; the build has -ffp-contract=off), one lane per (query, row): what the exact vector search
// returns.  The matrix-core pass above it only nominates rows; a result is the bits this function gives.
//   q: the query (LDS or global), v: the row, nq: the query's |q|^2 summed in the same order (host).
// (U 16-byte pieces of the row are requested before the first is used: a lane walks its own row, nothing is coalesced, and
//  the walk is latency bound -- 256 bytes per lane in flight instead of 64 took the rescoring of 150 nominations x 32 queries
//  at 768 dimensions from about 0.23 ms to 0.08.)
template <int SIM, int U>
__device__ __forceinline__ void knn_seq_block(const f32x4* __restrict__ vp, const f32x4* qp, int32_t c0, float& a, float& b) {
  f32x4 x[U];
#pragma unroll
  for (int u = 0; u < U; ++u) x[u] = vp[c0 + u];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const f32x4 y = qp[c0 + u];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (SIM == 2) {
        const float d = y[e] - x[u][e];
        const float sq = d * d;
        a = a + sq;
      } else {
        const float p = y[e] * x[u][e];
        a = a + p;
        if (SIM == 0) {
          const float vv = x[u][e] * x[u][e];
          b = b + vv;
        }
      }
    }
  }
}
template <int SIM>
__device__ __forceinline__ void knn_seq_sums(const float* q, const float* __restrict__ v, int32_t dim, float& a, float& b) {
  const f32x4* vp = (const f32x4*)v;
  const f32x4* qp = (const f32x4*)q;
  const int32_t n4 = dim >> 2;   // dim % 16 == 0: whole groups of four 16-byte pieces
  int32_t c0 = 0;
  for (; c0 + 16 <= n4; c0 += 16) knn_seq_block<SIM, 16>(vp, qp, c0, a, b);
  for (; c0 < n4; c0 += 4) knn_seq_block<SIM, 4>(vp, qp, c0, a, b);
}
__device__ __forceinline__ float knn_score_seq(int sim, const float* q, const float* __restrict__ v, int32_t dim, float nq, float boost) {
  float a = 0.f, b = 0.f;
  if (sim == 2) {   // squareDistance
    knn_seq_sums<2>(q, v, dim, a, b);
    return (1.0f / (1.0f + a)) * boost;
  }
  if (sim == 0) {   // cosine: dot and |v|^2 in one sweep, each its own chain
    knn_seq_sums<0>(q, v, dim, a, b);
    const float c = (float)((double)a / sqrt((double)nq * (double)b));
    return fmaxf((1.0f + c) / 2.0f, 0.0f) * boost;   // (a NaN cosine -- a zero vector -- scores 0, as the oracle's `s > 0 ? s : 0`)
  }
  knn_seq_sums<1>(q, v, dim, a, b);
  float s;
  if (sim == 1) s = fmaxf((1.0f + a) / 2.0f, 0.0f);
  else s = a < 0.0f ? 1.0f / (1.0f - a) : a + 1.0f;
  return s * boost;
}

// global docid -> its vector row (nullptr: the doc has no vector for the field, or lies in none of these leaves).
__device__ __forceinline__ const float* knn_row_of_doc(const DVecSeg* __restrict__ segs, int32_t n_segs, int32_t dim, uint32_t gdoc,
                                                       bool* in_these_leaves) {
  *in_these_leaves = false;
  for (int32_t si = 0; si < n_segs; ++si) {
    const DVecSeg sg = segs[si];
    const int64_t local = (int64_t)gdoc - (int64_t)sg.doc_base;
    if (local < 0 || local >= (int64_t)sg.max_doc) continue;
    *in_these_leaves = true;
    if (!sg.vecs) return nullptr;
    int64_t row = -1;
    if (!sg.ord_to_doc) {
      if (local < (int64_t)sg.n_vec) row = local;
    } else {  // lower_bound over the leaf's ascending ord -> doc map
      int32_t lo = 0, hi = sg.n_vec;
      while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if (sg.ord_to_doc[mid] < (int32_t)local) lo = mid + 1; else hi = mid;
      }
      if (lo < sg.n_vec && sg.ord_to_doc[lo] == (int32_t)local) row = lo;
    }
    return row >= 0 ? sg.vecs + row * dim : nullptr;
  }
  return nullptr;
}

// The dot products of one 16-row tile with the workgroup's query panel(s): acc0 / acc1 = C tiles of panel 0 / 1.
// vp: the lane's row chunk pointer (row j, 16-byte piece kk of every 64-byte chunk), qs: the panel in LDS in operand
// order.  kKnnDepth row chunks are in flight per lane; the body of the main loop has no control flow, so the LDS operand
// reads of the next chunks are scheduled under the MFMAs of the current one (with a branch per chunk the compiler issued
// them right before their use: every chunk waited for the LDS).
template <bool TWO>
__device__ __forceinline__ void knn_tile_dots(const f32x4* __restrict__ vp, const f32x4* qs, int32_t chunks, uint32_t lane,
                                              f32x4& acc0, f32x4& acc1) {
  constexpr int32_t kStride = TWO ? 128 : 64;   // float4s per chunk in LDS: one or two 16-query panels x 64 lanes
  f32x4 abuf[kKnnDepth];
#pragma unroll
  for (int i = 0; i < kKnnDepth; ++i) abuf[i] = vp[4 * min(i, chunks - 1)];
  int32_t c0 = 0;
  // the panel's operands of the NEXT chunk are read from LDS before the current chunk's MFMAs are issued (b0n / b1n)
  f32x4 b0n = qs[(int32_t)lane], b1n = b0n;
  if (TWO) b1n = qs[64 + (int32_t)lane];
  for (; c0 + kKnnDepth <= chunks; c0 += kKnnDepth) {
#pragma unroll
    for (int i = 0; i < kKnnDepth; ++i) {
      const int32_t c = c0 + i;
      const f32x4 a = abuf[i];
      abuf[i] = vp[4 * min(c + kKnnDepth, chunks - 1)];
      const f32x4 b0 = b0n, b1 = b1n;
      const int32_t cn = min(c + 1, chunks - 1);
      b0n = qs[cn * kStride + (int32_t)lane];
      if (TWO) b1n = qs[cn * kStride + 64 + (int32_t)lane];
      __builtin_amdgcn_sched_barrier(0);  // the reads stay ahead of the MFMAs: their LDS latency hides under 8 matrix instructions
      if (TWO) {  // two independent accumulation chains, interleaved (pinned: left alone the scheduler runs one chain after the other)
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b1[0], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b0[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b1[1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b0[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b1[2], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b0[3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b1[3], acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0[0], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b0[1], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b0[2], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b0[3], acc0, 0, 0, 0);
      }
    }
  }
  // the last chunks of a dimension that is no multiple of 16 * kKnnDepth (their rows are in abuf already)
#pragma unroll
  for (int i = 0; i < kKnnDepth; ++i) {
    const int32_t c = c0 + i;
    if (c < chunks) {  // wave-uniform
      const f32x4 a = abuf[i];
      const f32x4 b0 = qs[c * kStride + (int32_t)lane];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0[0], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b0[1], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b0[2], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b0[3], acc0, 0, 0, 0);
      if (TWO) {
        const f32x4 b1 = qs[c * kStride + 64 + (int32_t)lane];
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b1[0], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b1[1], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b1[2], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b1[3], acc1, 0, 0, 0);
      }
    }
  }
}

// Scores docs [row_begin, row_end) of one segment against <= 64 queries (panels of <= 32, see below); hits with
// key > theta[q] are appended to query q's candidate list.  dim must be a multiple of 16.
//   qpanel : n_q * dim floats (row-major), qnorm2 : n_q floats
//   cand   : n_q lists of `cap` keys, cand_cnt : n_q counters (may exceed cap => overflow, host redoes)
// Tiling: v_mfma_f32_16x16x4_f32, C[16 docs x 16 queries] per instruction, two query panels.  A operand:
// lane l supplies row (l & 15), k = 4 * (l >> 4) .. +4 of a 16-float chunk, i.e. FOUR lanes read 64
// contiguous bytes of one row per load instruction (the 32x32x2 shape would give 32: half a cache line
// per request), kKnnDepth chunks in flight per lane.
__global__ __launch_bounds__(kKnnThreads, 1)
void knn_score_kernel(const float* __restrict__ vecs, const float* __restrict__ vnorm2,
                      const int32_t* __restrict__ ord_to_doc, const uint64_t* __restrict__ live_bits,
                      int32_t dim, int64_t row_begin, int64_t row_end, int32_t doc_base,
                      const float* __restrict__ qpanel, const float* __restrict__ qnorm2, int32_t n_q, int32_t sim,
                      float boost, const unsigned long long* __restrict__ theta, uint64_t* __restrict__ cand,
                      uint32_t* __restrict__ cand_cnt, uint32_t cap, int32_t append_only) {
  // append_only: this launch is not followed by a selection (the host defers it once theta is tight): a query whose theta is
  // still unknown appends its live rows like everybody else instead of taking per-row slots a later launch would overwrite
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* qs = (f32x4*)smem;  // [dim/16][2][64] float4: chunk c, panel p, lane (j, kk) -> q[j + 16p][16c + 4kk .. +4]
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // 33 .. 64 queries: two panels of <= 32.  Workgroups b and b + 8 (the same XCD under round-robin dispatch, so the
  // same L2, and the Infinity Cache behind it) stream the SAME rows at the same pace, one panel each: the rows come
  // from HBM once.  The grid is a multiple of 16 then.
  uint32_t slice = blockIdx.x, n_slices = gridDim.x;
  if (n_q > 32) {
    const uint32_t b = blockIdx.x, panel = (b >> 3) & 1u;
    slice = (b & 7u) + 8u * (b >> 4);
    n_slices = gridDim.x >> 1;
    qpanel += (size_t)panel * 32u * (size_t)dim;
    qnorm2 += panel * 32u;
    theta += panel * 32u;
    cand += (size_t)panel * 32u * (size_t)cap;
    cand_cnt += panel * 32u;
    n_q = panel ? n_q - 32 : 32;
  }
  const uint32_t j = lane & 15u, kk = lane >> 4;
  const int32_t chunks = dim >> 4;
  const bool two_panels = n_q > 16;  // uniform.  <= 16 queries: one panel in LDS (half the bytes: dimensions up to 2048 fit)
  const int32_t pshift = two_panels ? 7 : 6;
  for (int32_t i = (int32_t)tid; i < (chunks << pshift); i += kKnnThreads) {
    const int32_t c = i >> pshift, p = two_panels ? (i >> 6) & 1 : 0, l = i & 63;
    const int32_t q = (l & 15) + 16 * p;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < n_q) v = *(const f32x4*)(qpanel + (int64_t)q * dim + 16 * c + 4 * (l >> 4));
    qs[i] = v;
  }
  __syncthreads();
  // D layout (per panel): query col = lane & 15 (+ 16p), doc row in tile = 4 * (lane >> 4) + reg
  float nq[2], th_lo[2];
  unsigned long long th[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int32_t q = (int32_t)j + 16 * p;
    nq[p] = q < n_q ? qnorm2[q] : 0.f;
    th[p] = q < n_q ? theta[q] : ~0ull;
    // cheap rejection before the exact (double) score mapping: a score more than 2^-16 relative below
    // theta's cannot become competitive through the mapping's rounding
    th_lo[p] = key_score(th[p]) * (1.0f - 1.0f / 65536.0f);
  }

  const int64_t rows_per_block = (int64_t)(kKnnThreads / 64) * 16;
  for (int64_t r0 = row_begin + (int64_t)slice * rows_per_block + (int64_t)wave * 16; r0 < row_end;
       r0 += (int64_t)n_slices * rows_per_block) {
    const int64_t row = min(r0 + (int64_t)j, row_end - 1);  // clamped: loads are unconditional
    const f32x4* vp = (const f32x4*)(vecs + row * dim) + kk;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (two_panels) knn_tile_dots<true>(vp, qs, chunks, lane, acc0, acc1);   // (wave-uniform; each variant's chunk loop is branch-free)
    else knn_tile_dots<false>(vp, qs, chunks, lane, acc0, acc1);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int32_t q = (int32_t)j + 16 * p;
      if (q < n_q) {
        const float inv_nq = nq[p] > 0.f ? 1.0f / sqrtf(nq[p]) : 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int64_t drow = r0 + 4 * (int32_t)kk + reg;
          if (drow < row_end) {
            const float dot = p == 0 ? acc0[reg] : acc1[reg];
            const float nv = vnorm2[drow];
            // fp32 estimate of the score (monotone maps): only near-competitive docs take the exact path
            float est;
            if (sim == 0) est = fmaxf((1.0f + dot * inv_nq * (1.0f / sqrtf(nv))) * 0.5f, 0.0f);
            else est = knn_map_score(sim, dot, nq[p], nv, 1.0f);
            if (est * boost >= th_lo[p]) {
              const int32_t ldoc = ord_to_doc ? ord_to_doc[drow] : (int32_t)drow;
              bool live = true;
              if (live_bits) live = (live_bits[ldoc >> 6] >> (ldoc & 63)) & 1ull;
              uint64_t key = 0;  // 0 = "nothing": never above a theta
              if (live) key = pack_key(knn_map_score(sim, dot, nq[p], nv, boost), (uint32_t)(doc_base + ldoc));
              if (th[p] == 0ull && !append_only) {
                // no theta yet (the query's first round, never longer than the list): every row is a
                // candidate and owns slot (row - row_begin) -- no counter traffic at all
                const uint64_t pos = (uint64_t)(drow - row_begin);
                if (pos < cap) cand[(size_t)q * cap + pos] = key;
                // (a round longer than the list while theta is still unknown -- too few live rows so far -- reports
                // its length: the select kernel flags the overflow and the host repeats the panel in bounded rounds)
                if (drow == row_end - 1) cand_cnt[q] = (uint32_t)min<int64_t>(row_end - row_begin, (int64_t)0xFFFFFFFFll);
              } else if (key > th[p]) {
                const uint32_t pos = atomicAdd(&cand_cnt[q], 1u);
                if (pos < cap) cand[(size_t)q * cap + pos] = key;
              }
            }
          }
        }
      }
    }
  }
}

// ---- the fp16 sketch: nomination at half the bytes ---------------------------------------------------------------------
// The answer's bits come from knn_score_seq over the fp32 rows; what the pass over ALL rows has to do is nominate, and a
// nomination only needs an estimate with a KNOWN error (knn_select_kernel<true> certifies against it).  So next to the fp32
// matrix a segment keeps the rows rounded to fp16 (scaled by a power of two so the largest |element| sits at 2^14), laid out in
// the order the matrix cores take them: tile t = rows 16t .. 16t+15, step s = dimensions 32s .. 32s+31, lane l supplies row
// (l & 15), dimensions 8 (l >> 4) .. +7 -- one v_mfma_f32_16x16x32_f16 A operand per lane, 1 KiB per (tile, step), the tiles of a
// row range contiguous.  A wave streams a contiguous run of tiles with kKnnDepth 16-byte pieces in flight per lane: perfectly
// coalesced, N * dim * 2 bytes per pass, and up to 64 queries (4 panels of 16, fp16 in LDS) ride on one pass because the fp16
// matrix rate is 16x the fp32 one.  |estimate - result| <= 2^-10 |q||v| (+ fp32 accumulation + flushed tiny elements): vectors.cpp.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr size_t kKnnSketchStaticLds = 1280;   // knn_sketch_kernel's per-query tables (static LDS next to the dynamic panel + queue)

// max |element| of the matrix as float bits (atomicMax on uints: magnitudes order like their bits)
__global__ __launch_bounds__(256) void knn_absmax_kernel(const float* __restrict__ vecs, int64_t n_elems, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (int64_t)gridDim.x * 256)
    m = max(m, __float_as_uint(fabsf(vecs[i])));
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
  if ((threadIdx.x & 63u) == 0 && m) atomicMax(out, m);
}
// *out = min over rows with a non-zero norm (bits; 0xFFFFFFFF when there is none)
__global__ __launch_bounds__(256) void knn_norm_min_kernel(const float* __restrict__ norm2, int64_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0xFFFFFFFFu;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t b = __float_as_uint(fabsf(norm2[i]));
    if (b) m = min(m, b);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, d, 64));
  if ((threadIdx.x & 63u) == 0) atomicMin(out, m);
}
// one thread per 16-byte piece of the sketch
__global__ __launch_bounds__(256) void knn_sketch_build_kernel(const float* __restrict__ vecs, int32_t dim, int64_t n, int32_t steps,
                                                              float scale, f16x8* __restrict__ sketch) {
  const int64_t piece = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t tiles = (n + 15) >> 4;
  if (piece >= tiles * steps * 64) return;
  const int32_t l = (int32_t)(piece & 63);
  const int64_t ts = piece >> 6;
  const int32_t s = (int32_t)(ts % steps);
  const int64_t row = (ts / steps) * 16 + (l & 15);
  const int32_t k0 = 32 * s + 8 * (l >> 4);
  f16x8 h;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int32_t k = k0 + e;
    const float x = (row < n && k < dim) ? vecs[row * dim + k] * scale : 0.0f;
    h[e] = (_Float16)x;   // round to nearest even; |x| <= 2^14
  }
  sketch[piece] = h;
}

// The query panel in the sketch kernel's B-operand order, once per panel (every launch of the pass and each of its 256 workgroups
// then copies it into LDS with 16-byte loads instead of converting it again from the fp32 panel with scalar ones):
//   out[(s * P + p) * 64 + l] = q[(l & 15) + 16p][32s + 8(l >> 4) .. +7] * qscale, rounded to fp16.
__global__ __launch_bounds__(256) void knn_panel_fp16_kernel(const float* __restrict__ qpanel, const float* __restrict__ qscale, int32_t dim,
                                                            int32_t steps, int32_t P, int32_t n_q, f16x8* __restrict__ out) {
  const int32_t i = (int32_t)(blockIdx.x * 256 + threadIdx.x);
  if (i >= steps * P * 64) return;
  const int32_t l = i & 63, p = (i >> 6) % P, sidx = (i >> 6) / P;
  const int32_t q = (l & 15) + 16 * p, k0 = 32 * sidx + 8 * (l >> 4);
  f16x8 h;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int32_t k = k0 + e;
    h[e] = (_Float16)((q < n_q && k < dim) ? qpanel[(int64_t)q * dim + k] * qscale[q] : 0.0f);
  }
  out[i] = h;
}

// The ring of row pieces is driven by hand: the requests are inline asm (the compiler's waitcnt insertion does not see them),
// and before slot i is consumed the wave waits until at most D - 1 requests are outstanding -- exactly the ones issued after slot
// i's.  Left to the compiler the loop either drained the ring at every group of D pieces (vmcnt(0) at the loop header) or copied
// the ring's registers at the top of the group, which needs the same wait.  The wait carries the slot as an in/out operand so
// that the matrix instruction reading it cannot be scheduled above the wait.
template <int OFF>
__device__ __forceinline__ void sk_request(f16x8& dst, const f16x8* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void sk_wait(f16x8& slot) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(slot) : "n"(N) : "memory");
}
template <int P, int D, int I>
__device__ __forceinline__ void sk_step(f16x8 (&abuf)[D], f32x4 (&acc)[P], const f16x8* qs_step, const f16x8* nxt_lo, const f16x8* nxt_hi) {
  // the step's P query operands are asked of the LDS together and BEFORE the wait for the row piece: one LDS round trip per
  // step, under the ring's wait -- read one by one between the matrix instructions (what the compiler does when it has no P x 4
  // registers to spare) each of them waits for its own (round 6: the whole cost of more queries per pass, see the epilogue)
  f16x8 b[P];
#pragma unroll
  for (int p = 0; p < P; ++p) b[p] = qs_step[(I * P + p) * 64];
  __builtin_amdgcn_sched_barrier(0);
  sk_wait<D - 1>(abuf[I]);
#pragma unroll
  for (int p = 0; p < P; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[I], b[p], acc[p], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  // into the registers the matrix instructions above have just read: no copy, D - 1 requests stay in flight
  if (I < 4) sk_request<(I & 3) * 1024>(abuf[I], nxt_lo);
  else sk_request<(I & 3) * 1024>(abuf[I], nxt_hi);
  __builtin_amdgcn_sched_barrier(0);
}

// Nominations from the sketch: the fp32 kernel's contract (candidate lists, theta, first-round slots) over the global tiles
// [tile_begin, tile_end) of the search's leaves (plan.h: DKnnLeaf) -- ONE launch walks every leaf.  P = panels of 16 queries
// (1 .. 4); D = pieces in flight per lane, `steps` (the sketch's, padded: a multiple of 4) is a multiple of D so that the ring of
// row pieces is indexed statically and the epilogue stands once per tile.  qscale[q]: the power of two the query was multiplied by
// before rounding to fp16; acc / (qscale * rows' scale) is the dot product in the vectors' own units (powers of two: exact).
// Rows are addressed by their PADDED position: (tile - tile_begin) * 16 + row in tile (a leaf's last tile may hold fewer than 16
// rows): a first-round slot, the row field of a queue entry.
// a value every lane holds, moved into scalar registers (the compiler cannot know that the wave's leaf is uniform)
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | (uint64_t)lo;
}
__device__ __forceinline__ int32_t knn_leaf_of_tile(const DKnnLeaf* __restrict__ leaves, int32_t n_leaves, int64_t tile) {
  int32_t lo = 0, hi = n_leaves;   // the last leaf whose tile_begin <= tile
  while (hi - lo > 1) {
    const int32_t mid = (lo + hi) >> 1;
    if (leaves[mid].tile_begin <= tile) lo = mid; else hi = mid;
  }
  return lo;
}

template <int P, int D>
__global__ __launch_bounds__(kKnnThreads, 1)
void knn_sketch_kernel(const DKnnLeaf* __restrict__ leaves, int32_t n_leaves, int32_t steps, int64_t tile_begin, int64_t tile_end,
                       const f16x8* __restrict__ panel16, const float* __restrict__ qnorm2, const float* __restrict__ qscale,
                       int32_t n_q, int32_t sim, float boost, const unsigned long long* __restrict__ theta,
                       uint64_t* __restrict__ cand, uint32_t* __restrict__ cand_cnt, uint32_t cap, int32_t append_only,
                       uint32_t qcap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16x8* qs = (f16x8*)smem;  // [steps][P][64]: step s, panel p, lane l -> q[(l & 15) + 16p][32s + 8(l >> 4) .. +7] * qscale
  // Behind the panel: the workgroup's queue of nominations (qcap entries: score bits << 32 | padded row << 6 | query).
  // A row that passes is pushed HERE (an LDS atomic: lgkmcnt) and the queue is written out once, when the workgroup has streamed
  // its rows: a global atomic with a return value in the tile epilogue is the newest vector-memory operation of the wave, and
  // waiting for it waits for every request of the ring before it -- with one passing row per tile (the round after the first
  // selection) the ring ran dry at every tile and the launch at 3.2 TB/s instead of 5.2.
  uint32_t* const q_n = (uint32_t*)(smem + (size_t)steps * P * 1024);
  uint64_t* const q_e = (uint64_t*)(smem + (size_t)steps * P * 1024 + 16);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  __shared__ float sk_nq[64], sk_dsc[64], sk_thhi[64];
  __shared__ unsigned long long sk_th[64];
  static_assert(sizeof(float) * 64 * 3 + sizeof(unsigned long long) * 64 == kKnnSketchStaticLds, "launch_knn_sketch sizes the queue by this");
  if (tid == 0) *q_n = 0u;
  if (tid < 64u) {
    const int32_t q = (int32_t)tid;
    const unsigned long long th_q = q < n_q ? theta[q] : ~0ull;
    sk_nq[q] = q < n_q ? qnorm2[q] : 0.f;
    sk_dsc[q] = q < n_q ? 1.0f / qscale[q] : 0.f;   // a power of two: exact
    sk_th[q] = th_q;
    // a key above theta carries a score >= theta's (equal scores: the docid decides): rows strictly below are rejected on the
    // score alone (theta = ~0, "nothing passes", is a NaN score: every compare with it is false)
    const uint32_t tsb = __float_as_uint(key_score(th_q));
    sk_thhi[q] = tsb ? __uint_as_float(tsb - 1u) : -1.0f;   // (a theta of score 0: ties among zero scores are the docid's business)
  }
  for (int32_t i = (int32_t)tid; i < steps * P * 64; i += kKnnThreads) qs[i] = panel16[i];   // (knn_panel_fp16_kernel's layout)
  __syncthreads();
  const uint32_t j = lane & 15u, kk = lane >> 4;
  // What the epilogue needs of query q = j + 16 p: |q|^2, 1 / its fp16 scale (a power of two: exact), theta and the largest score
  // theta rejects.  NOT held in registers through the stream (20 of the 128 a wave has: without them the compiler keeps a step's
  // P query operands in registers instead of reading them one by one): read where they are used -- per leaf for the filter's
  // thresholds, and in the epilogue proper, which few lanes reach once theta is known.  From LDS (kKnnSketchStaticLds bytes next to
  // the dynamic panel + queue): a vector load there would have the wave wait for its whole ring of row pieces.
  auto query_consts = [&](int p, float& nq_p, float& dsc_p, unsigned long long& th_p, float& th_hi_p) {
    const uint32_t q = j + 16u * (uint32_t)p;   // (< 64: the tables hold neutral values behind the panel's last query)
    nq_p = sk_nq[q];
    dsc_p = sk_dsc[q];
    th_p = sk_th[q];
    th_hi_p = sk_thhi[q];
  };
  const int64_t padded_rows = (tile_end - tile_begin) << 4;   // of this launch
  // a contiguous run of tiles per wave: one sequential stream of 1 KiB pieces per leaf it crosses
  const int64_t n_tiles = tile_end - tile_begin;
  const int64_t n_waves = (int64_t)gridDim.x * (kKnnThreads / 64), w = (int64_t)blockIdx.x * (kKnnThreads / 64) + wave;
  int64_t t_run = tile_begin + n_tiles * w / n_waves;
  const int64_t t_run_end = tile_begin + n_tiles * (w + 1) / n_waves;
  int32_t li = t_run < t_run_end ? knn_leaf_of_tile(leaves, n_leaves, t_run) : 0;
  while (t_run < t_run_end) {   // (a wave without tiles still meets the others at the queue's barrier)
    // the leaf, as scalars: its pointers feed the ring's base and uniform loads
    const DKnnLeaf& lf = leaves[li];
    const uint64_t u_sketch = uniform_u64((uint64_t)lf.sketch), u_norms = uniform_u64((uint64_t)lf.vnorm2);
    const uint64_t u_o2d = uniform_u64((uint64_t)lf.ord_to_doc), u_accept = uniform_u64((uint64_t)lf.accept);
    const int64_t leaf_t0 = (int64_t)uniform_u64((uint64_t)lf.tile_begin);
    const int32_t leaf_rows = __builtin_amdgcn_readfirstlane(lf.n_rows), doc_base = __builtin_amdgcn_readfirstlane(lf.doc_base);
    const float leaf_inv = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(lf.inv_rows_scale)));
    const int32_t* const ord_to_doc = (const int32_t*)u_o2d;
    const uint64_t* const live_bits = (const uint64_t*)u_accept;
    // Can any of a lane's 4 P estimates of a tile pass?  The estimate is a non-decreasing function of acc x (a per-row factor) for
    // COSINE, DOT_PRODUCT and MAXIMUM_INNER_PRODUCT, so "estimate > theta's score" has a necessary condition that costs one
    // multiply and one compare per element: acc x factor > xthr[p] -- xthr lowered by 2^-16 of the scale the estimate's own
    // roundings (a handful of ulps: 2^-22) live on.  A lane none of whose elements meets it skips the epilogue below; a lane
    // with one runs it UNCHANGED, so what is nominated is exactly what was (the filter is a superset test).  The epilogue was the
    // whole cost of more queries per pass: 2.76 ms at 64 queries against 2.50 without it and 2.50 at one query
    // (profiles/r06_knn_epilogue.log).  -inf: everything passes (no theta yet, EUCLIDEAN, a zero query, a boost <= 0) -- pass_all,
    // because a NaN product compares false with anything; +inf: nothing can (a column without a query, theta = "nothing passes").
    float xthr[P];
    bool pass_all = false;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      xthr[p] = -INFINITY;
      float nq_p, dsc_p, ts;   // ts: scores <= ts are rejected
      unsigned long long th_p;
      query_consts(p, nq_p, dsc_p, th_p, ts);
      const float c = dsc_p * leaf_inv * (sim == 0 ? (nq_p > 0.f ? __builtin_amdgcn_rsqf(nq_p) : 0.f) : 1.0f);   // estimate's dot = acc x c (x rsq|v|^2)
      if ((int32_t)j + 16 * p >= n_q || (th_p != 0ull && ts != ts)) {
        xthr[p] = INFINITY;
      } else if (th_p != 0ull && ts >= 0.0f && boost > 0.0f && c > 0.0f && c < INFINITY && sim != 2) {
        const float s = ts / boost;   // the estimate before the boost must exceed about this
        float x, scale;               // the dot product (cosine: the cosine) must exceed x; scale: what its roundings are relative to
        if (sim == 3) {               // dot < 0 ? 1 / (1 - dot) : dot + 1
          x = s >= 1.0f ? s - 1.0f : (s > 0.0f ? 1.0f - 1.0f / s : -INFINITY);
          scale = fabsf(x) + 1.0f;
        } else {                      // max((1 + dot) / 2, 0)
          x = 2.0f * s - 1.0f;
          scale = fabsf(x) + 1.0f;
        }
        if (x > -INFINITY && x < INFINITY) xthr[p] = (x - scale * 0x1p-16f) / c;   // (NaN / inf: stays -inf)
        if (!(xthr[p] == xthr[p])) xthr[p] = -INFINITY;
      }
      pass_all |= xthr[p] == -INFINITY;
    }
    // The norms' pointer.  Rebuilt from the leaf record it would be a GENERIC pointer to the compiler, which then loads a tile's
    // norms with flat VECTOR loads (vmcnt) and drains the ring at the tile's first use of one (round 3's build:
    // profiles/r03_knn_sketch_isa_note.txt).  In the constant address space the same loads are scalar (s_load, lgkmcnt), as they
    // were before the leaf table.  Round 4, same box, interleaved (profiles/r04_knn_scalar_norms_ab.log): the vector tests, 64 fuzz
    // rounds and the BASELINE-size tests bit-exact; kernel 2.66 -> 2.60 ms at 32 queries, 2.85 -> 2.82 at 64; no scratch in any
    // instantiation (<4, 8> had 36 B).
    typedef const float __attribute__((address_space(4))) cfloat_k;
    cfloat_k* const vnorm2 = (cfloat_k*)u_norms;
    const int64_t t0 = t_run - leaf_t0, t1 = min(t_run_end, leaf_t0 + (((int64_t)leaf_rows + 15) >> 4)) - leaf_t0;   // local tiles
    // the run as groups of D pieces (a tile is steps / D whole groups): `cur` walks the groups, the ring slot of piece i of a
    // group is i, and the piece D ahead -- the same slot of the NEXT group -- is requested the moment slot i has been consumed
    const f16x8* cur = (const f16x8*)u_sketch + (t0 * steps) * 64 + lane;
    const f16x8* const last_group = cur + ((t1 - t0) * steps - D) * 64;
    f16x8 abuf[D];
    sk_request<0>(abuf[0], cur);
    sk_request<1024>(abuf[1], cur);
    sk_request<2048>(abuf[2], cur);
    sk_request<3072>(abuf[3], cur);
    if (D == 8) {
      sk_request<0>(abuf[D - 4], cur + 256);
      sk_request<1024>(abuf[D - 3], cur + 256);
      sk_request<2048>(abuf[D - 2], cur + 256);
      sk_request<3072>(abuf[D - 1], cur + 256);
    }
    for (int64_t tile = t0; tile < t1; ++tile) {
      // |v|^2 of the tile's 16 rows: the tile is the same for the whole wave, so they are meant to come through the SCALAR
      // cache (s_load, counted by lgkmcnt, not by the ring's vmcnt) -- see the note at `vnorm2` above
      const uint32_t t_lo = __builtin_amdgcn_readfirstlane((uint32_t)(tile & 0xFFFFFFFFll));
      const uint32_t t_hi = __builtin_amdgcn_readfirstlane((uint32_t)(tile >> 32));
      const int64_t u_r0 = (int64_t)(((uint64_t)t_hi << 32) | t_lo) << 4;
      float nvt[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) nvt[c] = vnorm2[u_r0 + c];   // (the norms' allocation is padded: reads past row n - 1 stay inside it)
      f32x4 acc[P];
#pragma unroll
      for (int p = 0; p < P; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int32_t s0 = 0; s0 < steps; s0 += D) {
        const f16x8* nxt = cur < last_group ? cur + D * 64 : cur;   // (past the run's end: its last group again, never used)
        const f16x8* nxt_hi = nxt + 256;
        const f16x8* qg = qs + (size_t)s0 * P * 64 + lane;
        sk_step<P, D, 0>(abuf, acc, qg, nxt, nxt_hi);
        sk_step<P, D, 1>(abuf, acc, qg, nxt, nxt_hi);
        sk_step<P, D, 2>(abuf, acc, qg, nxt, nxt_hi);
        sk_step<P, D, 3>(abuf, acc, qg, nxt, nxt_hi);
        if (D == 8) {
          sk_step<P, D, D - 4>(abuf, acc, qg, nxt, nxt_hi);
          sk_step<P, D, D - 3>(abuf, acc, qg, nxt, nxt_hi);
          sk_step<P, D, D - 2>(abuf, acc, qg, nxt, nxt_hi);
          sk_step<P, D, D - 1>(abuf, acc, qg, nxt, nxt_hi);
        }
        cur = nxt;
      }
      float nv4[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        nv4[reg] = kk == 0 ? nvt[reg] : kk == 1 ? nvt[4 + reg] : kk == 2 ? nvt[8 + reg] : nvt[12 + reg];
      // D layout: query col = lane & 15 (+ 16p), row in tile = 4 * (lane >> 4) + reg
      const int64_t r0 = tile << 4;                                   // the tile's first row in its leaf
      const int64_t g0 = (leaf_t0 + tile - tile_begin) << 4;          // ... and its padded position in this launch
      float rn4[4];
      if (sim == 0) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) rn4[reg] = __builtin_amdgcn_rsqf(nv4[reg]);
      }
      bool maybe = pass_all;   // some element of mine may pass (see xthr above; a NaN product -- a zero row -- counts as "may")
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) maybe |= !(acc[p][reg] * (sim == 0 ? rn4[reg] : 1.0f) <= xthr[p]);
      if (maybe)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int32_t q = (int32_t)j + 16 * p;
        if (q < n_q) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            // element by element: the same test, then the estimate proper for the few that meet it
            if (!(xthr[p] == -INFINITY || !(acc[p][reg] * (sim == 0 ? rn4[reg] : 1.0f) <= xthr[p]))) continue;
            float nq_p, dsc_p, th_hi_p;
            unsigned long long th_p;
            query_consts(p, nq_p, dsc_p, th_p, th_hi_p);
            const float inv_nq = nq_p > 0.f ? __builtin_amdgcn_rsqf(nq_p) : 0.f;
            const bool slot_round = th_p == 0ull && !append_only;   // no theta yet: every padded row owns a slot, see knn_score_kernel
            // (the query's number, opaque to the optimiser from here on: the addresses of its list and counter are then computed
            //  HERE, where a row is written straight to the list -- hoisted out of the tile loop they cost the stream four registers
            //  it does not have: 20 B of scratch in the <4, 8> instantiation)
            uint32_t qv = (uint32_t)q;
            asm volatile("" : "+v"(qv));
            const int64_t drow = r0 + 4 * (int32_t)kk + reg;
            const int64_t gpos = g0 + 4 * (int32_t)kk + reg;
            const bool valid = drow < (int64_t)leaf_rows;
            float sc = 0.f;
            if (valid) {
              const float dot = acc[p][reg] * dsc_p * leaf_inv;
              const float nv = nv4[reg];
              float est;
              if (sim == 0) est = fmaxf((1.0f + dot * inv_nq * rn4[reg]) * 0.5f, 0.0f);
              else if (sim == 1) est = fmaxf((1.0f + dot) * 0.5f, 0.0f);
              else if (sim == 2) est = __builtin_amdgcn_rcpf(1.0f + fmaxf(nq_p + nv - 2.0f * dot, 0.0f));
              else est = dot < 0.0f ? __builtin_amdgcn_rcpf(1.0f - dot) : dot + 1.0f;
              // this estimate IS the nomination's score (hardware rsq / rcp, a few fp32 roundings: the bound's e_rel covers
              // them): nothing in double, nothing but compares until a row passes
              sc = est * boost;
            }
            if (slot_round || (valid && sc > th_hi_p)) {
              uint32_t qi = 0xFFFFFFFFu;
              if (!slot_round) qi = atomicAdd(q_n, 1u);
              if (qi < qcap) {
                q_e[qi] = ((uint64_t)__float_as_uint(sc) << 32) | ((uint64_t)gpos << 6) | (uint64_t)q;
              } else {   // the first round's slots (a padding row's holds "nothing"), or a full queue: straight to the list
                uint64_t key = 0;  // 0 = "nothing": never above a theta
                if (valid) {
                  const int32_t ldoc = ord_to_doc ? ord_to_doc[drow] : (int32_t)drow;
                  bool live = true;
                  if (live_bits) live = (live_bits[ldoc >> 6] >> (ldoc & 63)) & 1ull;
                  if (live) key = pack_key(sc, (uint32_t)(doc_base + ldoc));
                }
                if (slot_round) {
                  if ((uint64_t)gpos < (uint64_t)cap) cand[(size_t)qv * cap + (size_t)gpos] = key;
                  if (gpos == padded_rows - 1) cand_cnt[qv] = (uint32_t)min<int64_t>(padded_rows, (int64_t)0xFFFFFFFFll);
                } else if (key > th_p) {
                  const uint32_t pos = atomicAdd(&cand_cnt[qv], 1u);
                  if (pos < cap) cand[(size_t)qv * cap + pos] = key;
                }
                // (its loads and stores are complete here as far as the compiler's bookkeeping goes: with vector memory events
                // pending at the next tile's loop it waits vmcnt(0) in front of it -- the ring with them)
                __builtin_amdgcn_s_waitcnt(0x0F70);
              }
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last requests (never used) land before the registers are reused
    t_run = leaf_t0 + t1;
    ++li;
  }
  __syncthreads();
  // The queue goes out: a nomination's place in its query's list comes from the list's counter in global memory -- ONE atomic per
  // (workgroup, query) for all of the workgroup's nominations of that query, not one per nomination: with 64 queries the second
  // launch of a pass (theta still that of the first 65 k rows) queues ~560 nominations per workgroup, 144 k atomics on 64 words
  // of four cache lines, and the launch ended behind them (profiles/r06_knn_epilogue.log).  A thread keeps its (<= kFlushPerThread)
  // entries in registers between the count and the write.
  constexpr int kFlushPerThread = 4;   // qcap <= 4096 = 4 x kKnnThreads (launch_knn_sketch)
  static_assert(kFlushPerThread * kKnnThreads >= 4096, "the queue's capacity is bounded by what the flush holds in registers");
  const uint32_t n_queued = min(*q_n, qcap);
  uint32_t* const f_cnt = (uint32_t*)sk_nq;    // (the per-query tables are done with: their LDS holds the counts and the bases)
  uint32_t* const f_base = (uint32_t*)sk_dsc;
  if (tid < 64u) f_cnt[tid] = 0u;
  __syncthreads();
  uint64_t my_key[kFlushPerThread];
  uint32_t my_q[kFlushPerThread], my_rank[kFlushPerThread];
#pragma unroll
  for (int r = 0; r < kFlushPerThread; ++r) {
    const uint32_t i = tid + (uint32_t)r * kKnnThreads;
    my_q[r] = 0xFFFFFFFFu;
    my_key[r] = 0ull;
    my_rank[r] = 0u;
    if (i < n_queued) {
      const uint64_t e = q_e[i];
      const uint32_t q = (uint32_t)(e & 63ull);
      const int64_t gpos = (int64_t)((e >> 6) & 0x3FFFFFFull);
      const int64_t tile = tile_begin + (gpos >> 4);
      const DKnnLeaf lf = leaves[knn_leaf_of_tile(leaves, n_leaves, tile)];
      const int64_t drow = ((tile - lf.tile_begin) << 4) + (gpos & 15);
      const int32_t ldoc = lf.ord_to_doc ? lf.ord_to_doc[drow] : (int32_t)drow;
      bool live = true;
      if (lf.accept) live = (lf.accept[ldoc >> 6] >> (ldoc & 63)) & 1ull;
      const uint64_t key = pack_key(__uint_as_float((uint32_t)(e >> 32)), (uint32_t)(lf.doc_base + ldoc));
      if (live && key > theta[q]) {
        my_q[r] = q;
        my_key[r] = key;
        my_rank[r] = atomicAdd(&f_cnt[q], 1u);
      }
    }
  }
  __syncthreads();
  if (tid < 64u) {
    const uint32_t c = f_cnt[tid];
    f_base[tid] = c ? atomicAdd(&cand_cnt[tid], c) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kFlushPerThread; ++r)
    if (my_q[r] != 0xFFFFFFFFu) {
      const uint32_t pos = f_base[my_q[r]] + my_rank[r];
      if (pos < cap) cand[(size_t)my_q[r] * cap + pos] = my_key[r];
    }
}

// Per query: top-k of (running top-k  UNION  the round's candidate list) -> running top-k (sorted),
// theta[q] = k-th key once k hits are known.  One workgroup per query.
//
// REFINE: the candidates are NOMINATIONS (keys carrying the matrix-core estimate of the score); each is rescored in the
// oracle's order (knn_score_seq) before it competes, so the running top-k holds result bits.  Two uses (vectors.cpp):
//   certify = 1: the candidates are the sorted top-k_int of the estimates, the running list starts empty.  A row outside that
//     list has an estimate <= the list's last one, m, hence a result <= knn_result_upper(m) (plan.h; DESIGN §4.5: the
//     worst-case rounding bound of two length-dim fp32 sums), so the top-k of the rescored list IS the answer when its k-th
//     score lies above that or the list held every row; cert[q] says which.  If not, theta[q] becomes the lowest key a row
//     of the answer can carry as an estimate (k-th rescored score - E) and the host runs the second use:
//   certify = 0: a pass over the rows with theta fixed; every nomination is rescored and merged (theta is left alone).
struct KnnRefine {
  const DVecSeg* segs;
  const float* qpanel;      // the panel's queries, row-major
  const float* qnorm2;
  const float* ebound;      // per query: e_abs of knn_result_upper / knn_estimate_lower (plan.h)
  uint32_t* cert;
  int32_t n_segs, dim, sim, certify;
  float boost, erel, min_score;
  uint32_t k_int;
};

struct KnnSelSmem {
  uint64_t cand[kMergeCap];
  TopkScratch sc;
  uint64_t theta;
  uint32_t cnt;
  uint32_t pad;
};

template <bool REFINE>
__global__ __launch_bounds__(kScanThreads)
void knn_select_kernel(uint64_t* __restrict__ topk, uint32_t* __restrict__ topk_cnt, uint32_t k_stride, uint32_t k,
                       const uint64_t* __restrict__ cand, uint32_t* __restrict__ cand_cnt, uint32_t cap,
                       unsigned long long* __restrict__ theta, uint32_t* __restrict__ overflow, KnnRefine rf) {
  __shared__ KnnSelSmem s;
  __shared__ __attribute__((aligned(16))) float qv[REFINE ? 2048 : 4];
  const uint32_t tid = threadIdx.x, q = blockIdx.x;
  if (tid == 0) {
    s.theta = 0;
    s.cnt = 0;
  }
  if (REFINE)
    for (int32_t i = (int32_t)tid; i < rf.dim; i += kScanThreads) qv[i] = rf.qpanel[(size_t)q * rf.dim + i];
  __syncthreads();
  const uint32_t n_prev = topk_cnt[q];
  const uint32_t n_raw = cand_cnt[q];
  if (n_raw > cap && tid == 0) *overflow = 1u;  // the host shrinks the round and repeats it
  const uint32_t n_cand = min(n_raw, cap);
  for (int pass = 0; pass < 2; ++pass) {
    const uint64_t* src = pass == 0 ? topk + (size_t)q * k_stride : cand + (size_t)q * cap;
    const uint32_t c = pass == 0 ? n_prev : n_cand;
    for (uint32_t off = 0; off < c; off += kScanThreads) {
      const uint32_t i = off + tid;
      uint64_t key = (i < c) ? src[i] : 0;
      if (REFINE && pass == 1 && key != 0ull) {   // nomination -> result bits
        const uint32_t gdoc = 0xFFFFFFFFu - (uint32_t)key;
        bool mine;
        const float* v = knn_row_of_doc(rf.segs, rf.n_segs, rf.dim, gdoc, &mine);
        key = 0ull;
        if (v) {
          const float sc = knn_score_seq(rf.sim, qv, v, rf.dim, rf.qnorm2[q], rf.boost);
          if (!(rf.min_score > 0.0f) || sc >= rf.min_score) key = pack_key(sc, gdoc);
        }
      }
      const bool want = (i < c) && (key > s.theta);
      topk_append(s.cand, &s.cnt, want, key);
      __syncthreads();
      const uint32_t cn = s.cnt;
      __syncthreads();
      if (cn > (uint32_t)(kMergeCap - kScanThreads)) {
        uint64_t thr = 0;
        const uint32_t m = topk_compact<kScanThreads, kMergeCap>(s.cand, cn, k, &s.sc, &thr);
        if (tid == 0) {
          s.cnt = m;
          if (thr > s.theta) s.theta = thr;
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  {
    const uint32_t c = s.cnt;
    __syncthreads();
    if (c > k) {
      uint64_t thr = 0;
      const uint32_t m = topk_compact<kScanThreads, kMergeCap>(s.cand, c, k, &s.sc, &thr);
      if (tid == 0) s.cnt = m;
      __syncthreads();
    }
  }
  const uint32_t n = s.cnt;
  uint32_t n2 = 1;
  while (n2 < n) n2 <<= 1;
  for (uint32_t i = n + tid; i < n2; i += kScanThreads) s.cand[i] = 0;
  bitonic_sort_desc<kScanThreads>(s.cand, n2);
  uint64_t* out = topk + (size_t)q * k_stride;
  for (uint32_t i = tid; i < k_stride; i += kScanThreads) out[i] = (i < n) ? s.cand[i] : 0;
  if (tid == 0) {
    topk_cnt[q] = n;
    cand_cnt[q] = 0;
    if (!REFINE) {
      if (n == k) theta[q] = s.cand[k - 1];
    } else if (rf.certify) {
      // the nominations: sorted, at most k_int of them; a full list may have left rows outside, none with an estimate above m
      const bool full = n_raw >= rf.k_int;
      const double m = full ? (double)key_score(cand[(size_t)q * cap + rf.k_int - 1]) : 0.0;
      const double e_abs = (double)rf.ebound[q], e_rel = (double)rf.erel, b = (double)rf.boost;
      const bool ok = !full || (n == k && (double)key_score(s.cand[k - 1]) > knn_result_upper(rf.sim, m, e_abs, e_rel, b));
      rf.cert[q] = ok ? 1u : 0u;
      unsigned long long th = ~0ull;   // certified: the second pass nominates nothing for this query
      if (!ok) {
        // a row of the answer scores >= the k-th rescored score (or >= min_score while fewer than k are known): the lowest key
        // its estimate can carry
        const double base = n == k ? (double)key_score(s.cand[k - 1]) : (double)rf.min_score * b;
        const float lo = (float)knn_estimate_lower(rf.sim, base, e_abs, e_rel, b);
        th = lo > 0.0f ? pack_key(lo, 0xFFFFFFFFu) - 1ull : 0ull;
      }
      theta[q] = th;
    }
  }
}

// One wave, one (query, row): the similarity with the lanes striding the dimensions (coalesced row reads) and a butterfly
// sum.  EUCLIDEAN sums (q - v)^2 directly: |q|^2 + |v|^2 - 2 q.v cancels for near-duplicates and large norms.
__device__ __forceinline__ float knn_wave_score(int sim, const float* __restrict__ v, const float* __restrict__ q, int32_t dim,
                                                uint32_t lane, float nq, float nv, float boost) {
  float acc = 0.f;
  if (sim == 2) {
    for (int32_t k = (int32_t)lane; k < dim; k += 64) {
      const float d = q[k] - v[k];
      acc += d * d;
    }
  } else {
    for (int32_t k = (int32_t)lane; k < dim; k += 64) acc += v[k] * q[k];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (sim == 2) return (1.0f / (1.0f + acc)) * boost;
  return knn_map_score(sim, acc, nq, nv, boost);
}

// Rescore: one wave per candidate doc: exact similarity of the query vector with the doc's vector
// (vec_row[i] = row of candidate i in this segment's matrix, < 0: no vector => "second pass does
// not match"), combined as QueryRescore.combine: (float)(qw * first + rw * second) in double.
__global__ __launch_bounds__(256)
void rescore_vectors_kernel(const float* __restrict__ vecs, const float* __restrict__ vnorm2, int32_t dim,
                            const float* __restrict__ query, float qnorm2, int32_t sim, float boost,
                            const int64_t* __restrict__ vec_row, const float* __restrict__ first_scores, int32_t n,
                            double qw, double rw, float* __restrict__ out_scores) {
  const int32_t i = (int32_t)(blockIdx.x * 4 + (threadIdx.x >> 6));
  const uint32_t lane = threadIdx.x & 63u;
  if (i >= n) return;
  const int64_t row = vec_row[i];
  float second = 0.f;
  if (row >= 0) {
    second = knn_wave_score(sim, vecs + row * dim, query, dim, lane, qnorm2, vnorm2[row], boost);
  }
  if (lane == 0) {
    const double comb = row >= 0 ? qw * (double)first_scores[i] + rw * (double)second : qw * (double)first_scores[i];
    out_scores[i] = (float)comb;
  }
}

// Hybrid tail: the sorted first-pass hits of one query stay in HBM and are
// rescored in place of a host round trip -- one workgroup per query, one wave per hit: doc -> (leaf, row),
// exact similarity with the query's vector (same lane order and reduction as rescore_vectors_kernel, so
// both paths give the same bits), QueryRescore.combine in double, then QueryRescorer's sort
// (combined score desc, doc asc) in LDS and the window.
constexpr int kHybridThreads = 1024;
constexpr int kHybridRows = 4;   // hits a wave scores side by side (their rows' loads in flight together)
// knn_wave_score's partial sums for R rows at once: per row and lane the SAME additions in the same order (k = lane, lane + 64,
// ...), the R rows' loads issued together -- one wave, one row at a time waited a memory round trip per hit
template <int R>
__device__ __forceinline__ void knn_wave_partials(int sim, const float* const (&vp)[R], const float* __restrict__ q, int32_t dim, uint32_t lane,
                                                  float (&acc)[R]) {
  // (the rows' addresses come out of LDS: generic pointers to the compiler, flat loads -- they are global memory)
  typedef const __attribute__((address_space(1))) float* gfloat_ptr;
  gfloat_ptr v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = (gfloat_ptr)vp[r];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  int32_t k = (int32_t)lane;
  for (; k + 192 < dim; k += 256) {   // four strided elements per row and lane
    float x[R][4], qq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) qq[u] = q[k + 64 * u];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < 4; ++u) x[r][u] = v[r] ? v[r][k + 64 * u] : 0.f;   // (v[r]: wave-uniform)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (sim == 2) {
          const float d = qq[u] - x[r][u];
          acc[r] += d * d;
        } else {
          acc[r] += x[r][u] * qq[u];
        }
      }
  }
  for (; k < dim; k += 64) {
    const float qk = q[k];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float xv = v[r] ? v[r][k] : 0.f;
      if (sim == 2) {
        const float d = qk - xv;
        acc[r] += d * d;
      } else {
        acc[r] += xv * qk;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc[r] += __shfl_xor(acc[r], d, 64);
}
__global__ __launch_bounds__(kHybridThreads)
void hybrid_rescore_kernel(const uint64_t* __restrict__ first_keys, const uint32_t* __restrict__ first_counts,
                           uint32_t k_stride, const DVecSeg* __restrict__ segs, int32_t n_segs, int32_t dim,
                           const float* __restrict__ qvecs, const float* __restrict__ qnorm2, int32_t sim, float boost,
                           double qw, double rw, uint32_t window, uint64_t* __restrict__ out_keys,
                           uint32_t* __restrict__ out_counts, uint32_t w_stride, int32_t drop_foreign) {
  // drop_foreign (multi-GPU: the list is the MERGED first pass of all shards): a hit whose doc lies in none of these leaves
  // belongs to another rank -- that rank rescores it; here it is dropped (key 0 sorts last and is not counted)
  // Two phases (round 6; through round 5 a wave took a hit from its key to its score on its own, 63 hits one after the other, each
  // a chain of dependent loads -- key, leaf records, norm, row: 0.46 ms per 256 queries x 1000 hits, 1.7 TB/s):
  //   1. a THREAD per hit: key -> (leaf, row) -> the row's address and norm into LDS -- every hit's chain at once;
  //   2. a wave per FOUR hits: their rows' loads in flight together, knn_wave_score's sums and reduction per row unchanged.
  __shared__ uint64_t cand[1024];      // phase 1: the first-pass keys; phase 2: the combined keys
  __shared__ const float* h_v[1024];   // the hit's row (nullptr: the doc has no vector)
  __shared__ float h_nv[1024];
  __shared__ uint8_t h_mine[1024];
  __shared__ uint32_t n_foreign;
  if (threadIdx.x == 0) n_foreign = 0;
  const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n = min(first_counts[q], 1024u);
  const float* qv = qvecs + (size_t)q * dim;
  const float nq = qnorm2[q];
  for (uint32_t i = tid; i < n; i += (uint32_t)kHybridThreads) {
    const uint64_t key = first_keys[(size_t)q * k_stride + i];
    const uint32_t gdoc = 0xFFFFFFFFu - (uint32_t)key;
    int64_t row = -1;
    const float* v = nullptr;
    float nv = 0.f;
    bool mine = false;
    for (int32_t si = 0; si < n_segs; ++si) {
      const DVecSeg sg = segs[si];
      const int64_t local = (int64_t)gdoc - (int64_t)sg.doc_base;
      if (local < 0 || local >= (int64_t)sg.max_doc) continue;
      mine = true;
      if (sg.vecs) {
        if (!sg.ord_to_doc) {
          if (local < (int64_t)sg.n_vec) row = local;
        } else {  // lower_bound over the leaf's ascending ord -> doc map
          int32_t lo = 0, hi = sg.n_vec;
          while (lo < hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            if (sg.ord_to_doc[mid] < (int32_t)local) lo = mid + 1; else hi = mid;
          }
          if (lo < sg.n_vec && sg.ord_to_doc[lo] == (int32_t)local) row = lo;
        }
        if (row >= 0) {
          v = sg.vecs + row * dim;
          nv = sg.vnorm2[row];
        }
      }
      break;
    }
    cand[i] = key;
    h_v[i] = v;
    h_nv[i] = nv;
    h_mine[i] = mine ? 1u : 0u;
  }
  __syncthreads();
  constexpr uint32_t kWaves = (uint32_t)(kHybridThreads / 64);
  for (uint32_t i0 = wave; i0 < n; i0 += kWaves * (uint32_t)kHybridRows) {
    const float* v[kHybridRows];
    uint32_t idx[kHybridRows];
#pragma unroll
    for (int r = 0; r < kHybridRows; ++r) {
      idx[r] = i0 + (uint32_t)r * kWaves;
      const float* p = idx[r] < n ? h_v[idx[r]] : nullptr;   // (uniform LDS reads: a pointer as scalars)
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)p >> 32));
      v[r] = (const float*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
    }
    float acc[kHybridRows];
    knn_wave_partials<kHybridRows>(sim, v, qv, dim, lane, acc);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < kHybridRows; ++r)
        if (idx[r] < n) {
          const uint64_t key = cand[idx[r]];
          const uint32_t gdoc = 0xFFFFFFFFu - (uint32_t)key;
          const float first = key_score(key);
          float second = 0.f;
          if (v[r]) second = sim == 2 ? (1.0f / (1.0f + acc[r])) * boost : knn_map_score(sim, acc[r], nq, h_nv[idx[r]], boost);
          const double comb = v[r] ? qw * (double)first + rw * (double)second : qw * (double)first;
          const bool foreign = drop_foreign != 0 && h_mine[idx[r]] == 0u;
          cand[idx[r]] = foreign ? 0ull : pack_key((float)comb, gdoc);
          if (foreign) atomicAdd(&n_foreign, 1u);
        }
    }
  }
  uint32_t n2 = 1;
  while (n2 < n) n2 <<= 1;
  for (uint32_t i = n + tid; i < n2; i += (uint32_t)kHybridThreads) cand[i] = 0;
  bitonic_sort_desc<kHybridThreads>(cand, n2);  // starts with a barrier
  const uint32_t m = min(n - n_foreign, window);
  for (uint32_t i = tid; i < w_stride; i += (uint32_t)kHybridThreads) out_keys[(size_t)q * w_stride + i] = i < m ? cand[i] : 0;
  if (tid == 0) out_counts[q] = m;
}

// ---- launchers ------------------------------------------------------------------------------------
void launch_knn_row_norms(hipStream_t st, const float* vecs, int32_t dim, int64_t n, float* norm2) {
  if (n == 0) return;
  hipLaunchKernelGGL(knn_row_norms_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, vecs, dim, n, norm2);
}
void launch_knn_norm_max(hipStream_t st, const float* norm2, int64_t n, uint32_t* out_bits) {
  if (n == 0) return;
  hipLaunchKernelGGL(knn_norm_max_kernel, dim3((uint32_t)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, st, norm2, n, out_bits);
}
size_t knn_score_lds_bytes(int32_t dim, int32_t n_q) { return (size_t)(dim >> 4) * (((n_q > 32 ? 32 : n_q) > 16) ? 128 : 64) * 16; }
int launch_knn_score(hipStream_t st, uint32_t blocks, const float* vecs, const float* vnorm2, const int32_t* ord_to_doc,
                     const uint64_t* live_bits, int32_t dim, int64_t row_begin, int64_t row_end, int32_t doc_base,
                     const float* qpanel, const float* qnorm2, int32_t n_q, int32_t sim, float boost,
                     const unsigned long long* theta, uint64_t* cand, uint32_t* cand_cnt, uint32_t cap, int32_t append_only) {
  if (row_end <= row_begin) return 0;
  const size_t lds = knn_score_lds_bytes(dim, n_q);
  hipError_t e = hipFuncSetAttribute((const void*)knn_score_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(knn_score_kernel, dim3(blocks), dim3(kKnnThreads), lds, st, vecs, vnorm2, ord_to_doc, live_bits, dim,
                     row_begin, row_end, doc_base, qpanel, qnorm2, n_q, sim, boost, theta, cand, cand_cnt, cap, append_only);
  return 0;
}
int32_t knn_sketch_steps(int32_t dim) { return (((dim + 31) >> 5) + 3) & ~3; }   // 32 dimensions per step, whole groups of 4 steps (zero padded)
size_t knn_sketch_bytes(int32_t dim, int64_t n) { return (size_t)((n + 15) >> 4) * (size_t)knn_sketch_steps(dim) * 1024; }
void launch_knn_absmax(hipStream_t st, const float* vecs, int64_t n_elems, uint32_t* out_bits) {
  if (n_elems == 0) return;
  hipLaunchKernelGGL(knn_absmax_kernel, dim3((uint32_t)std::min<int64_t>((n_elems + 255) / 256, 4096)), dim3(256), 0, st, vecs, n_elems, out_bits);
}
void launch_knn_norm_min(hipStream_t st, const float* norm2, int64_t n, uint32_t* out_bits) {
  if (n == 0) return;
  hipLaunchKernelGGL(knn_norm_min_kernel, dim3((uint32_t)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, st, norm2, n, out_bits);
}
void launch_knn_sketch_build(hipStream_t st, const float* vecs, int32_t dim, int64_t n, float scale, void* sketch) {
  if (n == 0) return;
  const int32_t steps = knn_sketch_steps(dim);
  const int64_t pieces = ((n + 15) >> 4) * steps * 64;
  hipLaunchKernelGGL(knn_sketch_build_kernel, dim3((uint32_t)((pieces + 255) / 256)), dim3(256), 0, st, vecs, dim, n, steps, scale, (f16x8*)sketch);
}
void launch_knn_panel_fp16(hipStream_t st, const float* qpanel, const float* qscale, int32_t dim, int32_t n_q, void* panel16) {
  const int32_t steps = knn_sketch_steps(dim), panels = (n_q + 15) >> 4, pieces = steps * panels * 64;
  hipLaunchKernelGGL(knn_panel_fp16_kernel, dim3((uint32_t)((pieces + 255) / 256)), dim3(256), 0, st, qpanel, qscale, dim, steps, panels, n_q,
                     (f16x8*)panel16);
}
size_t knn_sketch_lds_bytes(int32_t dim, int32_t n_q) { return (size_t)knn_sketch_steps(dim) * (size_t)((n_q + 15) >> 4) * 1024; }
// the panel in fp16, the queue's counter, at least a small queue and the kernel's static tables must fit the CU's 160 KB
bool knn_sketch_fits(int32_t dim, int32_t n_q) { return knn_sketch_lds_bytes(dim, n_q) + 16 + 256 * 8 + kKnnSketchStaticLds <= 160 * 1024; }
int launch_knn_sketch(hipStream_t st, uint32_t blocks, const DKnnLeaf* leaves, int32_t n_leaves, int32_t dim, int64_t tile_begin,
                      int64_t tile_end, const void* panel16, const float* qnorm2, const float* qscale, int32_t n_q, int32_t sim,
                      float boost, const unsigned long long* theta, uint64_t* cand, uint32_t* cand_cnt, uint32_t cap, int32_t append_only) {
  if (tile_end <= tile_begin || n_leaves <= 0) return 0;
  const int32_t steps = knn_sketch_steps(dim), panels = (n_q + 15) >> 4;
  const size_t panel_bytes = knn_sketch_lds_bytes(dim, n_q);
  if (!knn_sketch_fits(dim, n_q)) return (int)hipErrorInvalidValue;   // (vectors.cpp only comes here when it fits)
  const uint32_t qcap = (uint32_t)std::min<size_t>(4096, (160 * 1024 - kKnnSketchStaticLds - panel_bytes - 16) / 8);   // the nomination queue behind the panel
  const size_t lds = panel_bytes + 16 + (size_t)qcap * 8;
#define NRT_SKETCH_LAUNCH(PANELS, DEPTH)                                                                                            \
  {                                                                                                                                 \
    hipError_t e = hipFuncSetAttribute((const void*)knn_sketch_kernel<PANELS, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                       (int)lds);                                                                                   \
    if (e != hipSuccess) return (int)e;                                                                                             \
    hipLaunchKernelGGL((knn_sketch_kernel<PANELS, DEPTH>), dim3(blocks), dim3(kKnnThreads), lds, st, leaves, n_leaves, steps,       \
                       tile_begin, tile_end, (const f16x8*)panel16, qnorm2, qscale, n_q, sim, boost, theta, cand, cand_cnt, cap,    \
                       append_only, qcap);                                                                                          \
  }
#define NRT_SKETCH_PANELS(PANELS) \
  if (steps % 8 == 0) NRT_SKETCH_LAUNCH(PANELS, 8) else NRT_SKETCH_LAUNCH(PANELS, 4)
  if (panels <= 1) { NRT_SKETCH_PANELS(1) }
  else if (panels == 2) { NRT_SKETCH_PANELS(2) }
  else if (panels == 3) { NRT_SKETCH_PANELS(3) }
  else { NRT_SKETCH_PANELS(4) }
#undef NRT_SKETCH_PANELS
#undef NRT_SKETCH_LAUNCH
  return 0;
}
void launch_knn_select(hipStream_t st, uint32_t n_q, uint64_t* topk, uint32_t* topk_cnt, uint32_t k_stride, uint32_t k,
                       const uint64_t* cand, uint32_t* cand_cnt, uint32_t cap, unsigned long long* theta,
                       uint32_t* overflow) {
  hipLaunchKernelGGL(knn_select_kernel<false>, dim3(n_q), dim3(kScanThreads), 0, st, topk, topk_cnt, k_stride, k, cand, cand_cnt,
                     cap, theta, overflow, KnnRefine{});
}
void launch_knn_refine_select(hipStream_t st, uint32_t n_q, uint64_t* topk, uint32_t* topk_cnt, uint32_t k_stride, uint32_t k,
                              const uint64_t* cand, uint32_t* cand_cnt, uint32_t cap, unsigned long long* theta, uint32_t* overflow,
                              const DVecSeg* segs, int32_t n_segs, int32_t dim, int32_t sim, const float* qpanel, const float* qnorm2,
                              float boost, const float* ebound, float erel, float min_score, uint32_t k_int, int32_t certify,
                              uint32_t* cert) {
  KnnRefine rf{};
  rf.segs = segs; rf.qpanel = qpanel; rf.qnorm2 = qnorm2; rf.ebound = ebound; rf.cert = cert;
  rf.n_segs = n_segs; rf.dim = dim; rf.sim = sim; rf.certify = certify;
  rf.boost = boost; rf.erel = erel; rf.min_score = min_score; rf.k_int = k_int;
  hipLaunchKernelGGL(knn_select_kernel<true>, dim3(n_q), dim3(kScanThreads), 0, st, topk, topk_cnt, k_stride, k, cand, cand_cnt,
                     cap, theta, overflow, rf);
}
void launch_rescore_vectors(hipStream_t st, const float* vecs, const float* vnorm2, int32_t dim, const float* query,
                            float qnorm2, int32_t sim, float boost, const int64_t* vec_row, const float* first_scores,
                            int32_t n, double qw, double rw, float* out_scores) {
  if (n == 0) return;
  hipLaunchKernelGGL(rescore_vectors_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, vecs, vnorm2, dim, query,
                     qnorm2, sim, boost, vec_row, first_scores, n, qw, rw, out_scores);
}
void launch_hybrid_rescore(hipStream_t st, uint32_t n_queries, const uint64_t* first_keys, const uint32_t* first_counts,
                           uint32_t k_stride, const DVecSeg* segs, int32_t n_segs, int32_t dim, const float* qvecs,
                           const float* qnorm2, int32_t sim, float boost, double qw, double rw, uint32_t window,
                           uint64_t* out_keys, uint32_t* out_counts, uint32_t w_stride, int32_t drop_foreign) {
  if (n_queries == 0) return;
  hipLaunchKernelGGL(hybrid_rescore_kernel, dim3(n_queries), dim3(kHybridThreads), 0, st, first_keys, first_counts, k_stride,
                     segs, n_segs, dim, qvecs, qnorm2, sim, boost, qw, rw, window, out_keys, out_counts, w_stride, drop_foreign);
}

// Multi-GPU hybrid: the hit totals a rank's rescored window carries into the second exchange.  The first pass's total (and
// its relation tag, plan.h: kHitsPrunedUnit) is the MERGED one, the same on every rank: rank 0 carries it, the others 0, so
// the merge's sum is that total.  GREATER_THAN_OR_EQUAL_TO needs the first pass's queue full (numHits hits): else the tag is
// dropped here, as unpack_topdocs does for one GPU.
__global__ __launch_bounds__(256)
void hybrid_hits_kernel(const uint64_t* __restrict__ first_hits, const uint32_t* __restrict__ first_counts,
                        const uint32_t* __restrict__ q_k, int32_t carries, uint64_t* __restrict__ out_hits, uint32_t n) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  uint64_t h = first_hits[q];
  if (first_counts[q] < q_k[q]) h &= kHitsPrunedUnit - 1ull;
  out_hits[q] = carries ? h : 0ull;
}
void launch_hybrid_hits(hipStream_t st, const uint64_t* first_hits, const uint32_t* first_counts, const uint32_t* q_k, int32_t carries,
                        uint64_t* out_hits, uint32_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(hybrid_hits_kernel, dim3((n + 255) / 256), dim3(256), 0, st, first_hits, first_counts, q_k, carries, out_hits, n);
}

}  // namespace nrtgpu
