"""One process per GPU: index sharding and the final exchange of SURVEY.md section 8e.

  * partition: rank r owns a contiguous docid range (tile aligned) of the index -- large segments are
    cut by docid range, exactly what Lucene 10's LeafReaderContextPartition allows
    (/root/reference/src/main/java/com/yelp/nrtsearch/server/search/MyIndexSearcher.java:172-187);
    BM25 statistics stay index-global (idf/avgdl are computed once on the host, never per GPU).
  * exchange: every rank holds, per query, its local top-k as packed keys
    (float_bits(score) << 32 | 0xFFFFFFFF - global_doc) plus counts and hit totals; ONE RCCL
    all-gather per array (torch.distributed backend "nccl" == RCCL over xGMI) puts all ranks'
    lists next to each other and every rank runs the same TopDocs.merge kernel
    (LazyQueueTopScoreDocCollectorManager.java:137-144): all_gather_topk.  all_to_all_topk is the split
    form of the same reduce: rank r receives every rank's lists for ITS slice of the batch's queries and
    merges only those (1/W of the bytes and of the merge work; xGMI is point-to-point).
    No other data-path collective.

The same code runs on CPU tensors with the gloo backend (tests/test_dist_gloo.py).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def pack_keys(docs: np.ndarray, scores: np.ndarray, k_stride: int) -> np.ndarray:
    """(score desc, doc asc) hits -> int64[k_stride] packed keys, unused tail 0 (device layout)."""
    out = np.zeros(k_stride, dtype=np.uint64)
    n = len(docs)
    bits = np.asarray(scores, dtype=np.float32).view(np.uint32).astype(np.uint64)
    out[:n] = (bits << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - np.asarray(docs, dtype=np.uint64))
    return out.view(np.int64)


def unpack_keys(keys: np.ndarray, n: int) -> Tuple[np.ndarray, np.ndarray]:
    u = np.asarray(keys[:n]).view(np.uint64)
    docs = (np.uint64(0xFFFFFFFF) - (u & np.uint64(0xFFFFFFFF))).astype(np.int32)
    scores = (u >> np.uint64(32)).astype(np.uint32).view(np.float32)
    return docs, scores


def topology(world: int, rank: int, doc_shards: int = 0):
    """N = D doc-shards x R query-groups (bench.py --doc-shards): D consecutive ranks share the index by docid range and
    exchange their top-k among themselves; the R = N / D groups hold the same index and take different batches.  doc_shards 0 = N
    (every GPU a shard of ONE search: BASELINE.json's north-star form); 1 = N replicas.  -> (D, R, group, rank within the group).
    The reference's analogue: virtual shards balance the leaves over D searcher threads (MyIndexSearcher.java:117-160) while R
    replicas of the index serve different requests (nrtsearch's replica nodes)."""
    d = doc_shards if doc_shards > 0 else world
    if d > world or world % d != 0:
        raise ValueError(f"doc_shards {d} does not divide the world size {world}")
    return d, world // d, rank // d, rank % d


def doc_shard_groups(world: int, doc_shards: int):
    """torch.distributed process groups of a D x R topology: every rank creates every group (in the same order) and gets ITS
    group back (None: the whole world, D == N; or no exchange at all, D == 1)."""
    import torch.distributed as dist

    d, r, group, _ = topology(world, dist.get_rank(), doc_shards)
    if d == world or d == 1:
        return None
    mine = None
    for g in range(r):
        h = dist.new_group(list(range(g * d, (g + 1) * d)))
        if g == group:
            mine = h
    return mine


def all_gather_topk(keys, counts, hits, group=None):
    """keys [B, k_stride] int64, counts [B] int32, hits [B] int64 (same device on every rank)
    -> gathered ([W, B, k_stride], [W, B], [W, B]) on every rank (of `group`: a doc-shard group, doc_shard_groups)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    b = keys.shape[0]
    # concatenation form (world * B rows): accepted by both RCCL and gloo
    g_keys = torch.empty((world * b,) + tuple(keys.shape[1:]), dtype=keys.dtype, device=keys.device)
    g_cnt = torch.empty((world * b,), dtype=counts.dtype, device=counts.device)
    g_hits = torch.empty((world * b,), dtype=hits.dtype, device=hits.device)
    dist.all_gather_into_tensor(g_keys, keys.contiguous(), group=group)
    dist.all_gather_into_tensor(g_cnt, counts.contiguous(), group=group)
    dist.all_gather_into_tensor(g_hits, hits.contiguous(), group=group)
    return g_keys.view((world, b) + tuple(keys.shape[1:])), g_cnt.view(world, b), g_hits.view(world, b)


def all_to_all_topk(keys, counts, hits, group=None):
    """keys [B, k_stride] int64, counts [B] int32, hits [B] int64 with B % world == 0.  Rank r receives every
    rank's lists for queries [r * B/W, (r + 1) * B/W): ([W, B/W, k_stride], [W, B/W], [W, B/W])."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    b = keys.shape[0]
    if b % world != 0:
        raise ValueError(f"batch of {b} queries does not divide by world size {world}")
    o_keys = torch.empty_like(keys)
    o_cnt = torch.empty_like(counts)
    o_hits = torch.empty_like(hits)
    dist.all_to_all_single(o_keys, keys.contiguous(), group=group)
    dist.all_to_all_single(o_cnt, counts.contiguous(), group=group)
    dist.all_to_all_single(o_hits, hits.contiguous(), group=group)
    per = b // world
    return o_keys.view((world, per) + tuple(keys.shape[1:])), o_cnt.view(world, per), o_hits.view(world, per)
