"""TopDocs.merge over MANY lists per query (nrtgpu_merge_topk_device -> merge_topk_kernel): since round 6 the kernel takes a thread
per list -- record and count at once, an exclusive scan of the counts, the lists' keys as one sequence -- in blocks of 768 lists.
A batch's own merge sees a few items plus up to ~250 helper slots per query; this drives the general shape: more lists than one
block, most of them empty, counts from 0 to the full stride, and compares with a sort on the host (TopDocs.merge's order: score
descending, docid ascending: LazyQueueTopScoreDocCollectorManager.java:137-144)."""
import numpy as np
import pytest

from nrtsearch_amd import api

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_lists,k", [(3, 100), (700, 100), (1000, 1000), (1900, 37)])
def test_merge_of_many_mostly_empty_lists(n_lists, k):
    import torch

    rng = np.random.default_rng(n_lists * 7 + k)
    n_q = 3
    k_stride = (k + 15) // 16 * 16
    ctx = api.GpuContext(0, max_batch=16)
    try:
        keys = np.zeros((n_lists, n_q, k_stride), dtype=np.uint64)
        cnt = np.zeros((n_lists, n_q), dtype=np.int32)
        hits = np.zeros((n_lists, n_q), dtype=np.int64)
        expect = []
        for qi in range(n_q):
            all_keys = []
            doc = 0
            for l in range(n_lists):
                # most lists empty, some short, a few full; query 2: every list empty but one
                r = rng.random()
                c = 0 if r < 0.6 else (int(rng.integers(1, 8)) if r < 0.9 else int(rng.integers(k_stride // 2, k_stride + 1)))
                if qi == 2:
                    c = min(k_stride, 5) if l == n_lists // 2 else 0
                c = min(c, k)   # (a shard returns at most numHits keys)
                scores = rng.integers(1, 50, size=c).astype(np.float32) * 0.25   # few distinct scores: docids decide
                docs = np.arange(doc, doc + c, dtype=np.uint64) * 3 + (l % 3)
                doc += c
                kk = (scores.view(np.uint32).astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - docs)
                kk = np.sort(kk)[::-1]   # a list arrives sorted, best first
                keys[l, qi, :c] = kk
                cnt[l, qi] = c
                hits[l, qi] = c + int(rng.integers(0, 3))
                all_keys.append(kk)
            merged = np.sort(np.concatenate(all_keys) if all_keys else np.zeros(0, np.uint64))[::-1][:k]
            expect.append(merged)
        d_keys = torch.from_numpy(keys.view(np.int64)).cuda()
        d_cnt = torch.from_numpy(cnt).cuda()
        d_hits = torch.from_numpy(hits).cuda()
        pm = api.PreparedMerge(ctx, n_lists, n_q, k_stride, [k] * n_q, [api.TOTAL_HITS_THRESHOLD] * n_q)
        pm.run(d_keys.data_ptr(), d_cnt.data_ptr(), d_hits.data_ptr())
        for qi in range(n_q):
            got = pm.topdocs(qi)
            e = expect[qi]
            edocs = (np.uint64(0xFFFFFFFF) - (e & np.uint64(0xFFFFFFFF))).astype(np.int64)
            escores = (e >> np.uint64(32)).astype(np.uint32)
            assert got.docs.astype(np.int64).tolist() == edocs.tolist(), f"query {qi}: docids / ranks ({n_lists} lists)"
            assert got.scores.view(np.uint32).tolist() == escores.tolist(), f"query {qi}: score bits"
            assert got.total_hits == int(hits[:, qi].sum()), f"query {qi}: totalHits is the sum of the lists'"
    finally:
        ctx.close()
