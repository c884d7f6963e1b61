"""CPU tests of the multi-GPU exchange logic (world_size 2 and 3, gloo backend).

The compute steps (local scan + top-k, merge) are injected: here they are test doubles backed by the
oracle, so what is exercised is the product's sharding plan, packed candidate block layout, the single
all-gather and the merge call order — the same `ShardedSearcher` code the GPU ranks run over RCCL.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from oracle import oracle as orc
from oramacore_amd import _native as N
from sharded_gloo_double import PAD_ID, ShardedSearcher, ShardPlan, block_views, packed_block_bytes


class OracleOps:
    """Test double of HipOps: same interface, CPU tensors, oracle arithmetic."""

    def __init__(self, corpus, row_doc):
        self.corpus = corpus
        self.row_doc = row_doc

    def local_topk(self, queries, k, block, out_n):
        q = queries.shape[0]
        ids, dst = block_views(block, q, k)
        for qi in range(q):
            d, v, _ = orc.vector_search(self.corpus, self.row_doc, queries[qi].numpy(), k)
            m = len(d)
            ids[qi, :m] = torch.from_numpy(d.view(np.int64).copy())
            dst[qi, :m] = torch.from_numpy(v.copy())
            ids[qi, m:] = -1  # UINT64_MAX padding
            dst[qi, m:] = float("inf")
            out_n[qi] = m

    def merge(self, blocks, lists, q, k, out_ids, out_dist, out_n):
        nb = packed_block_bytes(q, k)
        for qi in range(q):
            all_ids, all_d = [], []
            for l in range(lists):
                ids, dst = block_views(blocks[l * nb:(l + 1) * nb], q, k)
                all_ids.append(ids[qi].numpy().view(np.uint64))
                all_d.append(dst[qi].numpy())
            ids_c, d_c = np.concatenate(all_ids), np.concatenate(all_d)
            keep = ids_c != np.uint64(PAD_ID)
            md, ms = orc.top_n(ids_c[keep], -d_c[keep], k)  # distance asc == (-distance) desc, id asc
            m = len(md)
            out_ids[qi, :m] = torch.from_numpy(md.view(np.int64).copy())
            out_dist[qi, :m] = torch.from_numpy((-ms).copy())
            out_ids[qi, m:] = -1
            out_dist[qi, m:] = float("inf")
            out_n[qi] = m


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, k, q, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        corpus = util.gaussian_rows(n, d, seed=42)
        queries = util.gaussian_rows(q, d, seed=43)
        plan = ShardPlan(n, world)
        lo, hi = plan.range(rank)
        ops = OracleOps(corpus[lo:hi], np.arange(lo, hi, dtype=np.uint64))
        searcher = ShardedSearcher(ops, rank, world, torch.device("cpu"))
        ids, dst, cnt = searcher.search(torch.from_numpy(queries), k)
        ret[rank] = (ids.numpy().view(np.uint64).copy(), dst.numpy().copy(), cnt.numpy().copy())
        # a second search with another shape re-allocates the exchange buffers
        ids2, dst2, cnt2 = searcher.search(torch.from_numpy(queries[:1]), 3)
        assert ids2.shape == (1, 3)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,k", [(2, 1000, 10), (3, 1001, 7), (2, 5, 10)])
def test_sharded_search_matches_single_shard(world, n, k):
    d, q = 64, 4
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, d, k, q, ret), nprocs=world, join=True)
    corpus = util.gaussian_rows(n, d, seed=42)
    queries = util.gaussian_rows(q, d, seed=43)
    for qi in range(q):
        e_ids, e_d, _ = orc.vector_search(corpus, np.arange(n, dtype=np.uint64), queries[qi], k)
        m = len(e_ids)
        for r in range(world):  # every rank holds the identical global top-k
            ids, dst, cnt = ret[r]
            assert cnt[qi] == m == min(k, n)
            assert ids[qi, :m].tolist() == e_ids.tolist()
            assert np.array_equal(dst[qi, :m], e_d)


def test_shard_plan_covers_rows_exactly_once():
    for n, world in ((10_000_000, 8), (7, 3), (5, 8), (0, 2), (80_000_000, 8)):
        plan = ShardPlan(n, world)
        cover = 0
        prev_hi = 0
        for r in range(world):
            lo, hi = plan.range(r)
            assert lo == prev_hi and hi >= lo
            prev_hi = hi
            cover += hi - lo
        assert cover == n and prev_hi == n
        sizes = [plan.rows(r) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_packed_block_layout_matches_the_abi():
    lib = N.load()
    for q, k in ((1, 100), (64, 100), (256, 100), (3, 7), (1, 1)):
        assert lib.orama_packed_block_bytes(q, k) == packed_block_bytes(q, k)
        assert packed_block_bytes(q, k) % 8 == 0 and packed_block_bytes(q, k) >= q * k * 12
    blk = torch.zeros(packed_block_bytes(3, 7), dtype=torch.uint8)
    ids, dst = block_views(blk, 3, 7)
    ids[2, 6] = 5
    dst[0, 0] = 1.0
    assert blk[: 3 * 7 * 8].view(torch.int64)[-1] == 5
    assert blk[3 * 7 * 8: 3 * 7 * 8 + 4].view(torch.float32)[0] == 1.0


# ===================================================================== full-text / hybrid over a sharded index
from sharded_gloo_double import ShardedFulltextSearcher, post_block_bytes  # noqa: E402

F = np.float32


def _ordered(x) -> int:
    """f32 -> order-preserving u32 (csrc/device_utils.hpp f32_to_ordered)."""
    x = F(x)
    if x == 0:
        x = F(0.0)
    u = int(np.array([x], dtype=np.float32).view(np.uint32)[0])
    return (~u) & 0xffffffff if u & 0x80000000 else u | 0x80000000


def _unordered(k: int) -> np.float32:
    u = (k & 0x7fffffff) if k & 0x80000000 else (~k) & 0xffffffff
    return np.array([u], dtype=np.uint32).view(np.float32)[0]


class OraclePostOps:
    """Test double of HipPostOps: same stages on CPU tensors, f32 arithmetic in the oracle's operation order.
    `refs` of the query are the shard's own contributions [(token, docs, ntfs)]; `owned(doc)` is the doc range."""

    def __init__(self, lo_doc: int, hi_doc: int):
        self.lo, self.hi = lo_doc, hi_doc

    def begin(self, query, d_df):
        st = {"q": query, "S": [dict() for _ in range(query["n_tokens"])]}
        for tok, docs, ntfs in query["refs"]:
            acc = st["S"][tok]
            for d, v in zip(docs, ntfs):
                acc[int(d)] = F(acc.get(int(d), F(0.0)) + F(1.0) * F(v))
        for t in range(query["n_tokens"]):
            d_df[t] = len(st["S"][t])
        return st

    def score(self, st, df_global, d_minmax):
        q = st["q"]
        k = F(1.2)
        scores, mask = {}, {}
        for t in range(q["n_tokens"]):
            idf = orc.bm25_idf(q["total_documents"], max(int(df_global[t]), 1))
            for d, s in st["S"][t].items():
                if not np.isfinite(s) or s == 0 or abs(s) < np.finfo(np.float32).tiny:
                    continue
                term = orc.bm25f_score(s, k, idf)
                if np.isnan(term):
                    continue
                scores[d] = F(scores.get(d, F(0.0)) + term * F(1.0))
                mask[d] = mask.get(d, 0) | (1 << (t & 31))
        if q["threshold"] is not None:
            scores = {d: s for d, s in scores.items() if bin(mask[d]).count("1") >= q["threshold"]}
        st["scores"] = scores
        if d_minmax is not None:
            keys = [_ordered(s) for s in scores.values() if not np.isnan(s)]
            d_minmax[0] = max(keys) if keys else 0
            d_minmax[1] = 0xffffffff - (min(keys) if keys else 0xffffffff)

    def finish(self, st, d_minmax, vector, d_block):
        q = st["q"]
        scores = st["scores"]
        if vector is not None:
            mx = mn = F(0.0)
            for v in vector.values():
                mx, mn = max(mx, F(v)), min(mn, F(v))
            kmax, kmin = int(d_minmax[0]), 0xffffffff - int(d_minmax[1])
            if kmax != 0:
                mx = max(mx, _unordered(kmax))
            if kmin != 0xffffffff:
                mn = min(mn, _unordered(kmin))
            den = F(mx - mn)
            with np.errstate(invalid="ignore", divide="ignore"):
                scores = {d: F(F(s - mn) / den) for d, s in scores.items()}
                for d, v in vector.items():
                    if self.lo <= d < self.hi:
                        scores[d] = F(scores.get(d, F(0.0)) + F(F(F(v) - mn) / den))
        k = q["top_k"]
        docs = np.array(sorted(scores), dtype=np.uint64)
        vals = np.array([scores[int(d)] for d in docs], dtype=np.float32)
        td, ts = orc.top_n(docs, vals, k)
        ids = d_block[: k * 8].view(torch.int64)
        sc = d_block[k * 8: k * 12].view(torch.float32)
        ids[:] = -1
        sc[:] = float("-inf")
        ids[: len(td)] = torch.from_numpy(td.view(np.int64).copy())
        sc[: len(td)] = torch.from_numpy(ts.copy())
        d_block[post_block_bytes(k) - 8:].view(torch.int64)[0] = len(scores)

    def end(self, st):
        pass

    def merge(self, blocks, lists, k, out_ids, out_scores, out_n, out_count):
        nb = post_block_bytes(k)
        ids_c, sc_c, count = [], [], 0
        for l in range(lists):
            b = blocks[l * nb:(l + 1) * nb]
            ids_c.append(b[: k * 8].view(torch.int64).numpy().view(np.uint64))
            sc_c.append(b[k * 8: k * 12].view(torch.float32).numpy())
            count += int(b[nb - 8:].view(torch.int64)[0])
        ids_c, sc_c = np.concatenate(ids_c), np.concatenate(sc_c)
        keep = ids_c != np.uint64(PAD_ID)
        md, ms = orc.top_n(ids_c[keep], sc_c[keep], k)
        out_ids[: len(md)] = torch.from_numpy(md.view(np.int64).copy())
        out_scores[: len(md)] = torch.from_numpy(ms.copy())
        out_n[0] = len(md)
        out_count[0] = count


def _ft_corpus(n_docs=400, n_tokens=4, seed=5):
    """Small multi-entry contributions: per token 2 entries (fields) over random doc subsets; doc id = 10*i + 3."""
    rng = np.random.default_rng(seed)
    doc_ids = np.arange(n_docs, dtype=np.uint64) * np.uint64(10) + np.uint64(3)
    entries = []
    for t in range(n_tokens):
        for _ in range(2):
            m = int(rng.integers(5, n_docs // 2))
            pos = np.sort(rng.choice(n_docs, size=m, replace=False))
            entries.append((t, doc_ids[pos], rng.uniform(0.05, 3.0, size=m).astype(np.float32)))
    return doc_ids, entries


def _ft_worker(rank, world, port, top_k, threshold, hybrid, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        doc_ids, entries = _ft_corpus()
        plan = ShardPlan(len(doc_ids), world)
        lo, hi = plan.range(rank)
        lo_doc = int(doc_ids[lo]) if lo < len(doc_ids) else 1 << 62
        hi_doc = int(doc_ids[hi]) if hi < len(doc_ids) else 1 << 62
        local = []
        for t, d, v in entries:
            keep = (d >= lo_doc) & (d < hi_doc)
            local.append((t, d[keep], v[keep]))
        vec = {int(doc_ids[7]): 0.9, int(doc_ids[-1]): -0.1, int(doc_ids[len(doc_ids) // 2]): 0.45} if hybrid else None
        s = ShardedFulltextSearcher(OraclePostOps(lo_doc, hi_doc), rank, world, torch.device("cpu"))
        ids, sc, count = s.search(local, 4, float(len(doc_ids)), top_k, threshold=threshold, vector=vec)
        ret[rank] = (ids, sc, count)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,top_k,threshold,hybrid", [(2, 25, None, False), (3, 10, 2, False), (2, 25, None, True),
                                                          (3, 400, 3, True)])
def test_sharded_fulltext_matches_single_index(world, top_k, threshold, hybrid):
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_ft_worker, args=(world, port, top_k, threshold, hybrid, ret), nprocs=world, join=True)
    doc_ids, entries = _ft_corpus()
    fd, fs = orc.search_full_text(entries, 4, float(len(doc_ids)), 1.2, threshold)
    if hybrid:
        vec = {int(doc_ids[7]): 0.9, int(doc_ids[-1]): -0.1, int(doc_ids[len(doc_ids) // 2]): 0.45}
        fd, fs = orc.normalize_and_combine(list(vec), list(vec.values()), fd, fs)
    td, ts = orc.top_n(fd, fs, top_k)
    for r in range(world):
        ids, sc, count = ret[r]
        assert count == len(fd)
        assert ids.tolist() == td.tolist()
        assert np.array_equal(sc.view(np.uint32), ts.view(np.uint32))


def test_post_block_layout_matches_the_abi():
    lib = N.load()
    for k in (1, 7, 100, 4096):
        assert lib.orama_post_block_bytes(k) == post_block_bytes(k)
        assert post_block_bytes(k) % 8 == 0 and post_block_bytes(k) >= k * 12 + 8
