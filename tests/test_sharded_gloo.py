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
from oramacore_amd.sharded import PAD_ID, ShardedSearcher, ShardPlan, block_views, packed_block_bytes


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
