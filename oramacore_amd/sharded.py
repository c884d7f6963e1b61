"""Multi-GPU path: static corpus sharding + ONE all-gather of per-shard top-k candidates.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Rows are
partitioned into contiguous doc-id ranges; every rank scans its shard for the whole
query batch, writes its k candidates per query into one packed block
    [q*k u64 ids][q*k f32 distances]   (12 B per candidate, padded to 8 B)
and a single all-gather moves all blocks (payload q*k*12 B per rank — latency-bound on xGMI, every
peer pair has its own link).  Every rank then merges the world*k candidates with the same
deterministic rule as the single-GPU path (distance asc, id asc), so all ranks hold the identical
global top-k.

torch is plumbing here (device buffers, streams, the collective); the scan, the top-k and the merge
are the HIP kernels behind include/orama_hip.h.  The compute steps are injected as `ops` so that the
exchange logic can be exercised on CPU with the gloo backend (tests/test_sharded_gloo.py).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist

from . import _native as N

PAD_ID = (1 << 64) - 1


def packed_block_bytes(q: int, k: int) -> int:
    """Mirror of orama_packed_block_bytes (include/orama_hip.h)."""
    return (q * k * 12 + 7) & ~7


@dataclass(frozen=True)
class ShardPlan:
    """Static contiguous ranges: shard g holds rows [g*N/G, (g+1)*N/G)."""

    n_total: int
    world: int

    def range(self, rank: int) -> tuple[int, int]:
        lo = (self.n_total * rank) // self.world
        hi = (self.n_total * (rank + 1)) // self.world
        return lo, hi

    def rows(self, rank: int) -> int:
        lo, hi = self.range(rank)
        return hi - lo


class HipOps:
    """The product's compute steps: K1+K4 on the local shard, K6 merge — through the C ABI."""

    def __init__(self, ctx, store, scan_stream=None):
        self.lib = N.load()
        self.ctx = ctx
        self.store = store
        self.scan_stream = scan_stream  # torch.cuda.Stream shared by all searchers: corpus scans run there, in order

    def local_topk(self, queries: torch.Tensor, k: int, block: torch.Tensor, out_n: torch.Tensor) -> None:
        stream = torch.cuda.current_stream().cuda_stream
        if self.scan_stream is None:
            N.check(self.lib.orama_vec_search_packed_device(self.store.handle, queries.data_ptr(), queries.shape[0], k,
                                                            None, 0, block.data_ptr(), out_n.data_ptr(), stream))
        else:
            N.check(self.lib.orama_vec_search_packed_device2(self.store.handle, queries.data_ptr(), queries.shape[0],
                                                             k, None, 0, block.data_ptr(), out_n.data_ptr(),
                                                             self.scan_stream.cuda_stream, stream))

    def merge(self, blocks: torch.Tensor, lists: int, q: int, k: int, out_ids: torch.Tensor,
              out_dist: torch.Tensor, out_n: torch.Tensor) -> None:
        stream = torch.cuda.current_stream().cuda_stream
        N.check(self.lib.orama_merge_packed_device(self.ctx.handle, blocks.data_ptr(), lists, q, k,
                                                   out_ids.data_ptr(), out_dist.data_ptr(), out_n.data_ptr(),
                                                   stream))


def block_views(block: torch.Tensor, q: int, k: int):
    """(ids [q,k] int64 view, dist [q,k] float32 view) of one packed block (a uint8 tensor)."""
    ids = block[: q * k * 8].view(torch.int64).view(q, k)
    dst = block[q * k * 8: q * k * 12].view(torch.float32).view(q, k)
    return ids, dst


class ShardedSearcher:
    """search(): local top-k → one all-gather → merge. Buffers are allocated once per (q, k) shape."""

    def __init__(self, ops, rank: int, world: int, device: torch.device, group=None,
                 always_exchange: bool = False):
        self.ops = ops
        self.always_exchange = always_exchange  # run the all-gather + merge even at world == 1 (tests)
        self.rank = rank
        self.world = world
        self.device = device
        self.group = group
        self._shape = None

    def _alloc(self, q: int, k: int) -> None:
        if self._shape == (q, k):
            return
        dev = self.device
        nb = packed_block_bytes(q, k)
        self.block = torch.empty((nb,), dtype=torch.uint8, device=dev)
        self.loc_n = torch.empty((q,), dtype=torch.int32, device=dev)
        self.blocks = torch.empty((self.world * nb,), dtype=torch.uint8, device=dev)
        # ids are u64 on the device; torch carries them as int64 (same bits)
        self.out_ids = torch.empty((q, k), dtype=torch.int64, device=dev)
        self.out_dist = torch.empty((q, k), dtype=torch.float32, device=dev)
        self.out_n = torch.empty((q,), dtype=torch.int32, device=dev)
        self._shape = (q, k)

    def search(self, queries: torch.Tensor, k: int):
        """queries: [q, dim] f32 on self.device (replicated on every rank). Returns device tensors
        (ids [q,k] int64-as-u64, dist [q,k], n [q]); no host synchronisation."""
        q = queries.shape[0]
        self._alloc(q, k)
        self.ops.local_topk(queries, k, self.block, self.loc_n)
        if self.world == 1 and not self.always_exchange:
            ids, dst = block_views(self.block, q, k)
            return ids, dst, self.loc_n
        dist.all_gather_into_tensor(self.blocks, self.block, group=self.group)
        self.ops.merge(self.blocks, self.world, q, k, self.out_ids, self.out_dist, self.out_n)
        return self.out_ids, self.out_dist, self.out_n
