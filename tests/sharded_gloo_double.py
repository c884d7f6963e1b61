"""TEST DOUBLE (not part of the product): the round-1 torch.distributed form of the sharded search, kept so that the
exchange protocol (packed blocks, one all-gather, deterministic merge; df / min-max reductions) stays covered by
world_size-2 gloo processes on a CPU-only box.  The product's exchange lives in liborama_hip.so (shard_group.hip) and is
tested with >1 rank in tests/test_multirank_gpu.py and tests/test_mock_rccl.py.

Multi-GPU path: static corpus sharding + ONE all-gather of per-shard top-k candidates.

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

from oramacore_amd import _native as N

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


# ===================================================================== full-text / hybrid over a sharded index
def post_block_bytes(k: int) -> int:
    """Mirror of orama_post_block_bytes: [k u64 ids][k f32 scores][pad to 8][u64 count]."""
    return ((k * 12 + 7) & ~7) + 8


class HipPostOps:
    """The product's compute steps for one shard of a full-text index — K3/K5/K4 through the staged C ABI
    (orama_post_query_*), K6 + count sum through orama_post_merge_blocks_device."""

    def __init__(self, ctx, store):
        self.lib = N.load()
        self.ctx = ctx
        self.store = store

    def begin(self, query: dict, d_df: torch.Tensor):
        stream = torch.cuda.current_stream().cuda_stream
        return self.store.staged_query(d_df_ptr=d_df.data_ptr(), stream=stream, **query)

    def score(self, q, df_global, d_minmax) -> None:
        q.score(df_global, d_minmax.data_ptr() if d_minmax is not None else None)

    def finish(self, q, d_minmax, vector, d_block: torch.Tensor) -> None:
        q.finish(d_minmax.data_ptr() if d_minmax is not None else None, vector, d_block.data_ptr())

    def end(self, q) -> None:
        q.end()

    def merge(self, blocks: torch.Tensor, lists: int, k: int, out_ids, out_scores, out_n, out_count) -> None:
        stream = torch.cuda.current_stream().cuda_stream
        N.check(self.lib.orama_post_merge_blocks_device(self.ctx.handle, blocks.data_ptr(), lists, k,
                                                        out_ids.data_ptr(), out_scores.data_ptr(), out_n.data_ptr(),
                                                        out_count.data_ptr(), stream))


class ShardedFulltextSearcher:
    """search_full_text / search_hybrid (token_score.rs:186-387) over ONE index whose documents are split into
    contiguous doc-id ranges, one per rank.  The reference reads three index-wide quantities; each
    becomes one small collective between the stages of the local query:

        df per token (token_score.rs:262-275)            all-reduce SUM   int32[n_tokens]
        min/max of the score maps (token_score.rs:398-401) all-reduce MAX   int64[2]        (hybrid only)
        top-k + match count (sort.rs:260-279, search.rs:482) all-gather      [k ids][k scores][count] per rank

    N (`total_documents`) and the per-field average lengths are index-wide constants given at build time.  The
    vector map of a hybrid query is the GLOBAL map (ShardedSearcher + a2 epilogue), identical on every rank; each
    rank adds the entries whose document it owns."""

    def __init__(self, ops, rank: int, world: int, device: torch.device, group=None):
        self.ops = ops
        self.rank = rank
        self.world = world
        self.device = device
        self.group = group

    def search(self, refs, n_tokens: int, total_documents: float, top_k: int, threshold=None, allow=None,
               apply_omc: bool = True, vector=None, **kw):
        """Returns (ids u64[n], scores f32[n], count) — identical on every rank."""
        dev = self.device
        hybrid = vector is not None
        n_vec = 0 if vector is None else (len(vector) if isinstance(vector, dict) else len(vector[0]))
        d_df = torch.zeros((n_tokens,), dtype=torch.int32, device=dev)
        d_minmax = torch.zeros((2,), dtype=torch.int64, device=dev) if hybrid else None
        nb = post_block_bytes(top_k)
        block = torch.zeros((nb,), dtype=torch.uint8, device=dev)
        blocks = torch.zeros((self.world * nb,), dtype=torch.uint8, device=dev)
        out_ids = torch.zeros((top_k,), dtype=torch.int64, device=dev)
        out_sc = torch.zeros((top_k,), dtype=torch.float32, device=dev)
        out_n = torch.zeros((1,), dtype=torch.int32, device=dev)
        out_count = torch.zeros((1,), dtype=torch.int64, device=dev)
        query = dict(refs=refs, n_tokens=n_tokens, total_documents=total_documents, top_k=top_k,
                     threshold=threshold, allow=allow, apply_omc=apply_omc, hybrid=hybrid, n_vec_cap=n_vec, **kw)
        q = self.ops.begin(query, d_df)
        try:
            if self.world > 1:
                dist.all_reduce(d_df, op=dist.ReduceOp.SUM, group=self.group)
            df_global = d_df.cpu().numpy().astype("uint32")
            self.ops.score(q, df_global, d_minmax)
            if hybrid and self.world > 1:
                dist.all_reduce(d_minmax, op=dist.ReduceOp.MAX, group=self.group)
            self.ops.finish(q, d_minmax, vector, block)
            if self.world > 1:
                dist.all_gather_into_tensor(blocks, block, group=self.group)
            else:
                blocks.copy_(block)
            self.ops.merge(blocks, self.world, top_k, out_ids, out_sc, out_n, out_count)
            n = int(out_n.cpu()[0])
            ids = out_ids.cpu().numpy().view("uint64")[:n].copy()
            sc = out_sc.cpu().numpy()[:n].copy()
            count = int(out_count.cpu()[0])
        finally:
            self.ops.end(q)
        return ids, sc, count
