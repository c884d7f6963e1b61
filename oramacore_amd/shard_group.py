"""One index sharded over several GPUs — thin binding of the orama_shard_* entry points (include/orama_hip.h).

The exchange (RCCL all-gather / all-reduce over xGMI, or device-local reductions when the shards share one GPU)
runs INSIDE liborama_hip.so; this module only marshals arguments.  No torch.

    g = ShardGroup([0, 1, 2, 3])                    # one process, one shard per GPU (ncclCommInitAll)
    g = ShardGroup([0, 0, 0, 0])                    # four shards on one GPU (tests, 1-GPU boxes): no RCCL
    g = ShardGroup.from_rank(uid, rank, world, dev) # one process per GPU (bench.py under torch.distributed.run)
    stores = [EmbeddingFieldStorage(g.ctx(i), ...) for i in range(g.n_local)]
    ids, dist, n = g.vec_search(stores, queries, k)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .context import Context
from .fulltext import B_DEFAULT, K1_DEFAULT, _params

FORCE_RCCL = 1


class _BorrowedContext(Context):
    """Context owned by a shard group (not destroyed from Python)."""

    def __init__(self, lib, handle, device):  # noqa: D107 - no orama_ctx_create
        self._lib = lib
        self._h = C.c_void_p(handle)
        self.device = device

    def close(self) -> None:
        self._close_children()
        self._h = None


class ShardGroup:
    def __init__(self, devices, flags: int = 0, _handle=None):
        self._lib = N.load()
        if _handle is None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            h = C.c_void_p()
            N.check(self._lib.orama_shard_group_create(devs, len(devices), int(flags), C.byref(h)))
            _handle = h
        self._h = _handle
        w, nl, r0, rc = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int()
        N.check(self._lib.orama_shard_group_info(self._h, C.byref(w), C.byref(nl), C.byref(r0), C.byref(rc)))
        self.world, self.n_local, self.first_rank, self.uses_rccl = w.value, nl.value, r0.value, bool(rc.value)
        self._ctxs = [_BorrowedContext(self._lib, self._lib.orama_shard_group_ctx(self._h, i), -1)
                      for i in range(self.n_local)]

    @staticmethod
    def unique_id() -> bytes:
        lib = N.load()
        buf = C.create_string_buffer(128)
        N.check(lib.orama_shard_unique_id(buf))
        return buf.raw

    @classmethod
    def from_rank(cls, unique_id: bytes, rank: int, world: int, device: int) -> "ShardGroup":
        lib = N.load()
        assert len(unique_id) == 128
        h = C.c_void_p()
        N.check(lib.orama_shard_group_create_rank(C.create_string_buffer(unique_id, 128), int(rank), int(world),
                                                  int(device), C.byref(h)))
        return cls(None, _handle=h)

    def ctx(self, local_shard: int) -> Context:
        return self._ctxs[local_shard]

    def close(self) -> None:
        if getattr(self, "_h", None):
            for c in self._ctxs:
                c.close()  # what lives on the group's contexts goes first
            self._lib.orama_shard_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def barrier(self) -> None:
        N.check(self._lib.orama_shard_group_barrier(self._h))

    def allreduce_max(self, value: float) -> float:
        v = C.c_double(value)
        N.check(self._lib.orama_shard_group_allreduce_max_f64(self._h, C.byref(v)))
        return v.value

    # ------------------------------------------------------------------ helpers
    def _handles(self, stores):
        assert len(stores) == self.n_local, "one store per local shard"
        return (C.c_void_p * self.n_local)(*[s.handle if hasattr(s, "handle") else s._h for s in stores])

    def _allow(self, allow):
        """allow: None or a list of ResidentAllowBitmap, one per local shard."""
        if allow is None:
            return None, 0
        assert len(allow) == self.n_local
        toks = (C.c_void_p * self.n_local)(*[a.ffi_args()[0] for a in allow])
        return toks, allow[0].ffi_args()[1]

    @staticmethod
    def _refs(refs):
        arr = (N.TermRef * max(len(refs), 1))()
        for i, (tok, lst, boost) in enumerate(refs):
            arr[i].token, arr[i].list, arr[i].boost = int(tok), int(lst), float(boost)
        return arr

    # ------------------------------------------------------------------ searches
    def vec_search(self, stores, targets, limit: int, allow=None):
        dim = stores[0].dim
        t = np.ascontiguousarray(np.asarray(targets, dtype=np.float32).reshape(-1, dim))
        q, k = t.shape[0], int(limit)
        ids = np.zeros((q, max(k, 1)), dtype=np.uint64)
        dist = np.zeros((q, max(k, 1)), dtype=np.float32)
        cnt = np.zeros(q, dtype=np.uint32)
        toks, bits = self._allow(allow)
        N.check(self._lib.orama_shard_vec_search(self._h, self._handles(stores), t.ctypes.data, q, k, toks, bits,
                                                 ids.ctypes.data, dist.ctypes.data, cnt.ctypes.data))
        return ids, dist, cnt

    def post_search(self, stores, refs, n_tokens: int, total_documents: float, top_k: int, threshold=None, allow=None,
                    apply_omc: bool = True, vector: dict | None = None, b: float = B_DEFAULT, k: float = K1_DEFAULT):
        params = _params(total_documents, n_tokens, threshold, top_k, k)
        out_ids = np.zeros(max(top_k, 1), dtype=np.uint64)
        out_sc = np.zeros(max(top_k, 1), dtype=np.float32)
        out_n, out_count = C.c_uint32(), C.c_uint64()
        toks, bits = self._allow(allow)
        hybrid = vector is not None
        vd = np.ascontiguousarray(list(vector) if hybrid else [], dtype=np.uint64)
        vs = np.ascontiguousarray(list(vector.values()) if hybrid else [], dtype=np.float32)
        N.check(self._lib.orama_shard_post_search(self._h, self._handles(stores), self._refs(refs), len(refs), b,
                                                  C.byref(params), toks, bits, 1 if apply_omc else 0, 1 if hybrid else 0,
                                                  vd.ctypes.data, vs.ctypes.data, len(vd), out_ids.ctypes.data,
                                                  out_sc.ctypes.data, C.byref(out_n), C.byref(out_count)))
        return out_ids[: out_n.value], out_sc[: out_n.value], out_count.value

    def hybrid_search(self, vec_stores, post_stores, query, limit: int, similarity: float, refs, n_tokens: int,
                      total_documents: float, top_k: int, threshold=None, allow=None, apply_omc: bool = True,
                      rescale_e5: bool = False, b: float = B_DEFAULT, k: float = K1_DEFAULT):
        params = _params(total_documents, n_tokens, threshold, top_k, k)
        qv = np.ascontiguousarray(query, dtype=np.float32)
        out_ids = np.zeros(max(top_k, 1), dtype=np.uint64)
        out_sc = np.zeros(max(top_k, 1), dtype=np.float32)
        out_n, out_count = C.c_uint32(), C.c_uint64()
        toks, bits = self._allow(allow)
        N.check(self._lib.orama_shard_hybrid_search(self._h, self._handles(vec_stores), self._handles(post_stores),
                                                    qv.ctypes.data, int(limit), float(similarity), 1 if rescale_e5 else 0,
                                                    self._refs(refs), len(refs), b, C.byref(params), toks, bits,
                                                    1 if apply_omc else 0, out_ids.ctypes.data, out_sc.ctypes.data,
                                                    C.byref(out_n), C.byref(out_count)))
        return out_ids[: out_n.value], out_sc[: out_n.value], out_count.value

    def session(self, stores, queries, q_per_step: int, k: int, n_slots: int = 2, force_exchange: bool = False):
        return ShardSession(self, stores, queries, q_per_step, k, n_slots, force_exchange)

    def post_search_batch(self, stores, queries, total_documents: float, top_k: int, allow=None, apply_omc: bool = True,
                          b: float = B_DEFAULT, k: float = K1_DEFAULT, statuses: bool = False):
        """orama_shard_post_search_batch: `queries` as PostingsStore.search_batch (refs, n_tokens, threshold | None[, top_k]);
        returns the list of (ids, scores, count) — and the status array with `statuses`."""
        from .fulltext import PreparedBatch

        prep = PreparedBatch(stores[0], queries, total_documents, top_k, None, apply_omc, 0, b, k)
        toks, bits = self._allow(allow)
        st = np.zeros(max(prep.nq, 1), dtype=np.int32)
        rc = self._lib.orama_shard_post_search_batch(self._h, self._handles(stores), prep._descs, prep.nq, b, toks, bits,
                                                     1 if apply_omc else 0, prep._stride, prep.out_ids.ctypes.data,
                                                     prep.out_sc.ctypes.data, prep.out_n.ctypes.data, prep.out_count.ctypes.data,
                                                     st.ctypes.data)
        if not statuses:
            N.check(rc)
            return prep.results()
        return prep.results(), st[: prep.nq]

    def post_batcher(self, stores, max_batch: int = 256, max_wait_us: int = 0):
        from .fulltext import PostSearchBatcher

        return PostSearchBatcher(stores[0], max_batch, max_wait_us, group=self, shards=stores)

    def lanes(self) -> dict:
        """Sharded calls the group runs side by side (max) and the lanes made so far (orama_shard_group_lanes)."""
        mx, made = C.c_uint32(), C.c_uint32()
        N.check(self._lib.orama_shard_group_lanes(self._h, C.byref(mx), C.byref(made)))
        return {"max": mx.value, "created": made.value}

    def batcher(self, stores, max_batch: int = 64, max_wait_us: int = 0) -> "GroupSearchBatcher":
        return GroupSearchBatcher(self, stores, max_batch, max_wait_us)


class GroupSearchBatcher:
    """Request micro-batcher in front of a shard group (orama_batcher_create_group): concurrent single-query callers
    share sharded passes.  `search` blocks like vec_search(q = 1)."""

    def __init__(self, group: ShardGroup, stores, max_batch: int, max_wait_us: int):
        self._lib = group._lib
        self.group, self.stores = group, list(stores)
        self.dim = stores[0].dim
        h = C.c_void_p()
        N.check(self._lib.orama_batcher_create_group(group._h, group._handles(stores), int(max_batch), int(max_wait_us), C.byref(h)))
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orama_batcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def search(self, target, limit: int):
        t = np.ascontiguousarray(np.asarray(target, dtype=np.float32).reshape(self.dim))
        k = int(limit)
        ids = np.zeros(max(k, 1), dtype=np.uint64)
        dist = np.zeros(max(k, 1), dtype=np.float32)
        n = C.c_uint32()
        N.check(self._lib.orama_batcher_search(self._h, t.ctypes.data, k, ids.ctypes.data, dist.ctypes.data, C.byref(n)))
        return ids[: n.value], dist[: n.value]

    def stats(self) -> dict:
        r, b, l = C.c_uint64(), C.c_uint64(), C.c_uint32()
        N.check(self._lib.orama_batcher_stats(self._h, C.byref(r), C.byref(b), C.byref(l)))
        return {"requests": r.value, "batches": b.value, "largest_batch": l.value,
                "mean_batch": (r.value / b.value) if b.value else 0.0}


class ShardSession:
    """Pipelined vector-search steps over resident queries (orama_shard_session_*): bench.py's timed loop."""

    def __init__(self, group: ShardGroup, stores, queries, q_per_step: int, k: int, n_slots: int, force_exchange: bool):
        self._lib = group._lib
        self.group = group
        self.q, self.k, self.n_slots = int(q_per_step), int(k), int(n_slots)
        qs = np.ascontiguousarray(queries, dtype=np.float32)
        h = C.c_void_p()
        N.check(self._lib.orama_shard_session_create(group._h, group._handles(stores), qs.ctypes.data, qs.shape[0],
                                                     self.q, self.k, self.n_slots, 1 if force_exchange else 0,
                                                     C.byref(h)))
        self._h = h

    def step(self, i: int) -> None:
        N.check(self._lib.orama_shard_session_step(self._h, int(i)))

    def sync(self) -> None:
        N.check(self._lib.orama_shard_session_sync(self._h))

    def result(self, slot: int):
        ids = np.zeros((self.q, self.k), dtype=np.uint64)
        dist = np.zeros((self.q, self.k), dtype=np.float32)
        cnt = np.zeros(self.q, dtype=np.uint32)
        N.check(self._lib.orama_shard_session_result(self._h, int(slot), ids.ctypes.data, dist.ctypes.data,
                                                     cnt.ctypes.data))
        return ids, dist, cnt

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orama_shard_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
