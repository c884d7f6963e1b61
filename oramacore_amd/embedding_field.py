"""Host-side mirror of the reference's vector seam.

`EmbeddingFieldStorage` keeps the method names, argument meaning and error behaviour of
src/collection_manager/sides/read/index/embedding_field.rs:63-320; the third-party
`oramacore_fields::embedding::EmbeddingStorage` it wraps there is replaced by the HBM-resident
store behind `orama_vec_*` (include/orama_hip.h).  The epilogue (distance → similarity → rescale →
cut-off → per-doc sum, :268-276) is the reference's in-tree code and stays on the host, exactly as
the Rust shim in INTEGRATION.md keeps it.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _native as N
from .context import Context, Owner


class Model(enum.Enum):
    """src/python/embeddings.rs:13-63 — model → dimensions table."""

    BGESmall = "BGESmall"
    BGEBase = "BGEBase"
    BGELarge = "BGELarge"
    JinaEmbeddingsV2BaseCode = "JinaEmbeddingsV2BaseCode"
    MultilingualE5Small = "MultilingualE5Small"
    MultilingualE5Base = "MultilingualE5Base"
    MultilingualE5Large = "MultilingualE5Large"
    MultilingualMiniLML12V2 = "MultilingualMiniLML12V2"

    def dimensions(self) -> int:
        return {
            Model.BGESmall: 384,
            Model.BGEBase: 768,
            Model.BGELarge: 1024,
            Model.JinaEmbeddingsV2BaseCode: 768,
            Model.MultilingualE5Small: 384,
            Model.MultilingualE5Base: 768,
            Model.MultilingualE5Large: 1024,
            Model.MultilingualMiniLML12V2: 384,
        }[self]

    def is_e5(self) -> bool:
        return self in (Model.MultilingualE5Small, Model.MultilingualE5Base, Model.MultilingualE5Large)

    def rescale_score(self, score) -> np.float32:
        """src/python/embeddings.rs:71-92, evaluated in f32."""
        score = np.float32(score)
        if not self.is_e5():
            return score
        mn, mx = np.float32(0.7), np.float32(1.0)
        delta = np.float32(mx - mn)
        c = score
        if c < mn:
            c = mn
        if c > mx:
            c = mx
        return np.float32(np.float32(c - mn) / delta)


class AllowBitmap:
    """Materialised `FilterResult<DocumentId>` (index/filter.rs:344-392): bit d set <=> doc d passes.

    The reference hands the scan a predicate object (`DocumentFilter::contains`,
    embedding_field.rs:54-61); on the GPU the predicate is a bitmap over document ids.
    """

    def __init__(self, n_bits: int, allowed_ids=None):
        self.n_bits = int(n_bits)
        self.words = np.zeros((self.n_bits + 63) // 64, dtype=np.uint64)
        if allowed_ids is not None:
            ids = np.asarray(allowed_ids, dtype=np.uint64)
            ids = ids[ids < np.uint64(self.n_bits)]
            np.bitwise_or.at(self.words, (ids >> np.uint64(6)).astype(np.int64),
                             np.uint64(1) << (ids & np.uint64(63)))

    @classmethod
    def from_mask(cls, mask) -> "AllowBitmap":
        mask = np.asarray(mask, dtype=bool)
        return cls(mask.size, np.nonzero(mask)[0])

    def contains(self, doc_id: int) -> bool:
        if doc_id >= self.n_bits:
            return False
        return bool((int(self.words[doc_id >> 6]) >> (doc_id & 63)) & 1)

    def ffi_args(self):
        """(pointer, bits) for the `allow_bitmap` / `bitmap_bits` arguments: host words, uploaded per call."""
        return self.words.ctypes.data, self.n_bits

    def to_device(self, ctx) -> "ResidentAllowBitmap":
        return ResidentAllowBitmap(ctx, self)


class ResidentAllowBitmap:
    """The same filter kept in HBM (orama_allow_*): searches take its token in place of host
    words — no per-query upload.  `set(doc_ids, allowed)` flips single documents (delete / re-admit)."""

    def __init__(self, ctx, bitmap: AllowBitmap):
        self._lib = N.load()
        self.n_bits = bitmap.n_bits
        self._host = bitmap  # kept in sync so that `contains` still answers on the host
        h = C.c_void_p()
        N.check(self._lib.orama_allow_create(ctx.handle, bitmap.words.ctypes.data, bitmap.n_bits, C.byref(h)))
        self._h = h
        ctx._adopt(self)

    def ffi_args(self):
        return self._lib.orama_allow_token(self._h), self.n_bits

    def contains(self, doc_id: int) -> bool:
        return self._host.contains(doc_id)

    def set(self, doc_ids, allowed: bool) -> None:
        ids = np.ascontiguousarray(doc_ids, dtype=np.uint64)
        N.check(self._lib.orama_allow_set(self._h, ids.ctypes.data, ids.shape[0], 1 if allowed else 0))
        for d in ids.tolist():
            w, b = d >> 6, np.uint64(1) << np.uint64(d & 63)
            self._host.words[w] = (self._host.words[w] | b) if allowed else (self._host.words[w] & ~b)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orama_allow_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


@dataclass
class VectorSearchParams:
    """committed_field/vector.rs:10-15."""

    target: np.ndarray
    similarity: float = 0.7
    limit: int = 10
    filtered_doc_ids: Optional[AllowBitmap] = None


class EmbeddingFieldStorage(Owner):
    def __init__(self, ctx: Context, model: Model | None = None, *, dimensions: int | None = None,
                 metric: int = N.METRIC_COSINE, reserve_rows: int = 0, dtype: int = N.DTYPE_F32):
        self._lib = N.load()
        self.ctx = ctx
        self._model = model
        self.dim = int(dimensions if dimensions is not None else model.dimensions())
        h = C.c_void_p()
        N.check(self._lib.orama_vec_create(ctx.handle, self.dim, metric, int(dtype), int(reserve_rows),
                                           C.byref(h)))
        self.dtype = int(dtype)
        self._h = h
        ctx._adopt(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._close_children()  # batchers first
            self._lib.orama_vec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def model(self) -> Model | None:
        return self._model

    # --- embedding_field.rs:232-237
    def insert(self, doc_id: int, vectors) -> int:
        vecs = np.ascontiguousarray(np.asarray(vectors, dtype=np.float32).reshape(-1, self.dim))
        ids = np.full(vecs.shape[0], doc_id, dtype=np.uint64)
        return self.insert_rows(ids, vecs)

    def insert_rows(self, doc_ids, rows) -> int:
        """Bulk form of insert: row i belongs to doc_ids[i]. Returns the number of rows accepted."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        doc_ids = np.ascontiguousarray(doc_ids, dtype=np.uint64)
        if rows.ndim != 2 or rows.shape[1] != self.dim or rows.shape[0] != doc_ids.shape[0]:
            raise ValueError("rows must be [n, dim] with one doc id per row")
        acc = C.c_uint64()
        N.check(self._lib.orama_vec_insert(self._h, doc_ids.ctypes.data, rows.ctypes.data, rows.shape[0],
                                           C.byref(acc)))
        return acc.value

    # --- embedding_field.rs:240-242
    def delete(self, doc_id: int) -> None:
        ids = np.asarray([doc_id], dtype=np.uint64)
        N.check(self._lib.orama_vec_delete(self._h, ids.ctypes.data, 1))

    # --- embedding_field.rs:281-310
    def info(self) -> dict:
        st = N.VecInfo()
        N.check(self._lib.orama_vec_info(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in N.VecInfo._fields_}

    def has_pending_ops(self) -> bool:
        return self.info()["pending_ops"] > 0

    def compact(self, version: int) -> None:
        N.check(self._lib.orama_vec_compact(self._h, int(version)))

    def current_version_number(self) -> int:
        return self.info()["version"]

    def stats(self) -> dict:
        i = self.info()
        return {"dimensions": i["dimensions"], "vector_count": i["num_embeddings"], "model": self._model}

    # --- the storage-level call (oramacore_fields::embedding::EmbeddingStorage::search[_with_filter])
    def storage_search(self, targets, limit: int, allow: AllowBitmap | None = None):
        """Row-level k-NN: returns (doc_ids [q,k], distances [q,k], counts [q])."""
        t = np.ascontiguousarray(np.asarray(targets, dtype=np.float32).reshape(-1, self.dim))
        q = t.shape[0]
        k = int(limit)
        ids = np.zeros((q, max(k, 1)), dtype=np.uint64)
        dist = np.zeros((q, max(k, 1)), dtype=np.float32)
        cnt = np.zeros(q, dtype=np.uint32)
        bm_ptr, bm_bits = None, 0
        if allow is not None:
            bm_ptr, bm_bits = allow.ffi_args()
        N.check(self._lib.orama_vec_search(self._h, t.ctypes.data, q, k, bm_ptr, bm_bits,
                                           ids.ctypes.data, dist.ctypes.data, cnt.ctypes.data))
        return ids, dist, cnt

    # --- embedding_field.rs:250-278
    def search(self, params: VectorSearchParams, output: dict) -> None:
        ids, dist, cnt = self.storage_search(params.target, params.limit, params.filtered_doc_ids)
        sim_min = np.float32(params.similarity)
        for j in range(int(cnt[0])):
            similarity = np.float32(np.float32(1.0) - dist[0, j])
            score = self._model.rescale_score(similarity) if self._model is not None else similarity
            if score >= sim_min:
                d = int(ids[0, j])
                output[d] = np.float32(output.get(d, np.float32(0.0)) + score)

    # --- test / bench utilities
    def fill_synthetic(self, n_rows: int, seed: int, first_doc_id: int = 0) -> None:
        N.check(self._lib.orama_vec_fill_synthetic(self._h, int(n_rows), int(seed), int(first_doc_id)))

    def get_rows(self, row_idx):
        idx = np.ascontiguousarray(row_idx, dtype=np.uint64)
        out = np.empty((idx.shape[0], self.dim), dtype=np.float32)
        docs = np.empty(idx.shape[0], dtype=np.uint64)
        N.check(self._lib.orama_vec_get_rows(self._h, idx.ctypes.data, idx.shape[0], out.ctypes.data,
                                             docs.ctypes.data))
        return out, docs

    @property
    def handle(self):
        return self._h


class SearchBatcher:
    """Micro-batcher in front of one EmbeddingFieldStorage (orama_batcher_*): concurrent
    single-query callers (threads) share corpus passes.  `search` blocks like storage_search(q=1, no filter)."""

    def __init__(self, storage: EmbeddingFieldStorage, max_batch: int = 64, max_wait_us: int = 0):
        self._lib = N.load()
        self.storage = storage
        h = C.c_void_p()
        N.check(self._lib.orama_batcher_create(storage.handle, int(max_batch), int(max_wait_us), C.byref(h)))
        self._h = h
        storage._adopt(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orama_batcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def search(self, target, limit: int, allow=None):
        """Row-level k-NN of ONE target: (doc_ids [n], distances [n]).  `allow`: the request's filter — pass the SAME
        ResidentAllowBitmap object for the same filter so that those requests share corpus passes."""
        t = np.ascontiguousarray(np.asarray(target, dtype=np.float32).reshape(self.storage.dim))
        k = int(limit)
        ids = np.zeros(max(k, 1), dtype=np.uint64)
        dist = np.zeros(max(k, 1), dtype=np.float32)
        n = C.c_uint32()
        bm_ptr, bm_bits = allow.ffi_args() if allow is not None else (None, 0)
        N.check(self._lib.orama_batcher_search_filtered(self._h, t.ctypes.data, k, bm_ptr, bm_bits, ids.ctypes.data,
                                                        dist.ctypes.data, C.byref(n)))
        return ids[: n.value], dist[: n.value]

    def stats(self) -> dict:
        r, b, l = C.c_uint64(), C.c_uint64(), C.c_uint32()
        N.check(self._lib.orama_batcher_stats(self._h, C.byref(r), C.byref(b), C.byref(l)))
        return {"requests": r.value, "batches": b.value, "largest_batch": l.value,
                "mean_batch": (r.value / b.value) if b.value else 0.0}
