"""ctypes wrapper of oracle/liborama_oracle.so — the CPU restatement of the reference algorithm.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package (oramacore_amd) never imports it.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "liborama_oracle.so"


class _Entry(C.Structure):
    _fields_ = [("token", C.c_uint32), ("doc", C.c_void_p), ("ntf", C.c_void_p), ("len", C.c_uint64)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    src_m = max((HERE / "orama_oracle.c").stat().st_mtime, (HERE / "orama_oracle.h").stat().st_mtime)
    if not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < src_m:
        subprocess.run(["make", "-C", str(HERE), "-B"], check=True, capture_output=True)
    L = C.CDLL(str(LIB_PATH))
    vp = C.c_void_p
    L.orc_cosine_distance_f32.restype = C.c_float
    L.orc_cosine_distance_f32.argtypes = [vp, vp, C.c_uint32]
    L.orc_cosine_distance_f64.restype = C.c_double
    L.orc_cosine_distance_f64.argtypes = [vp, vp, C.c_uint32]
    L.orc_l2sq_distance_f32.restype = C.c_float
    L.orc_l2sq_distance_f32.argtypes = [vp, vp, C.c_uint32]
    L.orc_distances_f32.restype = None
    L.orc_distances_f32.argtypes = [vp, C.c_uint64, C.c_uint32, vp, C.c_int, vp]
    L.orc_distances_f32_mt.restype = None
    L.orc_distances_f32_mt.argtypes = [vp, C.c_uint64, C.c_uint32, vp, C.c_int, vp, C.c_int]
    L.orc_row_is_valid.restype = C.c_int
    L.orc_row_is_valid.argtypes = [vp, C.c_uint32]
    L.orc_vector_search.restype = C.c_uint32
    L.orc_vector_search.argtypes = [vp, C.c_uint64, C.c_uint32, vp, vp, vp, C.c_int, C.c_uint32, vp, C.c_uint64,
                                    vp, vp, vp]
    L.orc_rescale_score.restype = C.c_float
    L.orc_rescale_score.argtypes = [C.c_float, C.c_int]
    L.orc_embedding_epilogue.restype = None
    L.orc_embedding_epilogue.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_float, vp, vp, vp]
    L.orc_facet_count_buckets.restype = None
    L.orc_facet_count_buckets.argtypes = [vp, C.c_uint64, vp, vp, C.c_uint32, vp]
    L.orc_facet_count_ranges.restype = None
    L.orc_facet_count_ranges.argtypes = [vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, C.c_uint32, vp]
    L.orc_group_top.restype = None
    L.orc_group_top.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.orc_bm25_idf.restype = C.c_float
    L.orc_bm25_idf.argtypes = [C.c_float, C.c_uint64]
    L.orc_bm25f_normalized_tf.restype = C.c_float
    L.orc_bm25f_normalized_tf.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_float]
    L.orc_bm25f_score.restype = C.c_float
    L.orc_bm25f_score.argtypes = [C.c_float, C.c_float, C.c_float]
    L.orc_bm25_legacy_add.restype = C.c_float
    L.orc_bm25_legacy_add.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_uint64, C.c_float,
                                      C.c_float, C.c_float, C.c_float]
    L.orc_search_full_text.restype = C.c_uint64
    L.orc_search_full_text.argtypes = [C.POINTER(_Entry), C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_int,
                                       C.c_uint32, vp, vp]
    L.orc_normalize_and_combine.restype = C.c_uint64
    L.orc_normalize_and_combine.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp]
    L.orc_apply_omc.restype = None
    L.orc_apply_omc.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64]
    L.orc_top_n.restype = C.c_uint64
    L.orc_top_n.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, vp]
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def distances(corpus, q, metric: int = 0, threads: int = 1) -> np.ndarray:
    corpus = _f32(corpus)
    q = _f32(q)
    n, d = corpus.shape
    out = np.empty(n, dtype=np.float32)
    if threads > 1:
        lib().orc_distances_f32_mt(corpus.ctypes.data, n, d, q.ctypes.data, metric, out.ctypes.data, threads)
    else:
        lib().orc_distances_f32(corpus.ctypes.data, n, d, q.ctypes.data, metric, out.ctypes.data)
    return out


def cosine_distance_f64(q, x) -> float:
    q = _f32(q)
    x = _f32(x)
    return float(lib().orc_cosine_distance_f64(q.ctypes.data, x.ctypes.data, q.shape[0]))


def row_is_valid(x) -> bool:
    x = _f32(x)
    return bool(lib().orc_row_is_valid(x.ctypes.data, x.shape[0]))


def vector_search(corpus, row_doc, q, k: int, metric: int = 0, dead=None, allow_words=None, allow_bits: int = 0):
    """a1: returns (doc ids, distances, rows) of the k nearest rows."""
    corpus = _f32(corpus)
    n, d = corpus.shape
    q = _f32(q)
    row_doc = _u64(row_doc)
    out_doc = np.zeros(max(k, 1), dtype=np.uint64)
    out_dist = np.zeros(max(k, 1), dtype=np.float32)
    out_row = np.zeros(max(k, 1), dtype=np.uint64)
    dead_p = None
    if dead is not None:
        dead = np.ascontiguousarray(dead, dtype=np.uint8)
        dead_p = dead.ctypes.data
    allow_p = None
    if allow_words is not None:
        allow_words = _u64(allow_words)
        allow_p = allow_words.ctypes.data
    m = lib().orc_vector_search(corpus.ctypes.data, n, d, row_doc.ctypes.data, dead_p, q.ctypes.data, metric, k,
                                allow_p, allow_bits, out_doc.ctypes.data, out_dist.ctypes.data, out_row.ctypes.data)
    return out_doc[:m], out_dist[:m], out_row[:m]


def rescale_score(score: float, is_e5: bool) -> np.float32:
    return np.float32(lib().orc_rescale_score(C.c_float(score), 1 if is_e5 else 0))


def embedding_epilogue(hit_doc, hit_dist, is_e5: bool, min_similarity: float, init: dict | None = None) -> dict:
    """a2: returns {doc: score} after similarity/rescale/cut-off/per-doc sum."""
    hit_doc = _u64(hit_doc)
    hit_dist = _f32(hit_dist)
    init = init or {}
    cap = len(init) + hit_doc.shape[0] + 1
    io_doc = np.zeros(cap, dtype=np.uint64)
    io_score = np.zeros(cap, dtype=np.float32)
    for i, (d, s) in enumerate(sorted(init.items())):
        io_doc[i], io_score[i] = d, s
    n = C.c_uint64(len(init))
    lib().orc_embedding_epilogue(hit_doc.ctypes.data, hit_dist.ctypes.data, hit_doc.shape[0], 1 if is_e5 else 0,
                                 C.c_float(min_similarity), io_doc.ctypes.data, io_score.ctypes.data, C.byref(n))
    return {int(io_doc[i]): np.float32(io_score[i]) for i in range(n.value)}


def bm25_idf(total_documents: float, df: int) -> np.float32:
    return np.float32(lib().orc_bm25_idf(C.c_float(total_documents), df))


def bm25f_normalized_tf(tf: int, field_len: int, avg_len: float, b: float) -> np.float32:
    return np.float32(lib().orc_bm25f_normalized_tf(tf, field_len, C.c_float(avg_len), C.c_float(b)))


def bm25f_score(s: float, k: float, idf: float) -> np.float32:
    return np.float32(lib().orc_bm25f_score(C.c_float(s), C.c_float(k), C.c_float(idf)))


def bm25_legacy_add(tf, field_len, avg_len, total_docs, docs_with_term, k, weight, b, boost) -> np.float32:
    return np.float32(lib().orc_bm25_legacy_add(tf, field_len, C.c_float(avg_len), C.c_float(total_docs),
                                                docs_with_term, C.c_float(k), C.c_float(weight), C.c_float(b),
                                                C.c_float(boost)))


def search_full_text(entries, n_tokens: int, total_documents: float, k: float = 1.2, threshold: int | None = None):
    """a6–a8. entries: list of (token, doc_ids, ntf). Returns (docs asc, scores)."""
    keep = []
    arr = (_Entry * max(len(entries), 1))()
    total = 0
    for i, (tok, docs, ntf) in enumerate(entries):
        docs = _u64(docs)
        ntf = _f32(ntf)
        keep.append((docs, ntf))
        arr[i].token = tok
        arr[i].doc = docs.ctypes.data
        arr[i].ntf = ntf.ctypes.data
        arr[i].len = docs.shape[0]
        total += docs.shape[0]
    out_doc = np.zeros(max(total, 1), dtype=np.uint64)
    out_score = np.zeros(max(total, 1), dtype=np.float32)
    m = lib().orc_search_full_text(arr, len(entries), n_tokens, C.c_float(total_documents), C.c_float(k),
                                   0 if threshold is None else 1, 0 if threshold is None else int(threshold),
                                   out_doc.ctypes.data, out_score.ctypes.data)
    return out_doc[:m].copy(), out_score[:m].copy()


def normalize_and_combine(v_doc, v_score, f_doc, f_score):
    """a9. Inputs are maps as (doc, score) arrays in any order; returns (docs asc, scores)."""
    v_doc, v_score, f_doc, f_score = _u64(v_doc), _f32(v_score), _u64(f_doc), _f32(f_score)
    vo, fo = np.argsort(v_doc, kind="stable"), np.argsort(f_doc, kind="stable")
    v_doc, v_score, f_doc, f_score = v_doc[vo], v_score[vo], f_doc[fo], f_score[fo]
    v_doc, v_score, f_doc, f_score = map(np.ascontiguousarray, (v_doc, v_score, f_doc, f_score))
    cap = v_doc.shape[0] + f_doc.shape[0] + 1
    out_doc = np.zeros(cap, dtype=np.uint64)
    out_score = np.zeros(cap, dtype=np.float32)
    m = lib().orc_normalize_and_combine(v_doc.ctypes.data, v_score.ctypes.data, v_doc.shape[0], f_doc.ctypes.data,
                                        f_score.ctypes.data, f_doc.shape[0], out_doc.ctypes.data,
                                        out_score.ctypes.data)
    return out_doc[:m].copy(), out_score[:m].copy()


def apply_omc(doc, score, omc_doc, omc_mul):
    """a10. Returns the multiplied score array (docs unchanged)."""
    doc = _u64(doc)
    score = _f32(score).copy()
    omc_doc, omc_mul = _u64(omc_doc), _f32(omc_mul)
    o = np.argsort(omc_doc, kind="stable")
    omc_doc, omc_mul = np.ascontiguousarray(omc_doc[o]), np.ascontiguousarray(omc_mul[o])
    lib().orc_apply_omc(doc.ctypes.data, score.ctypes.data, doc.shape[0], omc_doc.ctypes.data, omc_mul.ctypes.data,
                        omc_doc.shape[0])
    return score


def top_n(doc, score, n: int):
    """a11. Returns (docs, scores) of the n best by (score desc, doc asc), NaN dropped."""
    doc, score = _u64(doc), _f32(score)
    out_doc = np.zeros(max(n, 1), dtype=np.uint64)
    out_score = np.zeros(max(n, 1), dtype=np.float32)
    m = lib().orc_top_n(doc.ctypes.data, score.ctypes.data, doc.shape[0], n, out_doc.ctypes.data,
                        out_score.ctypes.data)
    return out_doc[:m].copy(), out_score[:m].copy()


def facet_count_buckets(map_doc, bucket_off, bucket_doc) -> np.ndarray:
    map_doc, bucket_off, bucket_doc = _u64(map_doc), _u64(bucket_off), _u64(bucket_doc)
    out = np.zeros(len(bucket_off) - 1, dtype=np.uint64)
    lib().orc_facet_count_buckets(map_doc.ctypes.data, C.c_uint64(len(map_doc)), bucket_off.ctypes.data,
                                  bucket_doc.ctypes.data, C.c_uint32(len(out)), out.ctypes.data)
    return out


def facet_count_ranges(map_doc, doc, value, ranges) -> np.ndarray:
    map_doc, doc = _u64(map_doc), _u64(doc)
    value = np.ascontiguousarray(value, dtype=np.float64)
    fr = np.ascontiguousarray([r[0] for r in ranges], dtype=np.float64)
    to = np.ascontiguousarray([r[1] for r in ranges], dtype=np.float64)
    out = np.zeros(len(ranges), dtype=np.uint64)
    lib().orc_facet_count_ranges(map_doc.ctypes.data, C.c_uint64(len(map_doc)), doc.ctypes.data, value.ctypes.data,
                                 C.c_uint64(len(doc)), fr.ctypes.data, to.ctypes.data, C.c_uint32(len(ranges)),
                                 out.ctypes.data)
    return out


def group_top(map_doc, map_score, group_off, group_doc, max_results: int):
    map_doc, group_off, group_doc = _u64(map_doc), _u64(group_off), _u64(group_doc)
    map_score = _f32(map_score)
    g = len(group_off) - 1
    od = np.zeros((g, max_results), dtype=np.uint64)
    os_ = np.zeros((g, max_results), dtype=np.float32)
    on = np.zeros(g, dtype=np.uint32)
    lib().orc_group_top(map_doc.ctypes.data, map_score.ctypes.data, C.c_uint64(len(map_doc)), group_off.ctypes.data,
                        group_doc.ctypes.data, C.c_uint32(g), C.c_uint32(max_results), od.ctypes.data, os_.ctypes.data,
                        on.ctypes.data)
    return od, os_, on
