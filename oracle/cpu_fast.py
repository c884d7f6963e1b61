"""ctypes wrapper of oracle/orama_cpu_fast.c — the "fast CPU" baseline legs of bench.py.

MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline legs and tests/): the product never imports it.  The library is
compiled ON THE HOST THAT RUNS THE BENCHMARK (gcc -O3 -march=native, a second or two) into a temporary directory, so it
uses that host's vector width and never ships a binary built for another CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_lib = None
_flags = ""


class _Entry(C.Structure):
    _fields_ = [("token", C.c_uint32), ("doc", C.c_void_p), ("ntf", C.c_void_p), ("len", C.c_uint64)]


def lib() -> C.CDLL:
    global _lib, _flags
    if _lib is not None:
        return _lib
    out = Path(tempfile.mkdtemp(prefix="orama_cpu_fast_")) / "liborama_cpu_fast.so"
    err = ""
    for arch in ("-march=native", "-mavx2 -mfma", ""):
        flags = f"-O3 {arch} -std=gnu99 -fPIC -ffp-contract=off -fno-math-errno -pthread".split()
        r = subprocess.run(["gcc", *flags, "-shared", "-o", str(out), str(HERE / "orama_cpu_fast.c"), "-lm", "-lpthread"],
                           capture_output=True, text=True)
        if r.returncode == 0:
            _flags = " ".join(flags)
            break
        err = r.stderr
    else:
        raise RuntimeError(f"gcc could not build orama_cpu_fast.c: {err[-500:]}")
    L = C.CDLL(str(out))
    vp = C.c_void_p
    L.cpf_distances_f32.restype = None
    L.cpf_distances_f32.argtypes = [vp, C.c_uint64, C.c_uint32, vp, vp, C.c_int]
    L.cpf_place_rows.restype = vp
    L.cpf_place_rows.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_int]
    L.cpf_free.restype = None
    L.cpf_free.argtypes = [vp]
    L.cpf_bm25_hashmap.restype = C.c_uint64
    L.cpf_bm25_hashmap.argtypes = [C.POINTER(_Entry), C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_uint32, C.c_uint64,
                                   vp, vp, C.POINTER(C.c_uint64)]
    _lib = L
    return L


def build_flags() -> str:
    lib()
    return _flags


def distances(corpus, q, threads: int = 1) -> np.ndarray:
    corpus = np.ascontiguousarray(corpus, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    n, d = corpus.shape
    out = np.empty(n, dtype=np.float32)
    lib().cpf_distances_f32(corpus.ctypes.data, n, d, q.ctypes.data, out.ctypes.data, int(threads))
    return out


class PlacedRows:
    """A copy of `corpus` first-touched by the `threads` threads that will scan it (cpf_place_rows): `distances(q)` scans it with
    the same split.  Context manager; frees the copy on exit."""

    def __init__(self, corpus, threads: int):
        corpus = np.ascontiguousarray(corpus, dtype=np.float32)
        self.n, self.d, self.threads = corpus.shape[0], corpus.shape[1], int(threads)
        self.ptr = lib().cpf_place_rows(corpus.ctypes.data, self.n, self.d, self.threads)
        if not self.ptr:
            raise MemoryError("cpf_place_rows")

    def distances(self, q) -> np.ndarray:
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.empty(self.n, dtype=np.float32)
        lib().cpf_distances_f32(self.ptr, self.n, self.d, q.ctypes.data, out.ctypes.data, self.threads)
        return out

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.ptr:
            lib().cpf_free(self.ptr)
            self.ptr = None


def bm25_hashmap(entries, n_tokens: int, total_documents: float, k: float, threshold, top_k: int):
    """entries: list of (token, docs u64, ntf f32) as oracle.search_full_text takes them -> (ids, scores, count)."""
    keep = []
    arr = (_Entry * max(len(entries), 1))()
    for i, (tok, docs, ntf) in enumerate(entries):
        d = np.ascontiguousarray(docs, dtype=np.uint64)
        v = np.ascontiguousarray(ntf, dtype=np.float32)
        keep += [d, v]
        arr[i].token, arr[i].doc, arr[i].ntf, arr[i].len = int(tok), d.ctypes.data, v.ctypes.data, len(d)
    ids = np.zeros(max(top_k, 1), dtype=np.uint64)
    sc = np.zeros(max(top_k, 1), dtype=np.float32)
    count = C.c_uint64()
    m = lib().cpf_bm25_hashmap(arr, len(entries), int(n_tokens), C.c_float(total_documents), C.c_float(k),
                               0 if threshold is None else 1, 0 if threshold is None else int(threshold), int(top_k),
                               ids.ctypes.data, sc.ctypes.data, C.byref(count))
    return ids[:m], sc[:m], count.value
