"""ctypes binding of liborama_hip.so (the C ABI declared in include/orama_hip.h).

This is plumbing only: every function here forwards to one `extern "C"` entry point.  There is no
CPU fallback — if the shared object is missing and cannot be built, or no HIP device is present,
the calls raise `OramaError` / `RuntimeError` loudly.
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

from . import _build

ORAMA_OK = 0
ORAMA_ERR_INVALID = 1
ORAMA_ERR_HIP = 2
ORAMA_ERR_OOM = 3
ORAMA_ERR_UNSUPPORTED = 4
ORAMA_ERR_BUSY = 5

METRIC_COSINE = 0
METRIC_L2SQ = 1
DTYPE_F32 = 0
DTYPE_F16 = 1
DTYPE_F32_SHADOW16 = 2

HEADER = Path(__file__).resolve().parent.parent / "include" / "orama_hip.h"


class OramaError(RuntimeError):
    """A non-zero status from liborama_hip (mirrors the reference's anyhow::Error → ReadError::Generic)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"liborama_hip status {status}: {message}")
        self.status = status
        self.message = message


class VecInfo(C.Structure):
    _fields_ = [
        ("dimensions", C.c_uint32),
        ("num_embeddings", C.c_uint64),
        ("num_rows", C.c_uint64),
        ("pending_ops", C.c_uint64),
        ("version", C.c_uint64),
        ("hbm_bytes", C.c_uint64),
        ("two_stage_queries", C.c_uint64),
        ("two_stage_fallbacks", C.c_uint64),
    ]


class NtfEntry(C.Structure):
    _fields_ = [
        ("token", C.c_uint32),
        ("doc", C.POINTER(C.c_uint64)),
        ("ntf", C.POINTER(C.c_float)),
        ("len", C.c_uint64),
    ]


class Bm25Params(C.Structure):
    _fields_ = [
        ("total_documents", C.c_float),
        ("k", C.c_float),
        ("n_tokens", C.c_uint32),
        ("use_threshold", C.c_int),
        ("threshold", C.c_uint32),
        ("top_k", C.c_uint32),
    ]


class TermRef(C.Structure):
    _fields_ = [("token", C.c_uint32), ("list", C.c_uint32), ("boost", C.c_float)]


class PostQueryDesc(C.Structure):
    _fields_ = [("refs", C.POINTER(TermRef)), ("n_refs", C.c_uint32), ("params", Bm25Params)]


def declared_symbols() -> list[str]:
    """Every function name declared in include/orama_hip.h (used by the ABI export test)."""
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(orama_[a-z0-9_]+)\s*\(", text)
    seen, out = set(), []
    for n in names:
        if n not in seen:
            seen.add(n)
            out.append(n)
    return out


_lib = None


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load liborama_hip.so from the source tree (building it with hipcc when stale/missing)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.lib_path()  # liborama_hip.so; liborama_hip_cmp.so when ORAMA_COMPARISON_KERNELS=1 asks for the A/B flavour
    if build_if_missing and not _build.native_is_fresh():
        try:
            _build.build_native()
        except Exception as e:  # noqa: BLE001
            if not path.exists():
                raise RuntimeError(
                    f"liborama_hip.so is missing and could not be built ({e}); the HIP path has no fallback"
                ) from e
    if not path.exists():
        raise RuntimeError(f"{path} not found — run `python __graft_entry__.py` (build) first; no CPU fallback exists")
    lib = C.CDLL(str(path))
    _declare(lib)
    if lib.orama_abi_version() != 2:
        raise RuntimeError("liborama_hip ABI version mismatch")
    _lib = lib
    return lib


def _declare(lib: C.CDLL) -> None:
    vp, u64p, u32p, f32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    lib.orama_abi_version.restype = C.c_int
    lib.orama_last_error.restype = C.c_char_p
    sig = {
        "orama_ctx_create": [C.c_int, C.POINTER(vp)],
        "orama_ctx_synchronize": [vp],
        "orama_ctx_device_info": [vp, C.c_char_p, C.POINTER(C.c_int), u64p],
        "orama_ctx_set_scan_tuning": [vp, C.c_int, C.c_int, C.c_int],
        "orama_ctx_set_f16_tuning": [vp, C.c_int, C.c_int],
        "orama_ctx_set_f16_wide": [vp, C.c_int],
        "orama_ctx_set_bm25_ranges": [vp, C.c_int],
        "orama_ctx_set_two_stage": [vp, C.c_int],
        "orama_ctx_set_f32_batch": [vp, C.c_int],
        "orama_ctx_set_option": [vp, C.c_char_p, C.c_longlong],
        "orama_prof_enable": [vp, C.c_int],
        "orama_prof_reset": [vp],
        "orama_prof_get": [vp, C.c_char_p, C.POINTER(C.c_double), u64p],
        "orama_prof_samples": [vp, C.c_char_p, C.POINTER(C.c_float), C.c_uint64, u64p],
        "orama_ctx_pci_bus_id": [vp, C.c_char_p, C.c_int],
        "orama_dev_malloc": [vp, C.c_uint64, C.POINTER(vp)],
        "orama_dev_upload": [vp, vp, C.c_uint64, vp, C.c_uint64],
        "orama_dev_download": [vp, vp, C.c_uint64, vp, C.c_uint64],
        "orama_stream_create": [vp, C.c_int, C.POINTER(vp)],
        "orama_stream_synchronize": [vp, vp],
        "orama_vec_create": [vp, C.c_uint32, C.c_int, C.c_int, C.c_uint64, C.POINTER(vp)],
        "orama_vec_insert": [vp, vp, vp, C.c_uint64, u64p],
        "orama_vec_delete": [vp, vp, C.c_uint64],
        "orama_vec_compact": [vp, C.c_uint64],
        "orama_vec_info": [vp, C.POINTER(VecInfo)],
        "orama_vec_search": [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, vp, vp],
        "orama_vec_search_device": [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, vp, vp, vp],
        "orama_merge_candidates_device": [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp],
        "orama_vec_search_packed_device": [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, vp, vp],
        "orama_vec_search_packed_device2": [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, vp, vp, vp],
        "orama_merge_packed_device": [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp],
        "orama_vec_fill_synthetic": [vp, C.c_uint64, C.c_uint64, C.c_uint64],
        "orama_vec_get_rows": [vp, vp, C.c_uint64, vp, vp],
        "orama_allow_create": [vp, vp, C.c_uint64, C.POINTER(vp)],
        "orama_allow_set": [vp, vp, C.c_uint64, C.c_int],
        "orama_dict_create": [vp, vp, vp, C.c_uint32, C.POINTER(vp)],
        "orama_dict_expand": [vp, vp, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, vp, u32p],
        "orama_batcher_create": [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)],
        "orama_batcher_create_group": [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)],
        "orama_shard_group_lanes": [vp, u32p, u32p],
        "orama_batcher_search": [vp, vp, C.c_uint32, vp, vp, u32p],
        "orama_batcher_search_filtered": [vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, u32p],
        "orama_batcher_stats": [vp, u64p, u64p, u32p],
        "orama_bm25_score": [vp, C.POINTER(NtfEntry), C.c_uint32, C.POINTER(Bm25Params), vp, vp, C.c_uint64,
                             vp, vp, u32p, u64p],
        "orama_bm25_score_map": [vp, C.POINTER(NtfEntry), C.c_uint32, C.POINTER(Bm25Params), vp, vp, C.c_uint64, C.c_uint64,
                                 vp, vp, u64p],
        "orama_post_create": [vp, C.POINTER(vp)],
        "orama_post_build": [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_uint32, vp, vp, vp, vp, vp],
        "orama_post_set_omc": [vp, vp, vp, C.c_uint64],
        "orama_post_append": [vp, vp, C.c_uint64, vp, C.c_uint32, vp, vp, vp, vp, vp],
        "orama_post_get_list": [vp, C.c_uint32, C.c_uint64, vp, vp, vp, u64p],
        "orama_post_info": [vp, u64p, u32p, u64p, f32p],
        "orama_post_fill_synthetic": [vp, C.c_uint64, C.c_uint64, C.c_uint32, vp, C.c_uint64, u64p],
        "orama_post_search": [vp, C.POINTER(TermRef), C.c_uint32, C.c_float, C.POINTER(Bm25Params), vp,
                              C.c_uint64, C.c_int, vp, vp, u32p, u64p],
        "orama_post_search_batch": [vp, C.POINTER(PostQueryDesc), C.c_uint32, C.c_float, vp, C.c_uint64, C.c_int, C.c_uint32,
                                    C.c_uint32, vp, vp, vp, vp],
        "orama_post_search_batch_status": [vp, C.POINTER(PostQueryDesc), C.c_uint32, C.c_float, vp, C.c_uint64, C.c_int, C.c_uint32,
                                    C.c_uint32, vp, vp, vp, vp, vp],
        "orama_post_batcher_create": [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)],
        "orama_post_batcher_create_group": [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)],
        "orama_shard_post_search_batch": [vp, vp, C.POINTER(PostQueryDesc), C.c_uint32, C.c_float, vp, C.c_uint64, C.c_int,
                                          C.c_uint32, vp, vp, vp, vp, vp],
        "orama_post_batcher_search": [vp, C.POINTER(TermRef), C.c_uint32, C.c_float, C.POINTER(Bm25Params), vp,
                                      C.c_uint64, C.c_int, vp, vp, u32p, u64p],
        "orama_post_batcher_stats": [vp, u64p, u64p, u32p],
        "orama_post_search_hybrid": [vp, C.POINTER(TermRef), C.c_uint32, C.c_float, C.POINTER(Bm25Params), vp,
                                     C.c_uint64, vp, vp, C.c_uint32, C.c_int, vp, vp, u32p, u64p],
        "orama_hybrid_search": [vp, vp, vp, C.c_uint32, C.c_float, C.c_int, C.POINTER(TermRef), C.c_uint32, C.c_float,
                                C.POINTER(Bm25Params), vp, C.c_uint64, C.c_int, vp, vp, u32p, u64p],
        "orama_post_query_begin": [vp, C.POINTER(TermRef), C.c_uint32, C.c_float, C.POINTER(Bm25Params), vp, C.c_uint64,
                                   C.c_int, C.c_int, C.c_uint32, vp, vp, C.POINTER(vp)],
        "orama_post_query_score": [vp, vp, vp],
        "orama_post_query_finish": [vp, vp, vp, vp, C.c_uint32, vp],
        "orama_post_merge_blocks_device": [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp],
        "orama_post_set_avg_len": [vp, vp, C.c_uint32],
        "orama_shard_unique_id": [vp],
        "orama_shard_group_create": [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)],
        "orama_shard_group_create_rank": [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
        "orama_shard_group_info": [vp, u32p, u32p, u32p, C.POINTER(C.c_int)],
        "orama_shard_group_barrier": [vp],
        "orama_shard_group_allreduce_max_f64": [vp, C.POINTER(C.c_double)],
        "orama_shard_vec_search": [vp, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, vp, vp],
        "orama_shard_post_search": [vp, vp, C.POINTER(TermRef), C.c_uint32, C.c_float, C.POINTER(Bm25Params), vp,
                                    C.c_uint64, C.c_int, C.c_int, vp, vp, C.c_uint32, vp, vp, u32p, u64p],
        "orama_shard_hybrid_search": [vp, vp, vp, vp, C.c_uint32, C.c_float, C.c_int, C.POINTER(TermRef), C.c_uint32,
                                      C.c_float, C.POINTER(Bm25Params), vp, C.c_uint64, C.c_int, vp, vp, u32p, u64p],
        "orama_shard_session_create": [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)],
        "orama_shard_session_step": [vp, C.c_uint32],
        "orama_shard_session_sync": [vp],
        "orama_shard_session_result": [vp, C.c_uint32, vp, vp, vp],
        "orama_post_search_scores": [vp, C.POINTER(TermRef), C.c_uint32, C.c_float, C.POINTER(Bm25Params), vp, C.c_uint64,
                                     C.c_int, vp, vp, C.c_uint32, C.c_int, vp, vp, u32p, u64p, C.POINTER(vp)],
        "orama_scores_count": [vp, u64p],
        "orama_scores_export": [vp, C.c_uint64, vp, vp, u64p],
        "orama_scores_lookup": [vp, vp, C.c_uint32, vp, vp],
        "orama_facet_field_create_buckets": [vp, vp, vp, C.c_uint32, C.POINTER(vp)],
        "orama_facet_field_create_numbers": [vp, vp, vp, C.c_uint64, C.POINTER(vp)],
        "orama_facet_count": [vp, vp, vp],
        "orama_facet_count_ranges": [vp, vp, vp, vp, C.c_uint32, vp],
        "orama_group_top": [vp, vp, C.c_uint32, vp, vp, vp],
        "orama_hybrid_combine": [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.c_uint32, vp, vp, u32p, u64p],
        "orama_hybrid_rrf": [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.c_float, C.c_uint32, C.c_uint32, vp, vp, u32p,
                             u64p],
        "orama_top_n": [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, u32p],
    }
    sig["orama_device_count"] = [C.POINTER(C.c_int)]
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.orama_packed_block_bytes.argtypes = [C.c_uint32, C.c_uint32]
    lib.orama_packed_block_bytes.restype = C.c_uint64
    lib.orama_allow_token.argtypes = [vp]
    lib.orama_allow_token.restype = vp
    lib.orama_post_block_bytes.argtypes = [C.c_uint32]
    lib.orama_post_block_bytes.restype = C.c_uint64
    for name in ("orama_dev_free", "orama_stream_destroy"):
        fn = getattr(lib, name)
        fn.argtypes = [vp, vp]
        fn.restype = None
    lib.orama_shard_group_ctx.argtypes = [vp, C.c_uint32]
    lib.orama_shard_group_ctx.restype = vp
    for name in ("orama_ctx_destroy", "orama_scores_destroy", "orama_facet_field_destroy", "orama_shard_group_destroy", "orama_shard_session_destroy", "orama_vec_destroy", "orama_post_destroy", "orama_post_query_end",
                 "orama_batcher_destroy", "orama_post_batcher_destroy", "orama_allow_destroy",
                 "orama_dict_destroy"):
        fn = getattr(lib, name)
        fn.argtypes = [vp]
        fn.restype = None


def check(status: int) -> None:
    if status != ORAMA_OK:
        msg = load().orama_last_error().decode("utf-8", "replace")
        raise OramaError(status, msg)
