"""Per-GPU context (orama_ctx) — owns the device ordinal, stream/scratch pools and the profiler."""
from __future__ import annotations

import ctypes as C
import weakref

from . import _native as N


class Owner:
    """Handles die in dependency order: an owner closes what was created on it (stores on a context; batchers, score
    maps and facet fields on a store) before it releases its own handle — also when the interpreter tears objects down
    in arbitrary order at exit."""

    def _adopt(self, child) -> None:
        kids = self.__dict__.get("_children")
        if kids is None:
            kids = self.__dict__["_children"] = weakref.WeakSet()
        kids.add(child)

    def _close_children(self) -> None:
        kids = self.__dict__.get("_children")
        if kids:
            for c in list(kids):
                try:
                    c.close()
                except Exception:  # noqa: BLE001
                    pass
            kids.clear()


class Context(Owner):
    def __init__(self, device: int = 0):
        self._lib = N.load()
        h = C.c_void_p()
        N.check(self._lib.orama_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    @property
    def handle(self) -> C.c_void_p:
        if not self._h:
            raise RuntimeError("context already closed")
        return self._h

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._close_children()
            self._lib.orama_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def synchronize(self) -> None:
        N.check(self._lib.orama_ctx_synchronize(self.handle))

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cu = C.c_int()
        hbm = C.c_uint64()
        N.check(self._lib.orama_ctx_device_info(self.handle, name, C.byref(cu), C.byref(hbm)))
        return {"name": name.value.decode(), "compute_units": cu.value, "hbm_bytes": hbm.value}

    def set_scan_tuning(self, rows_per_wave: int, blocks_per_cu: int, nontemporal: bool) -> None:
        N.check(self._lib.orama_ctx_set_scan_tuning(self.handle, rows_per_wave, blocks_per_cu, 1 if nontemporal else 0))

    def set_f16_tuning(self, ksteps_per_chunk: int, ring_chunks: int) -> None:
        N.check(self._lib.orama_ctx_set_f16_tuning(self.handle, ksteps_per_chunk, ring_chunks))

    def set_two_stage(self, on: bool, always: bool = False) -> None:
        """DTYPE_F32_SHADOW16 stores: True = fp16 candidates + fp32 decision where it pays (default), False = plain fp32
        scan; always=True takes the two stages for small stores too."""
        N.check(self._lib.orama_ctx_set_two_stage(self.handle, (2 if always else 1) if on else 0))

    def set_bm25_ranges(self, on: bool, hybrid: bool = True, compact_keys: bool | str | None = None) -> None:
        """BM25 searches: True = K3r range-partitioned batch scorer (default; hybrid=False keeps it to the plain top-k search),
        False = K3 per-document records.  `compact_keys` (None = leave as it is): False = round 4's key lists, one slot per
        posting; True = compact lists for batches of 8 queries and more (the default); "always" = for every batch size — the
        option "k3r_compact", independent of the scorer choice."""
        N.check(self._lib.orama_ctx_set_bm25_ranges(self.handle, (1 if hybrid else 2) if on else 0))
        if compact_keys is not None:
            self.set_option("k3r_compact", 2 if compact_keys == "always" else 1 if compact_keys else 0)

    def set_option(self, name: str, value: int) -> None:
        """A tuning / test option by name (orama_ctx_set_option): alternative code paths with the same answers — nothing a deployment sets."""
        N.check(self._lib.orama_ctx_set_option(self.handle, name.encode(), int(value)))

    def set_f32_batch(self, min_queries: int = 9) -> None:
        """Plain fp32 stores: batches of >= min_queries queries take K1m (fp32 MFMA, <= 32 queries per corpus pass); 0 = never."""
        N.check(self._lib.orama_ctx_set_f32_batch(self.handle, int(min_queries)))

    def set_f16_wide(self, mode: int) -> None:
        """0 = K2 passes of 64, 1 = K2c, 2 / 3 = K2d geometry 1 / 2, 4 = K2q (default), 5 = K2h (orama_ctx_set_f16_wide)."""
        N.check(self._lib.orama_ctx_set_f16_wide(self.handle, int(mode)))

    # --- HIP-event profiler (bench.py roofline leg)
    def prof_enable(self, on: bool = True) -> None:
        N.check(self._lib.orama_prof_enable(self.handle, 1 if on else 0))

    def prof_reset(self) -> None:
        N.check(self._lib.orama_prof_reset(self.handle))

    def prof_get(self, kernel: str) -> tuple[float, int]:
        ms = C.c_double()
        n = C.c_uint64()
        N.check(self._lib.orama_prof_get(self.handle, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def prof_samples(self, kernel: str):
        """Per-launch durations (ms, oldest first) of `kernel` since the last reset — the median is what the roofline quotes."""
        import numpy as np

        n = C.c_uint64()
        N.check(self._lib.orama_prof_samples(self.handle, kernel.encode(), None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        if n.value:
            N.check(self._lib.orama_prof_samples(self.handle, kernel.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), n.value,
                                                 C.byref(n)))
        return out[:n.value]

    def pci_bus_id(self) -> str:
        """PCI address of this context's device ("0000:c5:00.0"): the key monitors find it by (HIP ordinals are not DRM cards)."""
        buf = C.create_string_buffer(64)
        N.check(self._lib.orama_ctx_pci_bus_id(self.handle, buf, 64))
        return buf.value.decode().lower()


class DeviceBuffer:
    """A raw HBM allocation (orama_dev_*): what callers of the *_device entry points pass as device pointers.
    Plumbing for tests / bench.py — replaces the torch tensors used for this in round 1."""

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        N.check(ctx._lib.orama_dev_malloc(ctx.handle, self.nbytes, C.byref(p)))
        self._p = p

    @property
    def ptr(self) -> int:
        if not self._p:
            raise RuntimeError("device buffer already freed")
        return self._p.value

    def upload(self, array, offset: int = 0) -> "DeviceBuffer":
        import numpy as np

        a = np.ascontiguousarray(array)
        assert offset + a.nbytes <= self.nbytes
        N.check(self.ctx._lib.orama_dev_upload(self.ctx.handle, self._p, offset, a.ctypes.data, a.nbytes))
        return self

    def download(self, dtype, count: int, offset: int = 0):
        import numpy as np

        out = np.empty(count, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        N.check(self.ctx._lib.orama_dev_download(self.ctx.handle, self._p, offset, out.ctypes.data, out.nbytes))
        return out

    def free(self) -> None:
        if getattr(self, "_p", None):
            self.ctx._lib.orama_dev_free(self.ctx.handle, self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class Stream:
    """A HIP stream created by the library (orama_stream_*)."""

    def __init__(self, ctx: Context, high_priority: bool = False):
        self.ctx = ctx
        p = C.c_void_p()
        N.check(ctx._lib.orama_stream_create(ctx.handle, 1 if high_priority else 0, C.byref(p)))
        self._p = p

    @property
    def ptr(self) -> int:
        return self._p.value or 0

    def synchronize(self) -> None:
        N.check(self.ctx._lib.orama_stream_synchronize(self.ctx.handle, self._p))

    def close(self) -> None:
        if getattr(self, "_p", None):
            self.ctx._lib.orama_stream_destroy(self.ctx.handle, self._p)
            self._p = None
