"""bench_legs.py — the measurement legs and helpers of bench.py (split out in round 6: the contract — arguments, the compact metric
line, main() — stays in bench.py; this file holds the CPU baseline, the clock / power sampler binding, the vector / two-stage / hybrid
legs, the post-run oracle checks and the live PMC traffic passes).  Nothing here is imported by the product."""
from __future__ import annotations

import csv
import ctypes
import glob
import json
import os
import shutil
import statistics
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
MFMA_F16_PEAK_TFLOPS = 2500.0
MFMA_F32_PEAK_TFLOPS = 157.3  # f32-input MFMA (v_mfma_f32_32x32x2_f32): MI355X_MICROARCH.md, 155 TF measured

WORKLOADS = {
    # name: (rows, dim, k, queries per step, storage dtype, description)
    "ns": (10_000_000, 768, 100, 1, "f32", "10M x 768 fp32 embeddings, single-query cosine top-100 (north-star)"),
    "nsb": (10_000_000, 768, 100, 64, "f32", "10M x 768 fp32 embeddings (the north-star rows, plain fp32 store), 64 concurrent queries per corpus pass: "
                                             "K1x (rows rounded to fp16 in registers, fp16 MFMA) proposes, K1 decides — answers bit-identical to single queries"),
    "c2": (1_000_000, 384, 100, 1, "f32", "1M x 384 fp32 embeddings, single-query cosine top-100 (BASELINE configs[1])"),
    "c3": (10_000_000, 768, 100, 64, "f16", "10M x 768 fp16 embeddings, batch-64 queries, MFMA scan + top-100 "
                                            "(BASELINE configs[2])"),
    "c5": (80_000_000, 768, 100, 256, "f16", "80M x 768 fp16 embeddings sharded over the ranks, batch-256 queries, "
                                             "MFMA scan + top-100 + RCCL all-gather (BASELINE configs[4])"),
}
EXTRA_CONFIGS = ("c4", "f32_batch", "c2", "c3", "c5_shard")



# ---------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(store, dim: int, n_total: int, k: int, sample_rows: int, details: bool = False) -> dict:
    """The oracle (C restatement of the reference algorithm — NOT the reference binary) timed on this
    host: sequential-f32 cosine distances over a bounded sample of the same corpus + top-k, single
    thread (the reference runs one search synchronously on one tokio worker) and, for
    context, row-parallel on all cores.  QPS is extrapolated linearly from rows/s to the full corpus."""
    from oracle import oracle as orc  # CPU baseline leg only

    sample_rows = min(sample_rows, store.info()["num_rows"])
    idx = np.arange(sample_rows, dtype=np.uint64)
    rows, _ = store.get_rows(idx)
    rng = np.random.default_rng(0xBEEF)
    qs = rng.standard_normal((4, dim)).astype(np.float32)
    orc.distances(rows[:1000], qs[0])  # warm the library

    def run(threads: int, min_seconds: float):
        t0 = time.perf_counter()
        passes = 0
        while True:
            d = orc.distances(rows, qs[passes % len(qs)], threads=threads)
            np.argpartition(d, min(k, sample_rows - 1))[:k]
            passes += 1
            el = time.perf_counter() - t0
            if el >= min_seconds and passes >= 3:
                return sample_rows * passes / el

    cores = os.cpu_count() or 1
    rows_per_s_1 = run(1, 8.0 if details else 6.0)
    rows_per_s_all = run(cores, 4.0 if details else 2.0)

    # BASELINE.md §3's "fast CPU" leg: the same scan as a CPU implementation would write it — -O3 -march=native, 8 independent
    # FMA accumulators per row (vectorised), row-parallel — compiled on THIS host (oracle/cpu_fast.py).  Not order-exact
    # (agrees with the oracle to ~1e-6); a thread-scaling line shows where the host's DRAM bandwidth saturates.
    from oracle import cpu_fast as cf

    err = float(np.max(np.abs(cf.distances(rows[:20000], qs[0], 1) - orc.distances(rows[:20000], qs[0]))))

    def run_fast(threads: int, min_seconds: float):
        t0 = time.perf_counter()
        passes = 0
        while True:
            d = cf.distances(rows, qs[passes % len(qs)], threads)
            np.argpartition(d, min(k, sample_rows - 1))[:k]
            passes += 1
            el = time.perf_counter() - t0
            if el >= min_seconds and passes >= 3:
                return sample_rows * passes / el

    # default run: four points round the thread count that won on every host so far (16-64); --details walks the whole ladder
    ladder = sorted({t for t in ((1, 2, 4, 8, 16, 32, 64, 128, cores) if details else (1, 16, 32, 64)) if t <= cores})
    scaling = []
    for t in ladder:
        rps = run_fast(t, 2.0 if t == 1 else 1.0)
        scaling.append({"threads": t, "value": rps / n_total, "gbytes_per_s": rps * dim * 4 / 1e9})
    # ... and the same scan over a copy of the sample whose pages were FIRST TOUCHED by the threads that scan them (round 4's
    # ladder fell beyond 16-32 threads: one numpy array, placed by one thread, read across the sockets — VERDICT r04 weak #9)
    placed = []
    for t in ([x for x in ladder if x >= 16][-4:] if details else []):
        with cf.PlacedRows(rows, t) as pr:
            pr.distances(qs[0])
            t0, passes = time.perf_counter(), 0
            while time.perf_counter() - t0 < 1.0 or passes < 3:
                d = pr.distances(qs[passes % len(qs)])
                np.argpartition(d, min(k, sample_rows - 1))[:k]
                passes += 1
            rps = sample_rows * passes / (time.perf_counter() - t0)
        placed.append({"threads": t, "value": rps / n_total, "gbytes_per_s": rps * dim * 4 / 1e9, "placement": "first touch by the scanning threads"})
    scaling_all = scaling + placed
    best = max(scaling_all, key=lambda e: e["value"])
    return {
        "value": rows_per_s_1 / n_total,
        "unit": "queries/s",
        "cores": 1,
        "kind": "port",
        "sample_rows": sample_rows,
        "sample": f"oracle orc_distances_f32 (scalar, sequential f32) + top-{k} over the first {sample_rows} rows "
                  f"of the same corpus, repeated >= {8 if details else 6} s; QPS = rows/s / {n_total}",
        "gbytes_per_s": rows_per_s_1 * dim * 4 / 1e9,
        "all_cores": {"value": rows_per_s_all / n_total, "cores": cores,
                      "gbytes_per_s": rows_per_s_all * dim * 4 / 1e9},
        "fast": {"value": best["value"], "unit": "queries/s", "cores": best["threads"], "kind": "port",
                 "gbytes_per_s": best["gbytes_per_s"], "placement": best.get("placement", "one array, first touched by one thread"),
                 "single_thread": scaling[0], "thread_scaling": scaling, "thread_scaling_numa_placed": placed,
                 "build": "gcc " + cf.build_flags() + " on this host", "max_abs_diff_vs_order_exact_oracle": err,
                 "sample": f"oracle/orama_cpu_fast.c cpf_distances_f32 (8 FMA accumulators per row, vectorised; fast CPU, NOT "
                           f"order-exact) + top-{k} over the same {sample_rows} rows, >= 1 s per thread count; the best "
                           "thread count is quoted"},
        "note": "restatements of the reference's algorithm, not the reference's (un-vendored, presumably SIMD) crate: reported "
                "baselines, never a speed-up claim.  `value` is the order-exact scalar oracle on one thread (the reference runs "
                "one search on one tokio worker); `fast` is what -O3 -march=native + all cores buy on this host",
    }


class GsSummary(ctypes.Structure):
    _C = ctypes
    _fields_ = [("available", _C.c_int), ("samples", _C.c_uint32), ("seconds", _C.c_double),
                ("gfxclk_mhz_median", _C.c_double), ("gfxclk_mhz_min", _C.c_double), ("gfxclk_mhz_max", _C.c_double),
                ("xcd_spread_mhz_max", _C.c_double), ("socket_power_w_mean", _C.c_double), ("socket_power_w_max", _C.c_double),
                ("energy_j", _C.c_double), ("energy_power_w", _C.c_double), ("ppt_residency_pct", _C.c_double),
                ("thm_residency_pct", _C.c_double), ("gfx_activity_pct_median", _C.c_double), ("xcds_reporting", _C.c_uint32),
                ("rsmi_index", _C.c_uint32), ("bdfid", _C.c_uint64),
                # round 6: the memory side (uclk, fabric clock, HBM temperature, memory-controller activity, per-XCD busy spread)
                ("uclk_mhz_median", _C.c_double), ("uclk_mhz_min", _C.c_double), ("socclk_mhz_median", _C.c_double),
                ("temp_hbm_c_max", _C.c_double), ("temp_mem_c_max", _C.c_double), ("temp_hotspot_c_max", _C.c_double),
                ("umc_activity_pct_median", _C.c_double), ("xcd_busy_spread_pct", _C.c_double)]


_SAMPLER_LIB = None


def sampler_lib():
    """scripts/native/libgpu_sampler.so (gpu_sampler.c: a native thread over librocm_smi64's decoded gpu_metrics table), built on
    first use with gcc when __graft_entry__.build() has not; None when it cannot be had — the record then says so."""
    global _SAMPLER_LIB
    if _SAMPLER_LIB is not None:
        return _SAMPLER_LIB or None
    src = ROOT / "scripts" / "native" / "gpu_sampler.c"
    lib = src.with_name("libgpu_sampler.so")
    try:
        if not lib.exists() or lib.stat().st_mtime < src.stat().st_mtime:
            subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-I/opt/rocm/include", str(src), "-o", str(lib), "-L/opt/rocm/lib",
                            "-lrocm_smi64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, timeout=120)
        l = ctypes.CDLL(str(lib))
        l.gs_last_error.restype = ctypes.c_char_p
        l.gs_open.argtypes = [ctypes.c_char_p]
        l.gs_start.argtypes = [ctypes.c_double]
        l.gs_stop.argtypes = [ctypes.POINTER(GsSummary)]
        _SAMPLER_LIB = l
    except (OSError, subprocess.SubprocessError):
        _SAMPLER_LIB = False
    return _SAMPLER_LIB or None


class ClockSampler:
    """Shader clock, socket power and throttle residency of the GPU at PCI address `bdf` (orama_ctx_pci_bus_id of the device the
    leg runs on) while a leg runs.  VERDICT r04 weak #4: round 4 indexed /sys/class/drm/card* by HIP ordinal and read hwmon's
    slow power1_average from a Python thread — on the driver's box that was another card (97 MHz during a saturated scan).  Now:
    the device is found by its PCI address, the firmware's gpu_metrics table (per-XCD gfxclk, current socket power, energy and
    PPT / thermal residency accumulators) is sampled every 2 ms by a native thread (no GIL), and a record whose clock reads
    under 500 MHz while the GPU was busy is marked unavailable instead of being quoted.  Fallback when librocm_smi64 cannot be
    used: hwmon under /sys/bus/pci/devices/<bdf>/ (labelled).  Everything is optional: a box without either says so."""

    BUSY_MIN_MHZ = 500.0

    def __init__(self, bdf: str | None, period_s: float = 0.002):
        import threading

        self.bdf = (bdf or "").lower()
        self.period = period_s
        self.lib = sampler_lib() if self.bdf else None
        self.native = False
        self.note = None
        if self.lib is not None:
            if self.lib.gs_open(self.bdf.encode()) >= 0:
                self.native = True
            else:
                self.note = "rocm_smi: " + self.lib.gs_last_error().decode()
        self.power_file = self.freq_file = None
        if not self.native and self.bdf:
            base = f"/sys/bus/pci/devices/{self.bdf}"
            for name in ("power1_input", "power1_average"):
                hits = glob.glob(os.path.join(base, "hwmon/hwmon*/" + name))
                if hits:
                    self.power_file = hits[0]
                    break
            hits = glob.glob(os.path.join(base, "hwmon/hwmon*/freq1_input"))
            self.freq_file = hits[0] if hits else None
        self.samples_w, self.samples_mhz = [], []
        self._stop = threading.Event()
        self._thread = None
        self._summary = None

    def _read(self):
        try:
            if self.power_file:
                self.samples_w.append(int(open(self.power_file).read().strip()) / 1e6)
            if self.freq_file:
                self.samples_mhz.append(int(open(self.freq_file).read().strip()) / 1e6)
        except (OSError, ValueError):
            pass

    def __enter__(self):
        import threading

        if self.native:
            self.native = self.lib.gs_start(self.period) == 0
        elif self.power_file or self.freq_file:
            def loop():
                while not self._stop.is_set():
                    self._read()
                    time.sleep(self.period)
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self.native:
            g = GsSummary()
            self.lib.gs_stop(g)
            self._summary = g
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=1.0)

    def summary(self) -> dict:
        base = {"pci_bus_id": self.bdf or None}
        if self.note:
            base["note"] = self.note
        g = self._summary
        if g is not None and g.available:
            out = dict(base, available=True, samples=int(g.samples), seconds=g.seconds,
                       source="gpu_metrics table via librocm_smi64, device matched by PCI address, native sampling thread",
                       sclk_mhz_median=g.gfxclk_mhz_median, sclk_mhz_min=g.gfxclk_mhz_min, sclk_mhz_max=g.gfxclk_mhz_max,
                       xcds_reporting=int(g.xcds_reporting), xcd_spread_mhz_max=g.xcd_spread_mhz_max,
                       power_w_mean=g.socket_power_w_mean, power_w_max=g.socket_power_w_max,
                       energy_j=g.energy_j or None, power_w_from_energy_counter=g.energy_power_w or None,
                       ppt_throttle_residency_pct=None if g.ppt_residency_pct < 0 else g.ppt_residency_pct,
                       thermal_throttle_residency_pct=None if g.thm_residency_pct < 0 else g.thm_residency_pct,
                       gfx_activity_pct_median=None if g.gfx_activity_pct_median < 0 else g.gfx_activity_pct_median,
                       uclk_mhz_median=g.uclk_mhz_median or None, uclk_mhz_min=g.uclk_mhz_min or None,
                       socclk_mhz_median=g.socclk_mhz_median or None,
                       temp_hbm_c_max=g.temp_hbm_c_max or None, temp_mem_c_max=g.temp_mem_c_max or None,
                       temp_hotspot_c_max=g.temp_hotspot_c_max or None,
                       umc_activity_pct_median=None if g.umc_activity_pct_median < 0 else g.umc_activity_pct_median,
                       xcd_busy_spread_pct=None if g.xcd_busy_spread_pct < 0 else g.xcd_busy_spread_pct)
            if g.gfxclk_mhz_median < self.BUSY_MIN_MHZ:
                out.update(available=False, reason=f"median clock {g.gfxclk_mhz_median:.0f} MHz during a busy region: not this GPU's "
                                                   "clock domain (or the table is stale) — record kept for inspection, not evidence")
            return out
        if not self.samples_w and not self.samples_mhz:
            return dict(base, available=False)
        out = dict(base, available=True, samples=max(len(self.samples_w), len(self.samples_mhz)),
                   source=f"hwmon under /sys/bus/pci/devices/{self.bdf}/ ({os.path.basename(self.power_file or '')}, Python thread)")
        if self.samples_mhz:
            out.update(sclk_mhz_median=float(np.median(self.samples_mhz)), sclk_mhz_min=float(np.min(self.samples_mhz)),
                       sclk_mhz_max=float(np.max(self.samples_mhz)))
            if out["sclk_mhz_median"] < self.BUSY_MIN_MHZ:
                out.update(available=False, reason="clock under 500 MHz during a busy region")
        if self.samples_w:
            out.update(power_w_mean=float(np.mean(self.samples_w)), power_w_max=float(np.max(self.samples_w)))
        return out


def pctl(lat_ms) -> dict:
    a = np.asarray(lat_ms, dtype=np.float64)
    return {"latency_ms_p50": float(np.percentile(a, 50)), "latency_ms_p95": float(np.percentile(a, 95)), "latency_samples": int(a.size)}


def vec_two_stage_ok(dim: int, k: int) -> bool:
    return dim % 4 == 0 and dim <= 1024 and 2 * k <= 4096


def shadow_store(oa, ctx, dim, n_local, lo, rank):
    """The north-star rows again, in a store that also keeps an fp16 copy of them (DTYPE_F32_SHADOW16, +50 % HBM)."""
    st = oa.EmbeddingFieldStorage(ctx, dimensions=dim, reserve_rows=n_local, dtype=oa.DTYPE_F32_SHADOW16)
    st.fill_synthetic(n_local, seed=0xC0FFEE + rank, first_doc_id=lo)
    return st


def two_stage_session(group, plain, st, k, qb, queries_h, steps=40, warmup=5) -> dict:
    """The shadow store through the SAME pipelined session as `value` (queries resident in HBM, results stay there, no host
    between the stages): the device form of the plan — the fallback for queries that are not proven is decided and run on
    the device.  Checked against the plain store's session, step by step, on the last two steps."""
    total = warmup + steps
    nq = (queries_h.shape[0] // qb)
    out = {}
    last = {}
    for name, store in (("shadow", st), ("plain", plain)):
        sess = group.session([store], queries_h, qb, k, n_slots=2)
        for i in range(warmup):
            sess.step(i)
        sess.sync()
        t0 = time.perf_counter()
        for i in range(warmup, total):
            sess.step(i)
        sess.sync()
        out[name] = steps * qb / (time.perf_counter() - t0)
        last[name] = [sess.result((total - 1 - j) % 2) for j in range(2)]
        sess.close()
    same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])
               for a, b in zip(last["shadow"], last["plain"]))
    assert same, "two-stage session answers differ from the fp32 session's"
    assert nq >= 1
    return {"value": out["shadow"], "unit": "queries/s", "plain_fp32_session": out["plain"], "queries_per_step": qb,
            "identical_to_fp32_session": same, "steps": steps}


def two_stage_leg(oa, ctx, plain, st, dim, k, qb, queries_h, group=None) -> dict:
    """Host-buffer API, one call per query batch: plain fp32 store vs fp32 rows + fp16 shadow (`st`, closed here)."""
    nq = min(40, queries_h.shape[0] // qb)
    identical = True
    session = two_stage_session(group, plain, st, k, qb, queries_h) if group is not None else None
    for i in range(3):
        a = plain.storage_search(queries_h[i * qb:(i + 1) * qb], k)
        b = st.storage_search(queries_h[i * qb:(i + 1) * qb], k)
        identical &= bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and
                          np.array_equal(a[2], b[2]))
    t0 = time.perf_counter()
    for i in range(nq):
        st.storage_search(queries_h[i * qb:(i + 1) * qb], k)
    el = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(nq):
        plain.storage_search(queries_h[i * qb:(i + 1) * qb], k)
    el_plain = time.perf_counter() - t0
    qs64 = np.random.default_rng(7).standard_normal((64, dim)).astype(np.float32)
    st.storage_search(qs64, k)
    t0 = time.perf_counter()
    for _ in range(5):
        st.storage_search(qs64, k)
    el64 = time.perf_counter() - t0
    info = st.info()
    st.close()
    assert identical, "two-stage answers differ from the fp32 scan"
    return {"value": nq * qb / el, "unit": "queries/s", "plain_fp32_same_api": nq * qb / el_plain,
            "batch64_queries_per_s": 5 * 64 / el64, "identical_to_fp32_scan": identical,
            "fallbacks": int(info["two_stage_fallbacks"]), "queries": int(info["two_stage_queries"]),
            "hbm_bytes": int(info["hbm_bytes"]), "session": session,
            "note": "fp32 rows + fp16 shadow (ORAMA_DTYPE_F32_SHADOW16): the fp16 scan proposes max(2k, k+256) candidates, "
                    "K1's arithmetic on the fp32 rows decides; a query whose candidate list cannot be proven complete "
                    "falls back to the fp32 scan; results bit-identical to the plain store (DESIGN.md K1s)"}


# ---------------------------------------------------------------------------------------------- vector legs
def check_vector_result(store, ids_all, dst_all, cnt, queries_last, k, qb, lo, hi, f16) -> None:
    """Post-run check of one step against the oracle (outside the timed region): sorted, exact on re-computation from
    the stored rows, and no row of a random sample of the shard beats the reported k-th distance unless it was returned
    (the size-independent property of SURVEY §8c).  Raises instead of letting a wrong number be printed."""
    from oracle import oracle as orc  # checker only

    n_local = hi - lo
    assert np.all(cnt == k), "bench result incomplete"
    for qi in sorted({0, qb - 1}):
        ids_h, dst_h = ids_all[qi], dst_all[qi]
        assert np.all(np.diff(dst_h) >= 0), "bench result not sorted"
        qv = queries_last[qi]
        if f16:  # the fp16 path scores the fp16-rounded query against the stored fp16 rows
            qv = qv.astype(np.float16).astype(np.float32)
        mine = (ids_h >= lo) & (ids_h < hi)
        if mine.any():
            rows, _ = store.get_rows((ids_h[mine] - np.uint64(lo)).astype(np.uint64))
            od = orc.distances(rows, qv)
            err = float(np.max(np.abs(od - dst_h[mine])))
            assert err <= 1e-4, f"bench parity check failed: {err}"
        srng = np.random.default_rng(1234 + qi)
        sample = srng.choice(n_local, size=min(20_000, n_local), replace=False).astype(np.uint64)
        srows, sdocs = store.get_rows(sample)
        sd = orc.distances(srows, qv, threads=8)
        inside = set(ids_h.tolist())
        missed = [int(dd) for dd, x in zip(sdocs.tolist(), sd.tolist()) if x < dst_h[-1] - 2e-4 and int(dd) not in inside]
        assert not missed, f"bench parity check failed: rows {missed[:5]} beat the reported k-th distance"


MEDIAN_MIN_LAUNCHES = 50  # the roofline is quoted on the median of >= 50 launches


def scan_step_samples(ctx, kern: str, launches_per_step: int) -> np.ndarray:
    """Scan time per STEP (ms) from the per-launch HIP-event samples: a step of the wide fp16 paths is several launches of
    different sizes (a dense head + super-chunks), so launches are summed step by step before any statistic is taken."""
    smp = ctx.prof_samples(kern).astype(np.float64)
    lps = max(1, launches_per_step)
    n = (smp.size // lps) * lps
    return smp[:n].reshape(-1, lps).sum(axis=1) if n else smp[:0]


def vector_leg(oa, group, name, n_total, steps, warmup, streams, force_exchange=False, rank=0, world=1, lo=0, hi=None,
               store=None, valid=True, desc=None, dump="", f16_slots=1, bdf=None, latency_steps=50):
    """One vector workload through the pipelined shard session: returns (bench-line dict, store, host queries)."""
    _, dim, k, qb, dtype, wdesc = WORKLOADS[name]
    desc = desc or wdesc
    hi = n_total if hi is None else hi
    n_local = hi - lo
    f16 = dtype == "f16"
    ctx = group.ctx(0)
    t_fill = 0.0
    if store is None:
        store = oa.EmbeddingFieldStorage(ctx, dimensions=dim, reserve_rows=n_local, dtype=oa.DTYPE_F16 if f16 else oa.DTYPE_F32)
        t_fill = time.perf_counter()
        store.fill_synthetic(n_local, seed=0xC0FFEE + rank, first_doc_id=lo)
        t_fill = time.perf_counter() - t_fill

    total_b = warmup + steps  # batches
    rng = np.random.default_rng(0xBEEF)
    queries_h = rng.standard_normal((total_b * qb, dim)).astype(np.float32)
    # Stream plan (inside the session): ONE scan stream (corpus scans of consecutive steps run back to back, never
    # concurrently, so each keeps the whole HBM bandwidth) + `--streams` high-priority tail streams used round-robin
    # for top-k / all-gather / merge, which are launch-bound and overlap the next step's scan.
    # fp16 workloads: the scan and its threshold-filter selections depend on each other step by step and all run on the
    # slot's one stream: ONE slot.  Two steps in flight (`--f16-slots 2`, round 4's experiment) cannot overlap their scans —
    # every scan kernel fills all CUs with full-register waves, the second one's workgroups wait for the first's to leave —
    # only one slot's launch-bound selection chain runs beside the other slot's scan: +2.4 % at 256 queries per step and
    # +4.1 % at 64 measured alone, nothing inside this script's full run, and every HIP-event duration then includes the
    # wait for the other slot (profiles/r04_f16_slots_experiment.log).  Not the default.
    n_streams = max(1, f16_slots) if f16 else max(1, streams)
    sess = group.session([store], queries_h, qb, k, n_slots=n_streams, force_exchange=force_exchange)

    def barrier():
        sess.sync()
        group.barrier()  # local devices drained + one all-reduced word over RCCL + drained again

    for i in range(warmup):
        sess.step(i)
    barrier()
    ctx.prof_reset()
    ctx.prof_enable(True)
    with ClockSampler(bdf) as clocks:
        t0 = time.perf_counter()
        for i in range(warmup, total_b):
            sess.step(i)
        barrier()
        elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    elapsed = group.allreduce_max(elapsed)  # the slowest rank's time

    ids_all, dst_all, cnt = sess.result((total_b - 1) % n_streams)
    check_vector_result(store, ids_all, dst_all, cnt, queries_h[(total_b - 1) * qb:total_b * qb], k, qb, lo, hi, f16)
    if dump:
        np.savez(f"{dump}.rank{rank}.npz", ids=ids_all, dist=dst_all, cnt=cnt, queries=queries_h[(total_b - 1) * qb:total_b * qb],
                 lo=lo, hi=hi, world=world)

    f32_mfma = not f16 and qb >= 9  # the library's default threshold (orama_ctx_set_f32_batch): batches take K1m
    kern = "vec_scan_f16" if f16 else "vec_scan_f32_cvt" if f32_mfma else "vec_scan_f32"  # (K1x where every row has a sound fp16 image: synthetic rows do)
    prof_steps, prof_note = steps, "HIP events on the launching stream over the timed region"
    if f16 and n_streams > 1:
        # Two steps in flight: a launch of one slot queues behind the other slot's scan (every scan kernel fills all CUs), and an
        # event pair on a stream then brackets that wait as well — the kernel's own duration (what rocprofv3 reports, what the
        # roofline is about) is measured on a short pass with ONE step in flight instead.
        sess.close()
        sess = group.session([store], queries_h, qb, k, n_slots=1, force_exchange=force_exchange)
        prof_steps = min(steps, 8)
        sess.step(0)
        sess.sync()
        ctx.prof_reset()
        ctx.prof_enable(True)
        for i in range(1, 1 + prof_steps):
            sess.step(i)
        sess.sync()
        ctx.prof_enable(False)
        prof_note = (f"HIP events over {prof_steps} extra steps with ONE step in flight (the timed region runs {n_streams}: a launch "
                     "then waits for the other slot's scan inside its event pair)")
    scan_ms, scan_n = ctx.prof_get(kern)
    sel_ms, _ = ctx.prof_get("topk_select")
    ag_ms, ag_n = ctx.prof_get("shard_all_gather")
    mg_ms, mg_n = ctx.prof_get("shard_merge")
    kpad = (dim + 127) // 128 * 128
    bytes_per_step = n_local * (kpad * 2 if f16 else dim * 4)  # one corpus pass per step
    launches_per_step = max(scan_n, 1) / prof_steps
    alg_bytes = bytes_per_step / launches_per_step
    avg_scan_s = scan_ms / max(scan_n, 1) / 1e3
    achieved_avg = alg_bytes / avg_scan_s / 1e9 if scan_n else 0.0

    # ---- the figure the roofline is quoted on: MEDIAN scan time per step over >= 50 steps.  The driver's run times
    # 20 steps: the timed region's samples are kept and the same pipelined session simply keeps stepping (profiler on, clocks
    # sampled again) until there are enough.  These steps are not part of `value`.
    lps_int = int(round(launches_per_step))
    uniform = scan_n > 0 and abs(launches_per_step - lps_int) < 1e-9
    extra = 0
    clocks_median = None
    if uniform:
        have = scan_step_samples(ctx, kern, lps_int).size
        extra = max(0, MEDIAN_MIN_LAUNCHES - have)
        if extra:
            ctx.prof_enable(True)
            with ClockSampler(bdf) as clocks2:
                for i in range(extra):
                    sess.step(i % total_b)
                sess.sync()
            ctx.prof_enable(False)
            clocks_median = clocks2.summary()
        per_step = scan_step_samples(ctx, kern, lps_int)
        med_step_ms = float(np.median(per_step))
        achieved = bytes_per_step / (med_step_ms / 1e3) / 1e9
        median_rec = {"median_scan_ms_per_step": med_step_ms, "median_launch_ms": med_step_ms / lps_int,
                      "steps_in_median": int(per_step.size), "of_which_behind_the_timed_region": int(extra),
                      "p05_scan_ms_per_step": float(np.percentile(per_step, 5)), "p95_scan_ms_per_step": float(np.percentile(per_step, 95)),
                      "min_scan_ms_per_step": float(per_step.min())}
    else:
        achieved, median_rec = achieved_avg, {"median_scan_ms_per_step": None, "note": "launch count per step not constant: average used"}

    # ---- per-step latency (BASELINE.json's metric names p50): ONE step in flight, queries resident in HBM, the clock stops when
    # the step's answer (after all-gather + K6 for N > 1) is complete on this rank; the slowest rank's time per step, then
    # percentiles.  The host-buffer API (adds PCIe both ways) is `latency_ms_p50_host_api`, N = 1 only.
    group.barrier()
    lat = []
    for i in range(max(5, latency_steps)):
        t1 = time.perf_counter()
        sess.step(i % total_b)
        sess.sync()
        dt = (time.perf_counter() - t1) * 1e3
        lat.append(group.allreduce_max(dt) if world > 1 else dt)
    lat = lat[min(3, len(lat) - 1):]  # the first steps re-warm a drained pipeline

    out = {
        "metric": f"queries/sec, cosine top-{k} scan ({n_total // 1_000_000}M x {dim} {dtype}) — HBM GB/s vs peak in "
                  "`roofline`",
        "value": steps * qb / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        **pctl(lat),
        "latency_definition": "one step in flight through the session (queries resident in HBM, answer complete on the rank incl. "
                              "all-gather + merge for N > 1), slowest rank per step; p50 / p95 over the samples",
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": desc, "rows_total": n_total, "rows_per_gpu": n_local, "dim": dim, "k": k,
                   "queries_per_step": qb, "parallelism": f"row-shard x{world} + all-gather(top-k) over RCCL (inside "
                   "liborama_hip.so, one process per GPU)" if world > 1 else "single GPU", "streams": n_streams,
                   "valid": bool(valid)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": (("vec_scan_f16_qs_kernel" if qb > 128 and dim <= 768 else "vec_scan_f16_pc_kernel") if f16 and qb > 64 else kern + "_kernel"),
                     "achieved_from": f"median scan time per step over {median_rec.get('steps_in_median')} steps (HIP events on the "
                                      "launching stream)" if uniform else "average launch",
                     "alg_bytes_per_launch": alg_bytes, "alg_bytes_per_step": bytes_per_step,
                     **median_rec,
                     "avg_launch_ms": avg_scan_s * 1e3, "launches": scan_n,
                     "achieved_from_avg_of_timed_region": achieved_avg, "frac_from_avg_of_timed_region": achieved_avg / HBM_PEAK_GBS,
                     "scan_launches_per_step": launches_per_step,
                     "topk_select_ms_per_step": sel_ms / prof_steps,
                     "kernel_durations_from": prof_note,
                     "clocks_during_timed_region": clocks.summary(),
                     "clocks_during_median_steps": clocks_median},
        "step_breakdown_us": {"scan": scan_ms / prof_steps * 1e3, "select": sel_ms / prof_steps * 1e3,
                              "all_gather": (ag_ms / ag_n * 1e3) if ag_n else None,
                              "merge_k6": (mg_ms / mg_n * 1e3) if mg_n else None,
                              "step_wall": elapsed / steps * 1e3,
                              "note": "HIP-event spans on this rank over the timed region; the select and exchange spans run on "
                                      "tail streams beside the next step's scan, so they do not add up to step_wall; all_gather "
                                      "includes the wait for the slowest peer's block"},
        "parity_check": "last step vs oracle: distances recomputed from the rows (<= 1e-4), no better row among 20 000 "
                        "sampled rows of the shard",
        "fill_seconds": t_fill,
    }
    if f16:
        flops = 2.0 * qb * n_local * kpad * prof_steps
        out["roofline"]["mfma_tflops"] = flops / (scan_ms / 1e3) / 1e12 if scan_ms else 0.0
        out["roofline"]["mfma_peak_tflops_dense_f16"] = MFMA_F16_PEAK_TFLOPS
        out["roofline"]["mfma_frac"] = out["roofline"]["mfma_tflops"] / MFMA_F16_PEAK_TFLOPS
    if f32_mfma:
        flops = 2.0 * 32 * ((qb + 31) // 32) * n_local * dim * prof_steps  # (the instruction computes whole 32-column tiles)
        out["roofline"]["mfma_tflops"] = flops / (scan_ms / 1e3) / 1e12 if scan_ms else 0.0
        out["roofline"]["mfma_peak_tflops_dense_f16"] = MFMA_F16_PEAK_TFLOPS
        out["roofline"]["mfma_frac"] = out["roofline"]["mfma_tflops"] / MFMA_F16_PEAK_TFLOPS
        out["roofline"]["note"] = ("K1x: the fp32 rows cross HBM once per 64 queries and are rounded to fp16 in registers (v_mfma_f32_32x32x16_f16); "
                                   "the candidates are re-scored by K1's arithmetic — the answers are the single-query scan's bits.  "
                                   "K1m (f32 x f32, <= 32 per pass, 5.3-5.5 ms) is scripts/k1m_probe.py --plan mfma")
    sess.close()
    return out, store, queries_h


def host_api_latency(store, queries_h, qb, k, n=50) -> dict:
    lat = []
    for i in range(min(n, queries_h.shape[0] // qb)):
        t1 = time.perf_counter()
        store.storage_search(queries_h[i * qb:(i + 1) * qb], k)
        lat.append((time.perf_counter() - t1) * 1e3)
    return {"latency_ms_p50_host_api": float(np.percentile(lat, 50)), "latency_ms_p95_host_api": float(np.percentile(lat, 95))}


# ---------------------------------------------------------------------------------------------- C4: hybrid
def hybrid_leg(oa, ctx, vec, n, dim, k, steps, warmup, n_lists=2048, tokens=12, shadow=None, bdf=None) -> dict:
    """BASELINE configs[3]: 10 M-doc BM25F (12-token queries) + the 10 M x 768 fp32 scan, min-max merge, top-100.
    `vec` is the north-star store (same rows), `shadow` the same rows with an fp16 copy (the vector leg takes the
    two-stage plan there; answers must be identical).  Synthetic postings per SURVEY §8d, generated in HBM."""
    from oramacore_amd import fulltext as ft

    T = tokens
    rng = np.random.default_rng(0xB26)
    ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=n_lists)).astype(np.uint32))
    post = ft.PostingsStore(ctx)
    t0 = time.perf_counter()
    n_post = post.fill_synthetic(n, ranks, seed=0xB25)
    t_fill = time.perf_counter() - t0
    info = post.info()
    total = warmup + steps
    qv = np.random.default_rng(0xBEEF).standard_normal((total, dim)).astype(np.float32)
    qlists = [rng.choice(len(ranks), size=T, replace=False) for _ in range(total)]
    refs = [[(t, int(l), 1.0) for t, l in enumerate(ql)] for ql in qlists]

    # one call per query: vector leg and BM25 leg overlap on two HIP streams.  The arguments are marshalled once, as a native
    # caller holds them (the ctypes / numpy marshalling of the Python wrapper cost ~40 us of every 4.5 ms call), and the timed
    # loop runs WITHOUT the HIP-event profiler (two event records around each of a call's ten launches): the scan's
    # duration for the roofline comes from a short profiled pass behind it.
    calls = [post.prepare_hybrid(vec, qv[i], k, 0.0, refs[i], T, float(n), k) for i in range(total)]

    for i in range(warmup):
        calls[i].run()
    ctx.synchronize()
    plain_results, lat_h = [], []
    with ClockSampler(bdf) as clocks:
        t0 = time.perf_counter()
        for i in range(warmup, total):
            t1 = time.perf_counter()
            plain_results.append(calls[i].run())  # one blocking call per query: its duration IS the request's latency
            lat_h.append((time.perf_counter() - t1) * 1e3)
        ctx.synchronize()
        el_h = time.perf_counter() - t0
    h_ids, h_sc, h_count = plain_results[-1]  # (checked against the oracle below)
    ctx.prof_reset()
    ctx.prof_enable(True)
    for i in range(warmup, min(total, warmup + 10)):
        calls[i].run()
    ctx.synchronize()
    ctx.prof_enable(False)
    scan_ms, scan_n = ctx.prof_get("vec_scan_f32")
    k3_launches = ctx.prof_get("bm25_accumulate")[1]  # 0 = every query took the range scorer + candidate tail
    shadow_out = None
    if shadow is not None:
        s_calls = [post.prepare_hybrid(shadow, qv[i], k, 0.0, refs[i], T, float(n), k) for i in range(total)]
        for i in range(warmup):
            s_calls[i].run()
        ctx.synchronize()
        t0 = time.perf_counter()
        s_results, lat_s = [], []
        for i in range(warmup, total):
            t1 = time.perf_counter()
            s_results.append(s_calls[i].run())
            lat_s.append((time.perf_counter() - t1) * 1e3)
        ctx.synchronize()
        el_s = time.perf_counter() - t0
        same = all(a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
                   for a, b in zip(plain_results, s_results))
        assert same, "hybrid answers on the shadow store differ from the plain store's"
        shadow_out = {"value": steps / el_s, "unit": "queries/s", "ms_per_step": el_s / steps * 1e3, **pctl(lat_s),
                      "identical_to_plain_store": same,
                      "note": "same call on the store with an fp16 shadow: the vector leg is the two-stage plan"}

    # BM25 alone: the batch entry (K3r scores 32 queries per set of launches) and single-query calls
    batch_q = [(refs[i], T, None) for i in range(warmup, total)] * max(1, 2048 // max(steps, 1))
    post.search_batch(batch_q[:64], float(n), k)
    prep = post.prepare_batch(batch_q, float(n), k)  # descriptors marshalled once, as a native caller holds them
    prep.run()
    runs_bb = []
    for _ in range(7):  # (one call is ~10 ms of wall time: the median of seven, the spread beside it)
        t0 = time.perf_counter()
        prep.run()
        runs_bb.append(time.perf_counter() - t0)
    el_bb = float(np.median(runs_bb))
    b_res = prep.results()
    t0 = time.perf_counter()
    post.search_batch(batch_q, float(n), k)
    el_bb_py = time.perf_counter() - t0
    # device time per query: chunks of 32 one at a time (a longer batch double-buffers its chunks on two streams, and
    # concurrent kernels stretch each other's event-to-event durations)
    chunks = [post.prepare_batch(batch_q[i:i + 32], float(n), k) for i in range(0, min(len(batch_q), 512), 32)]
    for c in chunks:  # (once untimed: scratch sets sized, code and tables warm — as for every other timed region of this file)
        c.run()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for c in chunks:
        c.run()
    ctx.prof_enable(False)
    n_dev_q = sum(c.nq for c in chunks)
    kb_ms, _ = ctx.prof_get("bm25_range_bounds")
    kd_ms, _ = ctx.prof_get("bm25_range_df")
    ks_ms, _ = ctx.prof_get("bm25_range_score")
    kt_ms, _ = ctx.prof_get("topk_select")
    t0 = time.perf_counter()
    for i in range(warmup, total):
        b_ids, b_sc, b_count = post.search(refs[i], T, float(n), k)
    el_b = time.perf_counter() - t0
    assert b_res[steps - 1][0].tolist() == b_ids.tolist() and b_res[steps - 1][2] == b_count

    # SURVEY §8(d) bytes of the full-text leg: 8 B per posting of the query's lists + 4 B per distinct document touched
    lens = {}
    postings_q, docs_q = [], []
    for j, i in enumerate(range(warmup, total)):
        for l in qlists[i]:
            if l not in lens:
                lens[l] = len(post.get_list(int(l))[0])
        postings_q.append(sum(lens[l] for l in qlists[i]))
        docs_q.append(b_res[j][2])  # `count` = distinct documents with a score (no threshold, no filter)
    avg_postings, avg_docs = float(np.mean(postings_q)), float(np.mean(docs_q))
    bm25_alg_bytes = avg_postings * 8 + avg_docs * 4
    dev_us = (kb_ms + kd_ms + ks_ms + kt_ms) * 1e3 / n_dev_q
    bm25_achieved = bm25_alg_bytes / (dev_us * 1e-6) / 1e9 if dev_us else 0.0

    # ---- parity of the last BM25 and hybrid query against the oracle (checker only)
    from oracle import oracle as orc

    i = total - 1
    entries = []
    for t, l in enumerate(qlists[i]):
        d, tf, ln = post.get_list(int(l))
        b_, avg_ = np.float32(0.75), np.float32(info["avg_field_length"])  # bm25.rs:99-110 in f32, vectorised
        ntf = np.float32(1.0) * (tf.astype(np.float32) / ((np.float32(1.0) - b_) + b_ * (ln.astype(np.float32) / avg_)))
        entries.append((t, d, ntf))
    od, os_ = orc.search_full_text(entries, T, float(n), 1.2, None)
    td, ts = orc.top_n(od, os_, k)
    reps, t_cpu0 = 0, time.perf_counter()
    while time.perf_counter() - t_cpu0 < 2.0:
        orc.top_n(*orc.search_full_text(entries, T, float(n), 1.2, None), k)
        reps += 1
    qsort_qps = reps / (time.perf_counter() - t_cpu0)
    # the loop the reference actually runs (token_score.rs:257-300): hash maps, not a sort — oracle/orama_cpu_fast.c, compiled
    # -O3 -march=native on this host; its answer must equal the oracle's bit for bit before its rate is quoted
    from oracle import cpu_fast as cf

    f_ids, f_sc, f_count = cf.bm25_hashmap(entries, T, float(n), 1.2, None, k)
    assert f_count == len(od) and f_ids.tolist() == td.tolist() and np.array_equal(f_sc.view(np.uint32), ts.view(np.uint32)), \
        "CPU hash-map baseline differs from the oracle"
    reps, t_cpu0 = 0, time.perf_counter()
    while time.perf_counter() - t_cpu0 < 3.0:
        cf.bm25_hashmap(entries, T, float(n), 1.2, None, k)
        reps += 1
    cpu_bm25 = {"value": reps / (time.perf_counter() - t_cpu0), "unit": "queries/s", "cores": 1, "kind": "port",
                "sample": f"oracle/orama_cpu_fast.c cpf_bm25_hashmap on the last query's contributions ({sum(len(e[1]) for e in entries)} "
                          "postings, ntf precomputed): per-token hash map doc -> sum, finalize_term into the document hash map, "
                          "bounded-heap top_n — the reference's loop (token_score.rs:257-300) with a cheaper hash than std's SipHash; "
                          "bit-identical to the oracle; repeated >= 3 s",
                "build": "gcc " + cf.build_flags() + " on this host",
                "order_exact_oracle_qsort_form": {"value": qsort_qps, "unit": "queries/s", "cores": 1}}
    assert b_count == len(od) and b_ids.tolist() == td.tolist(), "BM25 ids differ from the oracle"
    assert np.array_equal(b_sc.view(np.uint32), ts.view(np.uint32)), "BM25 scores differ from the oracle"
    ids, dist, _ = vec.storage_search(qv[i], k)
    sim = (np.float32(1.0) - dist[0]).astype(np.float32)
    cd, cs = orc.normalize_and_combine(ids[0], sim, od, os_)
    hd, hs = orc.top_n(cd, cs, k)
    assert h_count == len(cd) and h_ids.tolist() == hd.tolist(), "hybrid ids differ from the oracle"
    assert np.array_equal(h_sc.view(np.uint32), hs.view(np.uint32)), "hybrid scores differ from the oracle"
    post.close()

    alg_vec = n * dim * 4
    avg_scan_s = scan_ms / max(scan_n, 1) / 1e3
    achieved = alg_vec / avg_scan_s / 1e9 if scan_n else 0.0
    return {
        "metric": "queries/sec, hybrid search: 10M-doc BM25F (12 tokens) + 10M x 768 fp32 cosine scan, min-max merge, top-100",
        "value": steps / el_h, "unit": "queries/s", "steps": steps, "warmup": warmup, "ms_per_step": el_h / steps * 1e3,
        **pctl(lat_h),
        "latency_definition": "wall time of one blocking orama_hybrid_search call (query upload, both legs, merge, result in host "
                              "memory)",
        "dtype": "f32",
        "config": {"workload": "Hybrid: 10M docs BM25 (12 terms/query) + 10M x 768 vector, min-max merge (BASELINE configs[3])",
                   "docs": n, "dim": dim, "k": k, "tokens_per_query": T, "posting_lists": int(len(ranks)),
                   "postings_resident": int(n_post), "avg_postings_per_query": avg_postings,
                   "avg_docs_touched_per_query": avg_docs, "valid": n == 10_000_000 and dim == 768},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "vec_scan_f32_kernel",
                     "alg_bytes_per_launch": alg_vec, "avg_launch_ms": avg_scan_s * 1e3, "launches": scan_n,
                     "clocks_during_timed_region": clocks.summary(),
                     "note": "a hybrid query = the fp32 scan (dominant) + the full-text leg on a second stream"},
        "full_text_leg": ("range scorer (K3r) beside the scan + candidate tail after it" if k3_launches == 0 else
                          f"per-record scorer (K3) used by {k3_launches} launches"),
        "shadow_store": shadow_out,
        "bm25_only": {"value": len(batch_q) / el_bb, "unit": "queries/s",
                      "runs": {"n": len(runs_bb), "best": len(batch_q) / min(runs_bb), "worst": len(batch_q) / max(runs_bb), "statistic": "median"},
                      "note": "one orama_post_search_batch call over %d queries, descriptors built beforehand: K3r scores 32 queries "
                              "per set of launches, two sets in flight" % len(batch_q),
                      "through_python_wrapper": {"value": len(batch_q) / el_bb_py, "unit": "queries/s",
                                                 "note": "search_batch(): + ctypes marshalling of every query and result"},
                      "single_query_calls": {"value": steps / el_b, "unit": "queries/s", "ms_per_query": el_b / steps * 1e3},
                      "roofline": {"bound": "hbm", "achieved": bm25_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": bm25_achieved / HBM_PEAK_GBS, "traffic": None,
                                   "kernel": "range_bounds + range_score + keys_topk (K3r)",
                                   "alg_bytes_per_query": bm25_alg_bytes,
                                   "alg_bytes_definition": "SURVEY §8(d): 8 B per posting of the query's lists + 4 B per "
                                                           "distinct document touched",
                                   "device_us_per_query": dev_us,
                                   "device_us_by_kernel": {"range_bounds": kb_ms * 1e3 / n_dev_q,
                                                           "range_df": kd_ms * 1e3 / n_dev_q,
                                                           "range_score": ks_ms * 1e3 / n_dev_q,
                                                           "topk_select": kt_ms * 1e3 / n_dev_q},
                                   "note": "round 5: compact key lists — the scoring launch (2.5 VALU wave instructions per posting, bound by "
                                           "instruction issue and latency, not by HBM) appends only the keys at or above a floor, the top-k "
                                           "reads survivors only: chain traffic 0.92 x the algorithmic bytes (round 4: 2.2 x) — DESIGN K3r, "
                                           "profiles/r05_k3r_compact_ab_final.log, r05_pmc_k3r_*.json.  Since the second half of round 5 the "
                                           "top-k's final launch also DELIVERS the answers (ids, scores, counts, the queries' result words) to "
                                           "the pinned host block: topk_select includes that write over PCIe (~0.3 us per query), which until "
                                           "then was a launch of its own outside every bracket — the chain's end-to-end time fell, this figure "
                                           "rose (profiles/r05_k3r_direct_out_ab.log)"},
                      "cpu_baseline": cpu_bm25},
        "postings_fill_seconds": t_fill,
        "parity_check": "bit-exact vs oracle (last query): BM25 ids/scores/count, hybrid ids/scores/count",
    }


# ---------------------------------------------------------------------------------------------- live HBM traffic
def being_profiled() -> bool:
    return any("rocprof" in os.environ.get(v, "").lower() for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))


def pmc_child(args) -> None:
    """What the PMC passes run: the workload's store and a few scans — nothing else."""
    import oramacore_amd as oa
    from oramacore_amd.shard_group import ShardGroup

    n_total, dim, k, qb, dtype, _ = WORKLOADS[args.workload]
    n_total = args.rows or n_total
    group = ShardGroup([0])
    store = oa.EmbeddingFieldStorage(group.ctx(0), dimensions=dim, reserve_rows=n_total,
                                     dtype=oa.DTYPE_F16 if dtype == "f16" else oa.DTYPE_F32)
    store.fill_synthetic(n_total, seed=0xC0FFEE, first_doc_id=0)
    qs = np.random.default_rng(0xBEEF).standard_normal((4 * qb, dim)).astype(np.float32)
    sess = group.session([store], qs, qb, k, n_slots=1)
    for i in range(4):
        sess.step(i)
    sess.sync()
    sess.close()
    store.close()
    group.close()


def measure_traffic(args, kernel_substr: str, alg_bytes: float) -> dict | None:
    """HBM bytes per launch of the scan kernel from the PMC counters, measured in THIS run as MI355X_MICROARCH.md's
    HBM section prescribes: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (kernel trace only),
    values in KiB, FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced streaming reads)."""
    exe = shutil.which("rocprofv3")
    if args.no_pmc or being_profiled() or not exe:
        return None
    med = {}
    clock = None
    with tempfile.TemporaryDirectory(prefix="orama_pmc_") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE") + (("GRBM_GUI_ACTIVE",) if args.details else ()):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, str(ROOT / "bench.py"), "--pmc-child", "--workload", args.workload]
            if args.rows:
                cmd += ["--rows", str(args.rows)]
            try:
                r = subprocess.run(cmd, cwd=tmp, env=dict(os.environ, TMPDIR=tmp), capture_output=True, text=True, timeout=300)
            except subprocess.TimeoutExpired:
                return None
            vals, per_ns = [], []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path, newline="") as f:
                    for row in csv.DictReader(f):
                        if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            vals.append(float(row["Counter_Value"]))
                            try:
                                ns = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                                if ns > 0:
                                    per_ns.append(float(row["Counter_Value"]) / ns)
                            except (KeyError, ValueError):
                                pass
            if counter == "GRBM_GUI_ACTIVE":  # optional: the traffic stands without it
                if r.returncode == 0 and per_ns:
                    # the counter is summed over the 8 XCDs; busy cycles / wall = the clock the kernel ran at (MI355X_MICROARCH.md,
                    # "DVFS give-back") — of the PROFILED pass, which runs 2-4 % slower than the un-profiled timed region
                    clock = {"effective_gfxclk_mhz": statistics.median(per_ns) / 8 * 1e3, "launches": len(per_ns),
                             "source": "rocprofv3 --pmc GRBM_GUI_ACTIVE pass of `bench.py --pmc-child`: counter / 8 XCDs / kernel wall "
                                       "time, median over the launches (a profiled pass)"}
                continue
            if r.returncode != 0 or not vals:
                return None
            med[counter] = statistics.median(vals)
    read_b, write_b = med["FETCH_SIZE"] * 1024 * 2, med["WRITE_SIZE"] * 1024
    return {"traffic": read_b + write_b, "traffic_unit": "bytes/launch",
            "traffic_source": "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate, kernel "
                              "trace only) of `bench.py --pmc-child`; KiB units, FETCH_SIZE x2 per MI355X_MICROARCH.md",
            "traffic_over_algorithmic": (read_b + write_b) / alg_bytes,
            "effective_clock_profiled_pass": clock}


def committed_traffic(workload: str) -> dict | None:
    pmc_files = sorted(list((ROOT / "profiles").glob(f"r*_pmc_{workload}_vec_scan.json")) +
                       list((ROOT / "profiles").glob(f"r*_pmc_{workload}_shard_vec_scan.json")), key=lambda f: f.name[:3])
    if not pmc_files:
        return None
    rec = json.loads(pmc_files[-1].read_text())
    if "traffic_bytes_per_step" in rec:  # round 5's records: total over a fixed number of steps (scripts/pmc_total.py)
        return {"traffic": rec["traffic_bytes_per_step"], "traffic_unit": "bytes/step (one corpus pass; the fp16 scans split it into several launches)",
                "traffic_over_algorithmic": rec.get("traffic_over_algorithmic"),
                "traffic_source": f"NOT measured in this run — profiles/{pmc_files[-1].name} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                  "passes of scripts/pmc_scan_probe.py, total over 6 steps; FETCH_SIZE x2 per MI355X_MICROARCH.md)"}
    return {"traffic": rec.get("traffic_bytes_per_launch"), "traffic_unit": "bytes/launch",
            "traffic_source": f"NOT measured in this run — profiles/{pmc_files[-1].name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                              "passes of this command; FETCH_SIZE x2 per MI355X_MICROARCH.md)"}


