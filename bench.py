#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): queries/sec of the single-query cosine top-100 scan over a 10 M x 768 fp32
corpus (the north-star shape, 30.72 GB — fits one GPU), at 1/2/4/8 GPUs; HBM GB/s of the scan
kernel against the gfx950 peak; the CPU restatement of the reference timed beside it.

A "step" is one query: one pass of K1 (distance scan) over the rank's shard + K4 (top-100) + — for
N > 1 — one RCCL all-gather of the per-shard candidates + K6 (merge).  Corpus and queries are
resident in HBM before the timed region; results stay in HBM (the host-buffer API, which adds the
PCIe hop, is timed separately and reported as `latency_ms_p50_host_api`).

No torch in this process: buffers, streams and the RCCL exchange live inside liborama_hip.so
(orama_shard_*); under torch.distributed.run the ranks only read RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* from the environment and pass the 128-byte communicator id over a localhost socket.

Strong scaling: the 10 M rows are split statically over the N ranks, so queries/sec
should grow ~linearly with N.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ns|c2|c3] [--rows R]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)

WORKLOADS = {
    # name: (rows, dim, k, queries per step, storage dtype, description)
    "ns": (10_000_000, 768, 100, 1, "f32", "10M x 768 fp32 embeddings, single-query cosine top-100 (north-star)"),
    "c2": (1_000_000, 384, 100, 1, "f32", "1M x 384 fp32 embeddings, single-query cosine top-100 (BASELINE configs[1])"),
    "c3": (10_000_000, 768, 100, 64, "f16", "10M x 768 fp16 embeddings, batch-64 queries, MFMA scan + top-100 "
                                            "(BASELINE configs[2])"),
    "c5": (80_000_000, 768, 100, 256, "f16", "80M x 768 fp16 embeddings sharded over the ranks, batch-256 queries, "
                                             "MFMA scan + top-100 + RCCL all-gather (BASELINE configs[4])"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="ns")
    ap.add_argument("--rows", type=int, default=0, help="override the corpus size (debug only; marks the run invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the RCCL all-gather + merge even with one rank (plumbing test)")
    ap.add_argument("--cpu-sample-rows", type=int, default=200_000)
    ap.add_argument("--no-two-stage", action="store_true", help="skip the fp32 + fp16-shadow leg (N = 1, fp32 workloads)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are issued on round-robin (independent queries: the tiny top-k / "
                         "all-gather / merge kernels of step i overlap the corpus scan of step i+1)")
    return ap.parse_args()


def cpu_baseline(store, dim: int, n_total: int, k: int, sample_rows: int) -> dict:
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
    rows_per_s_1 = run(1, 8.0)
    rows_per_s_all = run(cores, 6.0)
    return {
        "value": rows_per_s_1 / n_total,
        "unit": "queries/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle orc_distances_f32 (scalar, sequential f32) + top-{k} over the first {sample_rows} rows "
                  f"of the same corpus, repeated >= 8 s; QPS = rows/s / {n_total}",
        "gbytes_per_s": rows_per_s_1 * dim * 4 / 1e9,
        "all_cores": {"value": rows_per_s_all / n_total, "cores": cores,
                      "gbytes_per_s": rows_per_s_all * dim * 4 / 1e9},
    }


def vec_two_stage_ok(dim: int, k: int) -> bool:
    return dim % 4 == 0 and dim <= 1024 and 2 * k <= 4096


def two_stage_leg(oa, ctx, plain, dim, n_local, k, qb, queries_h, lo, rank) -> dict:
    """Host-buffer API, one call per query batch: plain fp32 store vs fp32 rows + fp16 shadow (DTYPE_F32_SHADOW16)."""
    st = oa.EmbeddingFieldStorage(ctx, dimensions=dim, reserve_rows=n_local, dtype=oa.DTYPE_F32_SHADOW16)
    st.fill_synthetic(n_local, seed=0xC0FFEE + rank, first_doc_id=lo)
    nq = min(40, queries_h.shape[0] // qb)
    identical = True
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
            "hbm_bytes": int(info["hbm_bytes"]),
            "note": "fp32 rows + fp16 shadow (ORAMA_DTYPE_F32_SHADOW16): the fp16 scan proposes max(2k, k+256) candidates, "
                    "K1's arithmetic on the fp32 rows decides; a query whose candidate list cannot be proven complete "
                    "falls back to the fp32 scan; results bit-identical to the plain store (DESIGN.md K1s)"}


def main():
    args = parse_args()
    import oramacore_amd as oa
    from oramacore_amd.launch import RankEnv, ShardPlan, device_for, exchange_unique_id, self_launch
    from oramacore_amd.shard_group import FORCE_RCCL, ShardGroup

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher — N ranks of this script, one per GPU (the same process
        # model as under torch.distributed.run; rank 0 prints the JSON line)
        raise SystemExit(self_launch(args.gpus, [str(Path(__file__).resolve()), *sys.argv[1:]]))
    env = RankEnv.from_env()
    world, rank, local_rank = env.world, env.rank, env.local_rank
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # The exchange (RCCL all-gather of the per-shard candidates + K6) runs inside liborama_hip.so: one process per
    # GPU, rank 0 makes the communicator id and hands it to the others over a localhost socket.  No torch here.
    if world > 1:
        uid = exchange_unique_id(env, ShardGroup.unique_id)
        group = ShardGroup.from_rank(uid, rank, world, device_for(local_rank))
    else:
        group = ShardGroup([local_rank], flags=FORCE_RCCL if args.force_exchange else 0)
    ctx = group.ctx(0)

    n_total, dim, k, qb, dtype, desc = WORKLOADS[args.workload]
    if args.rows:
        n_total = args.rows
    plan = ShardPlan(n_total, world)
    lo, hi = plan.range(rank)
    n_local = hi - lo
    f16 = dtype == "f16"

    store = oa.EmbeddingFieldStorage(ctx, dimensions=dim, reserve_rows=n_local,
                                     dtype=oa.DTYPE_F16 if f16 else oa.DTYPE_F32)
    t_fill = time.perf_counter()
    store.fill_synthetic(n_local, seed=0xC0FFEE + rank, first_doc_id=lo)
    t_fill = time.perf_counter() - t_fill

    total_b = args.warmup + args.steps  # batches
    rng = np.random.default_rng(0xBEEF)
    queries_h = rng.standard_normal((total_b * qb, dim)).astype(np.float32)
    # Stream plan (inside the session): ONE scan stream (corpus scans of consecutive steps run back to back, never
    # concurrently, so each keeps the whole HBM bandwidth) + `--streams` high-priority tail streams used round-robin
    # for top-k / all-gather / merge, which are launch-bound and overlap the next step's scan.
    # fp16 workloads: the scan and its threshold-filter selections depend on each other step by step and all run on the
    # tail stream, so a second slot would only make two corpus scans share the HBM bandwidth (and inflate the per-launch
    # durations the roofline is computed from) — one slot.
    n_streams = 1 if f16 else max(1, args.streams)
    sess = group.session([store], queries_h, qb, k, n_slots=n_streams, force_exchange=args.force_exchange)

    def barrier():
        sess.sync()
        group.barrier()  # local devices drained + one all-reduced word over RCCL + drained again

    for i in range(args.warmup):
        sess.step(i)
    barrier()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_b):
        sess.step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    elapsed = group.allreduce_max(elapsed)  # the slowest rank's time

    # ---- sanity of the last step (outside the timed region): sorted, exact on re-computation, nothing missed
    ids_all, dst_all, cnt = sess.result((total_b - 1) % n_streams)
    assert np.all(cnt == k), "bench result incomplete"
    from oracle import oracle as orc  # checker only

    for qi in sorted({0, qb - 1}):
        ids_h, dst_h = ids_all[qi], dst_all[qi]
        assert np.all(np.diff(dst_h) >= 0), "bench result not sorted"
        mine = (ids_h >= lo) & (ids_h < hi)
        if mine.any():
            rows, _ = store.get_rows((ids_h[mine] - np.uint64(lo)).astype(np.uint64))
            qv = queries_h[(total_b - 1) * qb + qi]
            if f16:  # the fp16 path scores the fp16-rounded query against the stored fp16 rows
                qv = qv.astype(np.float16).astype(np.float32)
            od = orc.distances(rows, qv)
            err = float(np.max(np.abs(od - dst_h[mine])))
            assert err <= 1e-4, f"bench parity check failed: {err}"
        # "no better row was missed": no row of a random sample of this rank's shard beats the reported k-th
        # distance unless it is in the result (size-independent property, SURVEY §8c / tests/test_full_size_gpu.py)
        srng = np.random.default_rng(1234 + qi)
        sample = srng.choice(n_local, size=min(20_000, n_local), replace=False).astype(np.uint64)
        srows, sdocs = store.get_rows(sample)
        qv = queries_h[(total_b - 1) * qb + qi]
        if f16:
            qv = qv.astype(np.float16).astype(np.float32)
        sd = orc.distances(srows, qv, threads=8)
        inside = set(ids_h.tolist())
        missed = [int(dd) for dd, x in zip(sdocs.tolist(), sd.tolist()) if x < dst_h[-1] - 2e-4 and int(dd) not in inside]
        assert not missed, f"bench parity check failed: rows {missed[:5]} beat the reported k-th distance"

    kern = "vec_scan_f16" if f16 else "vec_scan_f32"
    scan_ms, scan_n = ctx.prof_get(kern)
    sel_ms, sel_n = ctx.prof_get("topk_select")
    kpad = (dim + 127) // 128 * 128
    bytes_per_step = n_local * (kpad * 2 if f16 else dim * 4)  # one corpus pass per step
    launches_per_step = max(scan_n, 1) / args.steps
    alg_bytes = bytes_per_step / launches_per_step
    avg_scan_s = scan_ms / max(scan_n, 1) / 1e3
    achieved = alg_bytes / avg_scan_s / 1e9 if scan_n else 0.0

    traffic, traffic_src = None, None
    pmc_files = sorted((ROOT / "profiles").glob(f"r*_pmc_{args.workload}_vec_scan.json"))
    if pmc_files and not args.rows and world == 1:
        rec = json.loads(pmc_files[-1].read_text())
        traffic = rec.get("traffic_bytes_per_launch")
        traffic_src = f"profiles/{pmc_files[-1].name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; " \
                      "FETCH_SIZE x2 per MI355X_MICROARCH.md)"

    out = {
        "metric": f"queries/sec, cosine top-{k} scan ({n_total // 1_000_000}M x {dim} {dtype}) — HBM GB/s vs peak in "
                  "`roofline`",
        "value": args.steps * qb / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": desc, "rows_total": n_total, "rows_per_gpu": n_local, "dim": dim, "k": k,
                   "queries_per_step": qb, "parallelism": f"row-shard x{world} + all-gather(top-k) over RCCL (inside "
                   "liborama_hip.so, one process per GPU)" if world > 1 else "single GPU", "streams": n_streams,
                   "valid": not bool(args.rows)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                     "kernel": ("vec_scan_f16_pc_kernel" if f16 and qb > 64 else kern + "_kernel"),
                     "alg_bytes_per_launch": alg_bytes,
                     "avg_launch_ms": avg_scan_s * 1e3, "launches": scan_n,
                     "scan_launches_per_step": launches_per_step,
                     "topk_select_ms_per_step": sel_ms / args.steps},
        "fill_seconds": t_fill,
    }
    if f16:
        flops = 2.0 * qb * n_local * kpad * args.steps
        out["roofline"]["mfma_tflops"] = flops / (scan_ms / 1e3) / 1e12 if scan_ms else 0.0
        out["roofline"]["mfma_peak_tflops_dense_f16"] = 2500.0
        out["roofline"]["mfma_frac"] = out["roofline"]["mfma_tflops"] / 2500.0
        if qb > 64:
            out["roofline"]["note"] = ("K2d (producer/consumer GEMM tiles, 256 queries per pass): the corpus crosses HBM once "
                                       "per batch; with the matrix pipes and the HBM stream both active the part sits at "
                                       "its package-power limit (DESIGN.md K2d, profiles/r02_power_probe.log)")

    if rank == 0 and world == 1:
        # host-buffer API latency (adds the PCIe hop for the query and the k results)
        lat = []
        for i in range(min(50, total_b)):
            t1 = time.perf_counter()
            store.storage_search(queries_h[i * qb:(i + 1) * qb], k)
            lat.append((time.perf_counter() - t1) * 1e3)
        out["latency_ms_p50_host_api"] = float(np.percentile(lat, 50))
        out["latency_ms_p95_host_api"] = float(np.percentile(lat, 95))
        if not args.no_cpu_baseline and not f16:
            out["cpu_baseline"] = cpu_baseline(store, dim, n_total, k, args.cpu_sample_rows)
        if not f16 and not args.no_two_stage and vec_two_stage_ok(dim, k):
            # NOT part of `value`: the same corpus in a store that also keeps an fp16 copy of its rows (+50 % HBM).  The
            # fp16 scan proposes candidates, the fp32 rows decide; the answers are compared with the plain store's here.
            out["two_stage_exact"] = two_stage_leg(oa, ctx, store, dim, n_local, k, qb, queries_h, lo, rank)
    device_name = ctx.device_info()["name"]
    sess.close()
    store.close()
    group.barrier()
    group.close()
    if rank == 0:
        out["device"] = device_name
        # RCCL writes its version banner through C stdio (fully buffered when stdout is a pipe): push it out first
        # so that the JSON line is the LAST line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
