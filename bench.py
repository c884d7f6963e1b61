#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): queries/sec of the single-query cosine top-100 scan over a 10 M x 768 fp32
corpus (the north-star shape, 30.72 GB — fits one GPU), at 1/2/4/8 GPUs; HBM GB/s of the scan
kernel against the gfx950 peak; the CPU restatement of the reference timed beside it.

A "step" is one query batch: one pass of the distance scan over the rank's shard + K4 (top-100) + — for
N > 1 — one RCCL all-gather of the per-shard candidates + K6 (merge).  Corpus and queries are
resident in HBM before the timed region; results stay in HBM (the host-buffer API, which adds the
PCIe hop, is timed separately and reported as `latency_ms_p50_host_api`).

`value` is the north-star workload.  At N = 1 the same run then times the other BASELINE configurations and
reports them as `configs: {c4, c2, c3, c5_shard}` — each with its own ms_per_step, roofline (HIP events on the
launching stream + SURVEY §8(d) algorithmic bytes) and post-run oracle check (`--configs none` skips them).

No torch in this process: buffers, streams and the RCCL exchange live inside liborama_hip.so
(orama_shard_*).  N > 1 is one process per GPU: under torch.distributed.run the ranks read RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment; a plain `python bench.py --gpus N` starts the N ranks itself
(oramacore_amd/launch.py).  The 128-byte communicator id travels over a localhost socket.

Strong scaling: the 10 M rows are split statically over the N ranks, so queries/sec
should grow ~linearly with N.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ns|c2|c3|c5] [--rows R] [--configs ...]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from bench_legs import (EXTRA_CONFIGS, HBM_PEAK_GBS, MEDIAN_MIN_LAUNCHES, MFMA_F16_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS, WORKLOADS,  # noqa: E402,F401
                        ClockSampler, GsSummary, being_profiled, check_vector_result, committed_traffic, cpu_baseline, host_api_latency,
                        hybrid_leg, measure_traffic, pctl, pmc_child, sampler_lib, scan_step_samples, shadow_store, two_stage_leg,
                        two_stage_session, vec_two_stage_ok, vector_leg)


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
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000,
                    help="rows of the corpus the CPU baseline legs scan (3 GB at 768 dimensions: beyond the host's caches, like the "
                         "30 GB corpus the QPS is extrapolated to)")
    ap.add_argument("--no-two-stage", action="store_true", help="skip the fp32 + fp16-shadow leg (N = 1, fp32 workloads)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are issued on round-robin (independent queries: the tiny top-k / "
                         "all-gather / merge kernels of step i overlap the corpus scan of step i+1)")
    ap.add_argument("--configs", default="all",
                    help="extra BASELINE configurations timed after the north-star leg at N = 1: 'all', 'none' or a comma "
                         "list of " + ",".join(EXTRA_CONFIGS))
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic live (two short rocprofv3 --pmc passes of `--pmc-child`)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dump-result", default="",
                    help="every rank writes the last step's queries and answer (ids, distances, counts) to "
                         "<path>.rank<r>.npz after the timed region (multi-rank parity tests compare the ranks)")
    ap.add_argument("--f16-slots", type=int, default=1,
                    help="fp16 workloads: steps in flight (each slot is one stream carrying its scans AND selections; the scans of two "
                         "slots cannot overlap — every scan kernel fills all CUs — but the launch-bound selections of one slot run beside "
                         "the other slot's work: +2.4 % at 256 queries per step and +4.1 % at 64 on its own, nothing inside the full bench "
                         "run — profiles/r04_f16_slots_experiment.log; default 1)")
    ap.add_argument("--no-preflight", action="store_true", help="self-launched N > 1: skip the pre-launch check of devices and RCCL")
    ap.add_argument("--details", action="store_true",
                    help="the long form: the CPU thread-scaling ladder with NUMA-placed rows and the GRBM_GUI_ACTIVE profiler pass "
                         "(adds ~15 s; the default run keeps the one-thread oracle, the all-core oracle and a 4-point fast-CPU ladder)")
    ap.add_argument("--details-file", default=str(ROOT / "bench_details.json"),
                    help="where the full record goes (the LAST stdout line is only the compact metric line, <= 4 KB)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------- the metric line
HEADLINE_BUDGET = 4096  # bytes: the driver keeps a bounded tail of stdout (round 5's 24.9 KB line did not parse)


def _r(x, nd=4):
    """Round floats for the metric line (6 significant figures at most); pass everything else through."""
    if isinstance(x, float):
        return float(f"{x:.6g}") if nd is None else round(x, nd)
    return x


def _leg_summary(leg: dict) -> dict:
    """The five numbers of one extra configuration."""
    rf = leg.get("roofline") or {}
    r = {"frac": _r(rf.get("frac")), "kernel": rf.get("kernel")}
    if rf.get("mfma_frac") is not None:
        r["mfma_frac"] = _r(rf["mfma_frac"])
    return {"value": _r(leg.get("value"), None), "ms_per_step": _r(leg.get("ms_per_step")),
            "latency_ms_p50": _r(leg.get("latency_ms_p50")), "roofline": r}


def headline_line(out: dict, details_file: str | None = None) -> dict:
    """The compact metric line (the LAST line of stdout): the contract's keys, the NS roofline and the CPU baseline as
    numbers, five numbers per extra configuration.  Everything else — sampler records, notes, traffic sources, the CPU
    thread ladder, the two-stage leg — is in the details file (and on stderr)."""
    cfg = out.get("config", {})
    rf = out.get("roofline", {})
    line = {k: _r(out.get(k), None) if k == "value" else _r(out.get(k)) for k in (
        "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "latency_ms_p50", "latency_ms_p95", "latency_samples",
        "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg.get(k) for k in ("workload", "rows_total", "rows_per_gpu", "dim", "k", "queries_per_step",
                                              "parallelism", "valid", "exchange", "comm_world") if k in cfg}
    if isinstance(cfg.get("ranks_seen"), list):
        line["config"]["ranks_seen"] = len(cfg["ranks_seen"])
    line["roofline"] = {k: _r(rf.get(k), None) if k in ("achieved", "traffic") else _r(rf.get(k)) for k in (
        "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "median_scan_ms_per_step")}
    if rf.get("traffic_over_algorithmic") is not None:
        line["roofline"]["traffic_over_algorithmic"] = _r(rf["traffic_over_algorithmic"])
    if rf.get("mfma_frac") is not None:
        line["roofline"]["mfma_frac"] = _r(rf["mfma_frac"])
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"], None), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "gbytes_per_s": _r(cb.get("gbytes_per_s"), 2),
                                "sample": f"oracle scalar cosine + top-k over {cb.get('sample_rows', '1e6')} rows of the corpus, "
                                          "QPS scaled to the full corpus"}
        if cb.get("fast"):
            line["cpu_baseline"]["fast"] = {"value": _r(cb["fast"]["value"], None), "cores": cb["fast"]["cores"],
                                            "gbytes_per_s": _r(cb["fast"].get("gbytes_per_s"), 2)}
    legs = {name: _leg_summary(leg) for name, leg in (out.get("configs") or {}).items()}
    c4 = (out.get("configs") or {}).get("c4") or {}
    if c4.get("bm25_only"):
        b = c4["bm25_only"]
        legs["bm25_batch"] = {"value": _r(b.get("value"), None), "unit": b.get("unit"),
                              "roofline": {"frac": _r((b.get("roofline") or {}).get("frac")),
                                           "kernel": (b.get("roofline") or {}).get("kernel")},
                              "us_per_query_device": _r((b.get("roofline") or {}).get("device_us_per_query"), 3),
                              "single_calls_per_s": _r((b.get("single_query_calls") or {}).get("value"), None)}
    if c4.get("shadow_store"):
        legs["c4_shadow"] = {"value": _r(c4["shadow_store"].get("value"), None),
                             "latency_ms_p50": _r(c4["shadow_store"].get("latency_ms_p50"))}
    ts = out.get("two_stage_exact")
    if ts:
        legs["two_stage_exact"] = {"value": _r(ts.get("value"), None), "identical_to_fp32_scan": ts.get("identical_to_fp32_scan")}
    if legs:
        line["configs"] = legs
    sb = out.get("step_breakdown_us")
    if sb:
        line["step_breakdown_us"] = {k: _r(sb.get(k), 1) for k in ("scan", "select", "all_gather", "merge_k6")}
    line["parity_check"] = "ok" if out.get("parity_check") else None
    if "error" in out:
        line["error"] = str(out["error"])[:300]
    if details_file:
        line["details"] = os.path.basename(details_file)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= HEADLINE_BUDGET:
        # shed the optional parts, largest first, until the line fits: the contract keys never go
        for victim in ("step_breakdown_us", "configs"):
            line.pop(victim, None)
            if len(json.dumps(line, separators=(",", ":"))) < HEADLINE_BUDGET:
                break
    assert len(json.dumps(line, separators=(",", ":"))) < HEADLINE_BUDGET, "bench.py: the metric line outgrew its budget"
    return line


def emit(out: dict, details_file: str) -> None:
    """Full record -> the details file and stderr; compact line -> the LAST line of stdout."""
    try:
        Path(details_file).write_text(json.dumps(out, indent=1) + "\n")
    except OSError as e:  # a read-only checkout must not cost the metric line
        print(f"bench.py: could not write {details_file}: {e}", file=sys.stderr)
        details_file = None
    print(json.dumps(out), file=sys.stderr, flush=True)
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    print(json.dumps(headline_line(out, details_file), separators=(",", ":")), flush=True)


# ---------------------------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args)
    import oramacore_amd as oa
    from oramacore_amd.launch import RankEnv, ShardPlan, device_for, exchange_unique_id, preflight, self_launch
    from oramacore_amd.shard_group import FORCE_RCCL, ShardGroup

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher — N ranks of this script, one per GPU (the same process
        # model as under torch.distributed.run; rank 0 prints the JSON line).  Checked first, in a throw-away child: enough
        # devices (or a loopback transport), RCCL resolvable — a launch that cannot work says why in ONE JSON line instead
        # of N tracebacks.
        if not args.no_preflight:
            pf = preflight(args.gpus)
            if not pf["ok"]:
                print(json.dumps({"error": "preflight: " + pf["error"], "n_gpus": args.gpus, "preflight": pf}), flush=True)
                raise SystemExit(2)
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
        device = device_for(local_rank)
        group = ShardGroup.from_rank(uid, rank, world, device)
    else:
        device = local_rank
        group = ShardGroup([local_rank], flags=FORCE_RCCL if args.force_exchange else 0)
    ctx = group.ctx(0)
    # RCCL writes its version banner through C stdio (fully buffered when stdout is a pipe) when the communicator is made:
    # every rank pushes it out NOW, so that nothing of it can land behind rank 0's JSON line when the ranks exit
    import ctypes
    ctypes.CDLL(None).fflush(None)
    # who is in the job, as the communicator itself reports it: one all-reduce(max) per rank slot carries that rank's device
    # ordinal and pid to everybody (a rank that is missing, or two ranks that believe they are the same one, show up here
    # — and in `n_gpus` — instead of as a plausible-looking number)
    ranks_seen = []
    for r in range(world):
        dev = group.allreduce_max(float(device + 1) if r == rank else 0.0)
        pid = group.allreduce_max(float(os.getpid()) if r == rank else 0.0)
        ranks_seen.append({"rank": r, "device": int(dev) - 1, "pid": int(pid)})
    if group.world != world or any(e["device"] < 0 for e in ranks_seen):
        raise SystemExit(f"bench.py: communicator reports world {group.world}, ranks {ranks_seen} — expected {world} ranks")

    try:
        bdf = ctx.pci_bus_id()
    except Exception:  # noqa: BLE001 — the sampler is optional
        bdf = None
    n_total, dim, k, qb, dtype, desc = WORKLOADS[args.workload]
    if args.rows:
        n_total = args.rows
    lo, hi = ShardPlan(n_total, world).range(rank)
    f16 = dtype == "f16"
    out, store, queries_h = vector_leg(oa, group, args.workload, n_total, args.steps, args.warmup, args.streams,
                                       force_exchange=args.force_exchange, rank=rank, world=world, lo=lo, hi=hi,
                                       valid=not bool(args.rows), dump=args.dump_result, f16_slots=args.f16_slots, bdf=bdf)
    out["config"]["ranks_seen"] = ranks_seen
    out["config"]["comm_world"] = group.world
    out["config"]["exchange"] = ("rccl" + (" (ORAMA_RCCL_LIB loopback: " + os.path.basename(os.environ["ORAMA_RCCL_LIB"]) + ")"
                                           if os.environ.get("ORAMA_RCCL_LIB") else "")) if group.uses_rccl else "none (one shard)"
    if f16 and qb > 64:
        out["roofline"]["note"] = ("K2d (producer/consumer GEMM tiles, 256 queries per pass): the corpus crosses HBM once "
                                   "per batch (DESIGN.md K2d)")

    solo = rank == 0 and world == 1
    if solo:
        out.update(host_api_latency(store, queries_h, qb, k))  # adds the PCIe hop for the query and the k results
        if not args.no_cpu_baseline and not f16:
            out["cpu_baseline"] = cpu_baseline(store, dim, n_total, k, args.cpu_sample_rows, details=args.details)
        want = EXTRA_CONFIGS if args.configs == "all" else () if args.configs == "none" else tuple(
            c for c in args.configs.split(",") if c)
        configs = {}
        # NOT part of `value`: the same corpus in a store that also keeps an fp16 copy of its rows (+50 % HBM).  The
        # fp16 scan proposes candidates, the fp32 rows decide; the answers are compared with the plain store's.
        shadow = (shadow_store(oa, ctx, dim, hi - lo, lo, rank)
                  if not f16 and not args.no_two_stage and vec_two_stage_ok(dim, k) else None)
        if args.workload == "ns" and not args.rows:
            if "c4" in want:  # shares the north-star rows
                configs["c4"] = hybrid_leg(oa, ctx, store, n_total, dim, k, steps=max(10, min(args.steps, 40)), warmup=3,
                                           shadow=shadow, bdf=bdf)
        if shadow is not None:
            out["two_stage_exact"] = two_stage_leg(oa, ctx, store, shadow, dim, k, qb, queries_h, group=group)
        if args.workload == "ns" and not args.rows and "f32_batch" in want:
            # the SAME plain fp32 store asked 64 queries at a time (what the request batcher does under load)
            configs["f32_batch"], _, qhb = vector_leg(oa, group, "nsb", n_total, 20, 3, 1, store=store, bdf=bdf, latency_steps=20)
            configs["f32_batch"].update(host_api_latency(store, qhb, 64, k, n=10))
        store.close()
        store = None
        if args.workload == "ns" and not args.rows:
            if "c2" in want:
                configs["c2"], st, qh2 = vector_leg(oa, group, "c2", WORKLOADS["c2"][0], 200, 10, args.streams, bdf=bdf)
                configs["c2"].update(host_api_latency(st, qh2, 1, k))
                st.close()
            st16 = None
            if "c3" in want:
                configs["c3"], st16, qh = vector_leg(oa, group, "c3", WORKLOADS["c3"][0], 30, 3, 1, f16_slots=args.f16_slots, bdf=bdf,
                                                         latency_steps=30)
                configs["c3"].update(host_api_latency(st16, qh, 64, k, n=10))
            if "c5_shard" in want:
                # the per-GPU shard of BASELINE configs[4]: 10 M of the 80 M x 768 fp16 rows, all 256 queries of a batch
                # (the rows are the ones C3 scans: the same store serves both legs)
                configs["c5_shard"], st16, _ = vector_leg(
                    oa, group, "c5", 10_000_000, 20, 3, 1, store=st16, f16_slots=args.f16_slots, bdf=bdf, latency_steps=30,
                    desc="per-GPU shard (10M rows) of: " + WORKLOADS["c5"][5])
                configs["c5_shard"]["config"]["note"] = ("one of the eight 10 M-row shards of configs[4] on one GPU; the 8-GPU "
                                                         "job adds one 307 KB all-gather + merge per batch")
            if st16 is not None:
                st16.close()
        for cname, wname in (("c2", "c2"), ("c3", "c3"), ("c5_shard", "c5_shard")):
            # (only NS's traffic is measured inside this run; the other legs point at their committed PMC record)
            if cname in configs and configs[cname]["roofline"].get("traffic") is None:
                t = committed_traffic(wname)
                if t:
                    configs[cname]["roofline"]["traffic_reference"] = t
        if configs:
            out["configs"] = configs
    device_name = ctx.device_info()["name"]
    if store is not None:
        store.close()
    group.barrier()
    group.close()
    if solo:
        # HBM traffic of the dominant kernel from the PMC counters — measured now (three short profiler passes of this
        # script's --pmc-child mode, after this process has released the GPU), else the committed record, labelled
        kern = "vec_scan_f16" if f16 else "vec_scan_f32"
        t = measure_traffic(args, kern, out["roofline"]["alg_bytes_per_launch"]) or (
            None if args.rows else committed_traffic(args.workload))
        if t:
            out["roofline"].update(t)
    elif rank == 0 and not args.rows:
        # N > 1: nothing is profiled inside a multi-rank job; the committed single-GPU record of the same kernel, labelled as such
        # (per launch of the FULL corpus: a rank's launch moves rows_per_gpu / rows_total of it)
        t = committed_traffic(args.workload)
        if t:
            t["traffic_source"] += " — a launch over the whole single-GPU corpus; this job's ranks scan 1/N of it each"
            out["roofline"]["traffic_reference"] = t
    if rank == 0:
        out["device"] = device_name
        # RCCL writes its version banner through C stdio (fully buffered when stdout is a pipe): emit() pushes it out first
        # so that the compact JSON line is the LAST line of stdout
        emit(out, args.details_file)


if __name__ == "__main__":
    main()
