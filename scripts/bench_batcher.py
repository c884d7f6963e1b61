#!/usr/bin/env python3
"""Serving-shaped bench of the micro-batcher: T client threads each issue single-query
cosine top-k searches (the reference API's shape — one target per call) against one store; with the batcher the
concurrent requests share corpus passes (K2 at Q <= 64 per pass for fp16 stores).

    python scripts/bench_batcher.py [--rows 10000000] [--dim 768] [--dtype f16] [--threads 128] [--per-thread 20]
"""
import argparse
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import oramacore_amd as oa  # noqa: E402
from oramacore_amd import _native as N  # noqa: E402


def run_clients(fn, threads, per_thread, queries):
    lat = [[] for _ in range(threads)]
    barrier = threading.Barrier(threads + 1)

    def client(t):
        barrier.wait()
        for j in range(per_thread):
            q = queries[(t * per_thread + j) % len(queries)]
            t0 = time.perf_counter()
            fn(q)
            lat[t].append((time.perf_counter() - t0) * 1e3)

    ths = [threading.Thread(target=client, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    barrier.wait()
    t0 = time.perf_counter()
    for th in ths:
        th.join()
    wall = time.perf_counter() - t0
    allv = np.concatenate([np.array(x) for x in lat])
    return {"qps": threads * per_thread / wall, "p50_ms": float(np.median(allv)), "p95_ms": float(np.percentile(allv, 95)),
            "p99_ms": float(np.percentile(allv, 99))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--threads", type=int, default=128)
    ap.add_argument("--per-thread", type=int, default=20)
    ap.add_argument("--max-batch", type=int, default=64)
    ap.add_argument("--solo-threads", type=int, default=8)
    args = ap.parse_args()
    ctx = oa.Context(0)
    st = oa.EmbeddingFieldStorage(ctx, dimensions=args.dim, reserve_rows=args.rows,
                                  dtype=N.DTYPE_F16 if args.dtype == "f16" else N.DTYPE_F32)
    st.fill_synthetic(args.rows, seed=0xC0FFEE)
    queries = np.random.default_rng(0xBEEF).standard_normal((256, args.dim)).astype(np.float32)
    b = oa.SearchBatcher(st, max_batch=args.max_batch, max_wait_us=0)
    for i in range(3):
        st.storage_search(queries[i], args.k)
        b.search(queries[i], args.k)
    # correctness: batched answers == solo answers
    ids0, d0, c0 = st.storage_search(queries[5], args.k)
    ids1, d1 = b.search(queries[5], args.k)
    same = ids0[0, :c0[0]].tolist() == ids1.tolist() and np.array_equal(d0[0, :c0[0]], d1)
    solo = run_clients(lambda q: st.storage_search(q, args.k), args.solo_threads, args.per_thread, queries)
    s0 = b.stats()
    batched = run_clients(lambda q: b.search(q, args.k), args.threads, args.per_thread, queries)
    s1 = b.stats()
    mean_batch = (s1["requests"] - s0["requests"]) / max(1, s1["batches"] - s0["batches"])
    print(json.dumps({"metric": "cosine top-%d QPS, single-query requests from concurrent clients" % args.k,
                      "config": {"rows": args.rows, "dim": args.dim, "dtype": args.dtype, "k": args.k},
                      "direct_calls": {"client_threads": args.solo_threads, **solo},
                      "through_batcher": {"client_threads": args.threads, "max_batch": args.max_batch,
                                          "mean_batch": mean_batch, "largest_batch": s1["largest_batch"], **batched},
                      "speedup": batched["qps"] / solo["qps"], "batched_equals_solo": bool(same)}))


if __name__ == "__main__":
    main()
