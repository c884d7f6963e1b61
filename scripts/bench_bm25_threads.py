#!/usr/bin/env python3
"""BM25-only throughput of the C4 full-text shape (10 M docs, 12 tokens, ~600 K postings per query) with T concurrent
caller threads — the reference serves searches from many tokio workers; every call is an independent
orama_post_search (own stream + scratch set)."""
import argparse, json, sys, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import fulltext as ft  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=10_000_000)
ap.add_argument("--threads", default="1,2,4,8,16")
ap.add_argument("--per-thread", type=int, default=150)
ap.add_argument("--budget-s", type=float, default=6.0, help="give up on a thread count after this many seconds")
ap.add_argument("--scorers", default="k3r,k3", help="k3r = range-partitioned batch scorer, k3 = per-document records")
ap.add_argument("--batch-callers", default="1,2,4", help="concurrent callers of the batch entry")
args = ap.parse_args()
n, T, k = args.docs, 12, 100
ctx = oa.Context(0)
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
post = ft.PostingsStore(ctx)
post.fill_synthetic(n, ranks, seed=0xB25)
qlists = [rng.choice(len(ranks), size=T, replace=False) for _ in range(512)]
refs = [[(t, int(l), 1.0) for t, l in enumerate(ql)] for ql in qlists]
lens = np.array([len(post.get_list(int(l))[0]) for l in range(len(ranks))])
avg_postings = float(np.mean([lens[ql].sum() for ql in qlists]))
print(f"postings referenced per query: {avg_postings:.0f} on average", flush=True)
all_out = {"avg_postings_per_query": avg_postings}
for scorer in args.scorers.split(","):
    ctx.set_bm25_ranges(scorer == "k3r")
    ctx.prof_reset()
    print(f"--- scorer {scorer}", flush=True)
    out = {}
    for nt in [int(x) for x in args.threads.split(",") if x and int(x) > 0]:
        for i in range(20):
            post.search(refs[i], T, float(n), k)

        done = [0] * nt
        deadline = time.perf_counter() + args.budget_s

        def worker(tid):
            for i in range(args.per_thread):
                if time.perf_counter() > deadline:
                    return
                post.search(refs[(tid * 131 + i) % len(refs)], T, float(n), k)
                done[tid] += 1

        ths = [threading.Thread(target=worker, args=(t,)) for t in range(nt)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        el = time.perf_counter() - t0
        out[nt] = sum(done) / el
        print(f"caller threads {nt:3d}: {out[nt]:9.0f} queries/s ({sum(done)} queries in {el:.2f} s)", flush=True)
    # the batch entry: one call per 1024 queries, from 1..4 concurrent callers
    batch = [(refs[i % len(refs)], T, None) for i in range(1024)]
    post.search_batch(batch[:64], float(n), k, max_parallel=8)
    for callers in [int(x) for x in args.batch_callers.split(",")]:
        def bworker():
            post.search_batch(batch, float(n), k, max_parallel=8)
        ths = [threading.Thread(target=bworker) for _ in range(callers)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        el = time.perf_counter() - t0
        out[f"batch_callers{callers}"] = callers * len(batch) / el
        print(f"orama_post_search_batch x {callers} callers: {callers * len(batch) / el:9.0f} queries/s", flush=True)
    # single-query callers through the request batcher (orama_post_batcher_*)
    if scorer == "k3r":
        for nt in (8, 32, 128):
            batcher = ft.PostSearchBatcher(post, max_batch=256)
            done = [0] * nt
            per = max(40, 2048 // nt)

            def bw(tid):
                for i in range(per):
                    batcher.search(refs[(tid * 131 + i) % len(refs)], T, float(n), k)
                    done[tid] += 1

            ths = [threading.Thread(target=bw, args=(t,)) for t in range(nt)]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            el = time.perf_counter() - t0
            st = batcher.stats()
            batcher.close()
            out[f"batcher_threads{nt}"] = sum(done) / el
            print(f"request batcher, {nt:3d} caller threads: {sum(done) / el:9.0f} queries/s (mean batch {st['mean_batch']:.1f})", flush=True)
    # device time per query of the batch path (HIP events around the kernels, one caller)
    ctx.prof_reset()
    ctx.prof_enable(True)
    post.search_batch(batch, float(n), k, max_parallel=8)
    ctx.prof_enable(False)
    prof = {}
    for name in ("bm25_range_bounds", "bm25_range_df", "bm25_range_score", "bm25_accumulate", "bm25_finalize", "topk_select"):
        ms, cnt = ctx.prof_get(name)
        if cnt:
            prof[name] = {"ms_total": ms, "launches": cnt, "us_per_query": ms * 1e3 / len(batch)}
    print("   device time of one 1024-query batch:", json.dumps(prof), flush=True)
    out["device_time"] = prof
    all_out[scorer] = out
print(json.dumps({"metric": "BM25-only queries/s, 10M docs, 12 tokens/query, top-100", "by_scorer": all_out}))
