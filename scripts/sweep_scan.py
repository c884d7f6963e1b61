#!/usr/bin/env python3
"""Tuning sweep of the K1 scan launch geometry on a real MI355X (one process, one corpus fill).

Prints one line per (rows_per_wave, blocks_per_cu, nontemporal): average kernel time (HIP events on
the launching stream) and achieved algorithmic GB/s.  Usage:
    python scripts/sweep_scan.py [--rows 10000000] [--dim 768] [--iters 20]
"""
import argparse
import itertools
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import oramacore_amd as oa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    ctx = oa.Context(0)
    st = oa.EmbeddingFieldStorage(ctx, dimensions=args.dim, reserve_rows=args.rows)
    st.fill_synthetic(args.rows, seed=0xC0FFEE, first_doc_id=0)
    rng = np.random.default_rng(0xBEEF)
    qs = rng.standard_normal((args.iters + 3, args.dim)).astype(np.float32)
    bytes_per = args.rows * args.dim * 4
    results = []
    ref = None
    for rows, bpc, nt in itertools.product((1, 2, 4, 8), (2, 4, 8, 16), (0, 1)):
        ctx.set_scan_tuning(rows, bpc, bool(nt))
        for i in range(3):
            st.storage_search(qs[i], 100)
        ctx.prof_reset()
        ctx.prof_enable(True)
        for i in range(args.iters):
            ids, dist, cnt = st.storage_search(qs[3 + i], 100)
        ctx.prof_enable(False)
        ms, n = ctx.prof_get("vec_scan_f32")
        sel_ms, sel_n = ctx.prof_get("topk_select")
        avg = ms / n
        gbs = bytes_per / (avg * 1e-3) / 1e9
        if ref is None:
            ref = (ids.copy(), dist.copy())
        else:  # every geometry must return the same answer
            assert np.array_equal(ref[0], ids) and np.allclose(ref[1], dist, atol=1e-6)
        rec = {"rows_per_wave": rows, "blocks_per_cu": bpc, "nt": nt, "scan_ms": avg, "GBps": gbs,
               "select_ms": sel_ms / sel_n}
        results.append(rec)
        print(json.dumps(rec), flush=True)
    best = max(results, key=lambda r: r["GBps"])
    print("BEST", json.dumps(best), flush=True)
    if args.out:
        Path(args.out).write_text(json.dumps({"args": vars(args), "results": results, "best": best}, indent=1))


if __name__ == "__main__":
    main()
