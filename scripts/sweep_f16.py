#!/usr/bin/env python3
"""Tuning sweep of the K2 (fp16 MFMA) register-ring geometry on a real MI355X (C3 shape by default).
    python scripts/sweep_f16.py [--rows 10000000] [--dim 768] [--q 64] [--iters 10]
"""
import argparse
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
    ap.add_argument("--q", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    ctx = oa.Context(0)
    st = oa.EmbeddingFieldStorage(ctx, dimensions=args.dim, reserve_rows=args.rows, dtype=oa.DTYPE_F16)
    st.fill_synthetic(args.rows, seed=0xC0FFEE)
    rng = np.random.default_rng(0xBEEF)
    qs = rng.standard_normal(((args.iters + 2) * args.q, args.dim)).astype(np.float32)
    kpad = (args.dim + 127) // 128 * 128
    bytes_per = args.rows * kpad * 2
    results, ref = [], None
    for kc, nbuf in ((8, 2), (8, 3), (8, 4), (12, 2), (12, 3), (16, 2)):
        ctx.set_f16_tuning(kc, nbuf)
        for i in range(2):
            st.storage_search(qs[i * args.q:(i + 1) * args.q], 100)
        ctx.prof_reset()
        ctx.prof_enable(True)
        for i in range(args.iters):
            ids, dist, cnt = st.storage_search(qs[(2 + i) * args.q:(3 + i) * args.q], 100)
        ctx.prof_enable(False)
        ms, n = ctx.prof_get("vec_scan_f16")
        sel_ms, sel_n = ctx.prof_get("topk_select")
        per_step = ms / args.iters
        if ref is None:
            ref = (ids.copy(), dist.copy())
        else:
            assert np.array_equal(ref[0], ids) and np.allclose(ref[1], dist, atol=1e-6)
        rec = {"kc": kc, "nbuf": nbuf, "scan_ms_per_step": per_step, "GBps": bytes_per / (per_step * 1e-3) / 1e9,
               "select_ms_per_step": sel_ms / args.iters}
        results.append(rec)
        print(json.dumps(rec), flush=True)
    best = max(results, key=lambda r: r["GBps"])
    print("BEST", json.dumps(best), flush=True)
    if args.out:
        Path(args.out).write_text(json.dumps({"args": vars(args), "results": results, "best": best}, indent=1))


if __name__ == "__main__":
    main()
