#!/usr/bin/env python3
"""C1 (BASELINE configs[0], plumbing): the benches/fulltext_simple.rs workload shape through the scoring
dispatcher mirror — docs "document content technology software development number {i}", N in {1000, 5000},
queries "technology", "technology software", "development", limit 10 (fulltext_simple.rs:383-400,436-465).

Reports host-API latency per query (tokenise → term lookup → K3/K4 on the GPU → top-10 back on the host) next to
the oracle (CPU restatement) on the same postings, and checks the two agree bit-for-bit.  This configuration is
launch/latency-bound (a few thousand postings): it measures plumbing overhead, not bandwidth.

    python scripts/bench_c1.py [--reps 200]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oramacore_amd as oa  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker + CPU baseline leg only)
from oramacore_amd.token_score import DEFAULT_EXACT_MATCH_BOOST  # noqa: E402
from oramacore_amd.token_score import (FulltextMode, Index, StringFieldStorage, TokenScoreContext,  # noqa: E402
                                       TokenScoreParams)

F = np.float32


def oracle_entries(idx, tokens):
    entries = []
    for ti, tok in enumerate(tokens):
        for fid in sorted(idx.string_fields):
            sf = idx.string_fields[fid]
            for term in [t for t in sorted(sf.postings) if t.startswith(tok)]:
                pl = sorted(sf.postings[term].items())
                docs = np.array([d for d, _ in pl], dtype=np.uint64)
                # (the exact-match factor of the store — the mirror's declared default — on the term that IS the token)
                bo = F(DEFAULT_EXACT_MATCH_BOOST) if term == tok else F(1.0)
                ntf = np.array([F(bo * orc.bm25f_normalized_tf(tf, sf.field_len[d], sf.avg_field_length(), 0.75))
                                for d, tf in pl], dtype=np.float32)
                entries.append((ti, docs, ntf))
    return entries


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    ctx = oa.Context(0)
    out = {"metric": "fulltext_simple host-API latency", "unit": "ms", "higher_is_better": False, "cases": []}
    for n in (1000, 5000):
        idx = Index(ctx)
        idx.string_fields[0] = StringFieldStorage()
        for i in range(n):
            idx.document_ids.add(i)
            idx.string_fields[0].insert(i, f"document content technology software development number {i}")
        idx.commit()
        tsc = TokenScoreContext(idx)
        for query in ("technology", "technology software", "development"):
            params = TokenScoreParams(mode=FulltextMode(query), limit=10)
            for _ in range(10):
                hits, count = tsc.execute(params)
            lat = []
            for _ in range(args.reps):
                t0 = time.perf_counter()
                hits, count = tsc.execute(params)
                lat.append((time.perf_counter() - t0) * 1e3)
            toks = [t for t, _ in tsc.text_parser.tokenize_and_stem(query)]
            entries = oracle_entries(idx, toks)
            cpu = []
            for _ in range(max(10, args.reps // 10)):
                t0 = time.perf_counter()
                od, os_ = orc.search_full_text(entries, len(toks), float(n), 1.2, None)
                td, ts = orc.top_n(od, os_, 10)
                cpu.append((time.perf_counter() - t0) * 1e3)
            ok = (count == len(od) and [h[0] for h in hits] == td.tolist()
                  and np.array_equal(np.array([h[1] for h in hits], dtype=np.float32).view(np.uint32),
                                     ts.view(np.uint32)))
            out["cases"].append({"docs": n, "query": query, "count": count, "postings": int(sum(len(e[1]) for e in entries)),
                                 "gpu_path_p50_ms": float(np.median(lat)), "gpu_path_p95_ms": float(np.percentile(lat, 95)),
                                 "oracle_scoring_only_p50_ms": float(np.median(cpu)),
                                 "bit_exact_vs_oracle": bool(ok)})
    out["note"] = ("GPU path = tokenise + dictionary lookup (Python) + K3/K4 + D2H; oracle column = scoring + top-n only on "
                   "prebuilt contributions (restatement of the reference algorithm, not the reference binary)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
