#!/usr/bin/env python3
"""Where the vector instructions of K3r's scoring launch go: a static budget from the gfx950 ISA of
range_score_compact_kernel (the plain top-k instantiation), phase by phase.

    python scripts/k3r_isa_budget.py > profiles/r06_k3r_isa_budget.md

hipcc compiles csrc/bm25_ranges.hip to assembly with the product flags (no GPU needed); the kernel's text is cut at its
`s_barrier`s (the phases of score_body are separated by barriers) and the segments are grouped into the six bodies the
kernel carries (2 / 4 / 5 / 6 / 7 / 8 postings per lane, told apart by the size of their rank phase).  Counted per
segment: VALU / SALU / LDS / VMEM instructions as written (loops counted once, both sides of a branch counted).  The
dynamic figure (SQ counters, profiles/r04_k3r_sq_counters_v6.md) is 2.50 VALU wave instructions per posting; the static
count of the 7-posting body is ~3.0 — the difference is the branch bodies most waves skip.
"""
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oramacore_amd import _build  # noqa: E402

KERNEL = "range_score_compact_kernel"
PHASES = ["gather + bitmap", "rank scan, part 1", "rank scan, part 2",
          "rank + presence mask per posting", "mask read-back", "cells of multi-posting documents", "score + floors (ballot chains)",
          "fold of multi-posting documents + survivor ballots"]


def main() -> None:
    src = ROOT / "oramacore_amd" / "csrc" / "bm25_ranges.hip"
    flags = [f for f in _build._flags() if not f.startswith("-Rpass")]
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k3r.s"
        subprocess.run([_build._hipcc(), *flags, "-ffp-contract=off", "--cuda-device-only", "-S", str(src), "-o", str(out)],
                       check=True, capture_output=True)
        text = out.read_text().split("\n")
    begin = next(i for i, l in enumerate(text) if l.startswith("_ZN") and KERNEL in l.split(":")[0])
    end = next(i for i in range(begin, len(text)) if text[i].startswith(".Lfunc_end"))
    segs, cur = [], dict(v=0, s=0, lds=0, vmem=0)
    for l in text[begin:end]:
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            segs.append(cur)
            cur = dict(v=0, s=0, lds=0, vmem=0)
        elif op == "ds_read_u16" and len(segs) >= 1 and (len(segs) - 1) % 8 == 0 and "cut" not in cur:
            # the first read of the block -> run table opens a body's gather: what came before it in this segment is the TAIL of
            # the body above it in the file (survivor append, publish) — or the dispatch on the round count for the first body
            cur["cut"] = dict(v=cur["v"], s=cur["s"], lds=cur["lds"], vmem=cur["vmem"])
            cur["lds"] += 1
        elif op.startswith("v_"):
            cur["v"] += 1
        elif op.startswith("s_"):
            cur["s"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            cur["vmem"] += 1
    segs.append(cur)
    print(f"# K3r scoring launch ({KERNEL}): static instruction budget by phase, from the gfx950 ISA\n")
    print(f"`{KERNEL}`: {sum(s['v'] for s in segs)} VALU, {sum(s['s'] for s in segs)} SALU, {sum(s['lds'] for s in segs)} LDS, "
          f"{sum(s['vmem'] for s in segs)} VMEM instructions as written, {len(segs) - 1} barriers; prologue (tables, bitmap clear, published floor): "
          f"{segs[0]['v']} VALU / {segs[0]['s']} SALU.\n")
    bodies = [segs[1 + 8 * i: 1 + 8 * (i + 1)] for i in range((len(segs) - 1) // 8)]
    tail = segs[1 + 8 * len(bodies):]
    # postings per lane of a body: its rank phase is ~10 VALU per posting
    order = [max(2, round((b[3]["v"] + 1) / 10)) for b in bodies if len(b) == 8]
    print("Bodies in file order (postings per lane): " + ", ".join(map(str, order)) + ".\n")
    # the tail of body i sits in front of body i + 1's gather (the last body's: the last segment)
    tails = [bodies[i + 1][0].get("cut") if i + 1 < len(bodies) else (tail[-1] if tail else None) for i in range(len(bodies))]
    for b, n, tl in zip(bodies, order, tails):
        if len(b) != 8:
            continue
        g = dict(b[0])
        if "cut" in g:
            for k in ("v", "s", "lds", "vmem"):
                g[k] -= g["cut"][k]
        b = [g, *b[1:]] + ([tl] if tl else [])
        print(f"## body of {n} postings per lane ({n * 64} posting slots per wave)\n")
        print("| phase | VALU | per posting | SALU | LDS | VMEM |")
        print("|---|---|---|---|---|---|")
        tot = 0
        for name, s in zip(PHASES + ["append of the survivors, publish, count"], b):
            print(f"| {name} | {s['v']} | {s['v'] / n:.1f} | {s['s']} | {s['lds']} | {s['vmem']} |")
            tot += s["v"]
        print(f"| **sum** | **{tot}** | **{tot / n:.1f}** (= {tot / (n * 64):.2f} wave instructions per posting slot) | | | |\n")


if __name__ == "__main__":
    main()
