#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` prints: calls, total / average / min / max duration, share.
Usage: python scripts/rocpd_summary.py <results.db> [> profiles/xxx_kernel_stats.md]"""
import sqlite3
import sys


def main(path: str) -> None:
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count "
                      "from kernels").fetchall() if _has_cols(db) else None
    if rows is None:
        rows = db.execute("select name, start, end from kernels").fetchall()
        rows = [(r[0], r[1], r[2], None, None, None, None, None) for r in rows]
    agg = {}
    for name, start, end, grid, wg, lds, vgpr, sgpr in rows:
        a = agg.setdefault(name, {"n": 0, "tot": 0, "min": 1 << 62, "max": 0, "grid": grid, "wg": wg, "lds": lds,
                                  "vgpr": vgpr, "sgpr": sgpr})
        d = end - start
        a["n"] += 1
        a["tot"] += d
        a["min"] = min(a["min"], d)
        a["max"] = max(a["max"], d)
    total = sum(a["tot"] for a in agg.values()) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | lds B | vgpr | sgpr |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"| `{short}` | {a['n']} | {a['tot'] / 1e6:.3f} | {a['tot'] / a['n'] / 1e3:.2f} | {a['min'] / 1e3:.2f} | "
              f"{a['max'] / 1e3:.2f} | {100.0 * a['tot'] / total:.2f} | {a['grid']} | {a['wg']} | {a['lds']} | "
              f"{a['vgpr']} | {a['sgpr']} |")


def _has_cols(db) -> bool:
    cols = [c[1] for c in db.execute("pragma table_info('kernels')")]
    return all(c in cols for c in ("grid_x", "workgroup_x", "lds_size", "vgpr_count", "sgpr_count"))


if __name__ == "__main__":
    main(sys.argv[1])
