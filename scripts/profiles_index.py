#!/usr/bin/env python3
"""Regenerate the machine-derived part of profiles/README.md FROM THE FILES (the hand-written index quoted
figures the files no longer held).  For every round it lists the kernel-stats tables with the dominant kernel's average
duration and call count as the file states them, the pytest logs with their pass counts, and the bench JSON lines with
their values; hand-written descriptions of the other records (and, since round 6, the kernel design histories moved out of
DESIGN.md) are kept in profiles/README.notes.md, which the index LINKS to (round 5 appended it verbatim: 150 KB by now).

    python scripts/profiles_index.py > profiles/README.md
"""
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "profiles"


def kernel_stats_line(path: Path) -> str:
    rows = []
    clocks = ""
    for line in path.read_text().splitlines():
        m = re.match(r"\| `([^`]+)` \| (\d+) \| ([0-9.]+) \| ([0-9.]+) \|", line)
        if m:
            rows.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4))))
        if line.startswith("clocks during the timed region"):
            c = re.search(r'"sclk_mhz_median": ([0-9.]+)', line)
            w = re.search(r'"power_w_from_energy_counter": ([0-9.]+)', line) or re.search(r'"power_w_mean": ([0-9.]+)', line)
            if c:
                clocks = f"; sclk {float(c.group(1)):.0f} MHz" + (f", {float(w.group(1)):.0f} W" if w else "") + " during the timed region"
    if not rows:
        return "(no kernel rows)"
    rows.sort(key=lambda r: -r[2])
    top = rows[:2]
    def short(n: str) -> str:
        m = re.search(r"(\w+_kernel\w*)", n)
        return m.group(1) if m else n.split("(")[0][-40:]

    return "; ".join(f"`{short(n)}` {c} calls, avg {a:.1f} us" for n, c, _, a in top) + clocks


def pytest_line(path: Path) -> str:
    m = re.findall(r"(\d+ passed[^\n]*)", path.read_text())
    return m[-1].strip() if m else "(no summary line)"


def bench_line(path: Path) -> str:
    try:
        whole = path.read_text().strip()
        try:
            d = json.loads(whole)  # (round 6: bench_details.json is an indented document)
        except Exception:  # noqa: BLE001
            d = json.loads(whole.splitlines()[-1])
    except Exception:  # noqa: BLE001
        return "(not a JSON line)"
    if not isinstance(d, dict):
        return "(not a bench line: see the notes)"
    if "value" not in d:
        return "(not a bench line: see the notes below)"
    out = f"value {d.get('value', 0):.1f} {d.get('unit', '')}, ms_per_step {d.get('ms_per_step', 0):.3f}, n_gpus {d.get('n_gpus')}"
    if d.get("latency_ms_p50") is not None:
        out += f", p50 {d['latency_ms_p50']:.3f} ms"
    r = d.get("roofline") or {}
    if r:
        out += f", roofline.frac {r.get('frac', 0):.3f} ({r.get('kernel', '')})"
    for name, c in (d.get("configs") or {}).items():
        out += f"; {name} {c.get('value', 0):.0f}"
        b = (c.get("bm25_only") or {})
        if b:
            out += f" (bm25 batch {b.get('value', 0):.0f}, single {b.get('single_query_calls', {}).get('value', 0):.0f})"
    return out


def main() -> None:
    print("# profiles/ — measurement evidence (one MI355X, ROCm 7.x, `gpurun`); files are named per round: `r01_*` … `r06_*`\n")
    print("The tables below are GENERATED from the files by `scripts/profiles_index.py` (figures are read out of each file, not "
          "typed).  Hand-written notes on every other record, and the kernel design histories of rounds 1-5, are in "
          "[`README.notes.md`](README.notes.md).\n")
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        files = sorted(P.glob(f"{rnd}_*"))
        if not files:
            continue
        print(f"## Round {int(rnd[1:])} — generated index\n")
        print("| File | What the file says |")
        print("|---|---|")
        for f in files:
            if f.name.endswith("_kernel_stats.md") or re.search(r"kernel_stats(_v\d+)?\.md$", f.name):
                print(f"| `{f.name}` | rocprofv3 kernel trace: {kernel_stats_line(f)} |")
            elif "pytest" in f.name:
                print(f"| `{f.name}` | {pytest_line(f)} |")
            elif f.suffix == ".json" and "bench" in f.name:
                print(f"| `{f.name}` | {bench_line(f)} |")
        print()
    print("## Round 6 — the other records\n")
    print((P / "README.r06.md").read_text() if (P / "README.r06.md").exists() else "(see README.notes.md)")


if __name__ == "__main__":
    main()
