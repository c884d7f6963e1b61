"""bench.py's measurement helpers that need no GPU: the clock / power sampler degrades to `available: false` instead of
failing a run (no device, no librocm_smi64 device, no sysfs), percentiles, and the per-step grouping of launch samples
(a step of the wide fp16 paths is several launches of different sizes: they are summed step by step before the median)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def test_sampler_without_a_device_says_so():
    for bdf in (None, "", "0000:ff:1f.7"):
        with bench.ClockSampler(bdf) as c:
            pass
        s = c.summary()
        assert s["available"] is False and s["pci_bus_id"] == (bdf or None)


def test_percentiles():
    p = bench.pctl([1.0, 2.0, 3.0, 4.0, 100.0])
    assert p["latency_ms_p50"] == 3.0 and p["latency_samples"] == 5 and 4.0 < p["latency_ms_p95"] <= 100.0


def test_launch_samples_are_grouped_by_step():
    class Ctx:
        def prof_samples(self, kernel):
            assert kernel == "vec_scan_f16"
            return np.array([0.2, 1.0, 1.0, 0.2, 1.1, 1.1, 0.2], dtype=np.float32)  # 2 whole steps of 3 launches + a stray one

    per_step = bench.scan_step_samples(Ctx(), "vec_scan_f16", 3)
    assert np.allclose(per_step, [2.2, 2.4])
    assert bench.scan_step_samples(Ctx(), "vec_scan_f16", 8).size == 0


def _recorded_runs():
    import json
    root = Path(__file__).resolve().parent.parent / "profiles"
    return [json.loads(p.read_text()) for p in sorted(root.glob("r0[5-9]_bench_all_configs*.json"))]


def test_headline_line_fits(capfd, tmp_path):
    """Round 5's line grew to 24.9 KB and the driver's bounded tail of stdout no longer held a whole JSON object.  The LAST
    stdout line is now a compact object under HEADLINE_BUDGET; the full record goes to the details file and stderr."""
    import json
    runs = _recorded_runs()
    assert runs, "no recorded bench run under profiles/"
    for rec in runs:
        line = bench.headline_line(rec, "bench_details.json")
        text = json.dumps(line, separators=(",", ":"))
        assert len(text) < bench.HEADLINE_BUDGET == 4096
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in line, key
        assert line["config"]["workload"] and "model" not in line["config"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in line["roofline"], key
        assert abs(line["roofline"]["frac"] - rec["roofline"]["frac"]) < 1e-4
        assert abs(line["value"] - rec["value"]) / rec["value"] < 1e-5
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in line["cpu_baseline"], key
        for leg in ("c2", "c3", "c4", "c5_shard"):
            assert set(line["configs"][leg]) == {"value", "ms_per_step", "latency_ms_p50", "roofline"}
    # what a driver that keeps the last 4 096 bytes of stdout would parse: emit() after 30 KB of other output
    rec = runs[-1]
    print("x" * 30000)
    bench.emit(rec, str(tmp_path / "d.json"))
    captured = capfd.readouterr()
    tail = captured.out[-4096:]
    parsed = json.loads(tail[tail.index("\n") + 1:] if "\n" in tail.rstrip("\n") else tail)
    assert parsed["roofline"]["frac"] and parsed["cpu_baseline"]["value"] and parsed["details"] == "d.json"
    assert json.loads((tmp_path / "d.json").read_text())["configs"]["c4"]["bm25_only"]["value"] == rec["configs"]["c4"]["bm25_only"]["value"]
    assert json.loads(captured.err.strip().splitlines()[-1]) == rec  # the long form went to stderr, whole


def test_headline_line_sheds_before_it_overflows():
    """A record with pathologically many legs still yields a line under the budget with the contract keys intact."""
    rec = _recorded_runs()[-1]
    rec = dict(rec, configs={f"leg{i}": rec["configs"]["c2"] for i in range(60)})
    line = bench.headline_line(rec)
    import json
    assert len(json.dumps(line, separators=(",", ":"))) < bench.HEADLINE_BUDGET and "roofline" in line and "cpu_baseline" in line
