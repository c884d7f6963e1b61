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
