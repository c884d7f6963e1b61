#!/usr/bin/env python3
"""Is bench.py's clock / power sampler looking at the right GPU, and do its figures agree with rocm-smi?

Prints: the PCI address HIP reports for device 0, every DRM card with ITS PCI address (HIP ordinal != card number on a
multi-GPU host: VERDICT r04 weak #4), one raw dump of the decoded gpu_metrics table, the sampler's summary over an idle
second and over a second of back-to-back fp32 scans, and rocm-smi's view of the same card during the scans."""
import glob
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ctypes  # noqa: E402

import bench  # noqa: E402
import oramacore_amd as oa  # noqa: E402

ctx = oa.Context(0)
bdf = ctx.pci_bus_id()
print("HIP device 0:", ctx.device_info(), "pci", bdf)
for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
    if "-" in os.path.basename(c):
        continue
    dev = os.path.realpath(os.path.join(c, "device"))
    hw = glob.glob(os.path.join(dev, "hwmon/hwmon*/power1_*"))
    print(" ", os.path.basename(c), "->", os.path.basename(dev), "gpu_metrics" if os.path.exists(os.path.join(dev, "gpu_metrics")) else "-",
          [os.path.basename(h) for h in hw])
lib = bench.sampler_lib()
print("sampler lib:", lib)
if lib is not None:
    print("gs_open:", lib.gs_open(bdf.encode()), lib.gs_last_error())
    buf = ctypes.create_string_buffer(2048)
    lib.gs_dump(buf, 2048)
    print("dump:", buf.value.decode())

n, d, k = 2_000_000, 768, 100
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=oa.DTYPE_F32)
st.fill_synthetic(n, seed=1)
q = np.random.default_rng(1).standard_normal((1, d)).astype(np.float32)
with bench.ClockSampler(bdf) as c0:
    time.sleep(1.0)
print("idle  :", c0.summary())

smi = []


def smi_loop():
    for _ in range(3):
        try:
            smi.append(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showbus"], capture_output=True, text=True,
                                      timeout=10).stdout)
        except Exception as e:  # noqa: BLE001
            smi.append(f"ERR {e}")


th = threading.Thread(target=smi_loop)
with bench.ClockSampler(bdf) as c1:
    t0 = time.perf_counter()
    th.start()
    reps = 0
    while time.perf_counter() - t0 < 3.0:
        st.storage_search(q, k)
        reps += 1
    el = time.perf_counter() - t0
th.join()
print(f"busy  : {reps} scans of {n} x {d} f32 in {el:.2f} s = {reps * n * d * 4 / el / 1e9:.0f} GB/s")
print("busy  :", c1.summary())
print("rocm-smi during the scans (last of three):")
print(smi[-1])
