#!/usr/bin/env python3
"""Package power, shader clock and JOULES PER PASS of the wide fp16 scans (10 M x 768 fp16) — K2d (mode 2), K2q (5), K2h (4),
their ablation builds and smaller batches (lower MFMA : byte ratio).  rocm-smi sampled from a side thread while one
configuration runs back to back; energy = mean package power x scan time (HIP events of the library)."""
import os, re, subprocess, sys, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import _native as N  # noqa: E402

ctx = oa.Context(0)
n, d, k = 10_000_000, 768, 100
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=N.DTYPE_F16)
st.fill_synthetic(n, seed=0xC0FFEE)
q = np.random.default_rng(1).standard_normal((256, d)).astype(np.float32)
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out)
        except Exception as e:  # noqa: BLE001
            samples.append(f"ERR {e}")
        time.sleep(0.1)


def run(label, nq, seconds=2.0):
    global stop, samples
    for _ in range(2):
        st.storage_search(q[:nq], k)
    ctx.prof_reset(); ctx.prof_enable(True)
    samples, stop = [], False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < seconds:
        st.storage_search(q[:nq], k); reps += 1
    el = time.perf_counter() - t0
    stop = True; th.join()
    ctx.prof_enable(False)
    scan_ms = ctx.prof_get("vec_scan_f16")[0] / reps
    pw, sclk = [], []
    for s in samples:
        pw += [float(x) for x in re.findall(r'"(?:Average|Current) (?:Graphics Package|Socket Graphics Package) Power \(W\)": "([0-9.]+)"', s)]
        sclk += [float(x) for x in re.findall(r'"sclk clock speed:": "\((\d+)Mhz\)"', s)]
    mean_w = float(np.mean(pw)) if pw else float("nan")
    duty = scan_ms / (el / reps * 1e3)
    print(f"{label:44s} scan {scan_ms:6.3f} ms/pass ({n*768*2/scan_ms/1e6:6.0f} GB/s) | call {el/reps*1e3:6.3f} ms | power mean {mean_w:6.0f} W max {max(pw) if pw else 0:5.0f} | "
          f"sclk {np.median(sclk) if sclk else 0:5.0f} MHz | {mean_w*el/reps:6.3f} J/call = {mean_w*el/reps/nq*1e3:6.2f} mJ/query | scan duty {duty:4.2f}", flush=True)


for mode, dbg, nq in ((2, 0, 256), (4, 0, 256), (5, 0, 256), (4, 32, 256), (4, 42, 256), (4, 9, 256), (4, 34, 256), (4, 40, 256),
                      (5, 32, 256), (5, 42, 256), (5, 9, 256), (2, 0, 128), (2, 0, 192), (4, 0, 192), (5, 0, 192)):
    ctx.set_f16_wide(mode)
    os.environ["ORAMA_K2C_DBG"] = str(dbg)
    run(f"mode {mode} DBG {dbg} Q={nq}", nq)
os.environ["ORAMA_K2C_DBG"] = "0"
run("K2 Q=64", 64)
