#!/usr/bin/env python3
"""Package power / clocks while one kernel configuration runs back to back for a few seconds (rocm-smi sampled from a
side thread).  MODE = f16 wide mode (1 = K2c, 3 = K2d geometry 2 ...), DBG = ablation build, CONST_ROWS/ZERO_Q as in
k2d_probe.py.  Also: MODE=f32 runs the fp32 K1 scan (no MFMA) for comparison."""
import os, subprocess, sys, threading, time, re
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import _native as N  # noqa: E402

ctx = oa.Context(0)
d, k = 768, 100
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), out))
        except Exception as e:  # noqa: BLE001
            samples.append((time.perf_counter(), f"ERR {e}"))
        time.sleep(0.15)


def run(label, fn, seconds=2.5):
    global stop, samples
    fn(); ctx.synchronize()
    samples, stop = [], False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < seconds:
        fn(); reps += 1
    ctx.synchronize()
    el = time.perf_counter() - t0
    stop = True; th.join()
    pw, sclk = [], []
    for _, s in samples:
        pw += [float(x) for x in re.findall(r'"(?:Average|Current) (?:Graphics Package|Socket Graphics Package) Power \(W\)": "([0-9.]+)"', s)]
        sclk += [float(x) for x in re.findall(r'"sclk clock speed:": "\((\d+)Mhz\)"', s)]
    print(f"{label:60s} {el/reps*1e3:8.3f} ms/call | power W: n={len(pw)} max={max(pw) if pw else None} mean={np.mean(pw) if pw else None} | sclk MHz: {sorted(set(sclk))[-3:] if sclk else None}", flush=True)
    if not pw and samples:
        print("   raw sample:", samples[len(samples)//2][1][:600].replace("\n", " "))


n16 = 10_000_000
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n16, dtype=N.DTYPE_F16)
st.fill_synthetic(n16, seed=0xC0FFEE)
q = np.random.default_rng(1).standard_normal((256, d)).astype(np.float32)
for mode, dbg in ((1, 0), (2, 0), (1, 9), (1, 2)):
    ctx.set_f16_wide(mode)
    os.environ["ORAMA_K2C_DBG"] = str(dbg)
    run(f"f16 Q=256 mode {mode} DBG {dbg}", lambda: st.storage_search(q, k))
os.environ["ORAMA_K2C_DBG"] = "0"
run("f16 Q=64 (K2)", lambda: st.storage_search(q[:64], k))
st.close()
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=5_000_000)
st.fill_synthetic(5_000_000, seed=0xC0FFEE)
run("f32 Q=1 (K1, 5M rows)", lambda: st.storage_search(q[0], k))
